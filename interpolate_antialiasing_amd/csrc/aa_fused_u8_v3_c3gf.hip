// aa_fused_u8_v3_c3gf.hip — plane-group instantiations (template parameter PL) of the fused uint8 kernel in float arithmetic: planar
// (NCHW) uint8 images of three channels in the reference harness's semantics (test.py's own layout and arithmetic: CHW bytes, float(),
// op, byte()) and the uint8 -> float32 conversion with planes out.
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_c3gf(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return dispatch_tw_planes_flt<3>(tw, maxc, p, q, lds);
}
