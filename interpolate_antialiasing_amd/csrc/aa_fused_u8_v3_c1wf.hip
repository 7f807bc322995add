// aa_fused_u8_v3_c1wf.hip — wide-window instantiations (17 .. 34 taps) of the fused uint8 kernel (aa_fused_u8_v3_impl.h) for 1 channel per
// pixel in FLOAT arithmetic: the reference harness's uint8 semantics and the uint8 -> float32 conversion at strong down-scaling (test.py's
// 906 -> 120 thumbnails).
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_c1wf(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return dispatch_tw_wide_flt<1>(tw, maxc, p, q, lds);
}
