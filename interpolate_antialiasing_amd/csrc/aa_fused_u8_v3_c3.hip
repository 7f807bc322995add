// aa_fused_u8_v3_c3.hip — instantiations of the fused uint8 kernel (aa_fused_u8_v3_impl.h) for 3 interleaved channels, Pillow
// (integer) arithmetic; the float-arithmetic (harness / float32-output) instantiations are in aa_fused_u8_v3_c3f.hip so that the two
// halves compile in parallel.
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_c3f(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);

int aa_v3_launch_c3(int tw, int maxc, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return flt ? aa_v3_launch_c3f(tw, maxc, p, q, lds) : dispatch_tw<3>(tw, maxc, p, q, lds, 0);
}
