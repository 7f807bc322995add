// aa_fused_u8_v3_c1.hip — instantiations of the fused uint8 kernel (aa_fused_u8_v3_impl.h) for 1 interleaved channel (planar bytes), Pillow
// (integer) arithmetic; the float-arithmetic (harness / float32-output) instantiations are in aa_fused_u8_v3_c1f.hip so that the two
// halves compile in parallel.
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_c1f(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds);

int aa_v3_launch_c1(int tw, int maxc, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return flt ? aa_v3_launch_c1f(tw, maxc, p, q, lds) : dispatch_tw<1>(tw, maxc, p, q, lds, 0);
}
