// aa_fused_u8_v3_c3g.hip — plane-group instantiations of the fused uint8 kernel (aa_fused_u8_v3_impl.h, template parameter PL): planar
// (NCHW) images of three channels, the same strip and band of all three planes in one wave; Pillow arithmetic, shrinking heights.
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_c3g(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return dispatch_tw_planes<3>(tw, maxc, p, q, lds);
}
