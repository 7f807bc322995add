// aa_fused_u8_v3_c1w.hip — wide-window instantiations (17 .. 34 taps) of the fused uint8 kernel (aa_fused_u8_v3_impl.h) for 1 channel per
// pixel, Pillow arithmetic: strong down-scaling such as test.py's 906 -> 120 thumbnails (17 bilinear / 33 bicubic taps).
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_c1w(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return dispatch_tw_wide<1>(tw, maxc, p, q, lds);
}
