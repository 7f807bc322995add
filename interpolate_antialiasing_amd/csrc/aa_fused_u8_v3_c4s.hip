// aa_fused_u8_v3_c4s.hip — split-window instantiations (template parameter SP: four lanes per output pixel, 35 .. 136 taps) of the fused
// uint8 kernel (aa_fused_u8_v3_impl.h) for 4 channels per pixel, Pillow arithmetic.
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_c4s(int tws, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return dispatch_tw_split<4>(tws, maxc, p, q, lds);
}
