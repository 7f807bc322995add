// aa_fused_u8_v3_c3ff.hip — the float-arithmetic instantiations of the fused uint8 kernel (aa_fused_u8_v3_impl.h) for 3 channels per
// pixel in the opt-in TOLERANCE mode (AA_FLAG_FAST): fused multiply-adds in both passes (see AA_V3_FLT_FAST in the header).
#define AA_V3_FLT_FAST 1
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_c3ff(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return dispatch_tw_flt<3>(tw, maxc, p, q, lds, 0);
}
