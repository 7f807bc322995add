// aa_fused_u8_v3_c3gff.hip — the float-arithmetic plane-group instantiations of the fused uint8 kernel in the opt-in TOLERANCE mode
// (AA_FLAG_FAST): fused multiply-adds in both passes (see AA_V3_FLT_FAST in the header).
#define AA_V3_FLT_FAST 1
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_c3gff(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return dispatch_tw_planes_flt<3>(tw, maxc, p, q, lds);
}
