// aa_fused_u8_v3_c4.hip — instantiations of the fused uint8 kernel (aa_fused_u8_v3_impl.h) for 4 interleaved channels.
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_c4(int tw, int maxc, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return flt ? dispatch_tw_flt<4>(tw, maxc, p, q, lds, 0) : dispatch_tw<4>(tw, maxc, p, q, lds, 0);
}
