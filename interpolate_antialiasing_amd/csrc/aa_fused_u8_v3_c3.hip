// aa_fused_u8_v3_c3.hip — instantiations of the fused uint8 kernel (aa_fused_u8_v3_impl.h) for 3 interleaved channels.
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_c3(int tw, int maxc, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return flt ? dispatch_tw_flt<3>(tw, maxc, p, q, lds, 0) : dispatch_tw<3>(tw, maxc, p, q, lds, 0);
}
