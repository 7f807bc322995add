// aa_fused_u8_v3_c1.hip — instantiations of the fused uint8 kernel (aa_fused_u8_v3_impl.h) for 1 interleaved channel (planar bytes).
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_c1(int tw, int maxc, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return flt ? dispatch_tw_flt<1>(tw, maxc, p, q, lds, 0) : dispatch_tw<1>(tw, maxc, p, q, lds, 0);
}
