// aa_fused_u8_v3_c4f.hip — float-arithmetic instantiations of the fused uint8 kernel (aa_fused_u8_v3_impl.h) for 4 interleaved channels:
// the reference harness's uint8 semantics and the uint8 -> float32 conversion.
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_c4f(int tw, int maxc, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return dispatch_tw_flt<4>(tw, maxc, p, q, lds, 0);
}
