// aa_fused_u8_v3_c3u.hip — instantiations of the fused uint8 kernel (aa_fused_u8_v3_impl.h) for 3 interleaved channels and
// heights that GROW: the gather-form vertical pass (template parameter UPK), Pillow and harness arithmetic.
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_up_c3(int tw, int upk, bool nonneg, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return dispatch_up<3>(tw, upk, nonneg, flt, p, q, lds);
}
