// aa_fused_u8_v3_c4u.hip — instantiations of the fused uint8 kernel (aa_fused_u8_v3_impl.h) for 4 interleaved channels and
// heights that GROW: the gather-form vertical pass (template parameter UPK), Pillow and harness arithmetic.
#include "aa_fused_u8_v3_impl.h"

int aa_v3_launch_up_c4(int tw, int upk, bool nonneg, bool flt, const FusedU8V3Params &p, const AAProblem &q, size_t lds) {
  return dispatch_up<4>(tw, upk, nonneg, flt, p, q, lds);
}
