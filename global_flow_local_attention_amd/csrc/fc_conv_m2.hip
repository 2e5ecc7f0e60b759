// fc_conv_kernel<2, *, *>: the FC-layer convolutions in arithmetic mode 2 (fc_gemm.h), see fc_conv_impl.h.
#include "fc_conv_impl.h"

namespace gfla {
GFLA_DEFINE_FC_CONV_MODE(2)

// the same arithmetic on a float32 input map (fc_conv_impl.h: SRC32)
int fc_conv_f32src(const PackedDesc &X32, const void *wk, int64_t w_split_stride, float *out, int64_t out_bs, int ldo,
                   int n_valid, int64_t B, int nch, int M, int Wv, int Wp, int k, const uint32_t *amax_x,
                   const uint32_t *amax_w, hipStream_t stream) {
  if ((k != 3 && k != 5) || !amax_x || !amax_w || X32.pix_stride < 64) return GFLA_ERR_UNSUPPORTED;
  if (B > 65535 || B <= 0 || M <= 0) return B == 0 ? GFLA_OK : GFLA_ERR_UNSUPPORTED;
  if (Wv <= 0 || Wv > Wp || !fc_conv_fits(Wv, Wp, k, 2)) return GFLA_ERR_UNSUPPORTED;
  ConvTiling tl;
  const int nmb = pick_conv_tiling(M, B, (int)ceil_div(n_valid, kFcTN), Wv, Wp, k, 2, &tl);
  if (nmb == 0) return GFLA_ERR_UNSUPPORTED;
  if (k == 3)
    return dispatch_conv<2, 3, true>(nmb, X32, wk, w_split_stride, out, out_bs, ldo, n_valid, B, nch, M, Wv, Wp, tl, amax_x, amax_w,
                                     stream);
  return dispatch_conv<2, 5, true>(nmb, X32, wk, w_split_stride, out, out_bs, ldo, n_valid, B, nch, M, Wv, Wp, tl, amax_x, amax_w,
                                   stream);
}
}  // namespace gfla
