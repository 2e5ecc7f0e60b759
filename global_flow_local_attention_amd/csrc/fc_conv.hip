// Mode dispatch of the FC-layer convolutions (kernels: fc_conv_impl.h, instantiated in fc_conv_m{0,2,3}.hip).
#include "fc_conv_impl.h"

namespace gfla {

bool fc_conv_fits(int Wv, int Wp, int k, int mode) {  // the smallest row tile (+ tap halo) has to fit the LDS of a CU
  return (int64_t)fc_nsplit(mode) * fc_conv_tile_pixels(kFcMinRowBlocks, Wv, Wp, k) * (mode ? 48 : 80) <= 156 * 1024;
}

int fc_conv(const PackedDesc &X, const void *wk, int64_t w_split_stride, float *out, int64_t out_bs, int ldo,
            int n_valid, int64_t B, int nch, int M, int Wv, int Wp, int k, int mode, const uint32_t *amax_x,
            const uint32_t *amax_w, hipStream_t stream) {
  if (!fc_mode_ok(mode) || (k != 3 && k != 5)) return GFLA_ERR_UNSUPPORTED;
  if (B > 65535 || B <= 0 || M <= 0) return B == 0 ? GFLA_OK : GFLA_ERR_UNSUPPORTED;
  if (Wv <= 0 || Wv > Wp || !fc_conv_fits(Wv, Wp, k, mode)) return GFLA_ERR_UNSUPPORTED;
  if (mode == 0)
    return fc_conv_mode<0>(X, wk, w_split_stride, out, out_bs, ldo, n_valid, B, nch, M, Wv, Wp, k, amax_x, amax_w, stream);
  if (mode == 1)
    return fc_conv_mode<1>(X, wk, w_split_stride, out, out_bs, ldo, n_valid, B, nch, M, Wv, Wp, k, amax_x, amax_w, stream);
  if (mode == 2)
    return fc_conv_mode<2>(X, wk, w_split_stride, out, out_bs, ldo, n_valid, B, nch, M, Wv, Wp, k, amax_x, amax_w, stream);
  return fc_conv_mode<3>(X, wk, w_split_stride, out, out_bs, ldo, n_valid, B, nch, M, Wv, Wp, k, amax_x, amax_w, stream);
}

}  // namespace gfla
