// Gradient of a replicate (clamp-to-edge) padding, gfx950.
//
// ExtractorAttn extracts the target patches at ZERO flow (base_function.py:806): block_target is the
// replicate-padded unfold of target, so its half of the first FC layer is a stride-1 convolution of the
// padded target (extractor_attn.py).  The padding's backward folds the border strips of the padded
// gradient back onto the edge pixels.  torch does that with one atomicAdd per padded element; written as a
// gather (one thread per UNPADDED element, edge threads sum their strip) it is a single coalesced pass.
#include "gfla_common.h"

namespace gfla {

template <typename T>
__global__ __launch_bounds__(kBlock) void replicate_pad_bwd_kernel(const T *__restrict__ gp, T *__restrict__ g,
                                                                  int64_t n, int H, int W, int pl, int pr, int pt,
                                                                  int pb) {
  using A = typename Num<T>::acc;
  const int64_t index = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (index >= n) return;
  const int x = (int)(index % W);
  const int y = (int)((index / W) % H);
  const int64_t plane = index / ((int64_t)W * H);
  const int Wp = W + pl + pr, Hp = H + pt + pb;
  // padded rows / columns that clamp onto (y, x)
  const int y0 = y == 0 ? 0 : y + pt, y1 = y == H - 1 ? Hp - 1 : y + pt;
  const int x0 = x == 0 ? 0 : x + pl, x1 = x == W - 1 ? Wp - 1 : x + pl;
  const T *src = gp + plane * (int64_t)Hp * Wp;
  A acc = 0;
  for (int yy = y0; yy <= y1; ++yy)
    for (int xx = x0; xx <= x1; ++xx) acc += Num<T>::ld(src + (int64_t)yy * Wp + xx);
  g[index] = Num<T>::from(acc);
}

template <typename T>
static int replicate_pad_bwd(const T *gp, T *g, int64_t planes, int64_t H, int64_t W, int pl, int pr, int pt, int pb,
                             gfla_stream_t stream) {
  if (!gp || !g) return GFLA_ERR_NULL_POINTER;
  if (planes < 0 || H <= 0 || W <= 0 || pl < 0 || pr < 0 || pt < 0 || pb < 0) return GFLA_ERR_BAD_SHAPE;
  if (planes == 0) return GFLA_OK;
  if ((H + pt + pb) * (W + pl + pr) > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  const int64_t n = planes * H * W;
  const int64_t blocks = ceil_div(n, kBlock);
  if (blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  replicate_pad_bwd_kernel<T><<<dim3((unsigned)blocks), dim3(kBlock), 0, static_cast<hipStream_t>(stream)>>>(
      gp, g, n, (int)H, (int)W, pl, pr, pt, pb);
  return launch_status();
}

}  // namespace gfla

extern "C" {
int gfla_replicate_pad_bwd_f32(const float *grad_padded, float *grad_in, int64_t planes, int64_t H, int64_t W,
                               int pad_left, int pad_right, int pad_top, int pad_bottom, gfla_stream_t stream) {
  return gfla::replicate_pad_bwd<float>(grad_padded, grad_in, planes, H, W, pad_left, pad_right, pad_top,
                                        pad_bottom, stream);
}
int gfla_replicate_pad_bwd_f64(const double *grad_padded, double *grad_in, int64_t planes, int64_t H, int64_t W,
                               int pad_left, int pad_right, int pad_top, int pad_bottom, gfla_stream_t stream) {
  return gfla::replicate_pad_bwd<double>(grad_padded, grad_in, planes, H, W, pad_left, pad_right, pad_top,
                                         pad_bottom, stream);
}
}
