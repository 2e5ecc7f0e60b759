// local_attn_reshape for gfx950: (B,k*k,H,W) <-> (B,1,k*H,k*W) depth-to-space permutation.
//
// Semantics: local_attn_reshape_kernel.cu:47-58 (forward), :94-106 (backward).  Pure data
// movement, bit-exact.  Both directions are written as a gather with lane <-> destination
// element, so every store is a full coalesced line and no atomics are needed (the reference's
// backward issues one atomicAdd per element onto a bijection, :106).
#include "gfla_common.h"

namespace gfla {

template <typename T>
__global__ __launch_bounds__(kBlock) void lar_fwd_kernel(const T *__restrict__ in, T *__restrict__ out,
                                                        int64_t n, int H, int W, int k) {
  const int64_t index = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (index >= n) return;
  const int Wo = k * W, Ho = k * H;
  const int x = (int)(index % Wo);
  const int y = (int)((index / Wo) % Ho);
  const int64_t b = index / ((int64_t)Wo * Ho);
  const int ys = y / k, xs = x / k;
  const int cs = (y - ys * k) * k + (x - xs * k);
  out[index] = in[((b * k * k + cs) * H + ys) * W + xs];
}

template <typename T>
__global__ __launch_bounds__(kBlock) void lar_bwd_kernel(const T *__restrict__ gout, T *__restrict__ gin,
                                                        int64_t n, int H, int W, int k) {
  const int64_t index = (int64_t)blockIdx.x * kBlock + threadIdx.x;  // over grad_in (B,k*k,H,W)
  if (index >= n) return;
  const int xs = (int)(index % W);
  const int ys = (int)((index / W) % H);
  const int cs = (int)((index / ((int64_t)W * H)) % (k * k));
  const int64_t b = index / ((int64_t)W * H * k * k);
  const int i = cs / k, j = cs - i * k;
  gin[index] = gout[(b * (k * H) + (ys * k + i)) * (int64_t)(k * W) + (xs * k + j)];
}

template <typename T>
static int reshape(bool fwd, const T *a, T *bptr, int64_t B, int64_t H, int64_t W, int k,
                   gfla_stream_t stream_) {
  if (!a || !bptr) return GFLA_ERR_NULL_POINTER;
  if (B <= 0 || H <= 0 || W <= 0 || k < 1) return GFLA_ERR_BAD_SHAPE;
  if ((k * H) * (k * W) > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  const int64_t n = B * k * k * H * W;
  const int64_t blocks = ceil_div(n, kBlock);
  if (blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (fwd)
    hipLaunchKernelGGL((lar_fwd_kernel<T>), dim3((unsigned)blocks), dim3(kBlock), 0, stream, a, bptr, n,
                       (int)H, (int)W, k);
  else
    hipLaunchKernelGGL((lar_bwd_kernel<T>), dim3((unsigned)blocks), dim3(kBlock), 0, stream, a, bptr, n,
                       (int)H, (int)W, k);
  return launch_status();
}

}  // namespace gfla

// The permutation only moves bit patterns, so bf16 reuses the 16-bit integer instantiation.
extern "C" {
int gfla_local_attn_reshape_fwd_f32(const float *in, float *out, int64_t B, int64_t H, int64_t W, int k,
                                    gfla_stream_t st) {
  return gfla::reshape<float>(true, in, out, B, H, W, k, st);
}
int gfla_local_attn_reshape_fwd_f64(const double *in, double *out, int64_t B, int64_t H, int64_t W,
                                    int k, gfla_stream_t st) {
  return gfla::reshape<double>(true, in, out, B, H, W, k, st);
}
int gfla_local_attn_reshape_fwd_bf16(const uint16_t *in, uint16_t *out, int64_t B, int64_t H, int64_t W,
                                     int k, gfla_stream_t st) {
  return gfla::reshape<uint16_t>(true, in, out, B, H, W, k, st);
}
int gfla_local_attn_reshape_bwd_f32(const float *go, float *gi, int64_t B, int64_t H, int64_t W, int k,
                                    gfla_stream_t st) {
  return gfla::reshape<float>(false, go, gi, B, H, W, k, st);
}
int gfla_local_attn_reshape_bwd_f64(const double *go, double *gi, int64_t B, int64_t H, int64_t W, int k,
                                    gfla_stream_t st) {
  return gfla::reshape<double>(false, go, gi, B, H, W, k, st);
}
int gfla_local_attn_reshape_bwd_bf16(const uint16_t *go, uint16_t *gi, int64_t B, int64_t H, int64_t W,
                                     int k, gfla_stream_t st) {
  return gfla::reshape<uint16_t>(false, go, gi, B, H, W, k, st);
}
}
