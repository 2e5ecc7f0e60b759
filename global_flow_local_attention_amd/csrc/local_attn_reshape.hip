// local_attn_reshape for gfx950: (B,k*k,H,W) <-> (B,1,k*H,k*W) depth-to-space permutation.
//
// Semantics: local_attn_reshape_kernel.cu:47-58 (forward), :94-106 (backward).  Pure data
// movement, bit-exact.  Both directions are written as a gather with lane <-> destination
// element, so every store is a full coalesced line and no atomics are needed (the reference's
// backward issues one atomicAdd per element onto a bijection, :106).
#include "gfla_common.h"
#include <algorithm>

namespace gfla {

// Round 5: the first version indexed with 64-bit divisions by run-time values (~100 instructions each, emulated) and was
// SLOWER than the reference's kernel at the layer-3 shape (7.1 us against 4.0: both launch-sized, ours arithmetic-bound).
// Now: batch on grid.y, 32-bit index arithmetic inside a sample, the kernel size a template parameter for 2..5 (divisions by
// constants), 0 = any.
template <typename T, int KT>
__global__ __launch_bounds__(kBlock) void lar_fwd_kernel(const T *__restrict__ in, T *__restrict__ out, int n1, int H, int W,
                                                        int k_rt) {
  const int k = KT ? KT : k_rt;
  const int index = blockIdx.x * kBlock + threadIdx.x;   // inside one sample's (k*H, k*W) map
  if (index >= n1) return;
  const int Wo = k * W;
  const int y = index / Wo, x = index - y * Wo;
  const int ys = y / k, xs = x / k;
  const int cs = (y - ys * k) * k + (x - xs * k);
  const int64_t base = (int64_t)blockIdx.y * n1;
  out[base + index] = in[base + (cs * H + ys) * W + xs];
}

template <typename T, int KT>
__global__ __launch_bounds__(kBlock) void lar_bwd_kernel(const T *__restrict__ gout, T *__restrict__ gin, int n1, int H, int W,
                                                        int k_rt) {
  const int k = KT ? KT : k_rt;
  const int index = blockIdx.x * kBlock + threadIdx.x;  // inside one sample's grad_in (k*k, H, W)
  if (index >= n1) return;
  const int HW = H * W;
  const int cs = index / HW, rem = index - cs * HW;
  const int ys = rem / W, xs = rem - ys * W;
  const int i = cs / k, j = cs - i * k;
  const int64_t base = (int64_t)blockIdx.y * n1;
  gin[base + index] = gout[base + (ys * k + i) * (k * W) + (xs * k + j)];
}

template <typename T>
static int reshape(bool fwd, const T *a, T *bptr, int64_t B, int64_t H, int64_t W, int k,
                   gfla_stream_t stream_) {
  if (!a || !bptr) return GFLA_ERR_NULL_POINTER;
  if (B <= 0 || H <= 0 || W <= 0 || k < 1) return GFLA_ERR_BAD_SHAPE;
  // 32-bit element index in the kernels: blockIdx.x * kBlock + threadIdx.x must not wrap in the last (partial) block
  if ((k * H) * (k * W) > 0x7fffff00LL) return GFLA_ERR_UNSUPPORTED;
  const int64_t n1 = (int64_t)k * k * H * W;
  const int64_t blocks = ceil_div(n1, kBlock);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  for (int64_t b0 = 0; b0 < B; b0 += 65535) {   // grid.y limit
    const unsigned nb = (unsigned)std::min<int64_t>(65535, B - b0);
    const dim3 grid((unsigned)blocks, nb), blk(kBlock);
    const T *ap = a + b0 * n1;
    T *bp = bptr + b0 * n1;
#define GFLA_LAR(KT_)                                                                                                   \
  if (fwd) lar_fwd_kernel<T, KT_><<<grid, blk, 0, stream>>>(ap, bp, (int)n1, (int)H, (int)W, k);                          \
  else lar_bwd_kernel<T, KT_><<<grid, blk, 0, stream>>>(ap, bp, (int)n1, (int)H, (int)W, k)
    switch (k) {
      case 2: GFLA_LAR(2); break;
      case 3: GFLA_LAR(3); break;
      case 4: GFLA_LAR(4); break;
      case 5: GFLA_LAR(5); break;
      default: GFLA_LAR(0); break;
    }
#undef GFLA_LAR
  }
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
