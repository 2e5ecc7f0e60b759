// The blend FaceTargetNet.forward puts behind each pair of ExtractorAttn blocks (generator.py:496-499):
//
//     out_p = out * (1 - mask_p) + attn_p * mask_p ;  out_r = out * (1 - mask_r) + attn_r * mask_r ;  out = out_p + out_r
//
// Nine elementwise torch kernels forward and ~14 backward per layer and frame; at the face model's batch they are
// launch-sized (5-13 us each: 16 % of the bf16 face step's GPU time, profiles/r4_face_bf16_kernel_stats.txt).  One kernel
// each way here.  Forward: every intermediate is rounded to the storage type exactly where the op-by-op evaluation rounds
// it (f32: separate multiply and add, no contraction; bf16: round-to-nearest-even after every op), so the result is the
// op-by-op result bit for bit.  Backward: d/d out, d/d attn_p, d/d attn_r per element; d/d mask_p, d/d mask_r are sums
// over the channels, accumulated in float32 (thread = pixel x channel chunk, one atomic per thread and mask).
#include "gfla_common.h"

namespace gfla {

template <typename T>
__device__ __forceinline__ float rnd(float v) {   // round a float result to the storage type and back
  return v;
}
template <>
__device__ __forceinline__ float rnd<bf16_t>(float v) {
  return __uint_as_float((uint32_t)Num<bf16_t>::pack(v) << 16);
}

constexpr int kBlendChunk = 16;   // channels per thread

template <typename T>
__global__ __launch_bounds__(256) void mask_blend_fwd_kernel(const T *__restrict__ out, const T *__restrict__ ap,
                                                            const T *__restrict__ ar, const T *__restrict__ mp,
                                                            const T *__restrict__ mr, T *__restrict__ y, int C, int HW,
                                                            int nchunk) {
#pragma clang fp contract(off)   // separate multiplies and adds, as the op-by-op evaluation has them (plain operators here:
                                 // HIP's __fmul_rn / __fadd_rn are inlined with the header's contraction setting)
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const int chunk = blockIdx.y, b = blockIdx.z;
  const float m_p = Num<T>::ld(mp + (int64_t)b * HW + p), m_r = Num<T>::ld(mr + (int64_t)b * HW + p);
  const float n_p = rnd<T>(1.f - m_p), n_r = rnd<T>(1.f - m_r);
  const int c0 = chunk * kBlendChunk, c1 = min(C, c0 + kBlendChunk);
  int64_t i = ((int64_t)b * C + c0) * HW + p;
  for (int c = c0; c < c1; ++c, i += HW) {
    const float o = Num<T>::ld(out + i), a_p = Num<T>::ld(ap + i), a_r = Num<T>::ld(ar + i);
    const float op = rnd<T>(rnd<T>(o * n_p) + rnd<T>(a_p * m_p));
    const float orr = rnd<T>(rnd<T>(o * n_r) + rnd<T>(a_r * m_r));
    y[i] = Num<T>::from(op + orr);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void mask_blend_bwd_kernel(const T *__restrict__ out, const T *__restrict__ ap,
                                                            const T *__restrict__ ar, const T *__restrict__ mp,
                                                            const T *__restrict__ mr, const T *__restrict__ g,
                                                            T *__restrict__ g_out, T *__restrict__ g_ap,
                                                            T *__restrict__ g_ar, float *__restrict__ g_mp,
                                                            float *__restrict__ g_mr, int C, int HW) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const int chunk = blockIdx.y, b = blockIdx.z;
  const float m_p = Num<T>::ld(mp + (int64_t)b * HW + p), m_r = Num<T>::ld(mr + (int64_t)b * HW + p);
  const float n_p = rnd<T>(1.f - m_p), n_r = rnd<T>(1.f - m_r);
  const int c0 = chunk * kBlendChunk, c1 = min(C, c0 + kBlendChunk);
  int64_t i = ((int64_t)b * C + c0) * HW + p;
  float s_p = 0.f, s_r = 0.f;
  for (int c = c0; c < c1; ++c, i += HW) {
    const float gv = Num<T>::ld(g + i);
    if (g_out) g_out[i] = Num<T>::from(rnd<T>(gv * n_p) + rnd<T>(gv * n_r));
    if (g_ap) g_ap[i] = Num<T>::from(gv * m_p);
    if (g_ar) g_ar[i] = Num<T>::from(gv * m_r);
    if (g_mp || g_mr) {
      const float o = Num<T>::ld(out + i);
      if (g_mp) s_p += gv * (Num<T>::ld(ap + i) - o);
      if (g_mr) s_r += gv * (Num<T>::ld(ar + i) - o);
    }
  }
  if (g_mp) atomic_add(g_mp + (int64_t)b * HW + p, s_p);
  if (g_mr) atomic_add(g_mr + (int64_t)b * HW + p, s_r);
}

template <typename T>
static int mask_blend_fwd(const T *out, const T *ap, const T *ar, const T *mp, const T *mr, T *y, int64_t B, int64_t C,
                          int64_t HW, gfla_stream_t stream_) {
  if (!out || !ap || !ar || !mp || !mr || !y) return GFLA_ERR_NULL_POINTER;
  if (B < 0 || C <= 0 || HW <= 0) return GFLA_ERR_BAD_SHAPE;
  if (B == 0) return GFLA_OK;
  const int64_t nchunk = ceil_div(C, kBlendChunk);
  if (B > 65535 || nchunk > 65535 || HW > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)ceil_div(HW, 256), (unsigned)nchunk, (unsigned)B);
  mask_blend_fwd_kernel<T><<<grid, 256, 0, static_cast<hipStream_t>(stream_)>>>(out, ap, ar, mp, mr, y, (int)C, (int)HW,
                                                                               (int)nchunk);
  return launch_status();
}

template <typename T>
static int mask_blend_bwd(const T *out, const T *ap, const T *ar, const T *mp, const T *mr, const T *g, T *g_out, T *g_ap,
                          T *g_ar, float *g_mp, float *g_mr, int64_t B, int64_t C, int64_t HW, gfla_stream_t stream_) {
  if (!out || !ap || !ar || !mp || !mr || !g) return GFLA_ERR_NULL_POINTER;
  if (B < 0 || C <= 0 || HW <= 0) return GFLA_ERR_BAD_SHAPE;
  if (B == 0 || (!g_out && !g_ap && !g_ar && !g_mp && !g_mr)) return GFLA_OK;
  const int64_t nchunk = ceil_div(C, kBlendChunk);
  if (B > 65535 || nchunk > 65535 || HW > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)ceil_div(HW, 256), (unsigned)nchunk, (unsigned)B);
  mask_blend_bwd_kernel<T><<<grid, 256, 0, static_cast<hipStream_t>(stream_)>>>(out, ap, ar, mp, mr, g, g_out, g_ap, g_ar,
                                                                               g_mp, g_mr, (int)C, (int)HW);
  return launch_status();
}

}  // namespace gfla

using gfla::bf16_t;

extern "C" {
int gfla_mask_blend_fwd_f32(const float *out, const float *ap, const float *ar, const float *mp, const float *mr, float *y,
                            int64_t B, int64_t C, int64_t HW, gfla_stream_t st) {
  return gfla::mask_blend_fwd<float>(out, ap, ar, mp, mr, y, B, C, HW, st);
}
int gfla_mask_blend_fwd_bf16(const uint16_t *out, const uint16_t *ap, const uint16_t *ar, const uint16_t *mp,
                             const uint16_t *mr, uint16_t *y, int64_t B, int64_t C, int64_t HW, gfla_stream_t st) {
  auto c = [](const uint16_t *p) { return reinterpret_cast<const bf16_t *>(p); };
  return gfla::mask_blend_fwd<bf16_t>(c(out), c(ap), c(ar), c(mp), c(mr), reinterpret_cast<bf16_t *>(y), B, C, HW, st);
}
int gfla_mask_blend_bwd_f32(const float *out, const float *ap, const float *ar, const float *mp, const float *mr,
                            const float *g, float *g_out, float *g_ap, float *g_ar, float *g_mp, float *g_mr, int64_t B,
                            int64_t C, int64_t HW, gfla_stream_t st) {
  return gfla::mask_blend_bwd<float>(out, ap, ar, mp, mr, g, g_out, g_ap, g_ar, g_mp, g_mr, B, C, HW, st);
}
int gfla_mask_blend_bwd_bf16(const uint16_t *out, const uint16_t *ap, const uint16_t *ar, const uint16_t *mp,
                             const uint16_t *mr, const uint16_t *g, uint16_t *g_out, uint16_t *g_ap, uint16_t *g_ar,
                             float *g_mp, float *g_mr, int64_t B, int64_t C, int64_t HW, gfla_stream_t st) {
  auto c = [](const uint16_t *p) { return reinterpret_cast<const bf16_t *>(p); };
  auto m = [](uint16_t *p) { return reinterpret_cast<bf16_t *>(p); };
  return gfla::mask_blend_bwd<bf16_t>(c(out), c(ap), c(ar), c(mp), c(mr), c(g), m(g_out), m(g_ap), m(g_ar), g_mp, g_mr, B, C,
                                      HW, st);
}
}
