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

// ---- storage-type conversion of up to four tensors in ONE launch -------------------------------------------------------
// The bf16 feature path of ExtractorAttn (extractor_attn.py: FusedAttnBf16Function) widens three operands on the way in and
// narrows three gradients on the way out, per call; at the face model's batch each conversion is a launch-sized kernel and
// the step is bound as much by the host's launch rate as by the GPU (tools/probe_face_host.py).  blockIdx.y = job.
struct ConvertJob {
  const void *src;
  void *dst;
  int64_t n;
};
struct ConvertJobs {
  ConvertJob j[4];
};
template <bool TO_BF16>
__global__ __launch_bounds__(256) void convert_multi_kernel(ConvertJobs jobs) {
  const ConvertJob &J = jobs.j[blockIdx.y];
  const int64_t n = J.n, stride = (int64_t)gridDim.x * 256;
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(J.src) | reinterpret_cast<uintptr_t>(J.dst)) & 15) == 0 ? n >> 2 : 0;
  if constexpr (TO_BF16) {   // float32 -> bfloat16, round to nearest even (torch's conversion)
    const float4 *s4 = static_cast<const float4 *>(J.src);
    uint2 *d4 = static_cast<uint2 *>(J.dst);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const float4 v = s4[i];
      d4[i] = make_uint2((uint32_t)Num<bf16_t>::pack(v.x) | (uint32_t)Num<bf16_t>::pack(v.y) << 16,
                         (uint32_t)Num<bf16_t>::pack(v.z) | (uint32_t)Num<bf16_t>::pack(v.w) << 16);
    }
    const float *s = static_cast<const float *>(J.src);
    uint16_t *d = static_cast<uint16_t *>(J.dst);
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d[i] = Num<bf16_t>::pack(s[i]);
  } else {                   // bfloat16 -> float32 (exact)
    const uint2 *s4 = static_cast<const uint2 *>(J.src);
    float4 *d4 = static_cast<float4 *>(J.dst);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      const uint2 v = s4[i];
      d4[i] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                          __uint_as_float(v.y & 0xffff0000u));
    }
    const uint16_t *s = static_cast<const uint16_t *>(J.src);
    float *d = static_cast<float *>(J.dst);
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
      d[i] = __uint_as_float((uint32_t)s[i] << 16);
  }
}

}  // namespace gfla

using gfla::bf16_t;

extern "C" {
/* dst_i[0..n_i) = convert(src_i[0..n_i)) for up to four tensors (unused jobs: n = 0); to_bf16 = 0: bfloat16 -> float32,
 * 1: float32 -> bfloat16 (round to nearest even). */
int gfla_convert_multi(const void *s0, void *d0, int64_t n0, const void *s1, void *d1, int64_t n1, const void *s2, void *d2,
                       int64_t n2, const void *s3, void *d3, int64_t n3, int to_bf16, gfla_stream_t st) {
  gfla::ConvertJobs jobs;
  const void *s[4] = {s0, s1, s2, s3};
  void *d[4] = {d0, d1, d2, d3};
  const int64_t n[4] = {n0, n1, n2, n3};
  int nj = 0;
  int64_t most = 0;
  for (int i = 0; i < 4; ++i) {
    if (n[i] < 0) return GFLA_ERR_BAD_SHAPE;
    if (n[i] == 0) continue;
    if (!s[i] || !d[i]) return GFLA_ERR_NULL_POINTER;
    jobs.j[nj++] = gfla::ConvertJob{s[i], d[i], n[i]};
    if (n[i] > most) most = n[i];
  }
  if (nj == 0) return GFLA_OK;
  for (int i = nj; i < 4; ++i) jobs.j[i] = gfla::ConvertJob{nullptr, nullptr, 0};
  int64_t blocks = gfla::ceil_div(most, 256 * 8);
  if (blocks > 4 * gfla::kNumCU) blocks = 4 * gfla::kNumCU;
  const dim3 grid((unsigned)blocks, (unsigned)nj);
  if (to_bf16)
    gfla::convert_multi_kernel<true><<<grid, 256, 0, static_cast<hipStream_t>(st)>>>(jobs);
  else
    gfla::convert_multi_kernel<false><<<grid, 256, 0, static_cast<hipStream_t>(st)>>>(jobs);
  return gfla::launch_status();
}

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
