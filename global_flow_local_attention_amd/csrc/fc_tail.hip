// Tail of ExtractorAttn's fully_connect_layer, gfx950: nonlinearity + 1x1 convolution.
//
// Reference (base_function.py:799-803): fully_connect_layer = [Conv2d(2C, 128, k, stride k), nonlinearity,
// Conv2d(128, k*k, 1), softmax].  The first convolution arrives here as two halves in two layouts -- the
// source half out of one GEMM, channel-outermost (128, B, H, W); the target half out of a stride-1 convolution
// of the padded target, (B, 128, H, W) -- and torch would add them (strided), apply the LeakyReLU, run the
// 1x1 convolution through a convolution library and, backwards, transpose the gradient back for the GEMM.
// One pass each way instead:
//   forward : logits[b,q,p] = b1[q] + sum_o W1[q,o] * lrelu(hs[b,o,p] + ht[b,o,p] + b0[o])
//   backward: g_pre = (W1^T g_logits) * lrelu'(pre), written ONCE PER LAYOUT (the GEMM's and the
//             convolution's), plus the activations (for dW1, a tiny batched GEMM left to the caller) and
//             per-workgroup partial sums of the two bias gradients.
// A lane owns one position p of one sample and a quarter of the hidden channels: all global accesses are
// coalesced over p, W1 is read from LDS as broadcast reads, the k*k accumulators live in registers.  HBM-bound: forward reads
// 2 x (B,128,HW) and writes (B,k*k,HW); backward reads the same + g_logits and writes up to 3 x (B,128,HW).
#include "gfla_common.h"

namespace gfla {

constexpr int kTailThreads = 256;
constexpr int kTailPix = 64;                          // positions per workgroup: one per lane
constexpr int kTailSlices = kTailThreads / kTailPix;  // the hidden channels are dealt to the 4 waves

template <typename A>
__device__ __forceinline__ A lrelu(A v, A slope) { return v > 0 ? v : v * slope; }

// sum of v over the 64 lanes of a wave (all lanes must take part)
template <typename A>
__device__ __forceinline__ A wave_sum(A v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// Work decomposition of both kernels: a workgroup = 64 positions of one sample x 4 waves; wave s handles the
// hidden channels o = s, s+4, ... (so four times as many waves are in flight as with one thread per position,
// which matters because every channel costs two dependent global loads).
template <typename T, int KK>
__global__ __launch_bounds__(kTailThreads) void fc_tail_fwd_kernel(const T *__restrict__ hs, int64_t hs_sb,
                                                                  int64_t hs_so, const T *__restrict__ ht,
                                                                  const T *__restrict__ b0, const T *__restrict__ w1,
                                                                  const T *__restrict__ b1, T *__restrict__ logits,
                                                                  int Hc, int HW, typename Num<T>::acc slope) {
  using A = typename Num<T>::acc;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  A *w_s = reinterpret_cast<A *>(gfla_smem);  // [Hc][KK] (o-major: the KK weights of one hidden channel together)
  A *b0_s = w_s + (size_t)Hc * KK;            // [Hc]
  A *red = w_s;                               // [kTailSlices][KK][kTailPix] partial logits, once w_s is dead
  for (int i = threadIdx.x; i < Hc * KK; i += kTailThreads) {
    const int o = i / KK, q = i - o * KK;
    w_s[i] = Num<T>::ld(w1 + (int64_t)q * Hc + o);
  }
  for (int o = threadIdx.x; o < Hc; o += kTailThreads) b0_s[o] = b0 ? Num<T>::ld(b0 + o) : (A)0;
  __syncthreads();
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int p = blockIdx.x * kTailPix + lane;
  const int64_t b = blockIdx.y;
  const int pc = min(p, HW - 1);
  A acc[KK];
#pragma unroll
  for (int q = 0; q < KK; ++q) acc[q] = 0;
  const T *hs_p = hs + b * hs_sb + pc;
  const T *ht_p = ht + b * (int64_t)Hc * HW + pc;
#pragma unroll 8
  for (int o = slice; o < Hc; o += kTailSlices) {
    const A a = lrelu<A>(Num<T>::ld(hs_p + (int64_t)o * hs_so) + Num<T>::ld(ht_p + (int64_t)o * HW) + b0_s[o], slope);
    const A *w = w_s + o * KK;
#pragma unroll
    for (int q = 0; q < KK; ++q) acc[q] = fma(w[q], a, acc[q]);
  }
  __syncthreads();  // every wave is done with the weights: their LDS is reused for the partial sums
#pragma unroll
  for (int q = 0; q < KK; ++q) red[(slice * KK + q) * kTailPix + lane] = acc[q];
  __syncthreads();
  T *lg = logits + b * (int64_t)KK * HW;
  for (int i = threadIdx.x; i < KK * kTailPix; i += kTailThreads) {
    const int q = i / kTailPix, l = i - q * kTailPix;
    const int pp = blockIdx.x * kTailPix + l;
    if (pp >= HW) continue;
    A v = b1 ? Num<T>::ld(b1 + q) : (A)0;
#pragma unroll
    for (int s = 0; s < kTailSlices; ++s) v += red[(s * KK + q) * kTailPix + l];
    lg[(int64_t)q * HW + pp] = Num<T>::from(v);
  }
}

// bias_partials: (B * tiles, Hc + KK), one row per workgroup -- its sums of g_pre over its positions for the
// Hc hidden channels, then of g_logits for the KK outputs.  The caller adds the rows up; atomics onto
// Hc + KK addresses from every wave of the launch would serialise in L2 instead.
template <typename T, int KK>
__global__ __launch_bounds__(kTailThreads) void fc_tail_bwd_kernel(
    const T *__restrict__ hs, int64_t hs_sb, int64_t hs_so, const T *__restrict__ ht, const T *__restrict__ b0,
    const T *__restrict__ w1, const T *__restrict__ g_logits, T *__restrict__ g_hs, T *__restrict__ g_ht,
    T *__restrict__ act, T *__restrict__ bias_partials, int Hc, int HW, typename Num<T>::acc slope) {
  using A = typename Num<T>::acc;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  A *w_s = reinterpret_cast<A *>(gfla_smem);  // [Hc][KK]
  A *b0_s = w_s + (size_t)Hc * KK;            // [Hc]
  for (int i = threadIdx.x; i < Hc * KK; i += kTailThreads) {
    const int o = i / KK, q = i - o * KK;
    w_s[i] = Num<T>::ld(w1 + (int64_t)q * Hc + o);
  }
  for (int o = threadIdx.x; o < Hc; o += kTailThreads) b0_s[o] = b0 ? Num<T>::ld(b0 + o) : (A)0;
  __syncthreads();
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int p = blockIdx.x * kTailPix + lane;
  const int64_t b = blockIdx.y;
  const bool live = p < HW;       // dead lanes still take part in the wave reductions (with zeros)
  const int pc = live ? p : 0;
  T *row = bias_partials ? bias_partials + (b * gridDim.x + blockIdx.x) * (int64_t)(Hc + KK) : nullptr;
  A gl[KK];
  const T *gl_p = g_logits + b * (int64_t)KK * HW + pc;
#pragma unroll
  for (int q = 0; q < KK; ++q) gl[q] = live ? Num<T>::ld(gl_p + (int64_t)q * HW) : (A)0;
  if (row && slice == 0) {
#pragma unroll
    for (int q = 0; q < KK; ++q) {
      const A s = wave_sum<A>(gl[q]);
      if (lane == 0) row[Hc + q] = Num<T>::from(s);
    }
  }
  const T *hs_p = hs + b * hs_sb + pc;
  const T *ht_p = ht + b * (int64_t)Hc * HW + pc;
  for (int o = slice; o < Hc; o += kTailSlices) {
    const A pre = Num<T>::ld(hs_p + (int64_t)o * hs_so) + Num<T>::ld(ht_p + (int64_t)o * HW) + b0_s[o];
    const A *w = w_s + o * KK;
    A ga = 0;
#pragma unroll
    for (int q = 0; q < KK; ++q) ga = fma(w[q], gl[q], ga);
    const A gp = live ? (pre > 0 ? ga : ga * slope) : (A)0;
    if (live) {
      g_hs[b * hs_sb + (int64_t)o * hs_so + p] = Num<T>::from(gp);
      if (g_ht) g_ht[(b * Hc + o) * (int64_t)HW + p] = Num<T>::from(gp);
      if (act) act[(b * Hc + o) * (int64_t)HW + p] = Num<T>::from(lrelu<A>(pre, slope));
    }
    if (row) {
      const A s = wave_sum<A>(gp);
      if (lane == 0) row[o] = Num<T>::from(s);
    }
  }
}

static int tail_check(int64_t B, int64_t Hc, int64_t HW, int KK) {
  if (B < 0 || Hc <= 0 || HW < 0 || KK <= 0) return GFLA_ERR_BAD_SHAPE;
  if (KK != 1 && KK != 4 && KK != 9 && KK != 16 && KK != 25) return GFLA_ERR_UNSUPPORTED;
  if (B > 65535 || HW > 0x7fffff00LL || Hc > 4096) return GFLA_ERR_UNSUPPORTED;
  return GFLA_OK;
}

#define GFLA_KK_SWITCH(KKV, ...)                              \
  switch (KKV) {                                              \
    case 1: { constexpr int KK = 1; __VA_ARGS__; } break;     \
    case 4: { constexpr int KK = 4; __VA_ARGS__; } break;     \
    case 9: { constexpr int KK = 9; __VA_ARGS__; } break;     \
    case 16: { constexpr int KK = 16; __VA_ARGS__; } break;   \
    default: { constexpr int KK = 25; __VA_ARGS__; } break;   \
  }

template <typename T>
static int fc_tail_fwd(const T *hs, int64_t hs_sb, int64_t hs_so, const T *ht, const T *b0, const T *w1, const T *b1,
                       T *logits, int64_t B, int64_t Hc, int64_t HW, int KK, double slope, gfla_stream_t stream) {
  using A = typename Num<T>::acc;
  if (!hs || !ht || !w1 || !logits) return GFLA_ERR_NULL_POINTER;
  if (int rc = tail_check(B, Hc, HW, KK)) return rc;
  if (B == 0 || HW == 0) return GFLA_OK;
  const dim3 grid((unsigned)ceil_div(HW, kTailPix), (unsigned)B);
  const int64_t words = Hc * KK + Hc > kTailSlices * KK * kTailPix ? Hc * KK + Hc : kTailSlices * KK * kTailPix;
  const unsigned lds = (unsigned)(words * sizeof(A));
  if (lds > 64 * 1024) return GFLA_ERR_UNSUPPORTED;
  GFLA_KK_SWITCH(KK, fc_tail_fwd_kernel<T, KK><<<grid, kTailThreads, lds, static_cast<hipStream_t>(stream)>>>(
                         hs, hs_sb, hs_so, ht, b0, w1, b1, logits, (int)Hc, (int)HW, (A)slope));
  return launch_status();
}

template <typename T>
static int fc_tail_bwd(const T *hs, int64_t hs_sb, int64_t hs_so, const T *ht, const T *b0, const T *w1,
                       const T *g_logits, T *g_hs, T *g_ht, T *act, T *bias_partials, int64_t B, int64_t Hc,
                       int64_t HW, int KK, double slope, gfla_stream_t stream) {
  using A = typename Num<T>::acc;
  if (!hs || !ht || !w1 || !g_logits || !g_hs) return GFLA_ERR_NULL_POINTER;
  if (int rc = tail_check(B, Hc, HW, KK)) return rc;
  if (B == 0 || HW == 0) return GFLA_OK;
  const dim3 grid((unsigned)ceil_div(HW, kTailPix), (unsigned)B);
  const unsigned lds = (unsigned)((Hc * KK + Hc) * sizeof(A));
  if (lds > 64 * 1024) return GFLA_ERR_UNSUPPORTED;
  GFLA_KK_SWITCH(KK, fc_tail_bwd_kernel<T, KK><<<grid, kTailThreads, lds, static_cast<hipStream_t>(stream)>>>(
                         hs, hs_sb, hs_so, ht, b0, w1, g_logits, g_hs, g_ht, act, bias_partials, (int)Hc, (int)HW,
                         (A)slope));
  return launch_status();
}

}  // namespace gfla

extern "C" {
#define GFLA_DEF_FC_TAIL(SFX, T)                                                                                      \
  int gfla_fc_tail_fwd_##SFX(const T *hs, int64_t hs_sb, int64_t hs_so, const T *ht, const T *b0, const T *w1,        \
                             const T *b1, T *logits, int64_t B, int64_t Hc, int64_t HW, int KK, double slope,          \
                             gfla_stream_t stream) {                                                                  \
    return gfla::fc_tail_fwd<T>(hs, hs_sb, hs_so, ht, b0, w1, b1, logits, B, Hc, HW, KK, slope, stream);              \
  }                                                                                                                   \
  int gfla_fc_tail_bwd_##SFX(const T *hs, int64_t hs_sb, int64_t hs_so, const T *ht, const T *b0, const T *w1,        \
                             const T *g_logits, T *g_hs, T *g_ht, T *act, T *bias_partials, int64_t B,                 \
                             int64_t Hc, int64_t HW, int KK, double slope, gfla_stream_t stream) {                     \
    return gfla::fc_tail_bwd<T>(hs, hs_sb, hs_so, ht, b0, w1, g_logits, g_hs, g_ht, act, bias_partials, B, Hc, HW,    \
                                KK, slope, stream);                                                                   \
  }
GFLA_DEF_FC_TAIL(f32, float)
GFLA_DEF_FC_TAIL(f64, double)
#undef GFLA_DEF_FC_TAIL
}
