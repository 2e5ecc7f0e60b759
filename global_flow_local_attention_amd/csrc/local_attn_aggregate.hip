// Fused local-attention tail of ExtractorAttn for gfx950.
//
// Reference composition (base_function.py:803-809): Softmax(dim=1) over the k*k logits,
// LocalAttnReshape to (B,1,kH,kW), multiply with block_source = BlockExtractor(source, flow)
// (B,C,kH,kW) and avg_pool2d(k,k).  Algebraically
//     out[b,c,y,x] = (1/k^2) * sum_{i,j} a_ij(b,y,x) * bilinear(source[b,c], tap_ij)
// where tap_ij is the block_extractor sample position (block_extractor_kernel.cu:57-76).  This
// file evaluates that sum directly from `source`, so neither block_source nor the product nor
// the reshaped attention map is ever written to HBM.
//
// One lane owns one flow pixel: it loads the k*k logits, does the softmax serially in registers
// (k*k <= 25 values; no cross-lane traffic needed), folds the k*k attention weights with the
// bilinear weights into a (k+1)x(k+1) coefficient patch (all taps of one pixel share the same
// fractional offset, so they read a dense patch), and then walks a chunk of channels doing
// (k+1)^2 loads + FMAs each.
#include "gfla_common.h"

namespace gfla {

template <typename A>
__device__ __forceinline__ A exp_t(A v);
template <>
__device__ __forceinline__ float exp_t<float>(float v) { return expf(v); }
template <>
__device__ __forceinline__ double exp_t<double>(double v) { return exp(v); }

// Per-pixel tap geometry shared by forward and backward.
template <typename A, int K>
struct PatchTaps {
  int xL[K], xR[K], yT[K], yB[K];  // clamped; yT/yB pre-multiplied by Ws
  A ax[K], ay[K];                  // un-clamped fractional parts
  int x0, y0;                      // floor of tap 0
  bool dense;                      // floor(tap t) == floor(tap 0) + t for every t

  __device__ __forceinline__ void init(A fx0, A fy0, int xf, int yf, int Hs, int Ws) {
    dense = true;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const A dx = (fx0 + (A)(t - K / 2)) + (A)xf;  // block_extractor_kernel.cu:62-67
      const A dy = (fy0 + (A)(t - K / 2)) + (A)yf;
      const A fdx = floor_t<A>(dx), fdy = floor_t<A>(dy);
      const int ix = (int)fdx, iy = (int)fdy;
      if (t == 0) {
        x0 = ix;
        y0 = iy;
      }
      dense = dense && (ix == x0 + t) && (iy == y0 + t);
      xL[t] = clampi(ix, 0, Ws - 1);
      xR[t] = clampi((int)(fdx + 1), 0, Ws - 1);
      yT[t] = clampi(iy, 0, Hs - 1) * Ws;
      yB[t] = clampi((int)(fdy + 1), 0, Hs - 1) * Ws;
      ax[t] = dx - fdx;
      ay[t] = dy - fdy;
    }
  }
};

template <typename T, int K>
__global__ __launch_bounds__(kBlock) void agg_fwd_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, const T *__restrict__ logits,
    T *__restrict__ out, T *__restrict__ attn_out, int C, int Hs, int Ws, int H, int W,
    int apply_softmax, int cpt, int ncg, int sp_blocks) {
  using A = typename Num<T>::acc;
  constexpr int KK = K * K;
  const int sp_blk = blockIdx.x % sp_blocks;
  const int bc = blockIdx.x / sp_blocks;
  const int p = sp_blk * kBlock + threadIdx.x;
  if (p >= H * W) return;
  const int b = bc / ncg, cg = bc - b * ncg;
  const int yf = p / W, xf = p - yf * W;
  const int64_t HW = (int64_t)H * W;

  // ---- softmax over the k*k logits of this pixel (base_function.py:803) -------------------
  A a[KK];
  const T *lg = logits + (int64_t)b * KK * HW + p;
#pragma unroll
  for (int t = 0; t < KK; ++t) a[t] = Num<T>::ld(lg + t * HW);
  if (apply_softmax) {
    A m = a[0];
#pragma unroll
    for (int t = 1; t < KK; ++t) m = fmax(m, a[t]);
    A s = 0;
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      a[t] = exp_t<A>(a[t] - m);
      s += a[t];
    }
    const A inv = (A)1 / s;
#pragma unroll
    for (int t = 0; t < KK; ++t) a[t] *= inv;
  }
  if (attn_out && cg == 0) {
    T *ao = attn_out + (int64_t)b * KK * HW + p;
#pragma unroll
    for (int t = 0; t < KK; ++t) ao[t * HW] = Num<T>::from(a[t]);
  }

  PatchTaps<A, K> tp;
  tp.init(Num<T>::ld(flow + ((int64_t)(b * 2 + 0) * H + yf) * W + xf),
          Num<T>::ld(flow + ((int64_t)(b * 2 + 1) * H + yf) * W + xf), xf, yf, Hs, Ws);

  const int c0 = cg * cpt, c1 = min(C, c0 + cpt);
  const int64_t plane_sz = (int64_t)Hs * Ws;
  const T *plane = src + ((int64_t)b * C + c0) * plane_sz;
  T *o = out + ((int64_t)b * C + c0) * HW + p;
  const A inv_kk = (A)1 / (A)KK;

  if (tp.dense) {
    // fold attention and bilinear weights into a (K+1)x(K+1) patch of coefficients
    A P[K + 1][K + 1];
#pragma unroll
    for (int r = 0; r <= K; ++r)
#pragma unroll
      for (int s = 0; s <= K; ++s) P[r][s] = 0;
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const A w = a[i * K + j];
        const A xL_P = 1 - tp.ax[j], xR_P = tp.ax[j], yT_P = 1 - tp.ay[i], yB_P = tp.ay[i];
        P[i][j] += w * (xL_P * yT_P);
        P[i][j + 1] += w * (xR_P * yT_P);
        P[i + 1][j] += w * (xL_P * yB_P);
        P[i + 1][j + 1] += w * (xR_P * yB_P);
      }
    int col[K + 1], row[K + 1];
#pragma unroll
    for (int s = 0; s <= K; ++s) {
      col[s] = clampi(tp.x0 + s, 0, Ws - 1);
      row[s] = clampi(tp.y0 + s, 0, Hs - 1) * Ws;
    }
    for (int c = c0; c < c1; ++c) {
      A acc = 0;
#pragma unroll
      for (int r = 0; r <= K; ++r)
#pragma unroll
        for (int s = 0; s <= K; ++s) acc += P[r][s] * Num<T>::ld(plane + row[r] + col[s]);
      *o = Num<T>::from(acc * inv_kk);
      plane += plane_sz;
      o += HW;
    }
  } else {
    // floor() of some tap landed one off the dense patch (a flow value within rounding of an
    // integer): evaluate tap by tap exactly as block_extractor does.
    for (int c = c0; c < c1; ++c) {
      A acc = 0;
#pragma unroll
      for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const A xL_P = 1 - tp.ax[j], xR_P = tp.ax[j], yT_P = 1 - tp.ay[i], yB_P = tp.ay[i];
          A s = (xL_P * yT_P) * Num<T>::ld(plane + tp.yT[i] + tp.xL[j]);
          s += (xR_P * yT_P) * Num<T>::ld(plane + tp.yT[i] + tp.xR[j]);
          s += (xL_P * yB_P) * Num<T>::ld(plane + tp.yB[i] + tp.xL[j]);
          s += (xR_P * yB_P) * Num<T>::ld(plane + tp.yB[i] + tp.xR[j]);
          acc += a[i * K + j] * s;
        }
      *o = Num<T>::from(acc * inv_kk);
      plane += plane_sz;
      o += HW;
    }
  }
}

// Backward of the fused tail.  attn = post-softmax weights saved by forward.
//   d out_c / d a_ij   = bs_ij,c / k^2
//   d out_c / d bs_ij,c = a_ij / k^2        -> block_extractor backward of (a_ij * g_c / k^2)
// and, with softmax,  d/d logit_ij = a_ij * (ga_ij - sum_mn a_mn ga_mn)  (linear in ga, so channel
// chunks combine by addition).
template <typename T, int K>
__global__ __launch_bounds__(kBlock) void agg_bwd_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, const T *__restrict__ attn,
    const T *__restrict__ gout, T *__restrict__ gsrc, T *__restrict__ gflow, T *__restrict__ glogits,
    int C, int Hs, int Ws, int H, int W, int apply_softmax, int cpt, int ncg, int sp_blocks) {
  using A = typename Num<T>::acc;
  constexpr int KK = K * K;
  const int sp_blk = blockIdx.x % sp_blocks;
  const int bc = blockIdx.x / sp_blocks;
  const int p = sp_blk * kBlock + threadIdx.x;
  if (p >= H * W) return;
  const int b = bc / ncg, cg = bc - b * ncg;
  const int yf = p / W, xf = p - yf * W;
  const int64_t HW = (int64_t)H * W;

  A a[KK], ga[KK];
  const T *at = attn + (int64_t)b * KK * HW + p;
#pragma unroll
  for (int t = 0; t < KK; ++t) {
    a[t] = Num<T>::ld(at + t * HW);
    ga[t] = 0;
  }
  PatchTaps<A, K> tp;
  tp.init(Num<T>::ld(flow + ((int64_t)(b * 2 + 0) * H + yf) * W + xf),
          Num<T>::ld(flow + ((int64_t)(b * 2 + 1) * H + yf) * W + xf), xf, yf, Hs, Ws);

  const int c0 = cg * cpt, c1 = min(C, c0 + cpt);
  const int64_t plane_sz = (int64_t)Hs * Ws;
  const T *plane = src + ((int64_t)b * C + c0) * plane_sz;
  T *gplane = gsrc ? gsrc + ((int64_t)b * C + c0) * plane_sz : nullptr;
  const T *g = gout + ((int64_t)b * C + c0) * HW + p;
  const A inv_kk = (A)1 / (A)KK;
  A gx_acc = 0, gy_acc = 0;
  for (int c = c0; c < c1; ++c) {
    const A go = Num<T>::ld(g) * inv_kk;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const A yT_P = 1 - tp.ay[i], yB_P = tp.ay[i];
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const A xL_P = 1 - tp.ax[j], xR_P = tp.ax[j];
        const A vTL = Num<T>::ld(plane + tp.yT[i] + tp.xL[j]);
        const A vTR = Num<T>::ld(plane + tp.yT[i] + tp.xR[j]);
        const A vBL = Num<T>::ld(plane + tp.yB[i] + tp.xL[j]);
        const A vBR = Num<T>::ld(plane + tp.yB[i] + tp.xR[j]);
        A bs = (xL_P * yT_P) * vTL;
        bs += (xR_P * yT_P) * vTR;
        bs += (xL_P * yB_P) * vBL;
        bs += (xR_P * yB_P) * vBR;
        ga[i * K + j] += go * bs;
        const A gb = go * a[i * K + j];  // gradient reaching block_source[b,c,yf*K+i,xf*K+j]
        gy_acc += gb * (-xL_P * vTL - xR_P * vTR + xL_P * vBL + xR_P * vBR);
        gx_acc += gb * (-yT_P * vTL - yB_P * vBL + yT_P * vTR + yB_P * vBR);
        if (gplane) {
          atomic_add(gplane + tp.yT[i] + tp.xL[j], gb * xL_P * yT_P);
          atomic_add(gplane + tp.yT[i] + tp.xR[j], gb * xR_P * yT_P);
          atomic_add(gplane + tp.yB[i] + tp.xL[j], gb * xL_P * yB_P);
          atomic_add(gplane + tp.yB[i] + tp.xR[j], gb * xR_P * yB_P);
        }
      }
    }
    plane += plane_sz;
    if (gplane) gplane += plane_sz;
    g += HW;
  }
  if (gflow) {
    atomic_add(gflow + ((int64_t)(b * 2 + 0) * H + yf) * W + xf, gx_acc);
    atomic_add(gflow + ((int64_t)(b * 2 + 1) * H + yf) * W + xf, gy_acc);
  }
  if (glogits) {
    T *gl = glogits + (int64_t)b * KK * HW + p;
    A dot = 0;
    if (apply_softmax) {
#pragma unroll
      for (int t = 0; t < KK; ++t) dot += a[t] * ga[t];
    }
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      const A v = apply_softmax ? a[t] * (ga[t] - dot) : ga[t];
      atomic_add(gl + t * HW, v);
    }
  }
}

struct AggGeo {
  int cpt, ncg, sp_blocks;
  int64_t blocks;
};
static AggGeo agg_geometry(int64_t B, int64_t C, int64_t H, int64_t W, int max_cpt, int64_t want_waves) {
  AggGeo g;
  g.sp_blocks = (int)ceil_div(H * W, kBlock);
  int cpt = tuning(1) > 0 ? tuning(1)
                          : pick_channels_per_thread((int64_t)g.sp_blocks * kBlock, C, B, max_cpt, want_waves);
  if (cpt > C) cpt = (int)C;
  g.cpt = cpt;
  g.ncg = (int)ceil_div(C, cpt);
  g.blocks = (int64_t)g.sp_blocks * g.ncg * B;
  return g;
}

static int agg_check(int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H, int64_t W, int k) {
  if (B <= 0 || C <= 0 || Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0 || k < 1) return GFLA_ERR_BAD_SHAPE;
  if (k > 5) return GFLA_ERR_UNSUPPORTED;  // fused path is instantiated for k = 1..5
  if (Hs * Ws > 0x7fffffffLL || H * W > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  return GFLA_OK;
}

#define GFLA_K_SWITCH(KV, ...)                     \
  switch (KV) {                                    \
    case 1: { constexpr int K = 1; __VA_ARGS__; } break;  \
    case 2: { constexpr int K = 2; __VA_ARGS__; } break;  \
    case 3: { constexpr int K = 3; __VA_ARGS__; } break;  \
    case 4: { constexpr int K = 4; __VA_ARGS__; } break;  \
    default: { constexpr int K = 5; __VA_ARGS__; } break; \
  }

template <typename T>
static int aggregate_fwd(const T *src, const T *flow, const T *logits, T *out, T *attn_out, int64_t B,
                         int64_t C, int64_t Hs, int64_t Ws, int64_t H, int64_t W, int k, int sm,
                         gfla_stream_t stream_) {
  if (!src || !flow || !logits || !out) return GFLA_ERR_NULL_POINTER;
  int st = agg_check(B, C, Hs, Ws, H, W, k);
  if (st != GFLA_OK) return st;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  AggGeo g = agg_geometry(B, C, H, W, 32, 4 * kNumCU * kWavesPerCU);
  if (g.blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  GFLA_K_SWITCH(k, agg_fwd_kernel<T, K><<<dim3((unsigned)g.blocks), dim3(kBlock), 0, stream>>>(src, flow, logits, out, attn_out, (int)C, (int)Hs, (int)Ws, (int)H, (int)W, sm, g.cpt, g.ncg, g.sp_blocks));
  return launch_status();
}

template <typename T>
static int aggregate_bwd(const T *src, const T *flow, const T *attn, const T *gout, T *gsrc, T *gflow,
                         T *glogits, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H, int64_t W,
                         int k, int sm, gfla_stream_t stream_) {
  if (!src || !flow || !attn || !gout) return GFLA_ERR_NULL_POINTER;
  int st = agg_check(B, C, Hs, Ws, H, W, k);
  if (st != GFLA_OK) return st;
  if (!gsrc && !gflow && !glogits) return GFLA_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  AggGeo g = agg_geometry(B, C, H, W, 32, 2 * kNumCU * kWavesPerCU);
  if (g.blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  GFLA_K_SWITCH(k, agg_bwd_kernel<T, K><<<dim3((unsigned)g.blocks), dim3(kBlock), 0, stream>>>(src, flow, attn, gout, gsrc, gflow, glogits, (int)C, (int)Hs, (int)Ws, (int)H, (int)W, sm, g.cpt, g.ncg, g.sp_blocks));
  return launch_status();
}

}  // namespace gfla

using gfla::bf16_t;

extern "C" {
int gfla_local_attn_aggregate_fwd_f32(const float *s, const float *f, const float *l, float *o, float *a,
                                      int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H, int64_t W,
                                      int k, int sm, gfla_stream_t st) {
  return gfla::aggregate_fwd<float>(s, f, l, o, a, B, C, Hs, Ws, H, W, k, sm, st);
}
int gfla_local_attn_aggregate_fwd_f64(const double *s, const double *f, const double *l, double *o,
                                      double *a, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H,
                                      int64_t W, int k, int sm, gfla_stream_t st) {
  return gfla::aggregate_fwd<double>(s, f, l, o, a, B, C, Hs, Ws, H, W, k, sm, st);
}
int gfla_local_attn_aggregate_fwd_bf16(const uint16_t *s, const uint16_t *f, const uint16_t *l,
                                       uint16_t *o, uint16_t *a, int64_t B, int64_t C, int64_t Hs,
                                       int64_t Ws, int64_t H, int64_t W, int k, int sm, gfla_stream_t st) {
  return gfla::aggregate_fwd<bf16_t>(reinterpret_cast<const bf16_t *>(s), reinterpret_cast<const bf16_t *>(f),
                                     reinterpret_cast<const bf16_t *>(l), reinterpret_cast<bf16_t *>(o),
                                     reinterpret_cast<bf16_t *>(a), B, C, Hs, Ws, H, W, k, sm, st);
}
int gfla_local_attn_aggregate_bwd_f32(const float *s, const float *f, const float *a, const float *go,
                                      float *gs, float *gf, float *gl, int64_t B, int64_t C, int64_t Hs,
                                      int64_t Ws, int64_t H, int64_t W, int k, int sm, gfla_stream_t st) {
  return gfla::aggregate_bwd<float>(s, f, a, go, gs, gf, gl, B, C, Hs, Ws, H, W, k, sm, st);
}
int gfla_local_attn_aggregate_bwd_f64(const double *s, const double *f, const double *a, const double *go,
                                      double *gs, double *gf, double *gl, int64_t B, int64_t C,
                                      int64_t Hs, int64_t Ws, int64_t H, int64_t W, int k, int sm,
                                      gfla_stream_t st) {
  return gfla::aggregate_bwd<double>(s, f, a, go, gs, gf, gl, B, C, Hs, Ws, H, W, k, sm, st);
}
}
