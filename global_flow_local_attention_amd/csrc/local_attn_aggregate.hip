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
#include <type_traits>

#include "gfla_common.h"
#include "be_bwd_lds.h"
#include "patch_mfma.h"

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

// ----------------------------------------------------------------------------------------------
// LDS-plane variants: workgroup <-> (b, group of G channels[, 1/split of the pixels]).  The G source
// planes are staged in LDS once; every lane then evaluates its pixels for the G channels with
// ds_read_b32 gathers ((K+1)^2 per output in the dense case).  Backward additionally keeps G gradient
// planes in LDS (ds_add_f32) and flushes them once.
// ----------------------------------------------------------------------------------------------
constexpr int kAggChunk = 4;  // channels whose accumulators one lane keeps in registers at a time

template <typename T, int K>
__global__ __launch_bounds__(kLdsThreads) void agg_fwd_lds_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, const T *__restrict__ logits,
    T *__restrict__ out, T *__restrict__ attn_out, int C, int Hs, int Ws, int H, int W,
    int apply_softmax, int G, int ngroups, int split) {
  using A = typename Num<T>::acc;
  constexpr int KK = K * K;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  A *planes = reinterpret_cast<A *>(gfla_smem);
  int bid = blockIdx.x;
  const int sp = bid % split;
  bid /= split;
  const int g = bid % ngroups;
  const int b = bid / ngroups;
  const int c0 = g * G;
  const int gc = min(G, C - c0);
  const int plane_sz = Hs * Ws;
  stage_planes<T, A>(src + ((int64_t)b * C + c0) * plane_sz, planes, gc * plane_sz);
  __syncthreads();
  const int HW = H * W;
  const int per = (HW + split - 1) / split;
  const int p_end = min(HW, (sp + 1) * per);
  const A inv_kk = (A)1 / (A)KK;
  for (int p = sp * per + threadIdx.x; p < p_end; p += blockDim.x) {
    const int yf = p / W, xf = p - yf * W;
    A a[KK];
    const T *lg = logits + (int64_t)b * KK * HW + p;
#pragma unroll
    for (int t = 0; t < KK; ++t) a[t] = Num<T>::ld(lg + (int64_t)t * HW);
    A sm_max = 0, sm_inv = 1;
    if (apply_softmax) {  // base_function.py:803
      sm_max = a[0];
#pragma unroll
      for (int t = 1; t < KK; ++t) sm_max = fmax(sm_max, a[t]);
      A ssum = 0;
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        a[t] = exp_t<A>(a[t] - sm_max);
        ssum += a[t];
      }
      sm_inv = (A)1 / ssum;
#pragma unroll
      for (int t = 0; t < KK; ++t) a[t] *= sm_inv;
    }
    if (attn_out && g == 0) {
      T *ao = attn_out + (int64_t)b * KK * HW + p;
#pragma unroll
      for (int t = 0; t < KK; ++t) ao[(int64_t)t * HW] = Num<T>::from(a[t]);
    }
    PatchTaps<A, K> tp;
    tp.init(Num<T>::ld(flow + (int64_t)(b * 2 + 0) * HW + p), Num<T>::ld(flow + (int64_t)(b * 2 + 1) * HW + p), xf,
            yf, Hs, Ws);
    T *o = out + ((int64_t)b * C + c0) * HW + p;
    if (tp.dense) {
      // Patch rows outermost, channels innermost: only ONE row of the (K+1)x(K+1) coefficient patch
      // and kAggChunk channel accumulators are live, instead of the whole patch.
      int col[K + 1];
#pragma unroll
      for (int q = 0; q <= K; ++q) col[q] = clampi(tp.x0 + q, 0, Ws - 1);
      for (int cb = 0; cb < gc; cb += kAggChunk) {
        A acc[kAggChunk];
#pragma unroll
        for (int c = 0; c < kAggChunk; ++c) acc[c] = 0;
#pragma unroll
        for (int r = 0; r <= K; ++r) {
          A Pr[K + 1];
#pragma unroll
          for (int q = 0; q <= K; ++q) Pr[q] = 0;
#pragma unroll
          for (int j = 0; j < K; ++j) {
            const A xL_P = 1 - tp.ax[j], xR_P = tp.ax[j];
            A wrow = 0;  // attention mass of tap column j that lands on patch row r
            if (r < K) wrow += a[r * K + j] * (1 - tp.ay[r]);          // tap row r, top weight
            if (r > 0) wrow += a[(r - 1) * K + j] * tp.ay[r - 1];       // tap row r-1, bottom weight
            Pr[j] += wrow * xL_P;
            Pr[j + 1] += wrow * xR_P;
          }
          const int rowoff = clampi(tp.y0 + r, 0, Hs - 1) * Ws;
#pragma unroll
          for (int c = 0; c < kAggChunk; ++c) {
            const A *pl = planes + (size_t)min(cb + c, gc - 1) * plane_sz + rowoff;
            A v = 0;
#pragma unroll
            for (int q = 0; q <= K; ++q) v += Pr[q] * pl[col[q]];
            acc[c] += v;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < kAggChunk; ++c)
          if (cb + c < gc) o[(int64_t)(cb + c) * HW] = Num<T>::from(acc[c] * inv_kk);
      }
    } else {
      // Some tap's floor() landed one off the dense patch (a flow value within rounding of an
      // integer).  Rare: evaluate tap by tap exactly as block_extractor does, with rolled loops that
      // re-derive a_ij from the logits so this path costs no registers.
      const A *pl = planes;
      const A fx0 = Num<T>::ld(flow + (int64_t)(b * 2 + 0) * HW + p);
      const A fy0 = Num<T>::ld(flow + (int64_t)(b * 2 + 1) * HW + p);
      for (int c = 0; c < gc; ++c) {
        A acc = 0;
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
          const A dy = (fy0 + (A)(i - K / 2)) + (A)yf;
          const A fdy = floor_t<A>(dy);
          const int yT = clampi((int)fdy, 0, Hs - 1) * Ws, yB = clampi((int)(fdy + 1), 0, Hs - 1) * Ws;
          const A yB_P = dy - fdy, yT_P = 1 - yB_P;
#pragma unroll 1
          for (int j = 0; j < K; ++j) {
            const A dx = (fx0 + (A)(j - K / 2)) + (A)xf;
            const A fdx = floor_t<A>(dx);
            const int xL = clampi((int)fdx, 0, Ws - 1), xR = clampi((int)(fdx + 1), 0, Ws - 1);
            const A xR_P = dx - fdx, xL_P = 1 - xR_P;
            A aij = Num<T>::ld(lg + (int64_t)(i * K + j) * HW);
            if (apply_softmax) aij = exp_t<A>(aij - sm_max) * sm_inv;
            A v = (xL_P * yT_P) * pl[yT + xL];
            v += (xR_P * yT_P) * pl[yT + xR];
            v += (xL_P * yB_P) * pl[yB + xL];
            v += (xR_P * yB_P) * pl[yB + xR];
            acc += aij * v;
          }
        }
        *o = Num<T>::from(acc * inv_kk);
        pl += plane_sz;
        o += HW;
      }
    }
  }
}

// d/d a_ij (the attention gradient before the softmax Jacobian):
//     ga[b,ij,p] = (1/K^2) * sum_c grad_out[b,c,p] * block_source_ij[b,c,p]
// workgroup <-> (b, channel super-group, tile of blockDim pixels); lane <-> ONE pixel, K*K register
// accumulators.  The workgroup walks its channels in sub-groups of G planes staged in LDS; the
// (K+1)x(K+1) source patch of the pixel is read row by row (ds_read_b32) and reused by the four taps
// that share each element.  One coalesced atomic per (ij, super-group) publishes the sums; the
// softmax Jacobian needs the totals over ALL channels and is applied by agg_softmax_bwd_kernel.
template <typename T, int K>
__global__ __launch_bounds__(512) void agg_ga_lds_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, const T *__restrict__ gout,
    typename Num<T>::acc *__restrict__ glogits, const T *__restrict__ attn, typename Num<T>::acc *__restrict__ gflow,
    int C, int Hs, int Ws, int H, int W, int G, int nsuper, int CS, int ntiles, int total) {
  using A = typename Num<T>::acc;
  constexpr int KK = K * K;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  A *planes = reinterpret_cast<A *>(gfla_smem);
  // The pixel tiles of one (sample, super-group) stage the SAME planes.  Workgroups are dealt round-robin to
  // the 8 XCDs, so give every XCD a contiguous run of the (b, sg, tile) index space: the tiles that share
  // planes then share one L2 instead of fetching them from HBM once per XCD.
  const int per_xcd = (total + kNumXCD - 1) / kNumXCD;
  int bid = (blockIdx.x % kNumXCD) * per_xcd + blockIdx.x / kNumXCD;
  if (bid >= total) return;
  const int tile = bid % ntiles;
  bid /= ntiles;
  const int sg = bid % nsuper;
  const int b = bid / nsuper;
  const int HW = H * W;
  const int plane_sz = Hs * Ws;
  const int p = tile * blockDim.x + threadIdx.x;
  const bool active = p < HW;
  const int yf = active ? p / W : 0, xf = active ? p - yf * W : 0;
  const int pc = active ? p : 0;
  const A fx0 = Num<T>::ld(flow + (int64_t)(b * 2 + 0) * HW + pc);
  const A fy0 = Num<T>::ld(flow + (int64_t)(b * 2 + 1) * HW + pc);
  PatchTaps<A, K> tp;
  tp.init(fx0, fy0, xf, yf, Hs, Ws);
  // Dense case: every tap (i, j) mixes the four patch elements (i..i+1, j..j+1) with weights that do not
  // depend on the channel, so  ga_ij = sum_ab w_ab(i,j) * P[i+a][j+b]  with  P[r][q] = sum_c g_c v_c[r][q].
  // Only the (K+1)^2 sums P are accumulated per channel ((K+1)^2 FMAs instead of 5 K^2); the 4-term mix
  // happens once per pixel at the end.
  int col[K + 1], rowoff[K + 1];
#pragma unroll
  for (int q = 0; q <= K; ++q) {
    col[q] = clampi(tp.x0 + q, 0, Ws - 1);
    rowoff[q] = clampi(tp.y0 + q, 0, Hs - 1) * Ws;
  }
  A P[(K + 1) * (K + 1)];
#pragma unroll
  for (int t = 0; t < (K + 1) * (K + 1); ++t) P[t] = 0;
  const A inv_kk = (A)1 / (A)KK;
  const int cs0 = sg * CS;
  const int cs1 = min(C, cs0 + CS);
  A *gl = glogits ? glogits + (int64_t)b * KK * HW + pc : nullptr;
  // d/d flow (block_extractor_kernel.cu:163-164) is linear in the same patch sums P, weighted by the attention:
  // computed here when asked for (attn, gflow non-NULL), one atomic pair per pixel and channel super-group
  const T *at = attn ? attn + (int64_t)b * KK * HW + pc : nullptr;
  A gx_acc = 0, gy_acc = 0;
  for (int cb = cs0; cb < cs1; cb += G) {
    const int gc = min(G, cs1 - cb);
    __syncthreads();  // previous sub-group fully consumed
    stage_planes<T, A>(src + ((int64_t)b * C + cb) * plane_sz, planes, gc * plane_sz);
    __syncthreads();
    if (!active) continue;
    const T *go_p = gout + ((int64_t)b * C + cb) * HW + p;
    if (tp.dense) {
      for (int c = 0; c < gc; ++c) {
        const A go = Num<T>::ld(go_p + (int64_t)c * HW) * inv_kk;
        const A *pl = planes + (size_t)c * plane_sz;
#pragma unroll
        for (int r = 0; r <= K; ++r)
#pragma unroll
          for (int q = 0; q <= K; ++q) P[r * (K + 1) + q] += go * pl[rowoff[r] + col[q]];
      }
    } else {
      // rare (a tap within rounding of an integer): tap by tap, published directly
      for (int c = 0; c < gc; ++c) {
        const A go = Num<T>::ld(go_p + (int64_t)c * HW) * inv_kk;
        const A *pl = planes + (size_t)c * plane_sz;
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
          const A dy = (fy0 + (A)(i - K / 2)) + (A)yf;
          const A fdy = floor_t<A>(dy);
          const int yT = clampi((int)fdy, 0, Hs - 1) * Ws, yB = clampi((int)(fdy + 1), 0, Hs - 1) * Ws;
          const A yB_P = dy - fdy, yT_P = 1 - yB_P;
#pragma unroll 1
          for (int j = 0; j < K; ++j) {
            const A dx = (fx0 + (A)(j - K / 2)) + (A)xf;
            const A fdx = floor_t<A>(dx);
            const int xL = clampi((int)fdx, 0, Ws - 1), xR = clampi((int)(fdx + 1), 0, Ws - 1);
            const A xR_P = dx - fdx, xL_P = 1 - xR_P;
            const A vTL = pl[yT + xL], vTR = pl[yT + xR], vBL = pl[yB + xL], vBR = pl[yB + xR];
            A bs = (xL_P * yT_P) * vTL;
            bs += (xR_P * yT_P) * vTR;
            bs += (xL_P * yB_P) * vBL;
            bs += (xR_P * yB_P) * vBR;
            if (gl) atomic_add(gl + (int64_t)(i * K + j) * HW, go * bs);
            if (gflow) {
              const A gv = Num<T>::ld(at + (int64_t)(i * K + j) * HW) * go;
              gy_acc += gv * (-xL_P * vTL - xR_P * vTR + xL_P * vBL + xR_P * vBR);
              gx_acc += gv * (-yT_P * vTL - yB_P * vBL + yT_P * vTR + yB_P * vBR);
            }
          }
        }
      }
    }
  }
  if (active && tp.dense) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const A yT_P = 1 - tp.ay[i], yB_P = tp.ay[i];
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const A xL_P = 1 - tp.ax[j], xR_P = tp.ax[j];
        A ga = (xL_P * yT_P) * P[i * (K + 1) + j];  // block_extractor_kernel.cu:78-84
        ga += (xR_P * yT_P) * P[i * (K + 1) + j + 1];
        ga += (xL_P * yB_P) * P[(i + 1) * (K + 1) + j];
        ga += (xR_P * yB_P) * P[(i + 1) * (K + 1) + j + 1];
        if (gl) atomic_add(gl + (int64_t)(i * K + j) * HW, ga);
        if (gflow) {
          const A a_ij = Num<T>::ld(at + (int64_t)(i * K + j) * HW);
          const A pTL = P[i * (K + 1) + j], pTR = P[i * (K + 1) + j + 1];
          const A pBL = P[(i + 1) * (K + 1) + j], pBR = P[(i + 1) * (K + 1) + j + 1];
          gy_acc += a_ij * (-xL_P * pTL - xR_P * pTR + xL_P * pBL + xR_P * pBR);
          gx_acc += a_ij * (-yT_P * pTL - yB_P * pBL + yT_P * pTR + yB_P * pBR);
        }
      }
    }
  }
  if (active && gflow) {
    atomic_add(gflow + (int64_t)(b * 2 + 0) * HW + p, gx_acc);
    atomic_add(gflow + (int64_t)(b * 2 + 1) * HW + p, gy_acc);
  }
}

// In place: glogits holds ga (d/d a_ij); turn it into d/d logit_ij = a_ij * (ga_ij - sum_mn a_mn ga_mn).
template <typename T, int K>
__global__ __launch_bounds__(kBlock) void agg_softmax_bwd_kernel(const T *__restrict__ attn,
                                                                typename Num<T>::acc *__restrict__ glogits, int64_t n,
                                                                int HW) {
  using A = typename Num<T>::acc;
  constexpr int KK = K * K;
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;  // over (b, p)
  if (idx >= n) return;
  const int64_t b = idx / HW;
  const int p = (int)(idx - b * HW);
  const T *at = attn + b * KK * HW + p;
  A *gl = glogits + b * KK * HW + p;
  A a[KK], ga[KK];
  A dot = 0;
#pragma unroll
  for (int t = 0; t < KK; ++t) {
    a[t] = Num<T>::ld(at + (int64_t)t * HW);
    ga[t] = gl[(int64_t)t * HW];
    dot += a[t] * ga[t];
  }
#pragma unroll
  for (int t = 0; t < KK; ++t) gl[(int64_t)t * HW] = a[t] * (ga[t] - dot);
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
  using A = typename Num<T>::acc;
  if (tuning(3) != 1) {
    PlaneGeo pg = plane_geometry(Hs * Ws, sizeof(A), B, C, H * W, true, kAggChunk);
    if (pg.G > 0) {
      const int64_t blocks = B * pg.ngroups * pg.split;
      if (blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
      GFLA_K_SWITCH(k, launch_lds(agg_fwd_lds_kernel<T, K>, dim3((unsigned)blocks), dim3(kLdsThreads), pg.lds_bytes, stream, 
                           src, flow, logits, out, attn_out, (int)C, (int)Hs, (int)Ws, (int)H, (int)W, sm, pg.G, pg.ngroups, pg.split));
      return launch_status();
    }
  }
  AggGeo g = agg_geometry(B, C, H, W, 32, 4 * kNumCU * kWavesPerCU);
  if (g.blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  GFLA_K_SWITCH(k, agg_fwd_kernel<T, K><<<dim3((unsigned)g.blocks), dim3(kBlock), 0, stream>>>(src, flow, logits, out, attn_out, (int)C, (int)Hs, (int)Ws, (int)H, (int)W, sm, g.cpt, g.ncg, g.sp_blocks));
  return launch_status();
}

// (2) d/d a_ij [+ d/d flow when gflow != NULL], then (3) the softmax Jacobian in place
template <typename T>
static int launch_agg_ga(const T *src, const T *flow, const T *attn, const T *gout, typename Num<T>::acc *glogits,
                         typename Num<T>::acc *gflow, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H, int64_t W, int k, int sm, hipStream_t stream) {
  using A = typename Num<T>::acc;
  const int threads = 512;
  const int64_t ntiles = ceil_div(H * W, threads);
  int64_t G = kLdsBudget / (Hs * Ws * (int64_t)sizeof(A));
  if (G < 1) return GFLA_ERR_UNSUPPORTED;
  if (G > C) G = C;
  int64_t nsuper = 1;  // split the channels until the launch has >= 4 workgroups per CU
  while (B * ntiles * nsuper < 4 * kNumCU && nsuper * 2 * G <= C) nsuper *= 2;
  const int64_t CS = ceil_div(C, nsuper);
  nsuper = ceil_div(C, CS);
  if (G > CS) G = CS;
  const int64_t blocks = B * nsuper * ntiles;
  if (blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  const unsigned lds = (unsigned)(G * Hs * Ws * sizeof(A));
  const int64_t padded = ceil_div(blocks, kNumXCD) * kNumXCD;  // the XCD remap needs a multiple of 8
  GFLA_K_SWITCH(k, agg_ga_lds_kernel<T, K><<<dim3((unsigned)padded), dim3(threads), lds, stream>>>(
                       src, flow, gout, glogits, gflow ? attn : nullptr, gflow, (int)C, (int)Hs, (int)Ws, (int)H, (int)W,
                       (int)G, (int)nsuper, (int)CS, (int)ntiles, (int)blocks));
  int st = launch_status();
  if (st == GFLA_OK && sm && glogits) {
    const int64_t n = B * H * W;
    GFLA_K_SWITCH(k, agg_softmax_bwd_kernel<T, K><<<dim3((unsigned)ceil_div(n, kBlock)), dim3(kBlock), 0, stream>>>(
                         attn, glogits, n, (int)(H * W)));
    st = launch_status();
  }
  return st;
}

// All three outputs are ACCUMULATED into (the caller zeroes them, or passes partial gradients to add to).
// workspace (gfla_scatter_workspace_bytes, may be NULL): enables the matrix-core scatter for d/d source (f32).
template <typename T>
static int aggregate_bwd(const T *src, const T *flow, const T *attn, const T *gout, T *gsrc,
                         typename Num<T>::acc *gflow, typename Num<T>::acc *glogits, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H, int64_t W,
                         int k, int sm, gfla_stream_t stream_, void *workspace = nullptr) {
  if (!src || !flow || !attn || !gout) return GFLA_ERR_NULL_POINTER;
  int st = agg_check(B, C, Hs, Ws, H, W, k);
  if (st != GFLA_OK) return st;
  if (!gsrc && !gflow && !glogits) return GFLA_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  using A = typename Num<T>::acc;
  const bool planes_fit = Hs * Ws * (int64_t)sizeof(A) <= kLdsBudget;
  if constexpr (std::is_same<T, float>::value) {
    // d/d source as a block-sparse product on the matrix cores (patch_mfma.hip); d/d flow then comes out of the
    // d/d a_ij pass, which holds the patch sums it needs
    if (gsrc && workspace && tuning(3) != 1 && (planes_fit || (!gflow && !glogits))) {
      // the LDS-atomic kernel is the device-side fallback for flows that spread the patches too far
      const bool lds_fallback = Hs * Ws * (int64_t)(sizeof(lds_acc_t) + sizeof(A)) <= kLdsBudget;
      const unsigned *skip_stat = nullptr;
      unsigned skip_limit = 0;
      st = agg_source_bwd_mfma(flow, attn, gout, gsrc, workspace, B, C, Hs, Ws, H, W, k, 1, lds_fallback ? 1 : 0, &skip_stat,
                               &skip_limit, stream);
      if (st == GFLA_OK) {
        if (lds_fallback && skip_limit != 0xffffffffu) {
          bool done = false;
          GFLA_K_SWITCH(k, st = launch_be_bwd_lds<T, K>(kGoutAttn, src, flow, gout, attn, gsrc, (A *)nullptr, B, C, Hs, Ws, H, W,
                                                        stream, &done, 0, 0, (const T *)nullptr, skip_stat, skip_limit));
          if (st != GFLA_OK) return st;
          if (!done) return GFLA_ERR_UNSUPPORTED;
        }
        if (gflow || glogits) st = launch_agg_ga<T>(src, flow, attn, gout, glogits, gflow, B, C, Hs, Ws, H, W, k, sm, stream);
        return st;
      }
      if (st != GFLA_ERR_UNSUPPORTED) return st;
    }
  }
  if (tuning(3) != 1 && Hs * Ws * (int64_t)(sizeof(lds_acc_t) + sizeof(A)) <= kLdsBudget) {
    // (1) grad_source + grad_flow: block_extractor backward of the factored gradient a_ij*g_c/k^2
    if (gsrc || gflow) {
      bool done = false;
      GFLA_K_SWITCH(k, st = launch_be_bwd_lds<T, K>(kGoutAttn, src, flow, gout, attn, gsrc, gflow, B, C, Hs, Ws, H, W, stream, &done));
      if (st != GFLA_OK) return st;
      if (!done) return GFLA_ERR_UNSUPPORTED;
    }
    if (glogits) st = launch_agg_ga<T>(src, flow, attn, gout, glogits, (A *)nullptr, B, C, Hs, Ws, H, W, k, sm, stream);
    return st;
  }
  if constexpr (sizeof(T) == 2) {
    return GFLA_ERR_UNSUPPORTED;  // bf16 storage: the planes-in-LDS kernels only
  } else {
    AggGeo g = agg_geometry(B, C, H, W, 32, 2 * kNumCU * kWavesPerCU);
    if (g.blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
    GFLA_K_SWITCH(k, agg_bwd_kernel<T, K><<<dim3((unsigned)g.blocks), dim3(kBlock), 0, stream>>>(src, flow, attn, gout, gsrc, gflow, glogits, (int)C, (int)Hs, (int)Ws, (int)H, (int)W, sm, g.cpt, g.ncg, g.sp_blocks));
    return launch_status();
  }
}

// Backward of everything in ExtractorAttn that flows into block_source(source, flow): the gradient of
// the unfold-layout FC operand (grad_unfold, may be NULL) plus the attention-weighted aggregation's
// (attn, grad_out; may be NULL), scattered into grad_source / grad_flow in one pass.
template <typename T>
static int local_attn_source_bwd(const T *src, const T *flow, const T *gunf, const T *attn, const T *gout,
                                 T *gsrc, typename Num<T>::acc *gflow, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H,
                                 int64_t W, int k, int layout, gfla_stream_t stream_) {
  using A = typename Num<T>::acc;
  if (!src || !flow || (!gunf && !(attn && gout)) || ((attn == nullptr) != (gout == nullptr))) return GFLA_ERR_NULL_POINTER;
  int st = agg_check(B, C, Hs, Ws, H, W, k);
  if (st != GFLA_OK) return st;
  if (Hs * Ws * (int64_t)(sizeof(lds_acc_t) + sizeof(A)) > kLdsBudget) return GFLA_ERR_UNSUPPORTED;
  if (!gsrc && !gflow) return GFLA_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int64_t HW = H * W;
  const int64_t cs = layout == 1 ? B * HW : HW;
  const int64_t bs = layout == 1 ? HW : C * k * k * HW;
  const int mode = gunf ? (attn ? kGoutUnfoldAttn : kGoutUnfold) : kGoutAttn;
  bool done = false;
  GFLA_K_SWITCH(k, st = launch_be_bwd_lds<T, K>(mode, src, flow, gunf ? gunf : gout, attn, gsrc, gflow, B, C, Hs, Ws, H, W,
                                                stream, &done, cs, bs, gout));
  if (st == GFLA_OK && !done) st = GFLA_ERR_UNSUPPORTED;
  return st;
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
int gfla_local_attn_aggregate_bwd_ws_f32(const float *s, const float *f, const float *a, const float *go, float *gs,
                                         float *gf, float *gl, void *workspace, int64_t B, int64_t C, int64_t Hs,
                                         int64_t Ws, int64_t H, int64_t W, int k, int sm, gfla_stream_t st) {
  return gfla::aggregate_bwd<float>(s, f, a, go, gs, gf, gl, B, C, Hs, Ws, H, W, k, sm, st, workspace);
}
/* bf16 storage: grad_source bf16; grad_flow and grad_logits FLOAT32 (reductions over channels, accumulated across
 * workgroups) */
int gfla_local_attn_aggregate_bwd_bf16(const uint16_t *s, const uint16_t *f, const uint16_t *a, const uint16_t *go,
                                       uint16_t *gs, float *gf, float *gl, int64_t B, int64_t C, int64_t Hs,
                                       int64_t Ws, int64_t H, int64_t W, int k, int sm, gfla_stream_t st) {
  return gfla::aggregate_bwd<bf16_t>(reinterpret_cast<const bf16_t *>(s), reinterpret_cast<const bf16_t *>(f),
                                     reinterpret_cast<const bf16_t *>(a), reinterpret_cast<const bf16_t *>(go),
                                     reinterpret_cast<bf16_t *>(gs), gf, gl, B, C, Hs, Ws, H, W, k, sm, st);
}
int gfla_local_attn_aggregate_bwd_f64(const double *s, const double *f, const double *a, const double *go,
                                      double *gs, double *gf, double *gl, int64_t B, int64_t C,
                                      int64_t Hs, int64_t Ws, int64_t H, int64_t W, int k, int sm,
                                      gfla_stream_t st) {
  return gfla::aggregate_bwd<double>(s, f, a, go, gs, gf, gl, B, C, Hs, Ws, H, W, k, sm, st);
}
int gfla_local_attn_source_bwd_f32(const float *s, const float *f, const float *gu, const float *a, const float *go,
                                   float *gs, float *gf, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H,
                                   int64_t W, int k, int layout, gfla_stream_t st) {
  return gfla::local_attn_source_bwd<float>(s, f, gu, a, go, gs, gf, B, C, Hs, Ws, H, W, k, layout, st);
}
int gfla_local_attn_source_bwd_bf16(const uint16_t *s, const uint16_t *f, const uint16_t *gu, const uint16_t *a,
                                    const uint16_t *go, uint16_t *gs, float *gf, int64_t B, int64_t C, int64_t Hs,
                                    int64_t Ws, int64_t H, int64_t W, int k, int layout, gfla_stream_t st) {
  return gfla::local_attn_source_bwd<bf16_t>(reinterpret_cast<const bf16_t *>(s), reinterpret_cast<const bf16_t *>(f),
                                             reinterpret_cast<const bf16_t *>(gu), reinterpret_cast<const bf16_t *>(a),
                                             reinterpret_cast<const bf16_t *>(go), reinterpret_cast<bf16_t *>(gs), gf, B, C,
                                             Hs, Ws, H, W, k, layout, st);
}
int gfla_local_attn_source_bwd_f64(const double *s, const double *f, const double *gu, const double *a,
                                   const double *go, double *gs, double *gf, int64_t B, int64_t C, int64_t Hs,
                                   int64_t Ws, int64_t H, int64_t W, int k, int layout, gfla_stream_t st) {
  return gfla::local_attn_source_bwd<double>(s, f, gu, a, go, gs, gf, B, C, Hs, Ws, H, W, k, layout, st);
}
}
