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
#include "rs_taps.h"

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

// ----------------------------------------------------------------------------------------------
// Forward through a per-pixel COEFFICIENT TABLE (the default for f32 / bf16 storage when the caller hands over a
// workspace).  What bounds agg_fwd_lds_kernel is neither HBM nor the LDS pipe but the per-pixel setup every channel
// group repeats -- softmax over k*k logits, tap geometry, folding the attention into the (K+1)x(K+1) patch
// coefficients: ~1100 VALU instructions per pixel against ~40 per pixel and channel, C/G times.  So:
//   1. agg_coef_kernel (once per pixel): softmax (-> attn_out), tap geometry, and the patch coefficients of the
//      pixel as (K+1) rows x (K+2) words of an EVEN-aligned window of source columns, with the replicate clamp in x
//      already folded in (coefficients of columns left / right of the map are added to the border column) -- a
//      dynamically indexed scatter, done in LDS.  Written as one record per pixel -- the coefficients + one packed word (window
//      start, first row) -- in the tile order of the main kernel; pixels whose taps do not form a dense patch (a flow within rounding of an integer) get
//      a sentinel and are evaluated tap by tap in the main kernel.
//   2. agg_fwd_stream_kernel (per sample, 1024 pixels and a range of channels): a lane loads the record of its pixel
//      ONCE (coalesced dwordx4) and keeps it in registers while the planes of its channels stream through LDS in
//      double-buffered chunks; a patch row is (K+3)/2 ds_read_b64 from an even word -- 256 B/clk against the 128 B/clk
//      of ds_read_b32, no per-column clamp -- consumed by v_pk_fma_f32 on the register pair a b64 read lands in.
// Non-finite source values: the window is one word wider than the patch and that word carries weight 0, so an
// inf / NaN there reaches this pixel (0 * inf), one column further than in the reference.
// ----------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned kAggNotDense = 0xffffffffu;  // packed word of a pixel whose taps are not a dense patch
constexpr unsigned kAggSkip = 0xfffffffeu;      // lane without a pixel (tile overhang)
constexpr int kAggCoefThreads = 256;

template <int K>
constexpr int agg_coef_slots() { return (K + 1) * (K + 2); }
// floats per pixel in the table: the coefficients + the packed word, padded to whole 16-byte vectors -- a lane fetches
// its record with a few dwordx4 loads (the texture addresser spends ~16 cycles per wave load whatever its width: one
// dword load per coefficient was the bottleneck of the first version of this kernel)
constexpr int agg_record_floats(int k) { return ((k + 1) * (k + 2) + 1 + 3) & ~3; }
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T, int K>
__global__ __launch_bounds__(kAggCoefThreads) void agg_coef_kernel(
    const T *__restrict__ flow, const T *__restrict__ logits, T *__restrict__ attn_out, float *__restrict__ table,
    int Hs, int Ws, int H, int W, int apply_softmax, int tw_log2, int ntile) {
  constexpr int KK = K * K, NS = agg_coef_slots<K>(), NR = agg_record_floats(K);
  __shared__ float coef[NS * kAggCoefThreads];  // [slot][thread]: a private, dynamically indexable column per lane
  const int HW = H * W;
  const int b = blockIdx.y;
  // wave <-> tile of the main kernel (same lane -> pixel mapping), record vectors written [tile][vector][lane]
  const int lane = threadIdx.x & 63, t = blockIdx.x * (kAggCoefThreads / 64) + (threadIdx.x >> 6);
  if (t >= ntile) return;
  const int tw = 1 << tw_log2, th = 64 >> tw_log2, tiles_x = (W + tw - 1) >> tw_log2;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int yf = ty * th + (lane >> tw_log2), xf = (tx << tw_log2) + (lane & (tw - 1));
  if (yf >= H || xf >= W) return;
  const int p = yf * W + xf;
  float a[KK];
  const T *lg = logits + (int64_t)b * KK * HW + p;
#pragma unroll
  for (int t = 0; t < KK; ++t) a[t] = Num<T>::ld(lg + (int64_t)t * HW);
  if (apply_softmax) {  // base_function.py:803
    float m = a[0];
#pragma unroll
    for (int t = 1; t < KK; ++t) m = fmaxf(m, a[t]);
    float ssum = 0;
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      a[t] = exp_t<float>(a[t] - m);
      ssum += a[t];
    }
    const float inv = 1.f / ssum;
#pragma unroll
    for (int t = 0; t < KK; ++t) a[t] *= inv;
  }
  if (attn_out) {
    T *ao = attn_out + (int64_t)b * KK * HW + p;
#pragma unroll
    for (int t = 0; t < KK; ++t) ao[(int64_t)t * HW] = Num<T>::from(a[t]);
  }
  PatchTaps<float, K> tp;
  tp.init(Num<T>::ld(flow + (int64_t)(b * 2 + 0) * HW + p), Num<T>::ld(flow + (int64_t)(b * 2 + 1) * HW + p), xf, yf,
          Hs, Ws);
  f32x4 *rec = reinterpret_cast<f32x4 *>(table) + ((int64_t)b * ntile + t) * (NR / 4) * 64 + lane;
  if (!tp.dense) {
    rec[(NS >> 2) * 64][NS & 3] = __uint_as_float(kAggNotDense);
    return;
  }
  // even-aligned window [xa, xa + K + 1] that holds every clamped column of the patch
  const int xs = clampi(tp.x0, 0, Ws - (K + 1));
  const int xa = xs & ~1;
  float *mine = coef + threadIdx.x;
#pragma unroll
  for (int s_ = 0; s_ < NS; ++s_) mine[s_ * kAggCoefThreads] = 0.f;
  int slot[K + 1];
#pragma unroll
  for (int q = 0; q <= K; ++q) slot[q] = (clampi(tp.x0 + q, 0, Ws - 1) - xa) * kAggCoefThreads;
#pragma unroll
  for (int r = 0; r <= K; ++r) {
    float Pr[K + 1];
#pragma unroll
    for (int q = 0; q <= K; ++q) Pr[q] = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      float wrow = 0;  // attention mass of tap column j that lands on patch row r
      if (r < K) wrow += a[r * K + j] * (1 - tp.ay[r]);
      if (r > 0) wrow += a[(r - 1) * K + j] * tp.ay[r - 1];
      Pr[j] += wrow * (1 - tp.ax[j]);
      Pr[j + 1] += wrow * tp.ax[j];
    }
    float *row = mine + r * (K + 2) * kAggCoefThreads;
#pragma unroll
    for (int q = 0; q <= K; ++q) row[slot[q]] += Pr[q];  // same lane, program order: no race
  }
  const int y0 = clampi(tp.y0, -(K + 1), Hs);  // rows are clamped to [0, Hs) anyway
  const unsigned packed = ((unsigned)(y0 + 16) << 16) | (unsigned)xa;
#pragma unroll
  for (int v = 0; v < NR / 4; ++v) {
    f32x4 q;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int s_ = 4 * v + e;
      q[e] = s_ < NS ? mine[s_ * kAggCoefThreads] : (s_ == NS ? __uint_as_float(packed) : 0.f);
    }
    rec[v * 64] = q;
  }
}

// LDS layout of a plane: rows in INTERLEAVED PAIRS,
//     word(y, x) = (y / 2) * pitch + (x / 2) * 4 + (y % 2) * 2 + (x % 2),     pitch = 32 (mod 64), >= 2 * Ws
// A word pair (even x) stays contiguous (one ds_read_b64), rows y and y+1 of a pair never share a bank (they occupy
// alternate 8-byte slots), and consecutive row pairs are 32 banks apart: a lane group whose addresses span <= 8 word
// pairs of <= 4 plane rows -- a 16 x 2 or 8 x 4 pixel tile with a coherent flow -- reads without bank conflicts (zero
// flow: 4 % conflict cycles; the bench's flow, which moves ~1 pixel per pixel, still loses 40 %, down from 60 % for
// row-major planes read by 64 pixels of one row).
// A chunk of planes is moved in two halves so that the global loads of the NEXT chunk are in flight while the current
// one is being read: agg_chunk_load (global -> registers, word pairs) ... agg_chunk_store (registers -> LDS).
constexpr int kAggPre = 8;  // word pairs per thread and chunk (the launcher sizes the chunk accordingly)

template <typename T>
__device__ __forceinline__ void agg_chunk_load(const T *__restrict__ g, int n, f32x2 (&pre)[kAggPre]) {
#pragma unroll
  for (int j = 0; j < kAggPre; ++j) {
    // unconditional (index clamped): a load under `if (i < n)` turns into a branch + s_waitcnt per load
    const int i = min(j * (int)blockDim.x + (int)threadIdx.x, n - 1);
    if constexpr (sizeof(T) == 4) {
      pre[j] = reinterpret_cast<const f32x2 *>(g)[i];
    } else {
      const unsigned raw = reinterpret_cast<const unsigned *>(g)[i];  // two bf16
      pre[j] = f32x2{__uint_as_float(raw << 16), __uint_as_float(raw & 0xffff0000u)};
    }
  }
}
// LDS word offsets of the thread's kAggPre word pairs inside a chunk buffer (the same for every chunk: computed once,
// two 16-bit offsets per register; the chunk buffer has < 2^16 words... in units of 2 words)
__device__ __forceinline__ void agg_chunk_offsets(int n_max, int per_plane, int wp, int pitch, int plane_sz,
                                                  unsigned (&off)[kAggPre / 2]) {
  const unsigned m_pl = 0xffffffffu / (unsigned)per_plane + 1u, m_wp = 0xffffffffu / (unsigned)wp + 1u;  // n * d < 2^32
#pragma unroll
  for (int j = 0; j < kAggPre; ++j) {
    const int i = min(j * (int)blockDim.x + (int)threadIdx.x, n_max - 1);
    const int c = (int)__umulhi((unsigned)i, m_pl);
    const int rem = i - c * per_plane;
    const int y = (int)__umulhi((unsigned)rem, m_wp);
    const int xp = rem - y * wp;
    const unsigned o = (unsigned)(c * plane_sz + (y >> 1) * pitch + (xp << 2) + ((y & 1) << 1)) >> 1;  // even word -> /2
    if (j & 1) off[j >> 1] |= o << 16; else off[j >> 1] = o;
  }
}
__device__ __forceinline__ void agg_chunk_store(float *lds, int n, const f32x2 (&pre)[kAggPre],
                                                const unsigned (&off)[kAggPre / 2]) {
#pragma unroll
  for (int j = 0; j < kAggPre; ++j) {
    const int i = j * (int)blockDim.x + (int)threadIdx.x;
    const unsigned o = (j & 1) ? off[j >> 1] >> 16 : off[j >> 1] & 0xffffu;
    if (i < n) *reinterpret_cast<f32x2 *>(lds + 2 * o) = pre[j];
  }
}

// workgroup <-> (sample, tile group = blockDim/64 tiles, channel range [sg*CS, sg*CS+CS)); wave <-> tile; lane <-> pixel.
// The lane's record is fetched ONCE and stays in registers while the channels of the range stream through LDS in
// chunks of CH planes, double buffered.
// <= 12 waves per workgroup: 170 VGPRs per lane (record 43 + two channels x two rows in flight 28 + next chunk 16 + ...)
//
// Two things the ISA of the first version showed (round 4; each cost an exposed round trip per step):
//   * a result store inside the channel loop: its address / data registers are overwritten by the next channel pair, so
//     hipcc waits for the store -- and the memory counter is in order: waiting for the YOUNGEST store waits for the next
//     chunk's prefetch loads as well (s_waitcnt vmcnt(0) at the top of the pair loop: the double buffering was synchronous).
//     Now the CHT results of a chunk stay in registers (CHT = planes per chunk, compile time) and are stored once per chunk,
//     behind the wait the staging needs anyway, from a uniform base + a per-lane offset (scalar-base store form);
//   * the last word of a row window is the low half of a 64-bit read whose high half is dead: hipcc reused that register
//     for the next address and had to wait for the read to land first (s_waitcnt lgkmcnt(0) in the middle of every burst of
//     reads).  The pair is kept alive until the row has been consumed.
constexpr int kAggStreamWaves = 12;
// timing ablations of agg_fwd_stream_kernel (tools/ubench/build_agg_abl.sh builds variant libraries with
// -DGFLA_AGG_ABL=bits; results are garbage): 1 = no LDS reads in the row loop, 2 = no prefetch / staging of the next chunk,
// 4 = no workgroup barrier in the chunk loop, 8 = no arithmetic on the rows
#ifndef GFLA_AGG_ABL
#define GFLA_AGG_ABL 0
#endif
constexpr int kAggAbl = GFLA_AGG_ABL;
constexpr int kAggMaxChunk = 8;   // planes per chunk the forward kernel is instantiated for (2, 4, 6, 8)
template <typename T, int K, int CHT>
__global__ __launch_bounds__(kAggStreamWaves * 64) void agg_fwd_stream_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, const T *__restrict__ logits,
    const float *__restrict__ table, T *__restrict__ out, int C, int Hs, int Ws, int H, int W, int apply_softmax,
    int CH, int CS, int nsuper, int tgroups, int pitch, int total, int tw_log2, int ntile) {
  constexpr int KK = K * K, NP = (K + 1) / 2, NS = agg_coef_slots<K>(), NR = agg_record_floats(K);
  static_assert(K % 2 == 1, "paired reads are laid out for odd K");
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  float *lds = reinterpret_cast<float *>(gfla_smem);
  // every XCD gets a contiguous run of the (sample, channel range, tile group) index space: the tile groups that
  // stage the same planes, and the channel ranges that read the same records, share one L2
  const int per_xcd = (total + kNumXCD - 1) / kNumXCD;
  int bid = (blockIdx.x % kNumXCD) * per_xcd + blockIdx.x / kNumXCD;
  if (bid >= total) return;
  const int tg = bid % tgroups;
  bid /= tgroups;
  const int sg = bid % nsuper;
  const int b = bid / nsuper;
  const int c_begin = sg * CS, c_end = min(C, c_begin + CS);
  const int plane_sz = ((Hs + 1) >> 1) * pitch;
  const int buf_sz = CH * plane_sz + 4;  // + the word pair a window may read past the last row
  const int HW = H * W;
  const float inv_kk = 1.f / (float)KK;
  const int tw = 1 << tw_log2, th = 64 >> tw_log2;
  const int tiles_x = (W + tw - 1) >> tw_log2;
  const int lane = threadIdx.x & 63;
  const int t = tg * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);

  // the lane's pixel: packed word + coefficients (w2[r][i] = words (2i, 2i+1) of the aligned row window, w1[r] = word K+1)
  f32x2 w2[K + 1][NP];
  float w1[K + 1];
  unsigned m = kAggSkip;
  int p = 0;
  if (t < ntile) {
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int yf = ty * th + (lane >> tw_log2), xf = (tx << tw_log2) + (lane & (tw - 1));
    if (yf < H && xf < W) {
      p = yf * W + xf;
      const f32x4 *rec = reinterpret_cast<const f32x4 *>(table) + ((int64_t)b * ntile + t) * (NR / 4) * 64 + lane;
      f32x4 q[NR / 4];
#pragma unroll
      for (int v = 0; v < NR / 4; ++v) q[v] = rec[v * 64];  // 1 KB per load instruction, fully coalesced
#pragma unroll
      for (int r = 0; r <= K; ++r) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          w2[r][i].x = q[(r * (K + 2) + 2 * i) >> 2][(r * (K + 2) + 2 * i) & 3];
          w2[r][i].y = q[(r * (K + 2) + 2 * i + 1) >> 2][(r * (K + 2) + 2 * i + 1) & 3];
        }
        w1[r] = q[(r * (K + 2) + K + 1) >> 2][(r * (K + 2) + K + 1) & 3];
      }
      m = __float_as_uint(q[NS >> 2][NS & 3]);
    }
  }

  const int wp = Ws >> 1;              // word pairs per row (Ws is even)
  const int per_plane = Hs * wp;       // word pairs per plane
  const T *s0 = src + ((int64_t)b * C + c_begin) * Hs * Ws;
  f32x2 pre[kAggPre];
  unsigned off[kAggPre / 2];
  agg_chunk_offsets(CH * per_plane, per_plane, wp, pitch, plane_sz, off);
  int gc = min(CH, c_end - c_begin);
  agg_chunk_load<T>(s0, gc * per_plane, pre);
  {  // words no load ever writes must be finite: they meet weight 0
    const int nrow = CH * ((Hs + 1) >> 1);
    for (int u = 0; u < 2; ++u) {
      float *bf = lds + u * buf_sz;
      if (pitch >= 2 * Ws + 4)
        for (int i = threadIdx.x; i < nrow * 4; i += blockDim.x) bf[(i >> 2) * pitch + 2 * Ws + (i & 3)] = 0.f;
      if (threadIdx.x < 4) bf[CH * plane_sz + threadIdx.x] = 0.f;
    }
  }
  agg_chunk_store(lds, gc * per_plane, pre, off);
  __syncthreads();

  int ro[K + 1];
  if (m < kAggSkip) {
    const int xa = (int)(m & 0xffffu), y0 = (int)(m >> 16) - 16;
#pragma unroll
    for (int r = 0; r <= K; ++r) {
      const int yc = clampi(y0 + r, 0, Hs - 1);
      ro[r] = (yc >> 1) * pitch + ((yc & 1) << 1) + (xa << 1);  // word pairs of a row sit 4 words apart
    }
  }
  T *ob = out + ((int64_t)b * C + c_begin) * HW;   // uniform: the stores take a scalar base + the lane's pixel offset
  __builtin_amdgcn_s_waitcnt(0x0F70);              // (record and first chunk: all landed; see the end of the loop body)
  int cur = 0;
  for (int cb = c_begin; cb < c_end; cb += CH) {
    const int gn = min(CH, c_end - cb - CH);  // planes of the next chunk (<= 0: none): in flight during this one
    // UNCONDITIONAL (the last chunk re-requests one word pair of its own first plane): with the loads under a branch hipcc
    // cannot count them, and every later wait for an older store becomes vmcnt(0) -- a wait for these loads
    if constexpr (!(kAggAbl & 2))
      agg_chunk_load<T>(s0 + (int64_t)(gn > 0 ? cb + CH - c_begin : 0) * Hs * Ws, gn > 0 ? gn * per_plane : 1, pre);
    gc = min(CH, c_end - cb);
    const float *pl = lds + cur * buf_sz;
    float res[CHT];
#pragma unroll
    for (int c = 0; c < CHT; ++c) res[c] = 0.f;
    if (m < kAggSkip) {
      // One patch row of one channel: NP + 1 ds_read_b64 (word K+1 as the first half of a 64-bit read: a ds_read_b32 is
      // banked mod 32 and conflicts on this layout).  The empty asm statements keep the reads apart: merged into
      // ds_read2_b64 they would run at half the LDS rate (MI355X_MICROARCH.md, LDS table).
      auto load_row = [&](const float *rp, f32x2(&v)[NP], f32x2 &u) {
        if constexpr (kAggAbl & 1) {
#pragma unroll
          for (int i = 0; i < NP; ++i) v[i] = f32x2{(float)lane, (float)(size_t)rp};
          u = f32x2{(float)lane, 1.f};
          return;
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          v[i] = *reinterpret_cast<const f32x2 *>(rp + 4 * i);
          asm volatile("" ::: "memory");
        }
        u = *reinterpret_cast<const f32x2 *>(rp + 4 * NP);
        asm volatile("" ::: "memory");
      };
#pragma unroll
      for (int c = 0; c < CHT; c += 2) {  // two channels at a time: independent accumulation chains
        // (a chunk shorter than CHT re-reads its last plane: finite values, results never stored)
        const float *p0_ = pl + min(c, gc - 1) * plane_sz, *p1_ = pl + min(c + 1, gc - 1) * plane_sz;
        f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
        float s0_ = 0.f, s1_ = 0.f;
        f32x2 v0[2][NP], v1[2][NP];
        f32x2 u0[2], u1[2];
        load_row(p0_ + ro[0], v0[0], u0[0]);
        load_row(p1_ + ro[0], v1[0], u1[0]);
#pragma unroll
        for (int r = 0; r <= K; ++r) {
          if (r < K) {  // next row in flight while this one is consumed
            load_row(p0_ + ro[r + 1], v0[(r + 1) & 1], u0[(r + 1) & 1]);
            load_row(p1_ + ro[r + 1], v1[(r + 1) & 1], u1[(r + 1) & 1]);
          }
          __builtin_amdgcn_sched_barrier(0);  // the next row's reads are issued before this row's arithmetic
          if constexpr (kAggAbl & 8) {
#pragma unroll
            for (int i = 0; i < NP; ++i) asm volatile("" ::"v"(v0[r & 1][i]), "v"(v1[r & 1][i]));
          } else {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
              acc0 = __builtin_elementwise_fma(w2[r][i], v0[r & 1][i], acc0);
              acc1 = __builtin_elementwise_fma(w2[r][i], v1[r & 1][i], acc1);
            }
            s0_ = fmaf(w1[r], u0[r & 1].x, s0_);
            s1_ = fmaf(w1[r], u1[r & 1].x, s1_);
          }
          // both halves of the last read stay allocated until here (see the note above the kernel)
          asm volatile("" ::"v"(u0[r & 1]), "v"(u1[r & 1]));
          __builtin_amdgcn_sched_barrier(0);
        }
        res[c] = (acc0.x + acc0.y + s0_) * inv_kk;
        if (c + 1 < CHT) res[c + 1] = (acc1.x + acc1.y + s1_) * inv_kk;
      }
    } else if (m == kAggNotDense) {
      // tap by tap exactly as block_extractor does; rolled loops that re-derive a_ij from the logits
      const int yf = p / W, xf = p - yf * W;
      const T *lg = logits + (int64_t)b * KK * HW + p;
      const float fx0 = Num<T>::ld(flow + (int64_t)(b * 2 + 0) * HW + p);
      const float fy0 = Num<T>::ld(flow + (int64_t)(b * 2 + 1) * HW + p);
      float sm_max = 0, sm_inv = 1;
      if (apply_softmax) {
        sm_max = Num<T>::ld(lg);
#pragma unroll 1
        for (int tt = 1; tt < KK; ++tt) sm_max = fmaxf(sm_max, Num<T>::ld(lg + (int64_t)tt * HW));
        float ssum = 0;
#pragma unroll 1
        for (int tt = 0; tt < KK; ++tt) ssum += exp_t<float>(Num<T>::ld(lg + (int64_t)tt * HW) - sm_max);
        sm_inv = 1.f / ssum;
      }
#pragma unroll
      for (int c = 0; c < CHT; ++c) {
        if (c >= gc) break;
        const float *plc = pl + c * plane_sz;
        float acc = 0;
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
          const float dy = (fy0 + (float)(i - K / 2)) + (float)yf;
          const float fdy = floorf(dy);
          const int yTc = clampi((int)fdy, 0, Hs - 1), yBc = clampi((int)(fdy + 1), 0, Hs - 1);
          const int yT = (yTc >> 1) * pitch + ((yTc & 1) << 1), yB = (yBc >> 1) * pitch + ((yBc & 1) << 1);
          const float yB_P = dy - fdy, yT_P = 1 - yB_P;
#pragma unroll 1
          for (int j = 0; j < K; ++j) {
            const float dx = (fx0 + (float)(j - K / 2)) + (float)xf;
            const float fdx = floorf(dx);
            const int xLc = clampi((int)fdx, 0, Ws - 1), xRc = clampi((int)(fdx + 1), 0, Ws - 1);
            const int xL = ((xLc >> 1) << 2) + (xLc & 1), xR = ((xRc >> 1) << 2) + (xRc & 1);
            const float xR_P = dx - fdx, xL_P = 1 - xR_P;
            float aij = Num<T>::ld(lg + (int64_t)(i * K + j) * HW);
            if (apply_softmax) aij = exp_t<float>(aij - sm_max) * sm_inv;
            float v = (xL_P * yT_P) * plc[yT + xL];
            v += (xR_P * yT_P) * plc[yT + xR];
            v += (xL_P * yB_P) * plc[yB + xL];
            v += (xR_P * yB_P) * plc[yB + xR];
            acc += aij * v;
          }
        }
        res[c] = acc * inv_kk;
      }
      // (rare branch, taken by a wave with a flow within rounding of an integer.)  Its loads have all been consumed; saying
      // so on every path out of it keeps them from turning the dense path's register reuse into waits for the prefetch
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    if constexpr (!(kAggAbl & 2)) cur ^= 1;
    if constexpr (!(kAggAbl & 2))
      if (gn > 0) agg_chunk_store(lds + cur * buf_sz, gn * per_plane, pre, off);
    // every memory load so far has landed (the staging above needed the prefetch; the tap-by-tap branch its own): said
    // unconditionally, so that nothing but the result stores below is pending when the loop comes round
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
    asm volatile("" ::: "memory");
    // the chunk's results, behind the wait the staging needed anyway; they fly during the next chunk
    if (m != kAggSkip) {
      T *oc = ob + (int64_t)(cb - c_begin) * HW;
#pragma unroll
      for (int c = 0; c < CHT; ++c)
        if (c < gc) oc[(int64_t)c * HW + p] = Num<T>::from(res[c]);
    }
    if constexpr (!(kAggAbl & 4))
      __syncthreads();  // the next chunk has landed, and nobody still reads the buffer the one after it will overwrite
  }
}

// Launch geometry of agg_fwd_stream_kernel.
struct AggStreamGeo {
  int CH, CS, nsuper, tgroups, threads, pitch, tw_log2, ntile;
  unsigned lds;
};
inline AggStreamGeo agg_stream_geometry(int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H, int64_t W, int k) {
  AggStreamGeo g{0, 0, 1, 1, 0, 0, 4, 0, 0};
  // tile width: the one that wastes the fewest lanes on the overhang (16 on ties); tuning key 16 overrides
  int twl = 4;
  double best_eff = -1;
  for (int cand : {4, 3, 5}) {
    const int64_t tw = 1 << cand, th = 64 >> cand;
    const double eff = (double)(H * W) / (double)(ceil_div(W, tw) * tw * ceil_div(H, th) * th);
    if (eff > best_eff + 1e-9) {
      best_eff = eff;
      twl = cand;
    }
  }
  if (tuning(16) == 8 || tuning(16) == 16 || tuning(16) == 32) twl = tuning(16) == 8 ? 3 : tuning(16) == 16 ? 4 : 5;
  const int64_t ntile = ceil_div(W, 1 << twl) * ceil_div(H, 64 >> twl);
  const int64_t tmax = tuning(9) >= 64 && tuning(9) <= kAggStreamWaves * 64 ? tuning(9) / 64 : kAggStreamWaves;  // tiles (waves) per workgroup
  const int64_t tgroups = ceil_div(ntile, tmax);
  const int64_t twg = ceil_div(ntile, tgroups);  // balanced
  const int64_t threads = twg * 64;
  int pitch = (int)(ceil_div(2 * Ws + 32, 64) * 64 - 32);  // smallest value >= 2 Ws that is 32 mod 64
  if (tuning(17) >= 2 * Ws && !(tuning(17) & 3)) pitch = tuning(17);  // experiment: pair pitch in words
  const int64_t per_plane = ceil_div(Hs, 2) * pitch * 4;
  const int64_t budget = 160 * 1024 - 64;
  // chunk: as many planes as two buffers fit and kAggPre word pairs per thread cover
  int64_t CH = std::min<int64_t>((budget / 2 - 16) / per_plane, kAggPre * threads / (Hs * (Ws / 2)));
  if (CH > C) CH = C;
  if (CH > kAggMaxChunk) CH = kAggMaxChunk;   // the forward kernel keeps a chunk's results in registers
  if (tuning(4) > 0 && tuning(4) < CH) CH = tuning(4);
  if (CH >= 2) CH &= ~1LL;  // channel pairs
  if (CH < 1) return g;
  // channel ranges: more of them = more workgroups, fewer chunks each (the first chunk of a workgroup is not overlapped)
  int64_t best_ns = 1;
  double best_cost = -1;
  for (int64_t ns = 1; ns * CH <= C || ns == 1; ns *= 2) {
    const int64_t CS = ceil_div(ceil_div(C, ns), CH) * CH;
    const int64_t nsr = ceil_div(C, CS);
    const int64_t wgs = B * tgroups * nsr;
    const double rounds = (double)ceil_div(wgs, kNumCU);
    const double cost = rounds * (1.3 + (double)(CS / CH));
    if (best_cost < 0 || cost < best_cost - 1e-9) {
      best_cost = cost;
      best_ns = ns;
    }
  }
  if (tuning(5) > 0) best_ns = tuning(5);
  const int64_t CS = ceil_div(ceil_div(C, best_ns), CH) * CH;
  g = AggStreamGeo{(int)CH, (int)CS, (int)ceil_div(C, CS), (int)tgroups, (int)threads, pitch, twl, (int)ntile,
                   (unsigned)(2 * (CH * per_plane + 16))};
  return g;
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
      // the next channel's upstream gradient is requested a channel ahead (raw: scaled where it is used), so the loop
      // never waits for a global round trip per channel
      A go_next = Num<T>::ld(go_p);
      for (int c = 0; c < gc; ++c) {
        const A go = go_next * inv_kk;
        if (c + 1 < gc) go_next = Num<T>::ld(go_p + (int64_t)(c + 1) * HW);
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

// d/d a_ij and d/d flow with the machinery of agg_fwd_stream_kernel (same workgroup <-> (sample, tile group, channel
// range) decomposition, same double-buffered interleaved planes, same paired reads): per channel the lane accumulates
// g_c * v_c over its (K+1) x (K+2)-word aligned WINDOW (48 packed-FMA accumulators for K = 5) instead of reading the
// (K+1)^2 patch with clamped per-column ds_read_b32; the window sums are mapped back to patch sums once per pixel at
// the end (an 8-way select per patch entry: the x clamp and the window's parity), then mixed into d/d a_ij and
// d/d flow exactly as agg_ga_lds_kernel does.  One atomic per (ij, channel range) publishes the sums.
// EPI = 1: the same accumulation for resample2d's d/d input2 (kernel_size 4, dilation 1: a K = 3 patch around
// floor(p + flow) - 1): `flow` is input2 (dx, dy, sigma), `gflow` its (B, 3, H, W) gradient, the epilogue is
// rs_bwd2_finish on the row / column sums of the patch sums (resample2d_kernel.cu:273-328); glogits / attn unused.
// CHT = planes per chunk (compile time, as in agg_fwd_stream_kernel): the chunk's CHT upstream-gradient values are
// requested one chunk AHEAD, in front of the plane prefetch -- the first version loaded them inside the channel-pair loop
// and waited for them at once (s_waitcnt vmcnt(0) per pair: a global round trip per pair, and the prefetch with it).
template <typename T, int K, int EPI, int CHT>
__global__ __launch_bounds__(kAggStreamWaves * 64) void agg_ga_stream_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, const T *__restrict__ gout, float *__restrict__ glogits,
    const T *__restrict__ attn, float *__restrict__ gflow, int C, int Hs, int Ws, int H, int W, int CH, int CS,
    int nsuper, int tgroups, int pitch, int total, int tw_log2, int ntile) {
  constexpr int KK = K * K, NP = (K + 1) / 2, NW = 2 * NP + 2;  // NW = words of the row window (K + 3)
  static_assert(K % 2 == 1, "paired reads are laid out for odd K");
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  float *lds = reinterpret_cast<float *>(gfla_smem);
  const int per_xcd = (total + kNumXCD - 1) / kNumXCD;
  int bid = (blockIdx.x % kNumXCD) * per_xcd + blockIdx.x / kNumXCD;
  if (bid >= total) return;
  const int tg = bid % tgroups;
  bid /= tgroups;
  const int sg = bid % nsuper;
  const int b = bid / nsuper;
  const int c_begin = sg * CS, c_end = min(C, c_begin + CS);
  const int plane_sz = ((Hs + 1) >> 1) * pitch;
  const int buf_sz = CH * plane_sz + 4;
  const int HW = H * W;
  const float inv_kk = EPI == 1 ? 1.f : 1.f / (float)KK;
  const int tw = 1 << tw_log2, th = 64 >> tw_log2;
  const int tiles_x = (W + tw - 1) >> tw_log2;
  const int lane = threadIdx.x & 63;
  const int t = tg * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
  bool active = false;
  int p = 0, yf = 0, xf = 0;
  if (t < ntile) {
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    yf = ty * th + (lane >> tw_log2);
    xf = (tx << tw_log2) + (lane & (tw - 1));
    active = yf < H && xf < W;
    if (active) p = yf * W + xf;
  }
  constexpr int NF = EPI == 1 ? 3 : 2;  // planes of `flow`
  const float fx0 = Num<T>::ld(flow + (int64_t)(b * NF + 0) * HW + p);
  const float fy0 = Num<T>::ld(flow + (int64_t)(b * NF + 1) * HW + p);
  PatchTaps<float, K> tp;
  Taps<float, 2> rt;
  int px0, py0;  // first column / row of the patch
  bool dense;
  if constexpr (EPI == 1) {
    static_assert(EPI == 0 || K == 3, "resample2d's 4 x 4 taps are a K = 3 patch");
    rt.template init<2>(fx0, fy0, Num<T>::ld(flow + (int64_t)(b * 3 + 2) * HW + p), xf, yf, Hs, Ws, 1, false);
    px0 = (int)fminf(fmaxf(floorf((float)xf + fx0), -1048576.f), 1048576.f) - 1;
    py0 = (int)fminf(fmaxf(floorf((float)yf + fy0), -1048576.f), 1048576.f) - 1;
    dense = active;
  } else {
    tp.init(fx0, fy0, xf, yf, Hs, Ws);
    px0 = tp.x0;
    py0 = tp.y0;
    dense = active && tp.dense;
  }
  // even-aligned window [xa, xa + K + 2] that holds every clamped column of the patch (as agg_coef_kernel)
  const int xa = clampi(px0, 0, Ws - (K + 1)) & ~1;
  int ro[K + 1];
#pragma unroll
  for (int r = 0; r <= K; ++r) {
    const int yc = clampi(py0 + r, 0, Hs - 1);
    ro[r] = (yc >> 1) * pitch + ((yc & 1) << 1) + (xa << 1);
  }
  f32x2 Pw[K + 1][NP + 1];  // window sums: words (2i, 2i+1) of patch row r
#pragma unroll
  for (int r = 0; r <= K; ++r)
#pragma unroll
    for (int i = 0; i <= NP; ++i) Pw[r][i] = f32x2{0.f, 0.f};
  float *gl = glogits ? glogits + (int64_t)b * KK * HW + p : nullptr;
  const T *at = attn ? attn + (int64_t)b * KK * HW + p : nullptr;
  float gx_acc = 0.f, gy_acc = 0.f;

  const int wp = Ws >> 1, per_plane = Hs * wp;
  const T *s0 = src + ((int64_t)b * C + c_begin) * Hs * Ws;
  f32x2 pre[kAggPre];
  unsigned off[kAggPre / 2];
  agg_chunk_offsets(CH * per_plane, per_plane, wp, pitch, plane_sz, off);
  int gc = min(CH, c_end - c_begin);
  agg_chunk_load<T>(s0, gc * per_plane, pre);
  {  // words no load ever writes must be finite: they meet weight 0 / are never selected
    const int nrow = CH * ((Hs + 1) >> 1);
    for (int u = 0; u < 2; ++u) {
      float *bf = lds + u * buf_sz;
      if (pitch >= 2 * Ws + 4)
        for (int i = threadIdx.x; i < nrow * 4; i += blockDim.x) bf[(i >> 2) * pitch + 2 * Ws + (i & 3)] = 0.f;
      if (threadIdx.x < 4) bf[CH * plane_sz + threadIdx.x] = 0.f;
    }
  }
  agg_chunk_store(lds, gc * per_plane, pre, off);
  __syncthreads();

  const T *go_b = gout + ((int64_t)b * C + c_begin) * HW;   // uniform base; the lane's pixel is the offset
  // raw upstream gradients of the chunk about to be computed (gcur) and of the one after it (gnxt); channels beyond the
  // range re-read the last one and are masked where they are used (a select behind a load would make it synchronous)
  float gcur[CHT], gnxt[CHT];
  const int cs_n = c_end - c_begin;
#pragma unroll
  for (int c = 0; c < CHT; ++c) gcur[c] = Num<T>::ld(go_b + (int64_t)min(c, cs_n - 1) * HW + p);
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): nothing pending when the loop is entered (see agg_fwd_stream_kernel)
  int cur = 0;
  for (int cb = c_begin; cb < c_end; cb += CH) {
    const int gn = min(CH, c_end - cb - CH);
#pragma unroll
    for (int c = 0; c < CHT; ++c) gnxt[c] = Num<T>::ld(go_b + (int64_t)min(cb + CH - c_begin + c, cs_n - 1) * HW + p);
    // unconditional, as in agg_fwd_stream_kernel (the last chunk re-requests one word pair)
    agg_chunk_load<T>(s0 + (int64_t)(gn > 0 ? cb + CH - c_begin : 0) * Hs * Ws, gn > 0 ? gn * per_plane : 1, pre);
    gc = min(CH, c_end - cb);
    const float *pl = lds + cur * buf_sz;
    if (dense) {
      auto load_row = [&](const float *rp, f32x2(&v)[NP + 1]) {
#pragma unroll
        for (int i = 0; i <= NP; ++i) {
          v[i] = *reinterpret_cast<const f32x2 *>(rp + 4 * i);
          asm volatile("" ::: "memory");
        }
      };
#pragma unroll
      for (int c = 0; c < CHT; c += 2) {
        // a chunk shorter than CHT: weight 0 on a re-read of its last plane
        const float g0 = c < gc ? gcur[c] * inv_kk : 0.f, g1 = c + 1 < gc ? gcur[c + 1 < CHT ? c + 1 : c] * inv_kk : 0.f;
        const float *p0_ = pl + min(c, gc - 1) * plane_sz, *p1_ = pl + min(c + 1, gc - 1) * plane_sz;
        const f32x2 gg0 = {g0, g0}, gg1 = {g1, g1};
        f32x2 v0[2][NP + 1], v1[2][NP + 1];
        load_row(p0_ + ro[0], v0[0]);
        load_row(p1_ + ro[0], v1[0]);
#pragma unroll
        for (int r = 0; r <= K; ++r) {
          if (r < K) {
            load_row(p0_ + ro[r + 1], v0[(r + 1) & 1]);
            load_row(p1_ + ro[r + 1], v1[(r + 1) & 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i <= NP; ++i) {
            Pw[r][i] = __builtin_elementwise_fma(gg0, v0[r & 1][i], Pw[r][i]);
            Pw[r][i] = __builtin_elementwise_fma(gg1, v1[r & 1][i], Pw[r][i]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if (active) {
      // rare (a tap within rounding of an integer): tap by tap, published directly
      for (int c = 0; c < gc; ++c) {
        const float go = Num<T>::ld(go_b + (int64_t)(cb - c_begin + c) * HW + p) * inv_kk;
        const float *plc = pl + (size_t)c * plane_sz;
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
          const float dy = (fy0 + (float)(i - K / 2)) + (float)yf;
          const float fdy = floorf(dy);
          const int yTc = clampi((int)fdy, 0, Hs - 1), yBc = clampi((int)(fdy + 1), 0, Hs - 1);
          const int yT = (yTc >> 1) * pitch + ((yTc & 1) << 1), yB = (yBc >> 1) * pitch + ((yBc & 1) << 1);
          const float yB_P = dy - fdy, yT_P = 1 - yB_P;
#pragma unroll 1
          for (int j = 0; j < K; ++j) {
            const float dx = (fx0 + (float)(j - K / 2)) + (float)xf;
            const float fdx = floorf(dx);
            const int xLc = clampi((int)fdx, 0, Ws - 1), xRc = clampi((int)(fdx + 1), 0, Ws - 1);
            const int xL = ((xLc >> 1) << 2) + (xLc & 1), xR = ((xRc >> 1) << 2) + (xRc & 1);
            const float xR_P = dx - fdx, xL_P = 1 - xR_P;
            const float vTL = plc[yT + xL], vTR = plc[yT + xR], vBL = plc[yB + xL], vBR = plc[yB + xR];
            float bs = (xL_P * yT_P) * vTL;
            bs += (xR_P * yT_P) * vTR;
            bs += (xL_P * yB_P) * vBL;
            bs += (xR_P * yB_P) * vBR;
            if (gl) atomic_add(gl + (int64_t)(i * K + j) * HW, go * bs);
            if (gflow) {
              const float gv = Num<T>::ld(at + (int64_t)(i * K + j) * HW) * go;
              gy_acc += gv * (-xL_P * vTL - xR_P * vTR + xL_P * vBL + xR_P * vBR);
              gx_acc += gv * (-yT_P * vTL - yB_P * vBL + yT_P * vTR + yB_P * vBR);
            }
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);   // (as in agg_fwd_stream_kernel: nothing of this branch stays pending)
    }
    cur ^= 1;
    if (gn > 0) agg_chunk_store(lds + cur * buf_sz, gn * per_plane, pre, off);
    __builtin_amdgcn_s_waitcnt(0x0F70);     // the prefetch and the next chunk's gradients have landed
#pragma unroll
    for (int c = 0; c < CHT; ++c) gcur[c] = gnxt[c];
    __syncthreads();
  }

  if (dense) {
    // window sums -> patch sums: P[r][q] = Pw[r][clamp(x0 + q) - xa]  (the slot is in [0, K + 2]; select chain on
    // statically indexed registers)
    int slot[K + 1];
#pragma unroll
    for (int q = 0; q <= K; ++q) slot[q] = clampi(px0 + q, 0, Ws - 1) - xa;
    float P[K + 1][K + 1];
#pragma unroll
    for (int r = 0; r <= K; ++r)
#pragma unroll
      for (int q = 0; q <= K; ++q) {
        float v = Pw[r][0].x;
#pragma unroll
        for (int w = 1; w < NW; ++w) v = slot[q] == w ? ((w & 1) ? Pw[r][w >> 1].y : Pw[r][w >> 1].x) : v;
        P[r][q] = v;
      }
    if constexpr (EPI == 1) {
      float Racc[K + 1], Cacc[K + 1];
#pragma unroll
      for (int r = 0; r <= K; ++r) Racc[r] = Cacc[r] = 0.f;
#pragma unroll
      for (int r = 0; r <= K; ++r)
#pragma unroll
        for (int q = 0; q <= K; ++q) {
          Racc[r] += rt.col_w(q) * P[r][q];
          Cacc[q] += rt.row_w(r) * P[r][q];
        }
      float rx, ry, rs;
      rs_bwd2_finish<float, 2>(rt, Racc, Cacc, rx, ry, rs);
      atomic_add(gflow + (int64_t)(b * 3 + 0) * HW + p, rx);
      atomic_add(gflow + (int64_t)(b * 3 + 1) * HW + p, ry);
      atomic_add(gflow + (int64_t)(b * 3 + 2) * HW + p, rs);
    } else {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const float yT_P = 1 - tp.ay[i], yB_P = tp.ay[i];
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float xL_P = 1 - tp.ax[j], xR_P = tp.ax[j];
        float ga = (xL_P * yT_P) * P[i][j];  // block_extractor_kernel.cu:78-84
        ga += (xR_P * yT_P) * P[i][j + 1];
        ga += (xL_P * yB_P) * P[i + 1][j];
        ga += (xR_P * yB_P) * P[i + 1][j + 1];
        if (gl) atomic_add(gl + (int64_t)(i * K + j) * HW, ga);
        if (gflow) {
          const float a_ij = Num<T>::ld(at + (int64_t)(i * K + j) * HW);
          gy_acc += a_ij * (-xL_P * P[i][j] - xR_P * P[i][j + 1] + xL_P * P[i + 1][j] + xR_P * P[i + 1][j + 1]);
          gx_acc += a_ij * (-yT_P * P[i][j] - yB_P * P[i + 1][j] + yT_P * P[i][j + 1] + yB_P * P[i + 1][j + 1]);
        }
      }
    }
    }
  }
  if (EPI == 0 && active && gflow) {
    atomic_add(gflow + (int64_t)(b * 2 + 0) * HW + p, gx_acc);
    atomic_add(gflow + (int64_t)(b * 2 + 1) * HW + p, gy_acc);
  }
}

// resample2d d/d input2 (kernel_size 4, dilation 1) on agg_ga_stream_kernel<T, 3, 1>; gin2 (B, 3, H, W) float32, accumulated
// into.  GFLA_ERR_UNSUPPORTED where the shape does not fit (the caller falls back to rs_lds_kernel<MODE 2>).
template <typename T>
int rs_bwd2_stream(const T *in1, const T *in2, const T *gout, float *gin2, int64_t B, int64_t C, int64_t Hi, int64_t Wi,
                   int64_t H, int64_t W, hipStream_t stream) {
  constexpr int k = 3;
  if (tuning(3) == 1 || tuning(8) == 1 || Wi < k + 1 || (Wi & 1) || Wi >= 32768 || Hi >= 32000 ||
      B * C * Hi * Wi >= (1LL << 31) || B * C * H * W >= (1LL << 31) || B > 65535)
    return GFLA_ERR_UNSUPPORTED;
  const AggStreamGeo pg = agg_stream_geometry(B, C, Hi, Wi, H, W, k);
  const int64_t total = B * pg.nsuper * pg.tgroups;
  const int64_t padded = ceil_div(total, kNumXCD) * kNumXCD;
  if (pg.CH <= 0 || padded > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
#define GFLA_RS2_LAUNCH(CV)                                                                                              \
  launch_lds(agg_ga_stream_kernel<T, 3, 1, CV>, dim3((unsigned)padded), dim3(pg.threads), pg.lds, stream, in1, in2, gout, \
             (float *)nullptr, (const T *)nullptr, gin2, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, pg.CH, pg.CS, pg.nsuper, \
             pg.tgroups, pg.pitch, (int)total, pg.tw_log2, pg.ntile)
  if (pg.CH <= 2) GFLA_RS2_LAUNCH(2);
  else if (pg.CH <= 4) GFLA_RS2_LAUNCH(4);
  else if (pg.CH <= 6) GFLA_RS2_LAUNCH(6);
  else GFLA_RS2_LAUNCH(8);
#undef GFLA_RS2_LAUNCH
  return launch_status();
}
template int rs_bwd2_stream<float>(const float *, const float *, const float *, float *, int64_t, int64_t, int64_t, int64_t,
                                   int64_t, int64_t, hipStream_t);
template int rs_bwd2_stream<bf16_t>(const bf16_t *, const bf16_t *, const bf16_t *, float *, int64_t, int64_t, int64_t,
                                    int64_t, int64_t, int64_t, hipStream_t);

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
                         gfla_stream_t stream_, void *workspace = nullptr) {
  if (!src || !flow || !logits || !out) return GFLA_ERR_NULL_POINTER;
  int st = agg_check(B, C, Hs, Ws, H, W, k);
  if (st != GFLA_OK) return st;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  using A = typename Num<T>::acc;
  if constexpr (std::is_same<A, float>::value) {
    // coefficient-table path: needs the caller's workspace (gfla_aggregate_fwd_workspace_bytes)
    // k = 3 has 16 patch words per output instead of 36: there the per-group setup of agg_fwd_lds_kernel costs less than
    // the extra pass (23 us against 20 + 5 at the bench shape); tuning key 8 = 2 forces the table path, 1 disables it
    if (workspace && tuning(3) != 1 && tuning(8) != 1 && (k >= 5 || tuning(8) == 2) && (k & 1) && Ws >= k + 1 && !(Ws & 1) && Ws < 32768 && Hs < 32000 &&
        B * C * Hs * Ws < (1LL << 31)) {
      AggStreamGeo pg = agg_stream_geometry(B, C, Hs, Ws, H, W, k);
      const int64_t total = B * pg.nsuper * pg.tgroups;
      const int64_t padded = ceil_div(total, kNumXCD) * kNumXCD;
      if (pg.CH > 0 && padded <= 0x7fffffffLL && B <= 65535) {
        float *table = static_cast<float *>(workspace);
        const dim3 cgrid((unsigned)ceil_div(pg.ntile, kAggCoefThreads / 64), (unsigned)B);
#define GFLA_AGG_TAB(KV)                                                                                                  \
  agg_coef_kernel<T, KV><<<cgrid, dim3(kAggCoefThreads), 0, stream>>>(flow, logits, attn_out, table, (int)Hs, (int)Ws,        \
                                                                      (int)H, (int)W, sm, pg.tw_log2, pg.ntile);           \
  GFLA_AGG_MAIN(KV)
#define GFLA_AGG_LAUNCH(KV, CV)                                                                                           \
  launch_lds(agg_fwd_stream_kernel<T, KV, CV>, dim3((unsigned)padded), dim3(pg.threads), pg.lds, stream, src, flow, logits, \
             (const float *)table, out, (int)C, (int)Hs, (int)Ws, (int)H, (int)W, sm, pg.CH, pg.CS, pg.nsuper, pg.tgroups, \
             pg.pitch, (int)total, pg.tw_log2, pg.ntile)
#define GFLA_AGG_MAIN(KV)                                          \
  if (pg.CH <= 2) GFLA_AGG_LAUNCH(KV, 2);                          \
  else if (pg.CH <= 4) GFLA_AGG_LAUNCH(KV, 4);                     \
  else if (pg.CH <= 6) GFLA_AGG_LAUNCH(KV, 6);                     \
  else GFLA_AGG_LAUNCH(KV, 8)
        switch (k) {
          case 1: GFLA_AGG_TAB(1); break;
          case 3: GFLA_AGG_TAB(3); break;
          default: GFLA_AGG_TAB(5); break;
        }
#undef GFLA_AGG_TAB
#undef GFLA_AGG_MAIN
#undef GFLA_AGG_LAUNCH
        return launch_status();
      }
    }
  }
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
  if constexpr (std::is_same<A, float>::value) {
    // streaming kernel (interleaved planes, paired reads); tuning key 8 = 1: round 1's kernel
    if (tuning(3) != 1 && tuning(8) != 1 && (k >= 5 || tuning(8) == 2) && (k & 1) && Ws >= k + 1 && !(Ws & 1) && Ws < 32768 && Hs < 32000 &&
        B * C * Hs * Ws < (1LL << 31)) {
      const AggStreamGeo pg = agg_stream_geometry(B, C, Hs, Ws, H, W, k);
      const int64_t total = B * pg.nsuper * pg.tgroups;
      const int64_t padded = ceil_div(total, kNumXCD) * kNumXCD;
      if (pg.CH > 0 && padded <= 0x7fffffffLL) {
#define GFLA_AGG_GA_LAUNCH(KV, CV)                                                                                       \
  launch_lds(agg_ga_stream_kernel<T, KV, 0, CV>, dim3((unsigned)padded), dim3(pg.threads), pg.lds, stream, src, flow, gout, \
             (float *)glogits, gflow ? attn : (const T *)nullptr, (float *)gflow, (int)C, (int)Hs, (int)Ws, (int)H, (int)W, \
             pg.CH, pg.CS, pg.nsuper, pg.tgroups, pg.pitch, (int)total, pg.tw_log2, pg.ntile)
#define GFLA_AGG_GA(KV)                                 \
  if (pg.CH <= 2) GFLA_AGG_GA_LAUNCH(KV, 2);            \
  else if (pg.CH <= 4) GFLA_AGG_GA_LAUNCH(KV, 4);       \
  else if (pg.CH <= 6) GFLA_AGG_GA_LAUNCH(KV, 6);       \
  else GFLA_AGG_GA_LAUNCH(KV, 8)
        switch (k) {
          case 1: GFLA_AGG_GA(1); break;
          case 3: GFLA_AGG_GA(3); break;
          default: GFLA_AGG_GA(5); break;
        }
#undef GFLA_AGG_GA
#undef GFLA_AGG_GA_LAUNCH
        int st = launch_status();
        if (st == GFLA_OK && sm && glogits) {
          const int64_t n = B * H * W;
          GFLA_K_SWITCH(k, agg_softmax_bwd_kernel<T, K><<<dim3((unsigned)ceil_div(n, kBlock)), dim3(kBlock), 0, stream>>>(
                               attn, glogits, n, (int)(H * W)));
          st = launch_status();
        }
        return st;
      }
    }
  }
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
  // bf16 storage exists for the planes-in-LDS kernels only: the forced-global test knob (key 3) does not apply to it
  if ((tuning(3) != 1 || sizeof(T) == 2) && Hs * Ws * (int64_t)(sizeof(lds_acc_t) + sizeof(A)) <= kLdsBudget) {
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
/* scratch for the coefficient-table forward: (k+1)(k+2) floats + one packed word per flow pixel */
int64_t gfla_aggregate_fwd_workspace_bytes(int64_t B, int64_t H, int64_t W, int k) {
  if (B <= 0 || H <= 0 || W <= 0 || k < 1 || k > 5) return 0;
  int64_t tiles = 0;  // records are stored per 64-pixel tile (8x8, 16x4 or 32x2, the launcher's choice), overhang included
  for (int l = 3; l <= 5; ++l) tiles = std::max<int64_t>(tiles, gfla::ceil_div(W, 1 << l) * gfla::ceil_div(H, 64 >> l));
  return B * tiles * 64 * (int64_t)gfla::agg_record_floats(k) * 4;
}
/* launch geometry of the table path (host logic only, for the CPU tests):
 * out[0..8] = channels per chunk, channels per range, ranges, tile groups, threads, LDS row-pair pitch (words), tile
 * width, tiles per sample, dynamic LDS bytes; returns GFLA_ERR_UNSUPPORTED where the plain kernels are used */
int gfla_aggregate_fwd_geometry(int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H, int64_t W, int k, int64_t *out) {
  if (!out) return GFLA_ERR_NULL_POINTER;
  if (B <= 0 || C <= 0 || Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0 || k < 1 || k > 5) return GFLA_ERR_BAD_SHAPE;
  if (!(k & 1) || Ws < k + 1 || (Ws & 1) || Ws >= 32768 || Hs >= 32000 || B * C * Hs * Ws >= (1LL << 31) || B > 65535)
    return GFLA_ERR_UNSUPPORTED;
  const gfla::AggStreamGeo g = gfla::agg_stream_geometry(B, C, Hs, Ws, H, W, k);
  if (g.CH <= 0) return GFLA_ERR_UNSUPPORTED;
  const int64_t v[9] = {g.CH, g.CS, g.nsuper, g.tgroups, g.threads, g.pitch, 1 << g.tw_log2, g.ntile, g.lds};
  for (int i = 0; i < 9; ++i) out[i] = v[i];
  return GFLA_OK;
}
int gfla_local_attn_aggregate_fwd_ws_f32(const float *s, const float *f, const float *l, float *o, float *a,
                                         void *workspace, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H,
                                         int64_t W, int k, int sm, gfla_stream_t st) {
  return gfla::aggregate_fwd<float>(s, f, l, o, a, B, C, Hs, Ws, H, W, k, sm, st, workspace);
}
int gfla_local_attn_aggregate_fwd_ws_bf16(const uint16_t *s, const uint16_t *f, const uint16_t *l, uint16_t *o,
                                          uint16_t *a, void *workspace, int64_t B, int64_t C, int64_t Hs, int64_t Ws,
                                          int64_t H, int64_t W, int k, int sm, gfla_stream_t st) {
  return gfla::aggregate_fwd<bf16_t>(reinterpret_cast<const bf16_t *>(s), reinterpret_cast<const bf16_t *>(f),
                                     reinterpret_cast<const bf16_t *>(l), reinterpret_cast<bf16_t *>(o),
                                     reinterpret_cast<bf16_t *>(a), B, C, Hs, Ws, H, W, k, sm, st, workspace);
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
/* 1 when gfla_local_attn_aggregate_bwd_<storage> takes source planes of Hs x Ws (elem_size 2 = bf16, 4 = f32, 8 = f64).
 * f32 / f64 always do (global-memory kernels behind the planes-in-LDS ones); bf16 storage exists for the planes-in-LDS
 * kernels only: a double accumulator plane + an arithmetic-type source plane per position within the LDS budget (tuning
 * key 10).  The host side asks here instead of duplicating the budget. */
int gfla_aggregate_bwd_supported(int64_t Hs, int64_t Ws, int elem_size) {
  if (Hs <= 0 || Ws <= 0 || (elem_size != 2 && elem_size != 4 && elem_size != 8)) return 0;
  if (elem_size != 2) return 1;
  return Hs * Ws * (int64_t)(sizeof(gfla::lds_acc_t) + sizeof(float)) <= gfla::lds_budget() ? 1 : 0;
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
