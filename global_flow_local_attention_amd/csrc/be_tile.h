// block_extractor for FEW, LARGE planes (BASELINE configs[1]: (1, 64, 256, 176), planes of 180 KB on 256 CUs), round 5.
// tile_map.h has the regime and the shared pieces.  Semantics: block_extractor_kernel.cu:20-85 (forward), :89-170 (backward).
//
// forward  be_fwd_gpix_kernel   lane = flow pixel (64 consecutive pixels per wave), the four waves of a workgroup take four
//          channel ranges of the SAME pixels; setup (flow pair, K fractions, patch origin) once per pixel for `cpw`
//          channels; the dense (K+1)^2 patch of a channel is read from global memory (row r: K+1 loads whose lanes are
//          consecutive addresses up to the flow's local variation), evaluated separably exactly as be_fwd_pix.h does, and
//          the K outputs of an output row leave as one 16-byte + one 4-byte store (k = 5).  The op is 96 % writes: what
//          the launch has to provide is enough waves streaming stores, which at B*C = 64 only spatial blocks can.
// backward be_bwd_tile_kernel   workgroup = (tile of th x tw flow pixels, G channels), lane = flow pixel.  The K*K incoming
//          gradients of a pixel are folded into its dense (K+1)^2 patch in registers (be_bwd_lds.h's fold) and added to
//          an LDS window = the bounding box of everything the tile's pixels reach, computed from the flow on the device;
//          the window leaves through one float atomic per touched element.  The window is processed in as many channel
//          rounds as fit the LDS budget; a window too large for one channel (wild flow) sends that tile to global atomics.
//          d/dflow is reduced in registers over taps and the G channels (source values from global memory / L1).
#pragma once

#include "be_fwd_pix.h"
#include "tile_map.h"

namespace gfla {

template <typename T, int K, int CH>
__global__ __launch_bounds__(256) void be_fwd_gpix_kernel(const T *__restrict__ src, const T *__restrict__ flow,
                                                         T *__restrict__ out, int C, int Hs, int Ws, int Hf, int Wf,
                                                         int cpw, int ncs4, int nblk, int64_t nwg) {
  using A = typename Num<T>::acc;
  const int64_t v = xcd_swizzle(blockIdx.x, nwg);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int cs = (int)(v % ncs4) * 4 + wave;
  const int64_t rest = v / ncs4;
  const int blk = (int)(rest % nblk), b = (int)(rest / nblk);
  const int c_begin = cs * cpw;
  if (c_begin >= C) return;   // (no barrier in this kernel)
  const int c_end = min(C, c_begin + cpw);
  const int HW = Hf * Wf, Wo = K * Wf;
  const int64_t oplane = (int64_t)(K * Hf) * Wo, plane = (int64_t)Hs * Ws;
  const int pl = (blk << 6) + lane;
  const bool active = pl < HW;
  const int p = active ? pl : HW - 1;
  const int yf = p / Wf, xf = p - yf * Wf;
  const A fx0 = Num<T>::ld(flow + (int64_t)(b * 2 + 0) * HW + p);
  const A fy0 = Num<T>::ld(flow + (int64_t)(b * 2 + 1) * HW + p);
  A ax[K], ay[K];
  int x0 = 0, y0 = 0;
  bool dense = true;
#pragma unroll
  for (int t = 0; t < K; ++t) {
    const A dx = (fx0 + (A)(t - K / 2)) + (A)xf;  // block_extractor_kernel.cu:62-67
    const A dy = (fy0 + (A)(t - K / 2)) + (A)yf;
    const A fdx = floor_t<A>(dx), fdy = floor_t<A>(dy);
    if (t == 0) {
      x0 = (int)fdx;
      y0 = (int)fdy;
    }
    dense &= ((int)fdx == x0 + t) & ((int)fdy == y0 + t);
    ax[t] = dx - fdx;
    ay[t] = dy - fdy;
  }
  const int ooff = (K * yf) * Wo + K * xf;
  const T *src_b = src + (int64_t)b * C * plane;
  T *out_b = out + (int64_t)b * C * oplane;
  if (dense) {
    // clamped columns / rows of the dense patch (:69-72); the clamp of the origin only keeps the sums in range
    const int x0c = clampi(x0, -(K + 1), Ws), y0c = clampi(y0, -(K + 1), Hs);
    int col[K + 1];
#pragma unroll
    for (int q = 0; q <= K; ++q) col[q] = clampi(x0c + q, 0, Ws - 1);
    for (int cb = c_begin; cb < c_end; cb += CH) {
      const int ncc = min(CH, c_end - cb);
      const T *plc = src_b + (int64_t)cb * plane;
      T *oc0 = out_b + (int64_t)cb * oplane + ooff;
      // the bilinear form separated (be_fwd_wrow.h has the derivation): patch rows interpolated along x once, output row
      // i = the blend of interpolated rows i and i + 1 -- the expressions of be_fwd_pix.h, operand for operand
      auto hrow = [&](int cc, int r, A (&h)[K]) {
        const T *pc = plc + (int64_t)min(cc, ncc - 1) * plane + clampi(y0c + r, 0, Hs - 1) * Ws;
        A vv[K + 1];
#pragma unroll
        for (int q = 0; q <= K; ++q) vv[q] = Num<T>::ld(pc + col[q]);
#pragma unroll
        for (int j = 0; j < K; ++j) h[j] = fma_t(ax[j], vv[j + 1], (1 - ax[j]) * vv[j]);
      };
      A hA[CH][K];
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) hrow(cc, 0, hA[cc]);
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const A yB_P = ay[i], yT_P = 1 - yB_P;
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) {
          A hB[K];
          hrow(cc, i + 1, hB);
          T o[K];
#pragma unroll
          for (int j = 0; j < K; ++j) o[j] = Num<T>::from(fma_t(yB_P, hB[j], yT_P * hA[cc][j]));
          if (active && cc < ncc) store_row<T, K, false>(oc0 + cc * oplane + (int64_t)i * Wo, o);
#pragma unroll
          for (int j = 0; j < K; ++j) hA[cc][j] = hB[j];
        }
      }
    }
  } else if (active) {
    // a coordinate within rounding of an integer: tap by tap, as the reference does (:69-84)
    int xL[K], xR[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const A dx = (fx0 + (A)(t - K / 2)) + (A)xf;
      const A fdx = floor_t<A>(dx);
      xL[t] = clampi((int)fdx, 0, Ws - 1);
      xR[t] = clampi((int)(fdx + 1), 0, Ws - 1);
    }
    for (int c = c_begin; c < c_end; ++c) {
      const T *pc = src_b + (int64_t)c * plane;
#pragma unroll 1
      for (int i = 0; i < K; ++i) {
        const A dy = (fy0 + (A)(i - K / 2)) + (A)yf;
        const A fdy = floor_t<A>(dy);
        const int yT = clampi((int)fdy, 0, Hs - 1) * Ws, yB = clampi((int)(fdy + 1), 0, Hs - 1) * Ws;
        const A yB_P = dy - fdy, yT_P = 1 - yB_P;
        T o[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const A xR_P = ax[j], xL_P = 1 - xR_P;
          A s = (xL_P * yT_P) * Num<T>::ld(pc + yT + xL[j]);
          s = fma_t(xR_P * yT_P, Num<T>::ld(pc + yT + xR[j]), s);
          s = fma_t(xL_P * yB_P, Num<T>::ld(pc + yB + xL[j]), s);
          s = fma_t(xR_P * yB_P, Num<T>::ld(pc + yB + xR[j]), s);
          o[j] = Num<T>::from(s);
        }
        store_row<T, K, false>(out_b + (int64_t)c * oplane + (int64_t)i * Wo + ooff, o);
      }
    }
  }
}

// channels per wave: as many as keep >= `want` waves in flight (tuning key 33 overrides)
inline int big_channels_per_wave(int64_t B, int64_t C, int64_t nblk, int ch, int64_t want) {
  if (tuning(33) > 0) return tuning(33) < C ? tuning(33) : (int)C;
  int64_t cpw = C;
  while (cpw > ch && B * nblk * ceil_div(C, cpw) < want) cpw = ceil_div(cpw, 2);
  cpw = ceil_div(cpw, ch) * ch;
  return (int)(cpw < C ? cpw : C);
}

template <typename T, int K>
static int launch_fwd_gpix(const T *src, const T *flow, T *out, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf,
                           int64_t Wf, hipStream_t stream, bool *done) {
  using A = typename Num<T>::acc;
  constexpr int CH = sizeof(A) == 8 ? (K >= 4 ? 1 : 2) : (K >= 5 ? 2 : 4);
  *done = false;
  const int64_t nblk = ceil_div(Hf * Wf, 64);
  const int cpw = big_channels_per_wave(B, C, nblk, CH, 24 * kNumCU);
  const int64_t ncs = ceil_div(C, cpw), ncs4 = ceil_div(ncs, 4);
  const int64_t nwg = B * nblk * ncs4;
  if (nwg > 0x7fffffffLL) return GFLA_OK;
  be_fwd_gpix_kernel<T, K, CH><<<dim3((unsigned)nwg), dim3(256), 0, stream>>>(src, flow, out, (int)C, (int)Hs, (int)Ws, (int)Hf,
                                                                             (int)Wf, cpw, (int)ncs4, (int)nblk, nwg);
  *done = true;
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------------
// Where a folded patch row goes: the LDS window (double planes: ds_add_f64) or, for a tile whose reach does not fit, the
// gradient plane itself (float / double atomics).
template <typename T>
struct BeWinSink {
  lds_acc_t *plane;  // this channel's window, element (row - ymin) * cols + (col - xmin)
  int cols, ymin, xmin;
  __device__ __forceinline__ void add(int row, int col, typename Num<T>::acc v) const {
    lds_add(plane + (row - ymin) * cols + (col - xmin), (lds_acc_t)v);
  }
};
template <typename T>
struct BeGlobalSink {
  T *plane;
  int Ws;
  __device__ __forceinline__ void add(int row, int col, typename Num<T>::acc v) const { atomic_add(plane + row * Ws + col, (T)v); }
};

// One channel of one flow pixel: fold the K x K gradients into the dense patch and hand its rows to `sink`; accumulate
// d/dflow.  `spl` = this channel's source plane (global), `gblk` = &grad_out[b, c, yf*K, xf*K].
template <typename T, int K, bool NEED_SRC, bool NEED_FLOW, typename Sink>
__device__ __forceinline__ void be_bwd_pixel_dense(const Sink &sink, const T *__restrict__ spl, const T *__restrict__ gblk,
                                                   int Wo, int Hs, int Ws, int y0c, const int (&col)[K + 1],
                                                   const typename Num<T>::acc (&ax)[K], typename Num<T>::acc fy0, int yf,
                                                   typename Num<T>::acc &gx_acc, typename Num<T>::acc &gy_acc) {
  using A = typename Num<T>::acc;
  A rowA[K + 1], vA[K + 1];
  int rA = clampi(y0c, 0, Hs - 1);
#pragma unroll
  for (int q = 0; q <= K; ++q) {
    rowA[q] = 0;
    vA[q] = NEED_FLOW ? Num<T>::ld(spl + rA * Ws + col[q]) : (A)0;
  }
#pragma unroll 1
  for (int i = 0; i < K; ++i) {
    const A dy = (fy0 + (A)(i - K / 2)) + (A)yf;  // block_extractor_kernel.cu:132-136
    const A yB_P = dy - floor_t<A>(dy), yT_P = 1 - yB_P;
    const int rB = clampi(y0c + i + 1, 0, Hs - 1);
    A gv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) gv[j] = Num<T>::ld(gblk + i * Wo + j);
    A rowB[K + 1], vB[K + 1];
#pragma unroll
    for (int q = 0; q <= K; ++q) {
      rowB[q] = 0;
      vB[q] = NEED_FLOW ? Num<T>::ld(spl + rB * Ws + col[q]) : (A)0;
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const A xL_P = 1 - ax[j], xR_P = ax[j];
      if (NEED_SRC) {  // :158-161, folded into the patch
        rowA[j] += gv[j] * xL_P * yT_P;
        rowA[j + 1] += gv[j] * xR_P * yT_P;
        rowB[j] += gv[j] * xL_P * yB_P;
        rowB[j + 1] += gv[j] * xR_P * yB_P;
      }
      if (NEED_FLOW) {  // :163-164
        gy_acc += gv[j] * (-xL_P * vA[j] - xR_P * vA[j + 1] + xL_P * vB[j] + xR_P * vB[j + 1]);
        gx_acc += gv[j] * (-yT_P * vA[j] - yB_P * vB[j] + yT_P * vA[j + 1] + yB_P * vB[j + 1]);
      }
    }
    if (NEED_SRC) {
#pragma unroll
      for (int q = 0; q <= K; ++q)
        if (rowA[q] != 0) sink.add(rA, col[q], rowA[q]);
    }
#pragma unroll
    for (int q = 0; q <= K; ++q) {
      rowA[q] = rowB[q];
      vA[q] = vB[q];
    }
    rA = rB;
  }
  if (NEED_SRC) {
#pragma unroll
    for (int q = 0; q <= K; ++q)
      if (rowA[q] != 0) sink.add(rA, col[q], rowA[q]);
  }
}

// the reference's own tap-by-tap form (a tap's floor() landed one off the dense patch)
template <typename T, int K, bool NEED_SRC, bool NEED_FLOW, typename Sink>
__device__ __forceinline__ void be_bwd_pixel_taps(const Sink &sink, const T *__restrict__ spl, const T *__restrict__ gblk, int Wo,
                                                  int Hs, int Ws, const int (&xL)[K], const int (&xR)[K],
                                                  const typename Num<T>::acc (&ax)[K], typename Num<T>::acc fy0, int yf,
                                                  typename Num<T>::acc &gx_acc, typename Num<T>::acc &gy_acc) {
  using A = typename Num<T>::acc;
#pragma unroll 1
  for (int i = 0; i < K; ++i) {
    const A dy = (fy0 + (A)(i - K / 2)) + (A)yf;
    const A fdy = floor_t<A>(dy);
    const int yT = clampi((int)fdy, 0, Hs - 1), yB = clampi((int)(fdy + 1), 0, Hs - 1);
    const A yB_P = dy - fdy, yT_P = 1 - yB_P;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const A g = Num<T>::ld(gblk + i * Wo + j);
      const A xL_P = 1 - ax[j], xR_P = ax[j];
      if (NEED_FLOW) {
        const A vTL = Num<T>::ld(spl + yT * Ws + xL[j]), vTR = Num<T>::ld(spl + yT * Ws + xR[j]);
        const A vBL = Num<T>::ld(spl + yB * Ws + xL[j]), vBR = Num<T>::ld(spl + yB * Ws + xR[j]);
        gy_acc += g * (-xL_P * vTL - xR_P * vTR + xL_P * vBL + xR_P * vBR);
        gx_acc += g * (-yT_P * vTL - yB_P * vBL + yT_P * vTR + yB_P * vBR);
      }
      if (NEED_SRC) {
        sink.add(yT, xL[j], g * xL_P * yT_P);
        sink.add(yT, xR[j], g * xR_P * yT_P);
        sink.add(yB, xL[j], g * xL_P * yB_P);
        sink.add(yB, xR[j], g * xR_P * yB_P);
      }
    }
  }
}

template <typename T, int K, bool NEED_SRC, bool NEED_FLOW>
__global__ __launch_bounds__(512) void be_bwd_tile_kernel(const T *__restrict__ src, const T *__restrict__ flow,
                                                         const T *__restrict__ gout, T *__restrict__ gsrc,
                                                         typename Num<T>::acc *__restrict__ gflow, int C, int Hs, int Ws,
                                                         int Hf, int Wf, int th, int tw, int ntx, int nty, int G, int ngroups,
                                                         int lds_elems, int64_t nwg) {
  using A = typename Num<T>::acc;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  lds_acc_t *planes = reinterpret_cast<lds_acc_t *>(gfla_smem);
  __shared__ int s_box[4];
  const int64_t v = xcd_swizzle(blockIdx.x, nwg);
  const int g = (int)(v % ngroups);
  const int64_t rest = v / ngroups;
  const int tile = (int)(rest % ((int64_t)ntx * nty)), b = (int)(rest / ((int64_t)ntx * nty));
  const int ty = tile / ntx, tx = tile - ty * ntx;
  const int c0 = g * G, gc = min(G, C - c0);
  const int ly = threadIdx.x / tw, lx = threadIdx.x - ly * tw;
  const int yf = ty * th + ly, xf = tx * tw + lx;
  const bool active = ly < th && yf < Hf && xf < Wf;
  const int HW = Hf * Wf, Wo = K * Wf;
  const int plane = Hs * Ws;
  const int64_t oplane = (int64_t)K * Hf * Wo;
  box_init(s_box);
  __syncthreads();
  // ---- per-pixel setup, once for the G channels
  A fx0 = 0, fy0 = 0, ax[K];
  int xL[K], xR[K], col[K + 1];
  int x0 = 0, y0 = 0, y0c = 0;
  bool dense = true;
  int bylo = 0x7fffffff, bxlo = 0x7fffffff, byhi = -1, bxhi = -1;
  const int p = active ? yf * Wf + xf : 0;
  if (active) {
    fx0 = Num<T>::ld(flow + (int64_t)(b * 2 + 0) * HW + p);
    fy0 = Num<T>::ld(flow + (int64_t)(b * 2 + 1) * HW + p);
  }
#pragma unroll
  for (int t = 0; t < K; ++t) {
    const A dx = (fx0 + (A)(t - K / 2)) + (A)xf;
    const A dy = (fy0 + (A)(t - K / 2)) + (A)yf;
    const A fdx = floor_t<A>(dx), fdy = floor_t<A>(dy);
    if (t == 0) {
      x0 = (int)fdx;
      y0 = (int)fdy;
    }
    dense = dense && ((int)fdx == x0 + t) && ((int)fdy == y0 + t);
    xL[t] = clampi((int)fdx, 0, Ws - 1);
    xR[t] = clampi((int)(fdx + 1), 0, Ws - 1);
    ax[t] = dx - fdx;
  }
  {
    const int x0c = clampi(x0, -(K + 2), Ws + 1);
    y0c = clampi(y0, -(K + 2), Hs + 1);
#pragma unroll
    for (int q = 0; q <= K; ++q) col[q] = clampi(x0c + q, 0, Ws - 1);
    if (active) {  // one row / column of slack covers the taps of the non-dense case
      bylo = clampi(y0c - 1, 0, Hs - 1), byhi = clampi(y0c + K + 1, 0, Hs - 1);
      bxlo = clampi(x0c - 1, 0, Ws - 1), bxhi = clampi(x0c + K + 1, 0, Ws - 1);
    }
  }
  if (NEED_SRC) box_reduce(s_box, bylo, bxlo, byhi, bxhi);
  __syncthreads();
  const int ymin = s_box[0], xmin = s_box[1];
  const int rows = s_box[2] - ymin + 1, cols = s_box[3] - xmin + 1;
  const int win = NEED_SRC ? rows * cols : 1;   // (rows <= 0: a tile without pixels, impossible by construction)
  const int g_fit = NEED_SRC ? min(gc, lds_elems / max(win, 1)) : gc;
  const T *src0 = src + ((int64_t)b * C + c0) * plane;
  T *gsrc0 = NEED_SRC ? gsrc + ((int64_t)b * C + c0) * plane : nullptr;
  const T *gblk0 = gout + ((int64_t)b * C + c0) * oplane + (int64_t)(yf * K) * Wo + xf * K;
  A gx_acc = 0, gy_acc = 0;
  if (g_fit == 0) {
    // the tile reaches further than one channel's window holds: global atomics for this tile
    if (active) {
      for (int c = 0; c < gc; ++c) {
        BeGlobalSink<T> sink{gsrc0 + (int64_t)c * plane, Ws};
        if (dense)
          be_bwd_pixel_dense<T, K, NEED_SRC, NEED_FLOW>(sink, src0 + (int64_t)c * plane, gblk0 + (int64_t)c * oplane, Wo, Hs, Ws,
                                                        y0c, col, ax, fy0, yf, gx_acc, gy_acc);
        else
          be_bwd_pixel_taps<T, K, NEED_SRC, NEED_FLOW>(sink, src0 + (int64_t)c * plane, gblk0 + (int64_t)c * oplane, Wo, Hs, Ws,
                                                       xL, xR, ax, fy0, yf, gx_acc, gy_acc);
      }
    }
  } else {
    for (int cb = 0; cb < gc; cb += g_fit) {
      const int n = min(g_fit, gc - cb);
      if (NEED_SRC) {
        zero_planes<lds_acc_t>(planes, n * win);
        __syncthreads();
      }
      if (active) {
        for (int c = 0; c < n; ++c) {
          BeWinSink<T> sink{planes + (size_t)c * win, cols, ymin, xmin};
          const T *spl = src0 + (int64_t)(cb + c) * plane;
          const T *gb = gblk0 + (int64_t)(cb + c) * oplane;
          if (dense)
            be_bwd_pixel_dense<T, K, NEED_SRC, NEED_FLOW>(sink, spl, gb, Wo, Hs, Ws, y0c, col, ax, fy0, yf, gx_acc, gy_acc);
          else
            be_bwd_pixel_taps<T, K, NEED_SRC, NEED_FLOW>(sink, spl, gb, Wo, Hs, Ws, xL, xR, ax, fy0, yf, gx_acc, gy_acc);
        }
      }
      if (NEED_SRC) {
        __syncthreads();
        for (int i = threadIdx.x; i < n * win; i += blockDim.x) {
          const lds_acc_t val = planes[i];
          if (val != 0) {
            const int c = i / win, e = i - c * win;
            const int wr = e / cols, wc = e - wr * cols;
            atomic_add(gsrc0 + (int64_t)(cb + c) * plane + (ymin + wr) * Ws + xmin + wc, (T)val);
          }
        }
        __syncthreads();
      }
    }
  }
  if (NEED_FLOW && active) {
    atomic_add(gflow + (int64_t)(b * 2 + 0) * HW + p, gx_acc);
    atomic_add(gflow + (int64_t)(b * 2 + 1) * HW + p, gy_acc);
  }
}

template <typename T, int K>
static int launch_be_bwd_tile(const T *src, const T *flow, const T *gout, T *gsrc, typename Num<T>::acc *gflow, int64_t B,
                              int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, hipStream_t stream, bool *done) {
  *done = false;
  if constexpr (sizeof(T) == 2) {
    return GFLA_OK;  // bf16 storage has no atomics: the planes-in-LDS kernels only
  } else {
    if (Hs * Ws > 0x3fffffffLL || (int64_t)K * Hf * K * Wf > 0x7fffffffLL) return GFLA_OK;
    const TileGeo tg = tile_geometry(Hf, Wf);
    int G = tuning(34) > 0 ? tuning(34) : 4;
    // fewer channels per workgroup while the launch has under three workgroups per CU
    while (G > 1 && B * tg.nty * tg.ntx * ceil_div(C, G) < 3 * kNumCU) G /= 2;
    if (G > C) G = (int)C;
    const int64_t ngroups = ceil_div(C, G);
    const int64_t nwg = B * tg.nty * tg.ntx * ngroups;
    if (nwg > 0x7fffffffLL) return GFLA_OK;
    const unsigned lds_bytes = gsrc ? (unsigned)lds_budget() : 0u;
    const int lds_elems = (int)(lds_bytes / sizeof(lds_acc_t));
    const dim3 grid((unsigned)nwg), blk((unsigned)tg.threads);
#define GFLA_BE_TILE_LAUNCH(S, F)                                                                                          \
  launch_lds(be_bwd_tile_kernel<T, K, S, F>, grid, blk, lds_bytes, stream, src, flow, gout, gsrc, gflow, (int)C, (int)Hs,  \
             (int)Ws, (int)Hf, (int)Wf, tg.th, tg.tw, tg.ntx, tg.nty, G, (int)ngroups, lds_elems, nwg)
    if (gsrc && gflow) GFLA_BE_TILE_LAUNCH(true, true);
    else if (gsrc) GFLA_BE_TILE_LAUNCH(true, false);
    else GFLA_BE_TILE_LAUNCH(false, true);
#undef GFLA_BE_TILE_LAUNCH
    *done = true;
    return launch_status();
  }
}

}  // namespace gfla
