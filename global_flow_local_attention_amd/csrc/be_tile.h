// block_extractor for FEW, LARGE planes (BASELINE configs[1]: (1, 64, 256, 176), planes of 180 KB on 256 CUs), round 5.
// tile_map.h has the regime and the shared pieces.  Semantics: block_extractor_kernel.cu:20-85 (forward), :89-170 (backward).
//
// Both directions: workgroup = (tile of th x tw flow pixels, G channels), lane = flow pixel, per-pixel setup once for the G
// channels.  The tile's BOUNDING BOX in the source plane (everything its pixels reach, from the flow, on the device) is the
// LDS window; the G channels pass through it in as many rounds as the LDS budget allows; a box too large for one channel
// (wild flow) sends that tile -- and only that tile -- to global memory.
//
// forward  be_fwd_tile_kernel   the window holds the source; a pixel's dense (K+1)^2 patch is read from it with ds_reads and
//          evaluated separably exactly as be_fwd_pix.h does; the K outputs of an output row leave as one 16-byte + one
//          4-byte store (k = 5).  Default tiles are WHOLE flow rows (tile_map.h: row_tile_geometry): the K*th output rows
//          of a channel are then one contiguous piece of the output plane per workgroup -- the op is 96 % writes.
//          (Measured and removed, round 5: staged stores for k = 5 -- the 20 bytes a lane writes per output row are a 16-byte
//          and a 4-byte store, two partial writes of every line; parking a row in a per-wave LDS tile and writing whole
//          16-byte pieces with consecutive lanes ran 71.7 -> 167.8 us: an LDS round trip per output row costs far more.)
//          (First version, measured and replaced: lane = pixel with the patch read from GLOBAL memory -- be_fwd_gpix_kernel,
//          kept as the per-tile fallback and under tuning key 38 = 1.  At (1,64,256,176) it ran at the store stream's rate
//          on a zero flow, 23 us = 0.62 of HBM, but at 47 us on a smooth one whatever the channels per wave: every per-tap
//          wave load costs ~40 CU cycles once the lanes' rows differ, 16 of them per pixel and channel;
//          profiles/r5_config2_first_kernels.txt.)
// backward be_bwd_tile_kernel   the K*K incoming gradients of a pixel are folded into its dense (K+1)^2 patch in registers
//          (be_bwd_lds.h's fold) and added to the window (double planes, ds_add_f64), which leaves through one float atomic
//          per touched element; d/dflow is reduced in registers over taps and the G channels, its source values come from a
//          second (float) window staged next to the accumulators.
#pragma once

#include <type_traits>

#include "be_fwd_pix.h"
#include "pin_regs.h"
#include "tile_map.h"

namespace gfla {

// Per-pixel setup of both directions: flow pair, K fractions per axis, patch origin, dense flag, clamped columns.
template <typename T, int K>
struct BePixel {
  using A = typename Num<T>::acc;
  A fx0, fy0, ax[K], ay[K];
  int x0c, y0c, xL[K], xR[K], col[K + 1];
  bool dense;
  __device__ __forceinline__ void init(const T *__restrict__ flow_b, int HW, int p, int xf, int yf, int Hs, int Ws, bool active) {
    fx0 = active ? Num<T>::ld(flow_b + p) : (A)0;
    fy0 = active ? Num<T>::ld(flow_b + HW + p) : (A)0;
    int x0 = 0, y0 = 0;
    dense = true;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const A dx = (fx0 + (A)(t - K / 2)) + (A)xf;  // block_extractor_kernel.cu:62-67 / :132-136
      const A dy = (fy0 + (A)(t - K / 2)) + (A)yf;
      const A fdx = floor_t<A>(dx), fdy = floor_t<A>(dy);
      if (t == 0) {
        x0 = (int)fdx;
        y0 = (int)fdy;
      }
      dense &= ((int)fdx == x0 + t) & ((int)fdy == y0 + t);
      xL[t] = clampi((int)fdx, 0, Ws - 1);  // :69-72
      xR[t] = clampi((int)(fdx + 1), 0, Ws - 1);
      ax[t] = dx - fdx;
      ay[t] = dy - fdy;
    }
    // the clamp of the origin only keeps the sums below in range
    x0c = clampi(x0, -(K + 2), Ws + 1);
    y0c = clampi(y0, -(K + 2), Hs + 1);
#pragma unroll
    for (int q = 0; q <= K; ++q) col[q] = clampi(x0c + q, 0, Ws - 1);
  }
  // A lane outside the tile (tile padding, or beyond the map) still walks the dense path's loads -- only its stores are masked.
  // Park its patch INSIDE the tile's bounding box (box = ymin, xmin, ymax, xmax of what the active lanes reach), so that its
  // window reads stay inside the staged window whatever the tile's flow is: a box shorter than K + 1 rows / columns was cut by
  // the plane's border on that side, where the index clamp of the patch rows / columns does the rest.
  __device__ __forceinline__ void park(const int *box, int Ws) {
    y0c = box[0] == 0 ? box[2] - K : box[0];
    x0c = box[1] == 0 ? box[3] - K : box[1];
#pragma unroll
    for (int q = 0; q <= K; ++q) col[q] = clampi(x0c + q, 0, Ws - 1);
  }
  // what the pixel can reach, one row / column of slack for the taps of the non-dense case
  __device__ __forceinline__ void reach(int Hs, int Ws, bool active, int &ylo, int &xlo, int &yhi, int &xhi) const {
    ylo = active ? clampi(y0c - 1, 0, Hs - 1) : 0x7fffffff;
    xlo = active ? clampi(x0c - 1, 0, Ws - 1) : 0x7fffffff;
    yhi = active ? clampi(y0c + K + 1, 0, Hs - 1) : -1;
    xhi = active ? clampi(x0c + K + 1, 0, Ws - 1) : -1;
  }
};

// ---- forward: per-pixel bodies shared by the window kernel (P = arithmetic type, LDS) and the global one (P = storage) ----
// row(cc, r): pointer p such that p[c] is the source value at plane column c of the CLAMPED patch row r (0..K) of chunk
// channel cc.  Output rows go to oc0 + cc * oplane + i * Wo.
#ifndef GFLA_BE_ROLL_FROM
#define GFLA_BE_ROLL_FROM 5
#endif
constexpr int kBeRollFrom = GFLA_BE_ROLL_FROM;   // output-row loop rolled from this kernel size on
template <typename T, typename P, int K, int CH, typename RowFn>
__device__ __forceinline__ void be_fwd_dense_chunk(RowFn row, int ncc, const BePixel<T, K> &px, int yf, T *__restrict__ oc0,
                                                   int64_t oplane, int Wo, bool active) {
  using A = typename Num<T>::acc;
  const int (&col)[K + 1] = px.col;
  const A (&ax)[K] = px.ax;
  // no lane of the wave has a patch column clamped at the border: a patch row is base + 0..K (immediate offsets, pairs)
  const bool contiguous = __all(col[K] - col[0] == K);
  // the bilinear form separated (be_fwd_wrow.h has the derivation): patch rows interpolated along x once, output row i = the
  // blend of interpolated rows i and i + 1 -- the expressions of be_fwd_pix.h, operand for operand.  A patch row of ALL CH
  // channels is requested at once (pin_regs.h), then interpolated.
  auto hrows = [&](int r, A (&h)[CH][K]) {
    A vv[CH * (K + 1)];
    if (contiguous) {
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) {
        const P *p0 = row(min(cc, ncc - 1), r) + col[0];
#pragma unroll
        for (int q = 0; q <= K; ++q) vv[cc * (K + 1) + q] = Num<P>::ld(p0 + q);
      }
    } else {
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) {
        const P *pc = row(min(cc, ncc - 1), r);
#pragma unroll
        for (int q = 0; q <= K; ++q) vv[cc * (K + 1) + q] = Num<P>::ld(pc + col[q]);
      }
    }
    pin_regs(vv);
#pragma unroll
    for (int cc = 0; cc < CH; ++cc)
#pragma unroll
      for (int j = 0; j < K; ++j)
        h[cc][j] = fma_t(ax[j], vv[cc * (K + 1) + j + 1], (1 - ax[j]) * vv[cc * (K + 1) + j]);
  };
  A hA[CH][K];
  hrows(0, hA);
  auto out_row = [&](int i, A yB_P) {
    const A yT_P = 1 - yB_P;
    A hB[CH][K];
    hrows(i + 1, hB);
#pragma unroll
    for (int cc = 0; cc < CH; ++cc) {
      T o[K];
#pragma unroll
      for (int j = 0; j < K; ++j) o[j] = Num<T>::from(fma_t(yB_P, hB[cc][j], yT_P * hA[cc][j]));
      if (active && cc < ncc) store_row<T, K, false>(oc0 + cc * oplane + (int64_t)i * Wo, o);
#pragma unroll
      for (int j = 0; j < K; ++j) hA[cc][j] = hB[cc][j];
    }
  };
  if constexpr (K >= kBeRollFrom) {
    // rolled: five unrolled output rows cost ~200 registers (two waves per SIMD: k = 5 ran 70.4 us unrolled, 60.8 rolled at
    // (1,64,256,176)); the row's fraction is re-derived with the expression that filled px.ay
    // (block_extractor_kernel.cu:62-67): the same bits
#pragma unroll 1
    for (int i = 0; i < K; ++i) {
      const A dy = (px.fy0 + (A)(i - K / 2)) + (A)yf;
      out_row(i, dy - floor_t<A>(dy));
    }
  } else {
#pragma unroll
    for (int i = 0; i < K; ++i) out_row(i, px.ay[i]);
  }
}
// a coordinate within rounding of an integer: tap by tap, as the reference does (:69-84).  at(yrow) -> pointer to plane row
// yrow (clamped) of this channel, indexed by clamped plane columns.
template <typename T, typename P, int K, typename RowAt>
__device__ __forceinline__ void be_fwd_taps_channel(RowAt at, int Hs, const int (&xL)[K], const int (&xR)[K],
                                                    const typename Num<T>::acc (&ax)[K], typename Num<T>::acc fy0, int yf,
                                                    T *__restrict__ oc, int Wo) {
  using A = typename Num<T>::acc;
#pragma unroll 1
  for (int i = 0; i < K; ++i) {
    const A dy = (fy0 + (A)(i - K / 2)) + (A)yf;
    const A fdy = floor_t<A>(dy);
    const P *rT = at(clampi((int)fdy, 0, Hs - 1)), *rB = at(clampi((int)(fdy + 1), 0, Hs - 1));
    const A yB_P = dy - fdy, yT_P = 1 - yB_P;
    T o[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const A xR_P = ax[j], xL_P = 1 - xR_P;
      A s = (xL_P * yT_P) * Num<P>::ld(rT + xL[j]);
      s = fma_t(xR_P * yT_P, Num<P>::ld(rT + xR[j]), s);
      s = fma_t(xL_P * yB_P, Num<P>::ld(rB + xL[j]), s);
      s = fma_t(xR_P * yB_P, Num<P>::ld(rB + xR[j]), s);
      o[j] = Num<T>::from(s);
    }
    store_row<T, K, false>(oc + (int64_t)i * Wo, o);
  }
}

// ---- forward, first version: patch read from global memory (the per-tile fallback of the window kernel; key 38 = 1) --------
template <typename T, int K, int CH>
__global__ __launch_bounds__(256) void be_fwd_gpix_kernel(const T *__restrict__ src, const T *__restrict__ flow,
                                                         T *__restrict__ out, int C, int Hs, int Ws, int Hf, int Wf,
                                                         int cpw, int ncs4, int nblk, int64_t nwg) {
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
  BePixel<T, K> px;
  px.init(flow + (int64_t)b * 2 * HW, HW, p, xf, yf, Hs, Ws, true);
  const int ooff = (K * yf) * Wo + K * xf;
  const T *src_b = src + (int64_t)b * C * plane;
  T *out_b = out + (int64_t)b * C * oplane;
  if (px.dense) {
    for (int cb = c_begin; cb < c_end; cb += CH) {
      const T *plc = src_b + (int64_t)cb * plane;
      const int y0c = px.y0c;
      auto row = [=](int cc, int r) { return plc + (int64_t)cc * plane + clampi(y0c + r, 0, Hs - 1) * Ws; };
      be_fwd_dense_chunk<T, T, K, CH>(row, min(CH, c_end - cb), px, yf, out_b + (int64_t)cb * oplane + ooff, oplane, Wo, active);
    }
  } else if (active) {
    for (int c = c_begin; c < c_end; ++c) {
      const T *pc = src_b + (int64_t)c * plane;
      auto at = [=](int yrow) { return pc + yrow * Ws; };
      be_fwd_taps_channel<T, T, K>(at, Hs, px.xL, px.xR, px.ax, px.fy0, yf, out_b + (int64_t)c * oplane + ooff, Wo);
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

// ---- forward: the window kernel ------------------------------------------------------------------------------------------
template <typename T, int K, int CH>
__global__ __launch_bounds__(512) void be_fwd_tile_kernel(const T *__restrict__ src, const T *__restrict__ flow,
                                                         T *__restrict__ out, int C, int Hs, int Ws, int Hf, int Wf, int th,
                                                         int tw, int ntx, int nty, int G, int ngroups, int lds_elems,
                                                         int64_t nwg) {
  using A = typename Num<T>::acc;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  A *planes = reinterpret_cast<A *>(gfla_smem);
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
  const int HW = Hf * Wf, Wo = K * Wf, plane = Hs * Ws;
  const int64_t oplane = (int64_t)(K * Hf) * Wo;
  const int p = active ? yf * Wf + xf : 0;
  box_init(s_box);
  __syncthreads();
  BePixel<T, K> px;
  px.init(flow + (int64_t)b * 2 * HW, HW, p, xf, yf, Hs, Ws, active);
  {
    int ylo, xlo, yhi, xhi;
    px.reach(Hs, Ws, active, ylo, xlo, yhi, xhi);
    box_reduce(s_box, ylo, xlo, yhi, xhi);
  }
  __syncthreads();
  if (!active) px.park(s_box, Ws);
  const T *src0 = src + ((int64_t)b * C + c0) * plane;
  const bool vec = window_vec_ok(src0, plane, Ws);
  const TileWin w = tile_window_vec(s_box, Ws, vec);
  const int g_fit = window_worth_staging(w, th, tw) ? min(gc, lds_elems / max(w.size, 1)) : 0;
  T *out0 = out + ((int64_t)b * C + c0) * oplane + (int64_t)(K * yf) * Wo + K * xf;
  if (g_fit == 0) {   // the tile reaches further than one channel's window holds: its patches come from global memory
    if (px.dense) {
      for (int cb = 0; cb < gc; cb += CH) {
        const T *plc = src0 + (int64_t)cb * plane;
        const int y0c = px.y0c;
        auto row = [=](int cc, int r) { return plc + (int64_t)cc * plane + clampi(y0c + r, 0, Hs - 1) * Ws; };
        be_fwd_dense_chunk<T, T, K, CH>(row, min(CH, gc - cb), px, yf, out0 + (int64_t)cb * oplane, oplane, Wo, active);
      }
    } else if (active) {
      for (int c = 0; c < gc; ++c) {
        const T *pc = src0 + (int64_t)c * plane;
        auto at = [=](int yrow) { return pc + yrow * Ws; };
        be_fwd_taps_channel<T, T, K>(at, Hs, px.xL, px.xR, px.ax, px.fy0, yf, out0 + (int64_t)c * oplane, Wo);
      }
    }
    return;
  }
  const A *win0 = planes - (w.ymin * w.cols + w.xmin);   // (plane row, plane column) -> win0[row * cols + column]
  for (int cb = 0; cb < gc; cb += g_fit) {
    const int n = min(g_fit, gc - cb);
    stage_windows<T, A>(src0 + (int64_t)cb * plane, plane, Ws, planes, w, n, vec);
    __syncthreads();
    if (px.dense) {
      for (int cc0 = 0; cc0 < n; cc0 += CH) {
        const A *wc = win0 + (size_t)cc0 * w.size;
        const int y0c = px.y0c, cols = w.cols, wsz = w.size;
        auto row = [=](int cc, int r) { return wc + cc * wsz + clampi(y0c + r, 0, Hs - 1) * cols; };
        be_fwd_dense_chunk<T, A, K, CH>(row, min(CH, n - cc0), px, yf, out0 + (int64_t)(cb + cc0) * oplane, oplane, Wo, active);
      }
    } else if (active) {
      for (int c = 0; c < n; ++c) {
        const A *wc = win0 + (size_t)c * w.size;
        const int cols = w.cols;
        auto at = [=](int yrow) { return wc + yrow * cols; };
        be_fwd_taps_channel<T, A, K>(at, Hs, px.xL, px.xR, px.ax, px.fy0, yf, out0 + (int64_t)(cb + c) * oplane, Wo);
      }
    }
    __syncthreads();
  }
}

template <typename T, int K>
static int launch_fwd_big(const T *src, const T *flow, T *out, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf,
                          int64_t Wf, hipStream_t stream, bool *done) {
  using A = typename Num<T>::acc;
  // (K = 3: 4 -> 2 measured 28.4 -> 27.2 us; 2 -> 1 in the tile kernel 28.1 -> 26.7 on a smooth flow, k = 5 on a wild flow 143 ->
  // 120, the other flows within 3 % either way: session s32)
  constexpr int CH = sizeof(A) == 8 ? (K >= 4 ? 1 : 2) : (K >= 3 ? 2 : 4);
  constexpr int CH_TILE = sizeof(A) == 8 ? CH : (K >= 3 ? 1 : 4);
  *done = false;
  if (Hs * Ws > 0x3fffffffLL) return GFLA_OK;
  if (tuning(38) == 1) {   // first version: global gathers, lane = pixel, four channel ranges per workgroup
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
  const BigGeo bg = big_geometry(0, B, C, Hf, Wf, K + 1, (int)sizeof(A));
  const TileGeo tg = bg.tg;
  const int G = bg.G;
  const int64_t ngroups = bg.ngroups, nwg = bg.nwg;
  if (nwg > 0x7fffffffLL) return GFLA_OK;
  const unsigned lds_bytes = bg.lds_bytes;
#define GFLA_BE_FWD_TILE(CH_)                                                                                                       \
  launch_lds(be_fwd_tile_kernel<T, K, CH_>, dim3((unsigned)nwg), dim3((unsigned)tg.threads), lds_bytes, stream, src, flow, out, (int)C, \
             (int)Hs, (int)Ws, (int)Hf, (int)Wf, tg.th, tg.tw, tg.ntx, tg.nty, G, (int)ngroups, (int)(lds_bytes / sizeof(A)), nwg)
  if constexpr (sizeof(A) == 4) {   // tuning key 40: channels evaluated together per pixel (registers against requests in flight)
    if (tuning(40) == 1) GFLA_BE_FWD_TILE(1);
    else if (tuning(40) == 2) GFLA_BE_FWD_TILE(2);
    else if (tuning(40) == 4) GFLA_BE_FWD_TILE(4);
    else GFLA_BE_FWD_TILE(CH_TILE);
  } else {
    GFLA_BE_FWD_TILE(CH_TILE);
  }
#undef GFLA_BE_FWD_TILE
  *done = true;
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------------
// Where a folded patch row goes: the LDS window (double planes: ds_add_f64) or, for a tile whose reach does not fit, the
// gradient plane itself (float / double atomics).  Both index (plane row, plane column) as base[row * pitch + col].
struct BeWinSink {
  lds_acc_t *base;
  int pitch;
  template <typename A>
  __device__ __forceinline__ void add(int row, int col, A v) const { lds_add(base + row * pitch + col, (lds_acc_t)v); }
};
template <typename T>
struct BeGlobalSink {
  T *base;
  int pitch;
  template <typename A>
  __device__ __forceinline__ void add(int row, int col, A v) const { atomic_add(base + row * pitch + col, (T)v); }
};

// One channel of one flow pixel: fold the K x K gradients into the dense patch and hand its rows to `sink`; accumulate
// d/dflow.  spl[row * spitch + col] = this channel's source value (a window or the plane), gblk = &grad_out[b, c, yf*K, xf*K].
// Neighbouring lanes of a wave whose patches are the same rows shifted by ONE column (the usual case: consecutive flow
// pixels, a flow that changes by less than a pixel between them) send K of their K + 1 patch-row values to the same
// addresses.  With FOLD those are summed across the lanes first -- K `v_add_f32_dpp wave_shr:1` per patch row -- so that a row
// leaves through ONE atomic per lane plus K sparse ones issued only by the lanes at the end of a run (nobody to hand the
// partial sums to).  An LDS atomic instruction costs the CU ~6-9 clocks conflict-free whatever the number of active lanes
// (tools/ubench/lds_atomics_sparse.hip) and 20-40 with the conflicts of real patch rows: the full-wave ones are what is saved.
// prev: the lane below holds the patch one column to the left (its column q + 1 = this lane's column q, same rows, both
// dense and inside the image); next: the lane above is linked to this one.
#ifndef GFLA_BE_BOTH_MAX
#define GFLA_BE_BOTH_MAX 16   // largest K*K whose both-gradient kernel takes the fold and the batched requests
#endif
struct BeLinks {
  bool prev, next;
};
template <typename T, int K>
__device__ __forceinline__ BeLinks be_links(const BePixel<T, K> &px, bool active, bool enable) {
  const int ok = (active && px.dense) ? 1 : 0;
  int same = __builtin_amdgcn_update_dpp(0, ok, 0x138, 0xf, 0xf, false) & ok;   // wave_shr:1 -- lane i reads lane i - 1
  same &= __builtin_amdgcn_update_dpp(0x7fffffff, px.y0c, 0x138, 0xf, 0xf, false) == px.y0c ? 1 : 0;
#pragma unroll
  for (int q = 0; q < K; ++q) same &= __builtin_amdgcn_update_dpp(-7, px.col[q + 1], 0x138, 0xf, 0xf, false) == px.col[q] ? 1 : 0;
  if ((threadIdx.x & 63) == 0 || !enable) same = 0;
  BeLinks l;
  l.prev = same != 0;
  l.next = __builtin_amdgcn_update_dpp(0, same, 0x130, 0xf, 0xf, false) != 0;   // wave_shl:1 -- lane i reads lane i + 1
  return l;
}

// gall[i * K + j] = grad_out[b, c, yf*K + i, xf*K + j]: the caller requests a channel's (or two channels') K x K block in ONE
// batch (be_load_gblock + pin_regs).  With a row's K values requested per row, as the first version did, a wave has 12-20 bytes
// per lane in flight, a CU 12 KB, the chip 3 MB -- a quarter of what 8 TB/s x ~1.5 us of latency needs: d/dflow alone ran at
// 2.3 TB/s of gradient reads whatever the tile shape (profiles/r5_config2_sweeps.txt).
template <typename T, int K>
__device__ __forceinline__ void be_load_gblock(const T *__restrict__ gblk, int Wo, typename Num<T>::acc *g) {
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) g[i * K + j] = Num<T>::ld(gblk + i * Wo + j);
}
// BATCH false: gall == nullptr, a row's K gradients are requested per row from gblk (what the k = 5 kernel with BOTH gradients
// keeps: the block in registers costs it 169 registers, or 128 and spills -- measured 166 against 160 us).
template <typename T, typename P, int K, bool NEED_SRC, bool NEED_FLOW, bool FOLD = false, bool BATCH = true, typename Sink>
__device__ __forceinline__ void be_bwd_pixel_dense(const Sink &sink, const P *__restrict__ spl, int spitch,
                                                   const typename Num<T>::acc *gall, const T *__restrict__ gblk, int Wo, int Hs,
                                                   const BePixel<T, K> &px, int yf, typename Num<T>::acc &gx_acc,
                                                   typename Num<T>::acc &gy_acc, BeLinks lk = BeLinks{false, false}) {
  using A = typename Num<T>::acc;
  // one patch row out: plain, or folded across linked lanes (see BeLinks)
  auto emit = [&](int r, const A (&rw)[K + 1]) {
    if constexpr (FOLD) {
      A t = rw[K];
#pragma unroll
      for (int q = K - 1; q >= 0; --q) {
        if (!lk.next && t != 0) sink.add(r, px.col[q + 1], t);
        const A from_prev = __builtin_amdgcn_update_dpp((A)0, t, 0x138, 0xf, 0xf, false);
        t = rw[q] + (lk.prev ? from_prev : (A)0);
      }
      if (t != 0) sink.add(r, px.col[0], t);
    } else {
#pragma unroll
      for (int q = 0; q <= K; ++q)
        if (rw[q] != 0) sink.add(r, px.col[q], rw[q]);
    }
  };
  A rowA[K + 1], vA[K + 1];
  int rA = clampi(px.y0c, 0, Hs - 1);
#pragma unroll
  for (int q = 0; q <= K; ++q) {
    rowA[q] = 0;
    vA[q] = NEED_FLOW ? (A)Num<P>::ld(spl + rA * spitch + px.col[q]) : (A)0;
  }
  // The row loop stays rolled (unrolled it costs 146 / 256+ registers at k = 3 / 5 against 100 / 122): the block is kept in
  // registers and ROTATED by one row per iteration -- K (K - 1) moves per row against the ~30 K operations of a row.
  A gq[BATCH ? K * K : 1];
  if constexpr (BATCH) {
#pragma unroll
    for (int e = 0; e < K * K; ++e) gq[e] = gall[e];
  }
#pragma unroll 1
  for (int i = 0; i < K; ++i) {
    const A dy = (px.fy0 + (A)(i - K / 2)) + (A)yf;  // block_extractor_kernel.cu:132-136
    const A yB_P = dy - floor_t<A>(dy), yT_P = 1 - yB_P;
    const int rB = clampi(px.y0c + i + 1, 0, Hs - 1);
    A gv[K], rowB[K + 1], vB[K + 1];
    if constexpr (BATCH) {
#pragma unroll
      for (int j = 0; j < K; ++j) gv[j] = gq[j];
#pragma unroll
      for (int e = 0; e + K < K * K; ++e) gq[e] = gq[e + K];
      if constexpr (NEED_FLOW) {   // the next patch row of the source, requested together (pin_regs.h)
#pragma unroll
        for (int q = 0; q <= K; ++q) vB[q] = (A)Num<P>::ld(spl + rB * spitch + px.col[q]);
        pin_regs(vB);
      }
    } else {   // this row's K incoming gradients and (for d/dflow) the next patch row, requested together
      A ld_[NEED_FLOW ? 2 * K + 1 : K];
#pragma unroll
      for (int j = 0; j < K; ++j) ld_[j] = Num<T>::ld(gblk + i * Wo + j);
      if constexpr (NEED_FLOW) {
#pragma unroll
        for (int q = 0; q <= K; ++q) ld_[K + q] = (A)Num<P>::ld(spl + rB * spitch + px.col[q]);
      }
      pin_regs(ld_);
#pragma unroll
      for (int j = 0; j < K; ++j) gv[j] = ld_[j];
      if constexpr (NEED_FLOW) {
#pragma unroll
        for (int q = 0; q <= K; ++q) vB[q] = ld_[K + q];
      }
    }
#pragma unroll
    for (int q = 0; q <= K; ++q) {
      rowB[q] = 0;
      if constexpr (!NEED_FLOW) vB[q] = 0;
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const A xL_P = 1 - px.ax[j], xR_P = px.ax[j];
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
    // (the test for zero stays: dropping it -- an exec-mask round trip per atomic -- measured 67 -> 66 us (k = 3) on a smooth
    // flow and costs integer flows 16 atomics instead of 9)
    if (NEED_SRC) emit(rA, rowA);
#pragma unroll
    for (int q = 0; q <= K; ++q) {
      rowA[q] = rowB[q];
      vA[q] = vB[q];
    }
    rA = rB;
  }
  if (NEED_SRC) emit(rA, rowA);
}

// the reference's own tap-by-tap form (a tap's floor() landed one off the dense patch)
template <typename T, typename P, int K, bool NEED_SRC, bool NEED_FLOW, typename Sink>
__device__ __forceinline__ void be_bwd_pixel_taps(const Sink &sink, const P *__restrict__ spl, int spitch,
                                                  const T *__restrict__ gblk, int Wo, int Hs, const BePixel<T, K> &px, int yf,
                                                  typename Num<T>::acc &gx_acc, typename Num<T>::acc &gy_acc) {
  using A = typename Num<T>::acc;
#pragma unroll 1
  for (int i = 0; i < K; ++i) {
    const A dy = (px.fy0 + (A)(i - K / 2)) + (A)yf;
    const A fdy = floor_t<A>(dy);
    const int yT = clampi((int)fdy, 0, Hs - 1), yB = clampi((int)(fdy + 1), 0, Hs - 1);
    const A yB_P = dy - fdy, yT_P = 1 - yB_P;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const A g = Num<T>::ld(gblk + i * Wo + j);
      const A xL_P = 1 - px.ax[j], xR_P = px.ax[j];
      const int xL = px.xL[j], xR = px.xR[j];
      if (NEED_FLOW) {
        const A vTL = Num<P>::ld(spl + yT * spitch + xL), vTR = Num<P>::ld(spl + yT * spitch + xR);
        const A vBL = Num<P>::ld(spl + yB * spitch + xL), vBR = Num<P>::ld(spl + yB * spitch + xR);
        gy_acc += g * (-xL_P * vTL - xR_P * vTR + xL_P * vBL + xR_P * vBR);
        gx_acc += g * (-yT_P * vTL - yB_P * vBL + yT_P * vTR + yB_P * vBR);
      }
      if (NEED_SRC) {
        sink.add(yT, xL, g * xL_P * yT_P);
        sink.add(yT, xR, g * xR_P * yT_P);
        sink.add(yB, xL, g * xL_P * yB_P);
        sink.add(yB, xR, g * xR_P * yB_P);
      }
    }
  }
}

template <typename T, int K, bool NEED_SRC, bool NEED_FLOW>
__global__ __launch_bounds__(512, (sizeof(T) == 4 ? 4 : 2)) void be_bwd_tile_kernel(const T *__restrict__ src, const T *__restrict__ flow,
                                                         const T *__restrict__ gout, T *__restrict__ gsrc,
                                                         typename Num<T>::acc *__restrict__ gflow, int C, int Hs, int Ws,
                                                         int Hf, int Wf, int th, int tw, int ntx, int nty, int G, int ngroups,
                                                         int lds_bytes, int64_t nwg, int abl, int fold) {
  // fold: patch rows summed across linked lanes before the atomics (BeLinks; tuning key 41 = 1 turns it off)
  // abl (tuning key 39, timing ablations, results garbage): 1 = stop after setup / box, 2 = no pixel loop, 4 = no flush,
  // 8 = no staging of the source window
  using A = typename Num<T>::acc;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
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
  const int p = active ? yf * Wf + xf : 0;
  box_init(s_box);
  __syncthreads();
  BePixel<T, K> px;
  px.init(flow + (int64_t)b * 2 * HW, HW, p, xf, yf, Hs, Ws, active);
  {
    int ylo, xlo, yhi, xhi;
    px.reach(Hs, Ws, active, ylo, xlo, yhi, xhi);
    box_reduce(s_box, ylo, xlo, yhi, xhi);
  }
  __syncthreads();
  const bool vec = NEED_FLOW && window_vec_ok(src + ((int64_t)b * C + g * G) * plane, plane, Ws);
  const TileWin w = tile_window_vec(s_box, Ws, vec);
  // float patch rows (one DPP add per fold); not where both gradients at k >= 5 share the registers: that kernel measured
  // slower with either change (zero flow 111 -> 123 us with the fold, -> 169 with the fold at 168 registers) and keeps the first form
  constexpr bool kFold = NEED_SRC && std::is_same<A, float>::value && !(NEED_FLOW && K * K > GFLA_BE_BOTH_MAX);
  BeLinks lk{false, false};
  if constexpr (kFold) lk = be_links<T, K>(px, active, fold != 0);
  constexpr int kPerElem = (NEED_SRC ? (int)sizeof(lds_acc_t) : 0) + (NEED_FLOW ? (int)sizeof(A) : 0);
  const int g_fit = min(gc, lds_bytes / max(w.size * kPerElem, 1));
  const T *src0 = src + ((int64_t)b * C + c0) * plane;
  T *gsrc0 = NEED_SRC ? gsrc + ((int64_t)b * C + c0) * plane : nullptr;
  const T *gblk0 = gout + ((int64_t)b * C + c0) * oplane + (int64_t)(yf * K) * Wo + xf * K;
  A gx_acc = 0, gy_acc = 0;
  if (abl & 1) {
    if (active && px.fx0 == (A)-1.2345e30) gflow[p] = px.ax[0];
    return;
  }
  if (g_fit == 0) {
    // the tile reaches further than one channel's window holds: global memory for this tile
    if (active) {
      for (int c = 0; c < gc; ++c) {
        BeGlobalSink<T> sink{gsrc0 + (int64_t)c * plane, Ws};
        if (px.dense)   // (requests per row: this rare path must not set the kernel's register count)
          be_bwd_pixel_dense<T, T, K, NEED_SRC, NEED_FLOW, false, false>(sink, src0 + (int64_t)c * plane, Ws, nullptr,
                                                                         gblk0 + (int64_t)c * oplane, Wo, Hs, px, yf, gx_acc, gy_acc);
        else
          be_bwd_pixel_taps<T, T, K, NEED_SRC, NEED_FLOW>(sink, src0 + (int64_t)c * plane, Ws, gblk0 + (int64_t)c * oplane, Wo, Hs,
                                                          px, yf, gx_acc, gy_acc);
      }
    }
  } else {
    // LDS: [g_fit windows of double accumulators][g_fit windows of source values]
    lds_acc_t *gplanes = reinterpret_cast<lds_acc_t *>(gfla_smem);
    A *splanes = reinterpret_cast<A *>(gfla_smem + (NEED_SRC ? sizeof(lds_acc_t) * (size_t)g_fit * w.size : 0));
    const int shift = w.ymin * w.cols + w.xmin;
    for (int cb = 0; cb < gc; cb += g_fit) {
      const int n = min(g_fit, gc - cb);
      if (NEED_SRC) zero_planes<lds_acc_t>(gplanes, n * w.size);
      if (NEED_FLOW && !(abl & 8)) stage_windows<T, A>(src0 + (int64_t)cb * plane, plane, Ws, splanes, w, n, vec);
      __syncthreads();
      if (active && !(abl & 2)) {
        // a channel's K x K gradient block in ONE batch of requests (be_load_gblock), except where both gradients at k >= 5
        // leave no registers for it
        constexpr bool kBatch = !(NEED_SRC && NEED_FLOW && K * K > GFLA_BE_BOTH_MAX);
        for (int c = 0; c < n; ++c) {
          const T *gb = gblk0 + (int64_t)(cb + c) * oplane;
          A gl[kBatch ? K * K : 1];
          if constexpr (kBatch) {
            be_load_gblock<T, K>(gb, Wo, gl);
            pin_regs(gl);
          }
          BeWinSink sink{gplanes + (size_t)c * w.size - shift, w.cols};
          const A *spl = splanes + (size_t)c * w.size - shift;
          if (px.dense)
            be_bwd_pixel_dense<T, A, K, NEED_SRC, NEED_FLOW, kFold, kBatch>(sink, spl, w.cols, gl, gb, Wo, Hs, px, yf, gx_acc, gy_acc, lk);
          else
            be_bwd_pixel_taps<T, A, K, NEED_SRC, NEED_FLOW>(sink, spl, w.cols, gb, Wo, Hs, px, yf, gx_acc, gy_acc);
        }
      }
      __syncthreads();
      if (NEED_SRC && !(abl & 4)) {
        flush_windows<T>(gsrc0 + (int64_t)cb * plane, plane, Ws, w, n, [gplanes](int i) { return (double)gplanes[i]; });
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
    const BigGeo bg = big_geometry(1, B, C, Hf, Wf, K + 1, (gsrc ? (int)sizeof(lds_acc_t) : 0) + (gflow ? (int)sizeof(typename Num<T>::acc) : 0));
    const TileGeo tg = bg.tg;
    const int G = bg.G;
    const int64_t ngroups = bg.ngroups, nwg = bg.nwg;
    if (nwg > 0x7fffffffLL) return GFLA_OK;
    const unsigned lds_bytes = bg.lds_bytes;
    const dim3 grid((unsigned)nwg), blk((unsigned)tg.threads);
#define GFLA_BE_TILE_LAUNCH(S, F)                                                                                          \
  launch_lds(be_bwd_tile_kernel<T, K, S, F>, grid, blk, lds_bytes, stream, src, flow, gout, gsrc, gflow, (int)C, (int)Hs,  \
             (int)Ws, (int)Hf, (int)Wf, tg.th, tg.tw, tg.ntx, tg.nty, G, (int)ngroups, (int)lds_bytes, nwg, tile_probe_bits(),     \
             tuning(41) != 1 ? 1 : 0)
    if (gsrc && gflow) GFLA_BE_TILE_LAUNCH(true, true);
    else if (gsrc) GFLA_BE_TILE_LAUNCH(true, false);
    else GFLA_BE_TILE_LAUNCH(false, true);
#undef GFLA_BE_TILE_LAUNCH
    *done = true;
    return launch_status();
  }
}

}  // namespace gfla
