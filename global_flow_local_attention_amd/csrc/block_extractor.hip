// block_extractor for gfx950: k x k bilinear patch gather around (x,y)+flow, and its gradient.
//
// Semantics follow the reference kernels (block_extractor_kernel.cu:20-85 forward, :89-170
// backward): flow channel 0 = x, channel 1 = y, in source pixels; taps are clamped to the
// source plane, weights come from the UN-clamped fractional part; (flow + offset) + index is
// evaluated in that order in the arithmetic type.  The decomposition onto the machine is new:
//
//  forward   one lane owns V consecutive output x of one output row (one 16-byte store) and
//            walks a chunk of channels with the SAME four tap offsets and weights, so the
//            per-pixel index/weight arithmetic is paid once per chunk, the dominant HBM stream
//            (the k^2-times-amplified output) is written in full 128-byte lines, and the
//            4-tap reads of the small source plane are served by L1/L2.
//  backward  one lane owns one flow pixel and a chunk of channels: d/dflow is reduced in
//            registers over the k^2 taps and the channel chunk (1 atomic per lane per component
//            instead of C*k^2 colliding atomics per address), grad_source is scattered with
//            double-precision LDS planes (ds_add_f64) flushed once, or relaxed device-scope float atomics when the plane does not fit.
#include "gfla_common.h"
#include "be_bwd_lds.h"
#include "be_fwd_pix.h"
#include "be_fwd_wrow.h"
#include "be_tile.h"

namespace gfla {

// ----------------------------------------------------------------------------------------
// forward: thread <-> (b, channel chunk, output row y, V consecutive x)
// ----------------------------------------------------------------------------------------
template <typename T, int V>
__global__ __launch_bounds__(kBlock) void be_fwd_rows_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, T *__restrict__ out, int C, int Hs,
    int Ws, int Hf, int Wf, int k, int cpt, int ncg, int sp_blocks) {
  using A = typename Num<T>::acc;
  const int Wo = k * Wf, Ho = k * Hf, WG = Wo / V;
  const int sp_blk = blockIdx.x % sp_blocks;
  const int bc = blockIdx.x / sp_blocks;
  const int sp = sp_blk * kBlock + threadIdx.x;
  if (sp >= Ho * WG) return;
  const int b = bc / ncg, cg = bc - b * ncg;
  const int c0 = cg * cpt;
  const int c1 = min(C, c0 + cpt);
  const int y = sp / WG;
  const int x0 = (sp - y * WG) * V;
  const int yf = y / k;
  const int oy = (y - yf * k) - k / 2;

  int off[V][4];
  A w[V][4];
  const T *flow_x = flow + ((int64_t)(b * 2 + 0) * Hf + yf) * Wf;
  const T *flow_y = flow + ((int64_t)(b * 2 + 1) * Hf + yf) * Wf;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    const int x = x0 + e;
    const int xf = x / k;
    const int ox = (x - xf * k) - k / 2;
    // block_extractor_kernel.cu:62-67
    const A fy = Num<T>::ld(flow_y + xf) + (A)oy;
    const A fx = Num<T>::ld(flow_x + xf) + (A)ox;
    const A dy = fy + (A)yf;
    const A dx = fx + (A)xf;
    const A fdx = floor_t<A>(dx), fdy = floor_t<A>(dy);
    // :69-76
    const int xL = clampi((int)fdx, 0, Ws - 1);
    const int xR = clampi((int)(fdx + 1), 0, Ws - 1);
    const int yT = clampi((int)fdy, 0, Hs - 1);
    const int yB = clampi((int)(fdy + 1), 0, Hs - 1);
    const A xL_P = 1 - (dx - fdx), xR_P = dx - fdx;
    const A yT_P = 1 - (dy - fdy), yB_P = dy - fdy;
    off[e][0] = yT * Ws + xL;
    off[e][1] = yT * Ws + xR;
    off[e][2] = yB * Ws + xL;
    off[e][3] = yB * Ws + xR;
    w[e][0] = xL_P * yT_P;
    w[e][1] = xR_P * yT_P;
    w[e][2] = xL_P * yB_P;
    w[e][3] = xR_P * yB_P;
  }

  const int64_t plane_sz = (int64_t)Hs * Ws;
  const int64_t oplane_sz = (int64_t)Ho * Wo;
  const T *plane = src + ((int64_t)b * C + c0) * plane_sz;
  T *orow = out + ((int64_t)b * C + c0) * oplane_sz + (int64_t)y * Wo + x0;
  for (int c = c0; c < c1; ++c) {
    Pack<T, V> r;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      // :78-84, same order of accumulation
      A s = w[e][0] * Num<T>::ld(plane + off[e][0]);
      s += w[e][1] * Num<T>::ld(plane + off[e][1]);
      s += w[e][2] * Num<T>::ld(plane + off[e][2]);
      s += w[e][3] * Num<T>::ld(plane + off[e][3]);
      r.v[e] = Num<T>::from(s);
    }
    *reinterpret_cast<Pack<T, V> *>(orow) = r;
    plane += plane_sz;
    orow += oplane_sz;
  }
}

template <typename T, int V>
static int launch_fwd_rows(const T *src, const T *flow, T *out, int64_t B, int64_t C, int64_t Hs,
                           int64_t Ws, int64_t Hf, int64_t Wf, int k, hipStream_t stream) {
  const int64_t Ho = k * Hf, Wo = k * Wf;
  const int64_t sp = Ho * (Wo / V);
  const int64_t sp_blocks = ceil_div(sp, kBlock);
  int cpt = tuning(1) > 0 ? tuning(1) : pick_channels_per_thread(sp_blocks * kBlock, C, B, 16);
  if (cpt > C) cpt = (int)C;
  const int64_t ncg = ceil_div(C, cpt);
  const int64_t blocks = sp_blocks * ncg * B;
  if (blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((be_fwd_rows_kernel<T, V>), dim3((unsigned)blocks), dim3(kBlock), 0, stream, src,
                     flow, out, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, k, cpt, (int)ncg,
                     (int)sp_blocks);
  return launch_status();
}

// ----------------------------------------------------------------------------------------
// forward, planes in LDS: workgroup <-> (b, group of G channels[, 1/split of the rows]).
// The G source planes are read from HBM once (16 B per lane, coalesced) into LDS; every lane then
// produces V consecutive outputs of one output row per channel: 4 ds_read_b32 + 4 FMA per
// output, one 16-byte store per channel.  HBM sees only the minimum traffic: source once,
// output once.
// ----------------------------------------------------------------------------------------
template <typename T, int V, bool WIN>
__global__ __launch_bounds__(kLdsThreads) void be_fwd_lds_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, T *__restrict__ out, int C, int Hs,
    int Ws, int Hf, int Wf, int k, int G, int ngroups, int split, int per, int margin) {
  using A = typename Num<T>::acc;
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
  const int Wo = k * Wf, Ho = k * Hf, WG = Wo / V;
  const int npos = Ho * WG;
  const int p_begin = sp * per;
  const int p_end = min(npos, p_begin + per);
  if (p_begin >= p_end) return;
  // rows of the source plane this workgroup keeps in LDS (all of them when margin < 0)
  const Window win = make_window((p_begin / WG) / k, ((p_end - 1) / WG) / k, k / 2, k - k / 2, WIN ? margin : -1, Hs);
  const int win_sz = win.rows * Ws;
  const T *gsrc0 = src + ((int64_t)b * C + c0) * plane_sz;
  for (int c = 0; c < gc; ++c)
    stage_planes<T, A>(gsrc0 + (int64_t)c * plane_sz + win.lo * Ws, planes + (size_t)c * win_sz, win_sz);
  __syncthreads();
  const A *lds0 = planes - win.lo * Ws;  // so that plane-relative offsets index the window
  const int64_t oplane_sz = (int64_t)Ho * Wo;
  for (int pos = p_begin + threadIdx.x; pos < p_end; pos += blockDim.x) {
    const int y = pos / WG;
    const int x0 = (pos - y * WG) * V;
    const int yf = y / k;
    const int oy = (y - yf * k) - k / 2;
    int off[V][4];
    A w[V][4];
    bool inside[V];
    const T *flow_x = flow + ((int64_t)(b * 2 + 0) * Hf + yf) * Wf;
    const T *flow_y = flow + ((int64_t)(b * 2 + 1) * Hf + yf) * Wf;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const int x = x0 + e;
      const int xf = x / k;
      const int ox = (x - xf * k) - k / 2;
      const A fy = Num<T>::ld(flow_y + xf) + (A)oy;  // block_extractor_kernel.cu:62-67
      const A fx = Num<T>::ld(flow_x + xf) + (A)ox;
      const A dy = fy + (A)yf;
      const A dx = fx + (A)xf;
      const A fdx = floor_t<A>(dx), fdy = floor_t<A>(dy);
      const int xL = clampi((int)fdx, 0, Ws - 1);  // :69-76
      const int xR = clampi((int)(fdx + 1), 0, Ws - 1);
      const int yT = clampi((int)fdy, 0, Hs - 1);
      const int yB = clampi((int)(fdy + 1), 0, Hs - 1);
      const A xL_P = 1 - (dx - fdx), xR_P = dx - fdx;
      const A yT_P = 1 - (dy - fdy), yB_P = dy - fdy;
      inside[e] = !WIN || (yT >= win.lo && yB < win.lo + win.rows);
      off[e][0] = yT * Ws + xL;
      off[e][1] = yT * Ws + xR;
      off[e][2] = yB * Ws + xL;
      off[e][3] = yB * Ws + xR;
      w[e][0] = xL_P * yT_P;
      w[e][1] = xR_P * yT_P;
      w[e][2] = xL_P * yB_P;
      w[e][3] = xR_P * yB_P;
    }
    T *orow = out + ((int64_t)b * C + c0) * oplane_sz + (int64_t)y * Wo + x0;
    const A *pl = lds0;
    const T *gpl = gsrc0;
    for (int c = 0; c < gc; ++c) {
      Pack<T, V> r;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        A v0, v1, v2, v3;
        if (inside[e]) {
          v0 = pl[off[e][0]]; v1 = pl[off[e][1]]; v2 = pl[off[e][2]]; v3 = pl[off[e][3]];
        } else {  // flow larger than the window's margin: this element gathers from global memory
          v0 = Num<T>::ld(gpl + off[e][0]); v1 = Num<T>::ld(gpl + off[e][1]);
          v2 = Num<T>::ld(gpl + off[e][2]); v3 = Num<T>::ld(gpl + off[e][3]);
        }
        A s = w[e][0] * v0;  // :78-84, same order of accumulation
        s += w[e][1] * v1;
        s += w[e][2] * v2;
        s += w[e][3] * v3;
        r.v[e] = Num<T>::from(s);
      }
      *reinterpret_cast<Pack<T, V> *>(orow) = r;
      pl += win_sz;
      gpl += plane_sz;
      orow += oplane_sz;
    }
  }
}

template <typename T, int V>
static int launch_fwd(const T *src, const T *flow, T *out, int64_t B, int64_t C, int64_t Hs, int64_t Ws,
                      int64_t Hf, int64_t Wf, int k, hipStream_t stream) {
  using A = typename Num<T>::acc;
  const int variant = tuning(0);
  // key 0: 0 auto, 1 global-gather kernel, 2 round 1's planes-in-LDS kernel (lane = output quad), 3 the lane-per-pixel
  // kernel with direct stores (be_fwd_pix.h), 4 the wave-per-flow-row kernel (be_fwd_wrow.h).
  // auto: flow rows of up to 64 pixels go to the wave-per-flow-row kernel (contiguous pieces whatever the width: 4.6-5.6
  // TB/s), wider ones to the lane-per-pixel kernel (5.2 TB/s when its output rows are whole 128-byte lines, 3.8-4.7
  // otherwise), planes beyond the LDS budget to round 1's windowed kernel
  if constexpr (sizeof(T) >= 4) {
    // few planes, each far beyond the LDS budget (BASELINE configs[1]): parallelism from pixel blocks, not planes (be_tile.h)
    if (variant == 0 && k >= 2 && k <= 5 && big_plane_regime(B, C, Hs * Ws * (int64_t)sizeof(A), lds_budget())) {
      bool done = false;
      int st = GFLA_OK;
      switch (k) {
        case 2: st = launch_fwd_big<T, 2>(src, flow, out, B, C, Hs, Ws, Hf, Wf, stream, &done); break;
        case 3: st = launch_fwd_big<T, 3>(src, flow, out, B, C, Hs, Ws, Hf, Wf, stream, &done); break;
        case 4: st = launch_fwd_big<T, 4>(src, flow, out, B, C, Hs, Ws, Hf, Wf, stream, &done); break;
        default: st = launch_fwd_big<T, 5>(src, flow, out, B, C, Hs, Ws, Hf, Wf, stream, &done); break;
      }
      if (done) note_path(GFLA_PATH_BE_FWD_GPIX);
      if (done || st != GFLA_OK) return st;
    }
    // small output planes (<= 32 KB: a whole plane is a few hundred lines that one workgroup writes within microseconds)
    // stream slightly faster from the lane-per-pixel kernel: (32,256,32,22) k=3 40 us against 41-45
    const bool small_plane = (int64_t)k * k * Hf * Wf * (int64_t)sizeof(T) <= 32 * 1024 && Wf <= 64;
    if ((variant == 4 || (variant == 0 && !small_plane)) && k >= 2 && k <= 5) {
      bool done = false;
      int st = GFLA_OK;
      switch (k) {
        case 2: st = launch_fwd_wrow<T, 2>(src, flow, out, B, C, Hs, Ws, Hf, Wf, stream, &done); break;
        case 3: st = launch_fwd_wrow<T, 3>(src, flow, out, B, C, Hs, Ws, Hf, Wf, stream, &done); break;
        case 4: st = launch_fwd_wrow<T, 4>(src, flow, out, B, C, Hs, Ws, Hf, Wf, stream, &done); break;
        default: st = launch_fwd_wrow<T, 5>(src, flow, out, B, C, Hs, Ws, Hf, Wf, stream, &done); break;
      }
      if (done) note_path(GFLA_PATH_BE_FWD_PIX);
      if (done || st != GFLA_OK) return st;
    }
    if (variant != 1 && variant != 2 && variant != 4 && k >= 2 && k <= 5) {  // lane = flow pixel, direct stores (be_fwd_pix.h)
      bool done = false;
      int st = GFLA_OK;
      switch (k) {
        case 2: st = launch_fwd_pix<T, 2>(src, flow, out, B, C, Hs, Ws, Hf, Wf, stream, &done); break;
        case 3: st = launch_fwd_pix<T, 3>(src, flow, out, B, C, Hs, Ws, Hf, Wf, stream, &done); break;
        case 4: st = launch_fwd_pix<T, 4>(src, flow, out, B, C, Hs, Ws, Hf, Wf, stream, &done); break;
        default: st = launch_fwd_pix<T, 5>(src, flow, out, B, C, Hs, Ws, Hf, Wf, stream, &done); break;
      }
      if (done) note_path(GFLA_PATH_BE_FWD_PIX);
      if (done || st != GFLA_OK) return st;
    }
  }
  if (variant != 1) {
    // work items = output positions; one flow row = k output rows of (k*Wf)/V positions
    PlaneGeo g = lds_geometry(Hs, Ws, sizeof(A), B, C, Hf, (int64_t)k * ((k * Wf) / V), k + 1, 1);
    if (g.G > 0) {
      const int64_t blocks = B * g.ngroups * g.split;
      if (blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
      if (g.margin < 0)
        launch_lds(be_fwd_lds_kernel<T, V, false>, dim3((unsigned)blocks), dim3(kLdsThreads), g.lds_bytes, stream, 
            src, flow, out, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, k, g.G, g.ngroups, g.split, g.per, g.margin);
      else
        launch_lds(be_fwd_lds_kernel<T, V, true>, dim3((unsigned)blocks), dim3(kLdsThreads), g.lds_bytes, stream, 
            src, flow, out, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, k, g.G, g.ngroups, g.split, g.per, g.margin);
      return launch_status();
    }
  }
  return launch_fwd_rows<T, V>(src, flow, out, B, C, Hs, Ws, Hf, Wf, k, stream);
}

template <typename T>
static int block_extractor_fwd(const T *src, const T *flow, T *out, int64_t B, int64_t C, int64_t Hs,
                               int64_t Ws, int64_t Hf, int64_t Wf, int k, gfla_stream_t stream_) {
  if (!src || !flow || !out) return GFLA_ERR_NULL_POINTER;
  if (B <= 0 || C <= 0 || Hs <= 0 || Ws <= 0 || Hf <= 0 || Wf <= 0 || k < 1) return GFLA_ERR_BAD_SHAPE;
  if (Hs * Ws > 0x7fffffffLL || (k * Hf) * (k * Wf) > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  constexpr int VMAX = 16 / sizeof(T) > 4 ? 4 : 16 / sizeof(T);  // f32:4  f64:2  bf16:4 (8-byte store)
  const int64_t Wo = k * Wf;
  const bool aligned = (reinterpret_cast<uintptr_t>(out) % (VMAX * sizeof(T))) == 0;
  if (aligned && Wo % VMAX == 0)
    return launch_fwd<T, VMAX>(src, flow, out, B, C, Hs, Ws, Hf, Wf, k, stream);
  if (VMAX == 4 && (reinterpret_cast<uintptr_t>(out) % (2 * sizeof(T))) == 0 && Wo % 2 == 0)
    return launch_fwd<T, 2>(src, flow, out, B, C, Hs, Ws, Hf, Wf, k, stream);
  return launch_fwd<T, 1>(src, flow, out, B, C, Hs, Ws, Hf, Wf, k, stream);
}

// ----------------------------------------------------------------------------------------
// backward, compile-time K: thread <-> (b, channel chunk, flow pixel)
// ----------------------------------------------------------------------------------------
template <typename T, int K>
__global__ __launch_bounds__(kBlock) void be_bwd_pix_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, const T *__restrict__ gout,
    T *__restrict__ gsrc, T *__restrict__ gflow, int C, int Hs, int Ws, int Hf, int Wf, int cpt,
    int ncg, int sp_blocks) {
  using A = typename Num<T>::acc;
  const int sp_blk = blockIdx.x % sp_blocks;
  const int bc = blockIdx.x / sp_blocks;
  const int p = sp_blk * kBlock + threadIdx.x;
  if (p >= Hf * Wf) return;
  const int b = bc / ncg, cg = bc - b * ncg;
  const int c0 = cg * cpt, c1 = min(C, c0 + cpt);
  const int yf = p / Wf, xf = p - yf * Wf;
  const int Wo = K * Wf;
  const int64_t oplane_sz = (int64_t)K * Hf * Wo;
  const int64_t plane_sz = (int64_t)Hs * Ws;

  const A fx0 = Num<T>::ld(flow + ((int64_t)(b * 2 + 0) * Hf + yf) * Wf + xf);
  const A fy0 = Num<T>::ld(flow + ((int64_t)(b * 2 + 1) * Hf + yf) * Wf + xf);
  int xL[K], xR[K], yT[K], yB[K];
  A ax[K], ay[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {
    const A dx = (fx0 + (A)(t - K / 2)) + (A)xf;  // block_extractor_kernel.cu:132-136
    const A dy = (fy0 + (A)(t - K / 2)) + (A)yf;
    const A fdx = floor_t<A>(dx), fdy = floor_t<A>(dy);
    xL[t] = clampi((int)fdx, 0, Ws - 1);
    xR[t] = clampi((int)(fdx + 1), 0, Ws - 1);
    yT[t] = clampi((int)fdy, 0, Hs - 1) * Ws;
    yB[t] = clampi((int)(fdy + 1), 0, Hs - 1) * Ws;
    ax[t] = dx - fdx;
    ay[t] = dy - fdy;
  }

  A gx_acc = 0, gy_acc = 0;
  const T *plane = src + ((int64_t)b * C + c0) * plane_sz;
  T *gplane = gsrc ? gsrc + ((int64_t)b * C + c0) * plane_sz : nullptr;
  const T *gblk = gout + ((int64_t)b * C + c0) * oplane_sz + (int64_t)(yf * K) * Wo + xf * K;
  for (int c = c0; c < c1; ++c) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const A yT_P = 1 - ay[i], yB_P = ay[i];
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const A xL_P = 1 - ax[j], xR_P = ax[j];
        const A g = Num<T>::ld(gblk + i * Wo + j);
        if (gflow) {
          const A vTL = Num<T>::ld(plane + yT[i] + xL[j]);
          const A vTR = Num<T>::ld(plane + yT[i] + xR[j]);
          const A vBL = Num<T>::ld(plane + yB[i] + xL[j]);
          const A vBR = Num<T>::ld(plane + yB[i] + xR[j]);
          // :163-164
          gy_acc += g * (-xL_P * vTL - xR_P * vTR + xL_P * vBL + xR_P * vBR);
          gx_acc += g * (-yT_P * vTL - yB_P * vBL + yT_P * vTR + yB_P * vBR);
        }
        if (gplane) {  // :158-161
          atomic_add(gplane + yT[i] + xL[j], g * xL_P * yT_P);
          atomic_add(gplane + yT[i] + xR[j], g * xR_P * yT_P);
          atomic_add(gplane + yB[i] + xL[j], g * xL_P * yB_P);
          atomic_add(gplane + yB[i] + xR[j], g * xR_P * yB_P);
        }
      }
    }
    plane += plane_sz;
    if (gplane) gplane += plane_sz;
    gblk += oplane_sz;
  }
  if (gflow) {
    T *gfx = gflow + ((int64_t)(b * 2 + 0) * Hf + yf) * Wf + xf;
    T *gfy = gflow + ((int64_t)(b * 2 + 1) * Hf + yf) * Wf + xf;
    if (ncg == 1) {  // sole writer of this flow pixel
      *gfx = *gfx + gx_acc;
      *gfy = *gfy + gy_acc;
    } else {
      atomic_add(gfx, gx_acc);
      atomic_add(gfy, gy_acc);
    }
  }
}

// backward, run-time k (any kernel size): thread <-> one grad_out element, the reference's own
// decomposition (block_extractor_kernel.cu:110-168), kept as the fallback for unusual k.
template <typename T>
__global__ __launch_bounds__(kBlock) void be_bwd_elem_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, const T *__restrict__ gout,
    T *__restrict__ gsrc, T *__restrict__ gflow, int64_t n, int C, int Hs, int Ws, int Hf, int Wf,
    int k) {
  using A = typename Num<T>::acc;
  const int64_t index = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (index >= n) return;
  const int Wo = k * Wf, Ho = k * Hf;
  const int x = (int)(index % Wo);
  const int y = (int)((index / Wo) % Ho);
  const int64_t bcI = index / ((int64_t)Wo * Ho);
  const int b = (int)(bcI / C);
  const int yf = y / k, xf = x / k;
  const A fy = Num<T>::ld(flow + ((int64_t)(b * 2 + 1) * Hf + yf) * Wf + xf) + (A)(y % k - k / 2);
  const A fx = Num<T>::ld(flow + ((int64_t)(b * 2 + 0) * Hf + yf) * Wf + xf) + (A)(x % k - k / 2);
  const A dy = fy + (A)yf, dx = fx + (A)xf;
  const A fdx = floor_t<A>(dx), fdy = floor_t<A>(dy);
  const int xL = clampi((int)fdx, 0, Ws - 1), xR = clampi((int)(fdx + 1), 0, Ws - 1);
  const int yT = clampi((int)fdy, 0, Hs - 1), yB = clampi((int)(fdy + 1), 0, Hs - 1);
  const A xL_P = 1 - (dx - fdx), xR_P = dx - fdx, yT_P = 1 - (dy - fdy), yB_P = dy - fdy;
  const int64_t pbase = bcI * Hs * Ws;
  const A g = Num<T>::ld(gout + index);
  if (gsrc) {
    atomic_add(gsrc + pbase + (int64_t)yT * Ws + xL, g * xL_P * yT_P);
    atomic_add(gsrc + pbase + (int64_t)yT * Ws + xR, g * xR_P * yT_P);
    atomic_add(gsrc + pbase + (int64_t)yB * Ws + xL, g * xL_P * yB_P);
    atomic_add(gsrc + pbase + (int64_t)yB * Ws + xR, g * xR_P * yB_P);
  }
  if (gflow) {
    const A vTL = Num<T>::ld(src + pbase + (int64_t)yT * Ws + xL);
    const A vTR = Num<T>::ld(src + pbase + (int64_t)yT * Ws + xR);
    const A vBL = Num<T>::ld(src + pbase + (int64_t)yB * Ws + xL);
    const A vBR = Num<T>::ld(src + pbase + (int64_t)yB * Ws + xR);
    atomic_add(gflow + ((int64_t)(b * 2 + 1) * Hf + yf) * Wf + xf,
               g * (-xL_P * vTL - xR_P * vTR + xL_P * vBL + xR_P * vBR));
    atomic_add(gflow + ((int64_t)(b * 2 + 0) * Hf + yf) * Wf + xf,
               g * (-yT_P * vTL - yB_P * vBL + yT_P * vTR + yB_P * vBR));
  }
}

template <typename T, int K>
static int launch_bwd_pix(const T *src, const T *flow, const T *gout, T *gsrc, typename Num<T>::acc *gflow,
                          int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf,
                          hipStream_t stream) {
  constexpr bool kBf16 = sizeof(T) == 2;
  if (tuning(2) != 1 && !kBf16 &&
      big_plane_regime(B, C, Hs * Ws * (int64_t)(sizeof(lds_acc_t) + sizeof(typename Num<T>::acc)), lds_budget())) {
    bool done = false;   // few planes far beyond the LDS budget: flow-pixel tiles with bounding-box windows (be_tile.h)
    int st = launch_be_bwd_tile<T, K>(src, flow, gout, gsrc, gflow, B, C, Hs, Ws, Hf, Wf, stream, &done);
    if (done) note_path(GFLA_PATH_BE_BWD_TILE);
    if (done || st != GFLA_OK) return st;
  }
  if (tuning(2) != 1 || kBf16) {
    bool done = false;
    int st = launch_be_bwd_lds<T, K>(kGoutTensor, src, flow, gout, static_cast<const T *>(nullptr), gsrc, gflow, B, C, Hs, Ws, Hf, Wf, stream, &done);
    if (done) note_path(GFLA_PATH_BE_BWD_LDS);
    if (done || st != GFLA_OK) return st;
  }
  if constexpr (kBf16) {
    return GFLA_ERR_UNSUPPORTED;  // bf16 storage: the planes-in-LDS kernel only (global atomics need f32 / f64)
  } else {
  const int64_t sp_blocks = ceil_div(Hf * Wf, kBlock);
  int cpt = tuning(1) > 0 ? tuning(1) : pick_channels_per_thread(sp_blocks * kBlock, C, B, 16, 2 * kNumCU * kWavesPerCU);
  if (cpt > C) cpt = (int)C;
  const int64_t ncg = ceil_div(C, cpt);
  const int64_t blocks = sp_blocks * ncg * B;
  if (blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  note_path(GFLA_PATH_BE_BWD_GLOBAL);
  hipLaunchKernelGGL((be_bwd_pix_kernel<T, K>), dim3((unsigned)blocks), dim3(kBlock), 0, stream, src,
                     flow, gout, gsrc, gflow, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, cpt,
                     (int)ncg, (int)sp_blocks);
  return launch_status();
  }
}

template <typename T>
static int block_extractor_bwd(const T *src, const T *flow, const T *gout, T *gsrc, typename Num<T>::acc *gflow,
                               int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, int k,
                               gfla_stream_t stream_) {
  if (!src || !flow || !gout) return GFLA_ERR_NULL_POINTER;
  if (B <= 0 || C <= 0 || Hs <= 0 || Ws <= 0 || Hf <= 0 || Wf <= 0 || k < 1) return GFLA_ERR_BAD_SHAPE;
  if (Hs * Ws > 0x7fffffffLL || (k * Hf) * (k * Wf) > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  if (!gsrc && !gflow) return GFLA_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  switch (k) {
    case 2: return launch_bwd_pix<T, 2>(src, flow, gout, gsrc, gflow, B, C, Hs, Ws, Hf, Wf, stream);
    case 3: return launch_bwd_pix<T, 3>(src, flow, gout, gsrc, gflow, B, C, Hs, Ws, Hf, Wf, stream);
    case 4: return launch_bwd_pix<T, 4>(src, flow, gout, gsrc, gflow, B, C, Hs, Ws, Hf, Wf, stream);
    case 5: return launch_bwd_pix<T, 5>(src, flow, gout, gsrc, gflow, B, C, Hs, Ws, Hf, Wf, stream);
    default: break;
  }
  if constexpr (sizeof(T) == 2) {
    return GFLA_ERR_UNSUPPORTED;
  } else {
    const int64_t n = B * C * (k * Hf) * (k * Wf);
    const int64_t blocks = ceil_div(n, kBlock);
    if (blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((be_bwd_elem_kernel<T>), dim3((unsigned)blocks), dim3(kBlock), 0, stream, src, flow,
                       gout, gsrc, gflow, n, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, k);
    return launch_status();
  }
}

// ----------------------------------------------------------------------------------------
// "unfold" layout: out (B, C*K*K, Hf, Wf), channel c*K*K + i*K + j = tap (i,j) of source channel c.
// Same values as the reference layout (B,C,K*Hf,K*Wf), arranged so that the stride-K convolution
// that consumes it in ExtractorAttn (base_function.py:800) is a plain batched GEMM
// W(128, C*K*K) @ out[b](C*K*K, Hf*Wf) and every lane writes/reads consecutive pixels.
// workgroup <-> (b, group of G channels in LDS[, 1/split of the pixels]); lane <-> pixel: the
// (K+1)x(K+1) patch is read once per channel ((K+1)^2 ds_read_b32) and yields all K*K outputs.
// ----------------------------------------------------------------------------------------
template <typename T, int K>
__global__ __launch_bounds__(kLdsThreads) void be_unfold_fwd_lds_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, T *__restrict__ out, int C, int Hs, int Ws,
    int Hf, int Wf, int G, int ngroups, int split, int64_t u_cs, int64_t u_bs) {
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
  const int HW = Hf * Wf;
  const int per = (HW + split - 1) / split;
  const int p_end = min(HW, (sp + 1) * per);
  for (int p = sp * per + threadIdx.x; p < p_end; p += blockDim.x) {
    const int yf = p / Wf, xf = p - yf * Wf;
    const A fx0 = Num<T>::ld(flow + (int64_t)(b * 2 + 0) * HW + p);
    const A fy0 = Num<T>::ld(flow + (int64_t)(b * 2 + 1) * HW + p);
    int xL[K], xR[K];
    A ax[K];
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
      dense = dense && ((int)fdx == x0 + t) && ((int)fdy == y0 + t);
      xL[t] = clampi((int)fdx, 0, Ws - 1);
      xR[t] = clampi((int)(fdx + 1), 0, Ws - 1);
      ax[t] = dx - fdx;
    }
    T *o = out + (int64_t)b * u_bs + (int64_t)c0 * KK * u_cs + p;  // element (b, ch, p) at b*u_bs + ch*u_cs + p
    if (dense) {
      int col[K + 1];
#pragma unroll
      for (int q = 0; q <= K; ++q) col[q] = clampi(x0 + q, 0, Ws - 1);
      for (int c = 0; c < gc; ++c) {
        const A *pl = planes + (size_t)c * plane_sz;
        A vA[K + 1];
        {
          const int off = clampi(y0, 0, Hs - 1) * Ws;
#pragma unroll
          for (int q = 0; q <= K; ++q) vA[q] = pl[off + col[q]];
        }
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
          const A dy = (fy0 + (A)(i - K / 2)) + (A)yf;
          const A yB_P = dy - floor_t<A>(dy), yT_P = 1 - yB_P;
          const int off = clampi(y0 + i + 1, 0, Hs - 1) * Ws;
          A vB[K + 1];
#pragma unroll
          for (int q = 0; q <= K; ++q) vB[q] = pl[off + col[q]];
          T *orow = o + (int64_t)(c * KK + i * K) * u_cs;
#pragma unroll
          for (int j = 0; j < K; ++j) {
            const A xL_P = 1 - ax[j], xR_P = ax[j];
            A s = (xL_P * yT_P) * vA[j];  // :78-84, same order of accumulation
            s += (xR_P * yT_P) * vA[j + 1];
            s += (xL_P * yB_P) * vB[j];
            s += (xR_P * yB_P) * vB[j + 1];
            orow[(int64_t)j * u_cs] = Num<T>::from(s);
          }
#pragma unroll
          for (int q = 0; q <= K; ++q) vA[q] = vB[q];
        }
      }
    } else {
      for (int c = 0; c < gc; ++c) {
        const A *pl = planes + (size_t)c * plane_sz;
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
          const A dy = (fy0 + (A)(i - K / 2)) + (A)yf;
          const A fdy = floor_t<A>(dy);
          const int yT = clampi((int)fdy, 0, Hs - 1) * Ws, yB = clampi((int)(fdy + 1), 0, Hs - 1) * Ws;
          const A yB_P = dy - fdy, yT_P = 1 - yB_P;
          T *orow = o + (int64_t)(c * KK + i * K) * u_cs;
#pragma unroll
          for (int j = 0; j < K; ++j) {
            const A xL_P = 1 - ax[j], xR_P = ax[j];
            A s = (xL_P * yT_P) * pl[yT + xL[j]];
            s += (xR_P * yT_P) * pl[yT + xR[j]];
            s += (xL_P * yB_P) * pl[yB + xL[j]];
            s += (xR_P * yB_P) * pl[yB + xR[j]];
            orow[(int64_t)j * u_cs] = Num<T>::from(s);
          }
        }
      }
    }
  }
}

#define GFLA_BE_K_SWITCH(KV, ...)                  \
  switch (KV) {                                    \
    case 1: { constexpr int K = 1; __VA_ARGS__; } break;  \
    case 2: { constexpr int K = 2; __VA_ARGS__; } break;  \
    case 3: { constexpr int K = 3; __VA_ARGS__; } break;  \
    case 4: { constexpr int K = 4; __VA_ARGS__; } break;  \
    default: { constexpr int K = 5; __VA_ARGS__; } break; \
  }

static int unfold_check(int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, int k,
                        int bytes_per_elem) {
  if (B <= 0 || C <= 0 || Hs <= 0 || Ws <= 0 || Hf <= 0 || Wf <= 0 || k < 1) return GFLA_ERR_BAD_SHAPE;
  if (k > 5) return GFLA_ERR_UNSUPPORTED;
  if (Hs * Ws * bytes_per_elem > kLdsBudget || Hf * Wf > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  return GFLA_OK;
}

static void unfold_strides(int layout, int64_t B, int64_t C, int64_t HW, int k, int64_t *cs, int64_t *bs) {
  if (layout == 1) {  // (C*k*k, B, Hf, Wf): one GEMM operand (C*k*k) x (B*Hf*Wf) for the whole batch
    *cs = B * HW;
    *bs = HW;
  } else {            // (B, C*k*k, Hf, Wf)
    *cs = HW;
    *bs = C * k * k * HW;
  }
}

template <typename T>
static int block_extractor_unfold_fwd(const T *src, const T *flow, T *out, int64_t B, int64_t C, int64_t Hs,
                                      int64_t Ws, int64_t Hf, int64_t Wf, int k, int layout,
                                      gfla_stream_t stream_) {
  using A = typename Num<T>::acc;
  if (!src || !flow || !out) return GFLA_ERR_NULL_POINTER;
  int st = unfold_check(B, C, Hs, Ws, Hf, Wf, k, sizeof(A));
  if (st != GFLA_OK) return st;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  PlaneGeo g = plane_geometry(Hs * Ws, sizeof(A), B, C, Hf * Wf, true);
  const int64_t blocks = B * g.ngroups * g.split;
  if (blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  int64_t cs, bs;
  unfold_strides(layout, B, C, (int64_t)(Hf * Wf), k, &cs, &bs);
  GFLA_BE_K_SWITCH(k, launch_lds(be_unfold_fwd_lds_kernel<T, K>, dim3((unsigned)blocks), dim3(kLdsThreads), g.lds_bytes, stream, 
                          src, flow, out, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, g.G, g.ngroups, g.split, cs, bs));
  return launch_status();
}

template <typename T>
static int block_extractor_unfold_bwd(const T *src, const T *flow, const T *gout, T *gsrc,
                                      typename Num<T>::acc *gflow, int64_t B, int64_t C, int64_t Hs, int64_t Ws,
                                      int64_t Hf, int64_t Wf, int k, int layout, gfla_stream_t stream_) {
  using A = typename Num<T>::acc;
  if (!src || !flow || !gout) return GFLA_ERR_NULL_POINTER;
  int st = unfold_check(B, C, Hs, Ws, Hf, Wf, k, sizeof(lds_acc_t) + sizeof(A));
  if (st != GFLA_OK) return st;
  if (!gsrc && !gflow) return GFLA_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  bool done = false;
  int64_t cs, bs;
  unfold_strides(layout, B, C, (int64_t)(Hf * Wf), k, &cs, &bs);
  GFLA_BE_K_SWITCH(k, st = launch_be_bwd_lds<T, K>(kGoutUnfold, src, flow, gout, static_cast<const T *>(nullptr), gsrc,
                                                   gflow, B, C, Hs, Ws, Hf, Wf, stream, &done, cs, bs));
  if (st == GFLA_OK && !done) st = GFLA_ERR_UNSUPPORTED;
  return st;
}

}  // namespace gfla

using gfla::bf16_t;

extern "C" {
int gfla_block_extractor_fwd_f32(const float *s, const float *f, float *o, int64_t B, int64_t C,
                                 int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, int k,
                                 gfla_stream_t st) {
  return gfla::block_extractor_fwd<float>(s, f, o, B, C, Hs, Ws, Hf, Wf, k, st);
}
int gfla_block_extractor_fwd_f64(const double *s, const double *f, double *o, int64_t B, int64_t C,
                                 int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, int k,
                                 gfla_stream_t st) {
  return gfla::block_extractor_fwd<double>(s, f, o, B, C, Hs, Ws, Hf, Wf, k, st);
}
int gfla_block_extractor_fwd_bf16(const uint16_t *s, const uint16_t *f, uint16_t *o, int64_t B,
                                  int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, int k,
                                  gfla_stream_t st) {
  return gfla::block_extractor_fwd<bf16_t>(reinterpret_cast<const bf16_t *>(s),
                                           reinterpret_cast<const bf16_t *>(f),
                                           reinterpret_cast<bf16_t *>(o), B, C, Hs, Ws, Hf, Wf, k, st);
}
int gfla_block_extractor_bwd_f32(const float *s, const float *f, const float *go, float *gs,
                                 float *gf, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf,
                                 int64_t Wf, int k, gfla_stream_t st) {
  return gfla::block_extractor_bwd<float>(s, f, go, gs, gf, B, C, Hs, Ws, Hf, Wf, k, st);
}
/* bf16 storage: grad_source bf16 (accumulated into, exclusive owner per plane); grad_flow FLOAT32 -- a sum over all
 * channels and taps, accumulated across channel groups with atomics */
int gfla_block_extractor_bwd_bf16(const uint16_t *s, const uint16_t *f, const uint16_t *go, uint16_t *gs, float *gf,
                                  int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, int k,
                                  gfla_stream_t st) {
  return gfla::block_extractor_bwd<bf16_t>(reinterpret_cast<const bf16_t *>(s), reinterpret_cast<const bf16_t *>(f),
                                           reinterpret_cast<const bf16_t *>(go), reinterpret_cast<bf16_t *>(gs), gf, B, C,
                                           Hs, Ws, Hf, Wf, k, st);
}
int gfla_block_extractor_bwd_f64(const double *s, const double *f, const double *go, double *gs,
                                 double *gf, int64_t B, int64_t C, int64_t Hs, int64_t Ws,
                                 int64_t Hf, int64_t Wf, int k, gfla_stream_t st) {
  return gfla::block_extractor_bwd<double>(s, f, go, gs, gf, B, C, Hs, Ws, Hf, Wf, k, st);
}
int gfla_unfold_supported(int64_t Hs, int64_t Ws, int k, int elem_size) {
  const int acc = elem_size == 8 ? 8 : 4;
  return (k >= 1 && k <= 5 && Hs > 0 && Ws > 0 && Hs * Ws * (int64_t)(8 + acc) <= gfla::lds_budget()) ? 1 : 0;
}
int gfla_block_extractor_unfold_fwd_f32(const float *s, const float *f, float *o, int64_t B, int64_t C,
                                        int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, int k, int layout,
                                        gfla_stream_t st) {
  return gfla::block_extractor_unfold_fwd<float>(s, f, o, B, C, Hs, Ws, Hf, Wf, k, layout, st);
}
int gfla_block_extractor_unfold_fwd_f64(const double *s, const double *f, double *o, int64_t B, int64_t C,
                                        int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, int k, int layout,
                                        gfla_stream_t st) {
  return gfla::block_extractor_unfold_fwd<double>(s, f, o, B, C, Hs, Ws, Hf, Wf, k, layout, st);
}
int gfla_block_extractor_unfold_fwd_bf16(const uint16_t *s, const uint16_t *f, uint16_t *o, int64_t B,
                                         int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, int k, int layout,
                                         gfla_stream_t st) {
  return gfla::block_extractor_unfold_fwd<bf16_t>(reinterpret_cast<const bf16_t *>(s),
                                                  reinterpret_cast<const bf16_t *>(f),
                                                  reinterpret_cast<bf16_t *>(o), B, C, Hs, Ws, Hf, Wf, k, layout, st);
}
int gfla_block_extractor_unfold_bwd_f32(const float *s, const float *f, const float *go, float *gs, float *gf,
                                        int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf,
                                        int k, int layout, gfla_stream_t st) {
  return gfla::block_extractor_unfold_bwd<float>(s, f, go, gs, gf, B, C, Hs, Ws, Hf, Wf, k, layout, st);
}
int gfla_block_extractor_unfold_bwd_bf16(const uint16_t *s, const uint16_t *f, const uint16_t *go, uint16_t *gs,
                                         float *gf, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf,
                                         int64_t Wf, int k, int layout, gfla_stream_t st) {
  return gfla::block_extractor_unfold_bwd<bf16_t>(reinterpret_cast<const bf16_t *>(s), reinterpret_cast<const bf16_t *>(f),
                                                  reinterpret_cast<const bf16_t *>(go), reinterpret_cast<bf16_t *>(gs), gf,
                                                  B, C, Hs, Ws, Hf, Wf, k, layout, st);
}
int gfla_block_extractor_unfold_bwd_f64(const double *s, const double *f, const double *go, double *gs,
                                        double *gf, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf,
                                        int64_t Wf, int k, int layout, gfla_stream_t st) {
  return gfla::block_extractor_unfold_bwd<double>(s, f, go, gs, gf, B, C, Hs, Ws, Hf, Wf, k, layout, st);
}
}
