// Building blocks of the "big plane" kernels (round 5): feature planes far beyond the LDS budget and FEW of them --
// BASELINE configs[1], (1, 64, 256, 176): 64 planes of 180 KB on 256 CUs.  The planes-in-LDS kernels get their parallelism
// from (batch x channel group) workgroups, each owning whole planes; here it has to come from SPACE:
//   * a workgroup owns a tile / a run of flow pixels and walks many channels with ONE per-pixel setup (at B*C = 64 the
//     windowed round-1 kernels spent 3/4 of their instructions recomputing tap geometry per group of 2-4 channels);
//   * workgroup -> tile mapping is XCD-aware: block b runs on XCD b % 8 (MI355X_MICROARCH.md, observed), so the blocks of
//     one XCD are given a CONTIGUOUS range of the (batch, tile) order with the channel groups of a tile adjacent -- each of
//     the eight L2s then holds one eighth of every plane (plus the flow's reach) instead of all of every plane;
//   * gathers read global memory directly (lanes = consecutive pixels: a wave's tap loads fall into 2-3 cache lines for any
//     flow that is locally coherent, and L1 / the XCD's L2 serve the overlap between taps); scatters accumulate in an LDS
//     window that is the BOUNDING BOX of what the tile's pixels actually reach (computed on the device from the flow, so a
//     smooth flow of any magnitude stays in LDS), processed in as many channel rounds as the box allows, and leave through
//     one float atomic per window element; a box too large for a single channel takes global atomics for that tile only.
#pragma once

#include "lds_plane.h"

namespace gfla {

// Bijective XCD remap (cdna_hip_programming.md T1): the blocks that land on XCD x get the x-th contiguous share of
// [0, nwg).  Correct for any nwg; a wrong placement guess only costs speed.
__host__ __device__ __forceinline__ int64_t xcd_swizzle(int64_t bid, int64_t nwg) {
  const int64_t q = nwg / kNumXCD, r = nwg % kNumXCD, xcd = bid % kNumXCD;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / kNumXCD;
}

// Bounding box of a tile's taps, reduced over the workgroup: box = {ymin, xmin, ymax, xmax} in LDS, initialised by
// box_init() before a barrier.  Inactive lanes pass an empty range.
__device__ __forceinline__ void box_init(int *box) {
  if (threadIdx.x == 0) {
    box[0] = box[1] = 0x7fffffff;
    box[2] = box[3] = -1;
  }
}
__device__ __forceinline__ void box_reduce(int *box, int ylo, int xlo, int yhi, int xhi) {
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    ylo = min(ylo, __shfl_xor(ylo, o));
    xlo = min(xlo, __shfl_xor(xlo, o));
    yhi = max(yhi, __shfl_xor(yhi, o));
    xhi = max(xhi, __shfl_xor(xhi, o));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&box[0], ylo);
    atomicMin(&box[1], xlo);
    atomicMax(&box[2], yhi);
    atomicMax(&box[3], xhi);
  }
}

// The LDS window of a tile: the bounding box [ymin, ymin + rows) x [xmin, xmin + cols) of the plane positions its pixels
// reach; element (y, x) of channel slot c lives at c * size + (y - ymin) * cols + (x - xmin).
struct TileWin {
  int ymin, xmin, rows, cols, size;
};
__device__ __forceinline__ TileWin tile_window(const int *box) {
  TileWin w;
  w.ymin = box[0];
  w.xmin = box[1];
  w.rows = box[2] - box[0] + 1;
  w.cols = box[3] - box[1] + 1;
  w.size = w.rows * w.cols;
  return w;
}
// Windows that are STAGED from global memory (gathers) want 16-byte traffic: when the plane allows it (row pitch and plane
// size multiples of four floats, base 16-byte aligned) the box is widened to whole groups of four columns, so that every
// window row starts on a 16-byte boundary in global memory and in LDS.
template <typename T>
__device__ __forceinline__ bool window_vec_ok(const T *src0, int64_t plane, int Ws) {
  return sizeof(T) == 4 && (Ws & 3) == 0 && (plane & 3) == 0 && (reinterpret_cast<uintptr_t>(src0) & 15) == 0;
}
__device__ __forceinline__ TileWin tile_window_vec(const int *box, int Ws, bool vec) {
  TileWin w = tile_window(box);
  if (vec) {
    const int x1 = min(Ws, (box[3] + 4) & ~3);   // one past the last column, rounded up (Ws is a multiple of 4)
    w.xmin = box[1] & ~3;
    w.cols = x1 - w.xmin;
    w.size = w.rows * w.cols;
  }
  return w;
}
// 16 bytes per lane, the (channel slot, row, group of four columns) space flattened over the workgroup, U requests per
// thread in flight.  (Measured with the loads of one window ROW per wave and request, 45 floats of 64 lanes: the staging of
// resample2d's forward at (1,64,256,176) took 17.5 of the kernel's 35 us whatever the rows per wave and the workgroups per
// CU -- ~60 CU cycles per wave load; profiles/r5_rs_fwd_tile_ablations.txt.)
template <typename A>
__device__ __forceinline__ void stage_windows_vec(const float *__restrict__ src0, int64_t plane, int Ws, A *lds, const TileWin &w, int n) {
  constexpr int U = 6;
  const int c4 = w.cols >> 2, per = w.rows * c4, total = n * per;
  const float inv_per = 1.0f / (float)per, inv_c4 = 1.0f / (float)c4;
  auto locate = [&](int f, int &goff, int &loff) {   // exact for the sizes at hand (f < 2^20): float reciprocal + fix-up
    int c = (int)(((float)f + 0.5f) * inv_per);
    c -= (c * per > f);
    c += ((c + 1) * per <= f);
    const int e = f - c * per;
    int r = (int)(((float)e + 0.5f) * inv_c4);
    r -= (r * c4 > e);
    r += ((r + 1) * c4 <= e);
    const int q = (e - r * c4) << 2;
    goff = (w.ymin + r) * Ws + w.xmin + q;
    loff = c * w.size + r * w.cols + q;
    return c;
  };
  for (int f0 = threadIdx.x; f0 < total; f0 += blockDim.x * U) {
    typedef float v4_t __attribute__((ext_vector_type(4)));   // (an array of HIP's float4 struct can end up in scratch)
    v4_t v[U];
    int lo[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int f = min(f0 + u * (int)blockDim.x, total - 1);
      int go;
      const int c = locate(f, go, lo[u]);
      v[u] = *reinterpret_cast<const v4_t *>(src0 + (int64_t)c * plane + go);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (f0 + u * (int)blockDim.x < total) *reinterpret_cast<v4_t *>(lds + lo[u]) = v[u];
  }
}

// global planes (row pitch Ws, `plane` elements apart) -> the windows of n channel slots; a wave per window row, lanes along
// the row (coalesced segments of `cols` elements).  EIGHT rows per wave are requested before the first is stored: written
// as load -> store per row the loop keeps one request in flight per wave, and a workgroup then spends its life waiting for
// ~30 dependent round trips (measured: the first window kernels were SLOWER than the global gathers they replaced,
// profiles/r5_config2_window_kernels_unpipelined_staging.txt).
template <typename T, typename A>
__device__ __forceinline__ void stage_windows(const T *__restrict__ src0, int64_t plane, int Ws, A *lds, const TileWin &w, int n,
                                              bool vec = false) {
  if constexpr (sizeof(T) == 4 && sizeof(A) == 4) {
    if (vec) {
      stage_windows_vec<A>(reinterpret_cast<const float *>(src0), plane, Ws, lds, w, n);
      return;
    }
  }
  constexpr int U = 8;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const int total = n * w.rows;
  for (int q = lane; q < w.cols; q += 64) {
    for (int rr0 = wave; rr0 < total; rr0 += nw * U) {
      A v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rr = min(rr0 + u * nw, total - 1);   // (clamped: a repeated load, no branch between the requests)
        const int c = rr / w.rows, r = rr - c * w.rows;
        v[u] = Num<T>::ld(src0 + (int64_t)c * plane + (int64_t)(w.ymin + r) * Ws + w.xmin + q);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int rr = rr0 + u * nw;
        if (rr < total) {
          const int c = rr / w.rows, r = rr - c * w.rows;
          lds[(size_t)c * w.size + r * w.cols + q] = v[u];
        }
      }
    }
  }
}
// the windows of n channel slots -> atomics into the global planes; val(i) = the value of window element i (0 = untouched:
// no atomic).  A wave per window row, lanes along the row.
template <typename T, typename F>
__device__ __forceinline__ void flush_windows(T *__restrict__ g0, int64_t plane, int Ws, const TileWin &w, int n, F val) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  for (int rr = wave; rr < n * w.rows; rr += nw) {
    const int c = rr / w.rows, r = rr - c * w.rows;
    const int base = c * w.size + r * w.cols;
    T *grow = g0 + (int64_t)c * plane + (int64_t)(w.ymin + r) * Ws + w.xmin;
    for (int q = lane; q < w.cols; q += 64) {
      const double v = val(base + q);
      if (v != 0) atomic_add(grow + q, (T)v);
    }
  }
}
// A gather tile whose bounding box is many times its own area (wild flow) is not worth staging: each staged element would
// be read less than once.  Such a tile reads global memory.
__device__ __forceinline__ bool window_worth_staging(const TileWin &w, int th, int tw) { return w.size <= 6 * th * tw + 512; }

// LDS a tile kernel asks for: G windows of the tile plus `span` taps and a flow reach of 4 positions either side -- not
// the whole budget, so that several workgroups share a CU and one's staging overlaps another's arithmetic (a tile that
// reaches further takes more channel rounds; correctness never depends on this estimate).
inline unsigned tile_lds_request(int th, int tw, int span, int G, int bytes_per_elem, int64_t budget) {
  int64_t want = (int64_t)(th + span + 8) * (tw + span + 8) * G * bytes_per_elem;
  if (want < 16 * 1024) want = 16 * 1024;
  if (want > budget) want = budget;
  return (unsigned)((want + 255) & ~(int64_t)255);
}

// Timing-ablation bits of the tile kernels (tuning key 39; results are garbage): honoured only in `make PROBES=1` builds, a
// default build ignores the key -- no user-reachable path that computes wrong results.
inline int tile_probe_bits() {
#ifdef GFLA_PROBES
  return tuning(39);
#else
  return 0;
#endif
}

// Host side: is this the regime of the big-plane kernels?  Few planes (the planes-in-LDS / windowed kernels get fewer
// than ~4 workgroups per CU out of batch x channel groups) AND planes beyond the LDS budget.  tuning key 30: 1 = never
// (round 1's windowed kernels), 2 = always (tests drive the kernels at small shapes).
inline bool big_plane_regime(int64_t B, int64_t C, int64_t plane_bytes, int64_t lds_budget) {
  const int t = tuning(30);
  if (t == 1) return false;
  if (t == 2) return true;
  return plane_bytes > lds_budget && B * C < 4 * kNumCU;
}

// Scatter tiles: th x tw flow pixels per workgroup, one pixel per thread (th * tw <= 512).  tuning keys 31 / 32.
struct TileGeo {
  int th, tw, nty, ntx, threads;
};
inline TileGeo tile_geometry(int64_t H, int64_t W, int dflt_th = 16, int dflt_tw = 32) {
  TileGeo g;
  int tw = tuning(32) > 0 ? tuning(32) : dflt_tw;
  if (tw > W) tw = (int)W;
  if (tw > 512) tw = 512;
  // NOT split evenly (176 -> 6 tiles of 30): tiles of 32 columns start on 128-byte lines of every per-pixel tensor, and what a
  // wave writes per row is then a whole line, not the tail of one and the head of the next (5 tiles of 32 and one of 16)
  const int ntx = (int)ceil_div(W, tw);
  int th = tuning(31) > 0 ? tuning(31) : dflt_th;
  if (th * tw > 512) th = 512 / tw;
  if (th > H) th = (int)H;
  if (th < 1) th = 1;
  g.th = th;
  g.tw = tw;
  g.ntx = ntx;
  g.nty = (int)ceil_div(H, th);
  g.threads = (int)ceil_div((int64_t)th * tw, 64) * 64;
  return g;
}

// Gather tiles of block_extractor's forward: 8 x 32 flow pixels (256 threads: many small workgroups in different phases keep
// the store stream busy).  Whole flow rows per workgroup -- one contiguous piece of the output plane per channel -- were the
// first default and measured at half the rate: their windows are 12+ full-width rows for 2 rows of pixels
// (profiles/r5_config2_sweeps.txt).  tuning keys 35 / 36.
inline TileGeo row_tile_geometry(int64_t H, int64_t W) {
  TileGeo g;
  int tw = tuning(36) > 0 ? tuning(36) : 32;
  if (tw > W) tw = (int)W;
  const int ntx = (int)ceil_div(W, tw);
  int th = tuning(35) > 0 ? tuning(35) : 8;
  if (th * tw > 512) th = 512 / tw;
  if (th > H) th = (int)H;
  if (th < 1) th = 1;
  g.th = th;
  g.tw = tw;
  g.ntx = ntx;
  g.nty = (int)ceil_div(H, th);
  g.threads = (int)ceil_div((int64_t)th * tw, 64) * 64;
  return g;
}

// channels per workgroup of a tile kernel: `dflt` (tuning key `key` overrides) unless that leaves the launch under three
// workgroups per CU
inline int tile_channels(int key, int dflt, int64_t tiles, int64_t C) {
  int G = tuning(key) > 0 ? tuning(key) : dflt;
  if (tuning(key) <= 0)
    while (G > 1 && tiles * ceil_div(C, G) < 3 * kNumCU) G /= 2;
  return G > C ? (int)C : G;
}

// The whole launch geometry of one tile kernel, in one place (the launchers use it; gfla_big_plane_geometry hands it to the
// CPU tests).  op: 0 block_extractor forward, 1 block_extractor backward, 2 resample2d forward / d/d input2 (gather),
// 3 resample2d d/d input1 (scatter).  span = taps per axis (K + 1, or (k - 1) * dilation + 1); bytes_per_elem = LDS bytes one
// window element needs per channel.
struct BigGeo {
  TileGeo tg;
  int G, ngroups;
  unsigned lds_bytes;
  int64_t nwg;
};
inline BigGeo big_geometry(int op, int64_t B, int64_t C, int64_t H, int64_t W, int span, int bytes_per_elem) {
  BigGeo g;
  // op: 0 block_extractor forward, 1 its backward, 2 resample2d forward, 3 its d/d input1 scatter, 4 its d/d input2 gather
  // (8 x 32 tiles: 29.5 -> 25.9 us against 16 x 32, session s33; the scatter and the forward measured best at 16 x 32)
  // resample2d's forward: 16 x 16 tiles of 8 channels (19.7 -> 17.5 us against 16 x 32 tiles of 16: a squarer tile has the
  // smaller bounding box per pixel; rows of 16 pixels still start on 64-byte boundaries; sessions s38 / s39)
  g.tg = op == 0 ? row_tile_geometry(H, W) : op == 2 ? tile_geometry(H, W, 16, 16) : tile_geometry(H, W, op == 4 ? 8 : 16);
  const int64_t tiles = B * g.tg.nty * g.tg.ntx;
  // (block_extractor's backward at k >= 5 -- span 6: windows of ~13 KB per channel -- measured 168 -> 153 us with 4 channels
  // per workgroup instead of 8: twice the workgroups, each with one channel round instead of two; profiles/r5_config2_sweeps.txt)
  g.G = op == 0 ? tile_channels(37, 8, tiles, C) : op == 2 ? tile_channels(37, 8, tiles, C) : op == 4 ? tile_channels(37, 16, tiles, C)
                                                            : tile_channels(34, op == 1 && span >= 6 ? 4 : 8, tiles, C);
  g.ngroups = (int)ceil_div(C, g.G);
  g.nwg = tiles * g.ngroups;
  g.lds_bytes = tile_lds_request(g.tg.th, g.tg.tw, span, g.G, bytes_per_elem, lds_budget());
  return g;
}

}  // namespace gfla
