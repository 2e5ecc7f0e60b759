// Flow-driven scatters as block-sparse products on the matrix cores, gfx950.
//
// The backward passes that scatter into a feature plane -- the attention-weighted aggregation's d/d source
// (base_function.py:808-809 through block_extractor_kernel.cu:123-168) and resample2d's d/d input1
// (resample2d_kernel.cu:98-202) -- have this shape:
//     dS[b, c, q] = sum_p  G[b, c, p] * w_b[p -> q]
// where the weights w_b[p -> q] do NOT depend on the channel: every flow pixel p spreads its gradient over a dense
// patch of (k+1)^2 (aggregation: attention x bilinear weights) or 4x4 (resample2d: normalised Gaussian weights)
// source positions around p + flow(p).  That is a sparse-matrix x dense-matrix product with ~36 non-zeros per column.
// The reference does it with 4 k^2 global atomics per pixel and channel; round 1 with (k+1)^2 LDS atomics per pixel
// and channel, which left the kernels on an LDS-atomic ceiling (59 % conflict cycles).  Here the sparse matrix is
// cut into dense tiles and multiplied on the f32 matrix cores, output-stationary, with no atomics at all:
//   * a workgroup OWNS a band of R source rows (<= 192 positions q) x 128 channels of one sample; its accumulators
//     (v_mfma_f32_32x32x2_f32, exact f32) stay in registers for the whole kernel and are stored once, coalesced;
//   * it walks the 32-pixel tiles of the flow field whose patches reach its band (a per-tile row range is known
//     from a pre-pass; smooth flows touch 2-3 bands per tile, wild flows more -- never wrong, only slower);
//   * per tile it builds W[32 p][band q] in LDS -- zero fill + one plain store per patch entry, entries that clamp
//     onto the same border position are pre-summed by the table pass -- stages G[128 c][32 p] and multiplies:
//     D[c][q] += G[c][p] W[p][q].  ~90 % of the multiplied entries are zeros, and it is still several times faster
//     than the atomics: the aggregate's d/d source 431 -> ~90 us at the bench shape.
// Sums are accumulated in a fixed order: the result is bit-reproducible run to run.
//
// Passes (all on the caller's stream, scratch from the caller):
//   1. table pass (op-specific): per flow pixel its patch entries (target position, weight) with border duplicates
//      folded, and per 32-pixel tile the range of source rows it reaches;
//   2. patch_scatter_mfma_kernel (generic);
//   3. aggregation only: pixels whose taps are not a dense patch (a tap within rounding of an integer position,
//      lds_plane.h / be_bwd_lds.h) are left out of the table and scattered tap by tap with global atomics.
#include "gfla_common.h"
#include "patch_mfma.h"
#include "rs_taps.h"

namespace gfla {

typedef float pm_f32x16 __attribute__((ext_vector_type(16)));

struct PatchEntry {
  int q;    // (ty << 16) | tx of the clamped target position, or -1: nothing to add
  float v;  // weight
};

constexpr int kPmStatSlots = 32;  // counters of the dispatch statistic (their sum is what counts)
constexpr int kPmTile = 32;     // flow pixels per tile = K extent of one MFMA pass
constexpr int kPmSub = 8;       // threads sharing the entries of one pixel
constexpr int kPmChannels = 128;  // channels per workgroup
constexpr int kPmMaxNB = 6;     // 32-column blocks of a band (<= 192 source positions)
constexpr int kPmGPitch = 33;   // floats; LDS pitch of a channel row of the G tile (odd: conflict-free column reads)

struct PmLayout {
  int64_t ntiles, table_bytes, rows_bytes, total;  // the last 256 bytes of the workspace hold the dispatch statistic
};
static PmLayout pm_layout(int64_t B, int64_t HW, int entries) {
  PmLayout L;
  L.ntiles = ceil_div(HW, kPmTile);
  L.table_bytes = ((B * L.ntiles * kPmTile * entries * (int64_t)sizeof(PatchEntry)) + 255) & ~(int64_t)255;
  L.rows_bytes = ((B * L.ntiles * (int64_t)sizeof(int2)) + 255) & ~(int64_t)255;
  L.total = L.table_bytes + L.rows_bytes + 256;
  return L;
}

__device__ __forceinline__ unsigned pm_stat_total(const unsigned *stat) {
  unsigned tot = 0;
#pragma unroll
  for (int i = 0; i < kPmStatSlots; ++i) tot += stat[i];
  return tot;
}

__device__ __forceinline__ int pm_safe_int(float v) {  // saturating float -> int that also keeps later +k defined
  return (int)fminf(fmaxf(v, -1048576.f), 1048576.f);
}

// Block-wide (lo, hi) of the rows a tile reaches -> tile_rows[slot]
__device__ __forceinline__ void pm_store_tile_rows(int lo, int hi, int2 *slot, unsigned *stat) {
  __shared__ int s_lo[4], s_hi[4];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    lo = min(lo, __shfl_xor(lo, m));
    hi = max(hi, __shfl_xor(hi, m));
  }
  if ((threadIdx.x & 63) == 0) {
    s_lo[threadIdx.x >> 6] = lo;
    s_hi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int l = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3])), h = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
    *slot = make_int2(l, h);
    // dispatch statistic: source rows reached, summed over the tiles (how far the flow spreads the tiles' patches);
    // spread over kPmStatSlots counters -- thousands of atomics on ONE address serialise (~12 ns each)
    if (h >= l) atomicAdd(stat + (blockIdx.x + blockIdx.y * gridDim.x) % kPmStatSlots, (unsigned)(h - l + 1));
  }
}

// Entry e = (r, s) of a P x P patch anchored at (y0, x0): is it the one entry that carries the sum of all entries
// clamping onto its target, and which patch rows / columns does that sum run over?  (Hs, Ws >= 2.)
struct Fold {
  int r_lo, r_hi, s_lo, s_hi, ty, tx;
  bool canonical;
};
__device__ __forceinline__ Fold pm_fold(int r, int s, int y0, int x0, int P, int Hs, int Ws) {
  Fold f;
  const int yy = y0 + r, xx = x0 + s;
  f.r_lo = f.r_hi = r;
  f.s_lo = f.s_hi = s;
  f.canonical = true;
  if (yy <= 0) {  // rows 0..r all clamp to 0: the carrier is the last of them
    f.r_lo = 0;
    f.canonical = f.canonical && (yy == 0 || r == P - 1);
  } else if (yy >= Hs - 1) {
    f.r_hi = P - 1;
    f.canonical = f.canonical && (yy == Hs - 1 || r == 0);
  }
  if (xx <= 0) {
    f.s_lo = 0;
    f.canonical = f.canonical && (xx == 0 || s == P - 1);
  } else if (xx >= Ws - 1) {
    f.s_hi = P - 1;
    f.canonical = f.canonical && (xx == Ws - 1 || s == 0);
  }
  f.ty = clampi(yy, 0, Hs - 1);
  f.tx = clampi(xx, 0, Ws - 1);
  return f;
}

// ------------------------------------------------------------------------------------------------ aggregation
// Taps of flow pixel (xf, yf) as block_extractor_kernel.cu:132-136 computes them; `dense` as lds_plane.h defines it.
template <int K>
struct AggTaps {
  float fx0, fy0;
  int xf, yf, x0, y0;
  bool dense;
  __device__ __forceinline__ float dx(int t) const { return (fx0 + (float)(t - K / 2)) + (float)xf; }
  __device__ __forceinline__ float dy(int t) const { return (fy0 + (float)(t - K / 2)) + (float)yf; }
  __device__ __forceinline__ float ax(int t) const { const float d = dx(t); return d - floorf(d); }
  __device__ __forceinline__ float ay(int t) const { const float d = dy(t); return d - floorf(d); }
  __device__ __forceinline__ void init(float fx, float fy, int x, int y) {
    fx0 = fx, fy0 = fy, xf = x, yf = y;
    dense = true;
    x0 = y0 = 0;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const int ix = (int)floorf(dx(t)), iy = (int)floorf(dy(t));
      if (t == 0) x0 = ix, y0 = iy;
      dense = dense && (ix == x0 + t) && (iy == y0 + t);
    }
  }
};

// grid (ceil(ntiles / 8), B), 256 threads: thread = flow pixel, a half wave = one 32-pixel tile of the scatter kernel.
// Everything of a pixel stays in registers: the (K+1)^2 patch coefficients (attention x bilinear weights / k^2,
// block_extractor_kernel.cu:158-161 folded), then the clamp fold -- entries that clamp onto the same border position
// are summed into one carrier entry (pm_fold) -- as two separable passes of statically indexed, predicated sums
// (columns, then rows).  (Round 2's first version walked five nested runtime loops over LDS per entry: 54 us.)
template <int K>
__global__ __launch_bounds__(256) void agg_patch_table_kernel(const float *__restrict__ flow,
                                                             const float *__restrict__ attn,
                                                             PatchEntry *__restrict__ table,
                                                             int2 *__restrict__ tile_rows,
                                                             unsigned *__restrict__ stat, int H, int W, int Hs, int Ws,
                                                             int ntiles) {
  constexpr int P = K + 1, E = P * P, KK = K * K;
  static_assert(E % 2 == 0, "entries are written two at a time");
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const int64_t b = blockIdx.y;
  const int HW = H * W;
  const int tile = blockIdx.x * 8 + (threadIdx.x >> 5), pl = threadIdx.x & 31;
  if (tile >= ntiles) return;  // whole half waves leave together
  const int p = tile * kPmTile + pl;
  const float inv_kk = 1.f / (float)KK;
  int lo = 0x7fffffff, hi = -1;
  i32x4 *out = reinterpret_cast<i32x4 *>(table + ((b * ntiles + tile) * kPmTile + pl) * (int64_t)E);
  bool dense = false;
  AggTaps<K> tp;
  if (p < HW) {
    const int yf = p / W, xf = p - yf * W;
    tp.init(flow[(b * 2 + 0) * HW + p], flow[(b * 2 + 1) * HW + p], xf, yf);
    dense = tp.dense;
  }
  if (dense) {
    const int y0 = pm_safe_int((float)tp.y0), x0 = pm_safe_int((float)tp.x0);
    lo = clampi(y0, 0, Hs - 1);
    hi = clampi(y0 + K, 0, Hs - 1);
    float a[KK];
    const float *at = attn + b * (int64_t)KK * HW + p;
#pragma unroll
    for (int t = 0; t < KK; ++t) a[t] = at[(int64_t)t * HW] * inv_kk;
    float ax[K], ay[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
      ax[t] = tp.ax(t);
      ay[t] = tp.ay(t);
    }
    float v[P][P];
#pragma unroll
    for (int r = 0; r < P; ++r) {
#pragma unroll
      for (int q = 0; q < P; ++q) v[r][q] = 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        float wrow = 0.f;  // attention mass of tap column j that lands on patch row r
        if (r < K) wrow += a[r * K + j] * (1.f - ay[r]);
        if (r > 0) wrow += a[(r - 1) * K + j] * ay[r - 1];
        v[r][j] += wrow * (1.f - ax[j]);
        v[r][j + 1] += wrow * ax[j];
      }
    }
    // pm_fold, one axis at a time: [lo_, hi_] = the entries summed into entry i, can_ = i carries that sum
    int clo[P], chi[P], rlo[P], rhi[P], tx[P], ty[P];
    bool ccan[P], rcan[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int xx = x0 + i, yy = y0 + i;
      clo[i] = chi[i] = rlo[i] = rhi[i] = i;
      ccan[i] = rcan[i] = true;
      if (xx <= 0) {
        clo[i] = 0;
        ccan[i] = xx == 0 || i == P - 1;
      } else if (xx >= Ws - 1) {
        chi[i] = P - 1;
        ccan[i] = xx == Ws - 1 || i == 0;
      }
      if (yy <= 0) {
        rlo[i] = 0;
        rcan[i] = yy == 0 || i == P - 1;
      } else if (yy >= Hs - 1) {
        rhi[i] = P - 1;
        rcan[i] = yy == Hs - 1 || i == 0;
      }
      tx[i] = clampi(xx, 0, Ws - 1);
      ty[i] = clampi(yy, 0, Hs - 1);
    }
    float cf[P][P];
#pragma unroll
    for (int r = 0; r < P; ++r)
#pragma unroll
      for (int q = 0; q < P; ++q) {
        float acc = 0.f;
#pragma unroll
        for (int q2 = 0; q2 < P; ++q2) acc += (q2 >= clo[q] && q2 <= chi[q]) ? v[r][q2] : 0.f;
        cf[r][q] = acc;
      }
#pragma unroll
    for (int r = 0; r < P; ++r)
#pragma unroll
      for (int q = 0; q < P; q += 2) {
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int r2 = 0; r2 < P; ++r2) {
          const bool in = r2 >= rlo[r] && r2 <= rhi[r];
          acc0 += in ? cf[r2][q] : 0.f;
          acc1 += in ? cf[r2][q + 1] : 0.f;
        }
        const bool c0 = rcan[r] && ccan[q], c1 = rcan[r] && ccan[q + 1];
        i32x4 e;
        e.x = c0 ? (ty[r] << 16) | tx[q] : -1;
        e.y = c0 ? __float_as_int(acc0) : 0;
        e.z = c1 ? (ty[r] << 16) | tx[q + 1] : -1;
        e.w = c1 ? __float_as_int(acc1) : 0;
        out[(r * P + q) >> 1] = e;
      }
  } else {
    const i32x4 none = {-1, 0, -1, 0};
#pragma unroll
    for (int e = 0; e < E / 2; ++e) out[e] = none;
  }
  // (lo, hi) of the rows the tile reaches: reduction over the 32 lanes of the half wave
#pragma unroll
  for (int msk = 16; msk >= 1; msk >>= 1) {
    lo = min(lo, __shfl_xor(lo, msk));
    hi = max(hi, __shfl_xor(hi, msk));
  }
  if (pl == 0) {
    tile_rows[b * ntiles + tile] = make_int2(lo, hi);
    // dispatch statistic: source rows reached, summed over the tiles (how far the flow spreads the tiles' patches);
    // spread over kPmStatSlots counters -- thousands of atomics on ONE address serialise (~12 ns each)
    if (hi >= lo) atomicAdd(stat + (tile + (int)b * ntiles) % kPmStatSlots, (unsigned)(hi - lo + 1));
  }
}

// Pixels that are not a dense patch: the reference's tap-by-tap scatter with global atomics (rare: a tap within
// rounding of an integer position).  One thread per (b, p); dense pixels return at once.
template <int K>
__global__ __launch_bounds__(256) void agg_scatter_nondense_kernel(const float *__restrict__ flow,
                                                                  const float *__restrict__ attn,
                                                                  const float *__restrict__ gout,
                                                                  float *__restrict__ gsrc, int C, int H, int W,
                                                                  int Hs, int Ws, const unsigned *__restrict__ stat,
                                                                  unsigned limit) {
  constexpr int KK = K * K;
  const int HW = H * W;
  if (pm_stat_total(stat) > limit) return;  // the LDS-atomic kernel took this launch, all pixels included
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int64_t b = blockIdx.y;
  if (p >= HW) return;
  const int yf = p / W, xf = p - yf * W;
  AggTaps<K> tp;
  tp.init(flow[(b * 2 + 0) * HW + p], flow[(b * 2 + 1) * HW + p], xf, yf);
  if (tp.dense) return;
  const float *at = attn + b * (int64_t)KK * HW + p;
  const float inv_kk = 1.f / (float)KK;
  for (int c = 0; c < C; ++c) {
    const float go = gout[(b * C + c) * (int64_t)HW + p] * inv_kk;
    float *gp = gsrc + (b * C + c) * (int64_t)Hs * Ws;
#pragma unroll 1
    for (int i = 0; i < K; ++i) {
      const float dy = tp.dy(i), fdy = floorf(dy);
      const int yT = clampi(pm_safe_int(fdy), 0, Hs - 1) * Ws, yB = clampi(pm_safe_int(fdy + 1), 0, Hs - 1) * Ws;
      const float yB_P = dy - fdy, yT_P = 1.f - yB_P;
#pragma unroll 1
      for (int j = 0; j < K; ++j) {
        const float dx = tp.dx(j), fdx = floorf(dx);
        const int xL = clampi(pm_safe_int(fdx), 0, Ws - 1), xR = clampi(pm_safe_int(fdx + 1), 0, Ws - 1);
        const float xR_P = dx - fdx, xL_P = 1.f - xR_P;
        const float gv = at[(int64_t)(i * K + j) * HW] * go;
        atomic_add(gp + yT + xL, gv * xL_P * yT_P);
        atomic_add(gp + yT + xR, gv * xR_P * yT_P);
        atomic_add(gp + yB + xL, gv * xL_P * yB_P);
        atomic_add(gp + yB + xR, gv * xR_P * yB_P);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ resample2d
// d/d input1 (resample2d_kernel.cu:195-198): weight of tap (r, s) = SAFE_DIV(wy[r], sum) * wx[s]; the 2*KH x 2*KH
// taps of a pixel are a dense patch when dilation == 1.  Weights as in the LDS kernels (Taps::init<true>).
template <int KH>
__global__ __launch_bounds__(256) void rs_patch_table_kernel(const float *__restrict__ in2,
                                                            PatchEntry *__restrict__ table,
                                                            int2 *__restrict__ tile_rows,
                                                            unsigned *__restrict__ stat, int H, int W, int Hi, int Wi,
                                                            int trunc) {
  constexpr int P = 2 * KH, E = P * P;
  const int tile = blockIdx.x, ntiles = gridDim.x;
  const int64_t b = blockIdx.y;
  const int HW = H * W;
  const int pl = threadIdx.x >> 3, sub = threadIdx.x & (kPmSub - 1);
  const int p = tile * kPmTile + pl;
  PatchEntry *out = table + ((b * ntiles + tile) * kPmTile + pl) * (int64_t)E;
  int lo = 0x7fffffff, hi = -1;
  if (p < HW) {
    const int y = p / W, x = p - y * W;
    const float *i2 = in2 + b * 3 * (int64_t)HW + p;
    Taps<float, KH> t;
    t.template init<2>(i2[0], i2[HW], i2[2 * (int64_t)HW], x, y, Hi, Wi, 1, trunc != 0);
    const int y0 = pm_safe_int((float)t.iy0) - (KH - 1), x0 = pm_safe_int((float)t.ix0) - (KH - 1);
    lo = clampi(y0, 0, Hi - 1);
    hi = clampi(y0 + P - 1, 0, Hi - 1);
    float wy[P], wx[P];
#pragma unroll
    for (int r = 0; r < P; ++r) {
      wy[r] = (float)safe_div<float>(t.row_w(r), t.sum);
      wx[r] = t.col_w(r);
    }
    for (int e = sub; e < E; e += kPmSub) {
      const int r = e / P, s = e - r * P;
      const Fold f = pm_fold(r, s, y0, x0, P, Hi, Wi);
      float v = 0.f;
      if (f.canonical) {
#pragma unroll
        for (int rr = 0; rr < P; ++rr)
#pragma unroll
          for (int ss = 0; ss < P; ++ss)
            if (rr >= f.r_lo && rr <= f.r_hi && ss >= f.s_lo && ss <= f.s_hi) v += wy[rr] * wx[ss];
      }
      out[e] = f.canonical ? PatchEntry{(f.ty << 16) | f.tx, v} : PatchEntry{-1, 0.f};
    }
  } else {
    for (int e = sub; e < E; e += kPmSub) out[e] = PatchEntry{-1, 0.f};
  }
  pm_store_tile_rows(lo, hi, tile_rows + b * ntiles + tile, stat);
}

// ------------------------------------------------------------------------------------------- the generic product
// dS[b, c, band rows] (+)= sum over the tiles reaching the band of G[b, c, tile pixels] * W[tile pixels -> band]
// grid (bands, B, ceil(C / 128)); 256 threads = 4 waves, wave w owns channels 32w..32w+31 of the slice.
constexpr int kPmMaxEntriesPerThread = 5;  // ceil(36 / 8)

typedef float pm_f32x4 __attribute__((ext_vector_type(4)));

// global -> registers for one tile: this thread's 16 pixels of one channel of G and its patch-table entries
__device__ __forceinline__ void pm_fetch(pm_f32x4 (&gv)[4], int (&peq)[kPmMaxEntriesPerThread],
                                         float (&pev)[kPmMaxEntriesPerThread], int tile, const float *g_row, bool g_ok,
                                         bool g_vec, int gh, int HW, const PatchEntry *tab, int sub, int E) {
  const int pbase = tile * kPmTile + gh * 16;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = pbase + 4 * q;
    pm_f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (g_ok) {
      const float *src = g_row + (int64_t)tile * kPmTile + 4 * q;
      if (g_vec && p + 3 < HW) {
        v = *reinterpret_cast<const pm_f32x4 *>(src);
      } else {
        if (p < HW) v[0] = src[0];
        if (p + 1 < HW) v[1] = src[1];
        if (p + 2 < HW) v[2] = src[2];
        if (p + 3 < HW) v[3] = src[3];
      }
    }
    gv[q] = v;
  }
  const unsigned long long *tp = reinterpret_cast<const unsigned long long *>(tab + (int64_t)tile * kPmTile * E);
#pragma unroll
  for (int m = 0; m < kPmMaxEntriesPerThread; ++m) {
    const int e = sub + kPmSub * m;
    unsigned long long raw = 0xffffffffull;  // q = -1
    if (e < E) raw = tp[e];
    peq[m] = (int)(unsigned)(raw & 0xffffffffull);
    pev[m] = __uint_as_float((unsigned)(raw >> 32));
  }
}

template <int NB>
__global__ __launch_bounds__(256, 2) void patch_scatter_mfma_kernel(const float *__restrict__ G, float *__restrict__ dS,
                                                                   const PatchEntry *__restrict__ table,
                                                                   const int2 *__restrict__ tile_rows, int C, int HW,
                                                                   int Hs, int Ws, int R, int ntiles, int E,
                                                                   int accumulate, const unsigned *__restrict__ stat,
                                                                   unsigned limit) {
  constexpr int NQ = NB * 32;
  // all-zero (K step, block) pairs are skipped where a band has three or more blocks; with two (64-wide planes: one row per
  // band) the bookkeeping cost more than the MFMAs it saved (aggregation backward at 64x64: 490 -> 570 us)
  constexpr bool kSkip = NB >= 3;
  if (pm_stat_total(stat) > limit) return;  // adaptive dispatch: the flow spreads the patches too far, the LDS-atomic kernel runs
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  float *wt = reinterpret_cast<float *>(gfla_smem);                   // [32][NQ]
  float *gt = wt + kPmTile * NQ;                                      // [128][33]
  unsigned short *list = reinterpret_cast<unsigned short *>(gt + kPmChannels * kPmGPitch);  // tiles reaching the band
  __shared__ int s_nlist;
  // which (K step, 32-column block) pairs of the current tile's W hold a non-zero: a tile's 32 pixels reach a band with a few
  // of their patch rows only, and a pixel's six columns fall into one or two of the band's blocks -- most pairs are all zero,
  // and their MFMAs (64 cycles each) are skipped
  __shared__ unsigned s_mask[3];   // 16 K steps x NB <= 6 blocks = up to 96 bits
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, kh = lane >> 5;
  const int band = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int c0 = blockIdx.z * kPmChannels;
  const int by0 = band * R, by1 = min(Hs, by0 + R) - 1;
  const int nq = (by1 - by0 + 1) * Ws;

  // ascending list of the tiles whose patches reach rows [by0, by1]: built by wave 0, 64 tiles per step
  if (wave == 0) {
    const int2 *tr = tile_rows + b * ntiles;
    int n = 0;
    for (int base = 0; base < ntiles; base += 64) {
      const int tile = base + lane;
      bool f = false;
      if (tile < ntiles) {
        const int2 rr = tr[tile];
        f = rr.x <= by1 && rr.y >= by0;
      }
      const unsigned long long m = __ballot(f);
      if (f) list[n + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)tile;
      n += __popcll(m);
    }
    if (lane == 0) s_nlist = n;
  }
  __syncthreads();
  const int nlist = s_nlist;

  pm_f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // staging roles: G tile -- thread = (channel t >> 1, 16-pixel half t & 1); W tile -- thread = (pixel t >> 3, slot t & 7)
  const int gc = t >> 1, gh = t & 1;
  const bool g_ok = c0 + gc < C;
  const float *g_row = G + (b * C + (g_ok ? c0 + gc : 0)) * (int64_t)HW + gh * 16;
  const bool g_vec = (HW & 3) == 0 && (reinterpret_cast<uintptr_t>(G) & 15) == 0;
  const int pl = t >> 3, sub = t & (kPmSub - 1);
  const PatchEntry *tab = table + (b * ntiles * kPmTile + pl) * (int64_t)E;

  pm_f32x4 gv[4];
  int peq[kPmMaxEntriesPerThread];
  float pev[kPmMaxEntriesPerThread];
  if (nlist > 0) pm_fetch(gv, peq, pev, list[0], g_row, g_ok, g_vec, gh, HW, tab, sub, E);
  for (int li = 0; li < nlist; ++li) {
    __syncthreads();  // the previous tile's MFMAs are done with wt / gt
#pragma unroll
    for (int i = 0; i < NB; ++i) reinterpret_cast<float4 *>(wt)[t + 256 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kSkip && t < 3) s_mask[t] = 0u;
    {
      float *dst = gt + gc * kPmGPitch + gh * 16;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        dst[4 * q + 0] = gv[q][0];
        dst[4 * q + 1] = gv[q][1];
        dst[4 * q + 2] = gv[q][2];
        dst[4 * q + 3] = gv[q][3];
      }
    }
    __syncthreads();  // wt is zero everywhere before any entry lands
    unsigned blocks = 0;   // the column blocks this thread's entries fall into
#pragma unroll
    for (int m = 0; m < kPmMaxEntriesPerThread; ++m) {
      const int q = peq[m];
      if (q >= 0) {
        const int ty = q >> 16, tx = q & 0xffff;
        if (ty >= by0 && ty <= by1) {
          const int col = (ty - by0) * Ws + tx;
          wt[pl * NQ + col] = pev[m];
          blocks |= 1u << (col >> 5);
        }
      }
    }
    // 16 consecutive threads hold the two pixels of ONE K step: their blocks are OR-ed across the 16 lanes and published by
    // one lane (a thread-per-entry atomicOr put ~1000 same-address LDS atomics into every tile pass: at 64-wide planes that
    // cost more than the skipped MFMAs saved)
    if constexpr (kSkip) {
      blocks |= (unsigned)__shfl_xor((int)blocks, 1);
      blocks |= (unsigned)__shfl_xor((int)blocks, 2);
      blocks |= (unsigned)__shfl_xor((int)blocks, 4);
      blocks |= (unsigned)__shfl_xor((int)blocks, 8);
    }
    if (kSkip && (t & 15) == 0 && blocks) {
      const int base = (t >> 4) * NB, w = base >> 5, sh = base & 31;   // bits [base, base + NB) of the mask
      atomicOr(&s_mask[w], blocks << sh);
      if (sh + NB > 32) atomicOr(&s_mask[w + 1], blocks >> (32 - sh));
    }
    if (li + 1 < nlist)  // in flight during the MFMAs
      pm_fetch(gv, peq, pev, list[li + 1], g_row, g_ok, g_vec, gh, HW, tab, sub, E);
    __syncthreads();
    // D[c][q] += G[c][p] W[p][q]: A = G (rows = channels), B = W (columns = band positions), K = the 32 pixels
    // wave-uniform: scalar registers
    const unsigned mk0 = kSkip ? (unsigned)__builtin_amdgcn_readfirstlane((int)s_mask[0]) : 0xffffffffu;
    const unsigned mk1 = kSkip ? (unsigned)__builtin_amdgcn_readfirstlane((int)s_mask[1]) : 0xffffffffu;
    const unsigned mk2 = kSkip ? (unsigned)__builtin_amdgcn_readfirstlane((int)s_mask[2]) : 0xffffffffu;
    const float *ga = gt + (wave * 32 + l31) * kPmGPitch + kh;
    const float *wb = wt + kh * NQ + l31;
    float a = ga[0];
    float bq[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bq[j] = wb[32 * j];
#pragma unroll 4
    for (int ks = 0; ks < kPmTile / 2; ++ks) {
      const float ac = a;
      float bc[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) bc[j] = bq[j];
      const int kn = ks + 1 < kPmTile / 2 ? ks + 1 : ks;  // operands of the next K step, ahead of this step's MFMAs
      a = ga[2 * kn];
#pragma unroll
      for (int j = 0; j < NB; ++j) bq[j] = wb[2 * kn * NQ + 32 * j];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NB; ++j)
      {
        const int bit = ks * NB + j;
        const unsigned w = bit < 32 ? mk0 : (bit < 64 ? mk1 : mk2);
        if (!kSkip || ((w >> (bit & 31)) & 1u)) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac, bc[j], acc[j], 0, 0, 0);
      }
    }
  }
  // C/D layout: column = lane & 31 (band position), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (channel)
  // One base pointer per lane, the 16 rows as multiples of the plane pitch; when accumulating, ALL old values are requested
  // before the first is used (a load + wait + add + store per element was 16 NB dependent round trips per workgroup)
  const int64_t plane = (int64_t)Hs * Ws;
  const int cb = c0 + wave * 32 + 4 * kh;
  float *ob = dS + ((b * C + cb) * (int64_t)Hs + by0) * Ws + l31;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const bool col_ok = 32 * j + l31 < nq;
    float oldv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dc = (r & 3) + 8 * (r >> 2);
      const bool ok = col_ok && cb + dc < C;
      oldv[r] = (accumulate && ok) ? ob[dc * plane + 32 * j] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dc = (r & 3) + 8 * (r >> 2);
      if (col_ok && cb + dc < C) ob[dc * plane + 32 * j] = oldv[r] + acc[j][r];
    }
  }
}

// Band height: the largest R with R * Ws <= 96 positions (three 32-column blocks), one row for wider planes.  Measured
// at the bench shapes (tools/bench_scatter.py, profiles/r2_scatter_paths.jsonl): 64x44 planes 2 rows 256 us,
// 4 rows 293 us; 32x22 planes 4 rows 128 us, 2 rows 141 us, 1 row 151 us.
struct PmGeo {
  int R, NB, bands;
};
static PmGeo pm_geometry(int64_t B, int64_t C, int Hs, int Ws, int W, int patch) {
  (void)B, (void)C, (void)W, (void)patch;
  int R = tuning(13) > 0 ? tuning(13) : 96 / Ws;
  // planes wider than 48: one row would be a band of two blocks -- too little work per tile pass, and no skipping of its zero
  // slices; up to six blocks instead (aggregation backward at 64x64: 501 us with one row per band, 428 with two, 415 with three)
  if (tuning(13) <= 0 && R < 2) R = (kPmMaxNB * 32) / Ws;
  if (R < 1) R = 1;
  if (R > Hs) R = Hs;
  while (R > 1 && R * Ws > kPmMaxNB * 32) --R;
  if (R * Ws > kPmMaxNB * 32) return PmGeo{0, 0, 0};
  return PmGeo{R, (int)ceil_div((int64_t)R * Ws, 32), (int)ceil_div(Hs, R)};
}

// tuning key 14: 0 = where it measured faster (pm_auto), 1 = never, 2 = wherever the shape is supported
static bool pm_auto(bool aggregate, int64_t Ws) {
  if (tuning(14) == 2) return true;
  // bench shapes, smooth / coherent flows, with the all-zero K slices skipped (profiles/r4_scatter_paths_after_zero_skip.jsonl,
  // r4_scatter_paths_mfma_everywhere.jsonl): aggregation 64x44 376 -> 153 us, 32x22 95 -> 69 (round 3, without the skipping:
  // 90..100, and the LDS-atomic kernel was kept there); resample2d 32x22 138 -> 91, 64x44 210 (LDS planes fed from tap
  // records) vs 214 (coherent: 194 vs 229)
  return aggregate ? true : Ws <= 24;
}

// dynamic LDS of patch_scatter_mfma_kernel: the W / G tiles plus 2 bytes per flow tile; plain <<<>>> launches get 64 KB
static int64_t pm_lds_bytes(int NB, int64_t ntiles) {
  return (int64_t)(kPmTile * NB * 32 + kPmChannels * kPmGPitch) * (int64_t)sizeof(float) + ((ntiles + 7) & ~(int64_t)7) * 2;
}

static bool pm_supported(int64_t B, int64_t C, int64_t H, int64_t W, int64_t Hs, int64_t Ws) {
  if (!(B > 0 && B <= 65535 && C > 0 && Hs >= 2 && Ws >= 2 && Ws <= kPmMaxNB * 32 && Hs < 32768 && Ws < 32768 &&
        H * W <= (int64_t)kPmTile * 65535 && ceil_div(C, kPmChannels) <= 65535 && tuning(14) != 1))
    return false;
  // the per-tile row table lives in LDS: a flow field with more tiles than fit the 64 KB launch limit is UNSUPPORTED
  // here (the callers fall back to the LDS-atomic kernels) instead of failing at launch
  const PmGeo g = pm_geometry(B, C, (int)Hs, (int)Ws, (int)W, 0);
  return g.R > 0 && pm_lds_bytes(g.NB, ceil_div(H * W, kPmTile)) <= 64 * 1024;
}

static int pm_scatter(const float *G, float *dS, const PatchEntry *table, const int2 *tile_rows, int64_t B, int64_t C,
                      int64_t H, int64_t W, int64_t Hs, int64_t Ws, int entries, int patch, int accumulate,
                      const unsigned *stat, unsigned limit, hipStream_t stream) {
  const PmGeo g = pm_geometry(B, C, (int)Hs, (int)Ws, (int)W, patch);
  if (g.R == 0) return GFLA_ERR_UNSUPPORTED;
  const int ntiles = (int)ceil_div(H * W, kPmTile);
  const dim3 grid((unsigned)g.bands, (unsigned)B, (unsigned)ceil_div(C, kPmChannels));
  const unsigned lds = (unsigned)pm_lds_bytes(g.NB, ntiles);
  if (lds > 64 * 1024) return GFLA_ERR_UNSUPPORTED;
#define GFLA_PM(NB_)                                                                                              \
  case NB_:                                                                                                       \
    patch_scatter_mfma_kernel<NB_><<<grid, 256, lds, stream>>>(G, dS, table, tile_rows, (int)C, (int)(H * W), (int)Hs, \
                                                               (int)Ws, g.R, ntiles, entries, accumulate, stat, limit); \
    break
  switch (g.NB) {
    GFLA_PM(1);
    GFLA_PM(2);
    GFLA_PM(3);
    GFLA_PM(4);
    GFLA_PM(5);
    GFLA_PM(6);
    default: return GFLA_ERR_UNSUPPORTED;
  }
#undef GFLA_PM
  return launch_status();
}

// Adaptive dispatch.  The product's cost grows with how far the flow spreads a tile's patches over the source rows
// (tile passes ~ rows reached / band height); the LDS-atomic kernels do not care.  The table pass sums the rows
// reached per tile into a device counter; the matrix-core kernels run only if the average stays within `limit`
// rows per tile, otherwise they return at once and the caller's LDS-atomic kernel (launched right behind, with the
// inverse predicate) does the work -- no host synchronisation either way.
// limit_rows (tuning key 15, in rows per tile; 0 = default 2.6 x the rows a tile reaches under a constant flow;
// >= 100000 = always the matrix-core path).
static unsigned pm_limit(int64_t B, int64_t ntiles, int W, int patch, bool adaptive) {
  if (!adaptive || tuning(15) >= 100000) return 0xffffffffu;
  const double still = (double)kPmTile / W + 1.0 + patch;  // rows reached with a constant flow
  const double per_tile = tuning(15) > 0 ? (double)tuning(15) : 2.6 * still;
  const double total = per_tile * (double)B * (double)ntiles;
  return total >= 4.0e9 ? 0xffffffffu : (unsigned)total;
}

// d/d source of the attention-weighted aggregation (f32).  Returns GFLA_ERR_UNSUPPORTED (nothing launched) when the
// shape is outside this path; gsrc is accumulated into (+=) when accumulate != 0, overwritten otherwise.
// adaptive != 0: *skip_stat / *skip_limit receive the predicate for the caller's fallback kernel (skip if
// *stat <= limit); with adaptive == 0 the product always runs.
int agg_source_bwd_mfma(const float *flow, const float *attn, const float *gout, float *gsrc, void *workspace,
                        int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H, int64_t W, int k, int accumulate,
                        int adaptive, const unsigned **skip_stat, unsigned *skip_limit, hipStream_t stream) {
  if ((k != 3 && k != 5) || !pm_supported(B, C, H, W, Hs, Ws) || !workspace || !pm_auto(true, Ws)) return GFLA_ERR_UNSUPPORTED;
  const int entries = (k + 1) * (k + 1);
  const PmLayout L = pm_layout(B, H * W, entries);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  PatchEntry *table = reinterpret_cast<PatchEntry *>(ws);
  int2 *rows = reinterpret_cast<int2 *>(ws + L.table_bytes);
  unsigned *stat = reinterpret_cast<unsigned *>(ws + L.table_bytes + L.rows_bytes);
  const unsigned limit = pm_limit(B, L.ntiles, (int)W, k + 1, adaptive != 0);
  if (hipMemsetAsync(stat, 0, kPmStatSlots * 4, stream) != hipSuccess) return GFLA_ERR_LAUNCH;
  const dim3 tg((unsigned)ceil_div(L.ntiles, 8), (unsigned)B);
  if (k == 3)
    agg_patch_table_kernel<3><<<tg, 256, 0, stream>>>(flow, attn, table, rows, stat, (int)H, (int)W, (int)Hs, (int)Ws, (int)L.ntiles);
  else
    agg_patch_table_kernel<5><<<tg, 256, 0, stream>>>(flow, attn, table, rows, stat, (int)H, (int)W, (int)Hs, (int)Ws, (int)L.ntiles);
  int st = launch_status();
  if (st != GFLA_OK) return st;
  st = pm_scatter(gout, gsrc, table, rows, B, C, H, W, Hs, Ws, entries, k + 1, accumulate, stat, limit, stream);
  if (st != GFLA_OK) return st;
  const dim3 ng((unsigned)ceil_div(H * W, 256), (unsigned)B);
  if (k == 3)
    agg_scatter_nondense_kernel<3><<<ng, 256, 0, stream>>>(flow, attn, gout, gsrc, (int)C, (int)H, (int)W, (int)Hs, (int)Ws, stat, limit);
  else
    agg_scatter_nondense_kernel<5><<<ng, 256, 0, stream>>>(flow, attn, gout, gsrc, (int)C, (int)H, (int)W, (int)Hs, (int)Ws, stat, limit);
  if (skip_stat) *skip_stat = stat;
  if (skip_limit) *skip_limit = limit;
  return launch_status();
}

// d/d input1 of resample2d (f32, dilation 1).  Same conventions.
int rs_input1_bwd_mfma(const float *in2, const float *gout, float *gin1, void *workspace, int64_t B, int64_t C,
                       int64_t Hi, int64_t Wi, int64_t H, int64_t W, int k, int trunc, int accumulate, int adaptive,
                       const unsigned **skip_stat, unsigned *skip_limit, hipStream_t stream) {
  if ((k != 2 && k != 4) || !pm_supported(B, C, H, W, Hi, Wi) || !workspace || !pm_auto(false, Wi)) return GFLA_ERR_UNSUPPORTED;
  const int entries = k * k;
  const PmLayout L = pm_layout(B, H * W, entries);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  PatchEntry *table = reinterpret_cast<PatchEntry *>(ws);
  int2 *rows = reinterpret_cast<int2 *>(ws + L.table_bytes);
  unsigned *stat = reinterpret_cast<unsigned *>(ws + L.table_bytes + L.rows_bytes);
  const unsigned limit = pm_limit(B, L.ntiles, (int)W, k, adaptive != 0);
  if (hipMemsetAsync(stat, 0, kPmStatSlots * 4, stream) != hipSuccess) return GFLA_ERR_LAUNCH;
  const dim3 tg((unsigned)L.ntiles, (unsigned)B);
  if (k == 2)
    rs_patch_table_kernel<1><<<tg, 256, 0, stream>>>(in2, table, rows, stat, (int)H, (int)W, (int)Hi, (int)Wi, trunc);
  else
    rs_patch_table_kernel<2><<<tg, 256, 0, stream>>>(in2, table, rows, stat, (int)H, (int)W, (int)Hi, (int)Wi, trunc);
  const int st = launch_status();
  if (st != GFLA_OK) return st;
  if (skip_stat) *skip_stat = stat;
  if (skip_limit) *skip_limit = limit;
  return pm_scatter(gout, gin1, table, rows, B, C, H, W, Hi, Wi, entries, k, accumulate, stat, limit, stream);
}

int64_t pm_table_bytes(int64_t B, int64_t H, int64_t W, int entries) {
  if (B <= 0 || H <= 0 || W <= 0 || entries <= 0) return 0;
  return (pm_layout(B, H * W, entries).total + 255) & ~(int64_t)255;
}
int64_t pm_workspace_bytes(int64_t B, int64_t H, int64_t W, int entries) {
  if (B <= 0 || H <= 0 || W <= 0 || entries <= 0) return 0;
  return pm_table_bytes(B, H, W, entries) + B * H * W * kRsTapRecBytes;
}

}  // namespace gfla

extern "C" {
/* scratch for the matrix-core scatter paths: the patch table of one op invocation */
int64_t gfla_scatter_workspace_bytes(int64_t B, int64_t H, int64_t W, int patch_entries) {
  return gfla::pm_workspace_bytes(B, H, W, patch_entries);
}
}
