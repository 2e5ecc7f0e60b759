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
#include "rs_taps.h"

namespace gfla {

typedef float pm_f32x16 __attribute__((ext_vector_type(16)));

struct PatchEntry {
  int q;    // (ty << 16) | tx of the clamped target position, or -1: nothing to add
  float v;  // weight
};

constexpr int kPmTile = 32;     // flow pixels per tile = K extent of one MFMA pass
constexpr int kPmSub = 8;       // threads sharing the entries of one pixel
constexpr int kPmChannels = 128;  // channels per workgroup
constexpr int kPmMaxNB = 6;     // 32-column blocks of a band (<= 192 source positions)
constexpr int kPmGPitch = 33;   // floats; LDS pitch of a channel row of the G tile (odd: conflict-free column reads)

struct PmLayout {
  int64_t ntiles, table_bytes, rows_bytes, total;
};
static PmLayout pm_layout(int64_t B, int64_t HW, int entries) {
  PmLayout L;
  L.ntiles = ceil_div(HW, kPmTile);
  L.table_bytes = ((B * L.ntiles * kPmTile * entries * (int64_t)sizeof(PatchEntry)) + 255) & ~(int64_t)255;
  L.rows_bytes = ((B * L.ntiles * (int64_t)sizeof(int2)) + 255) & ~(int64_t)255;
  L.total = L.table_bytes + L.rows_bytes;
  return L;
}

__device__ __forceinline__ int pm_safe_int(float v) {  // saturating float -> int that also keeps later +k defined
  return (int)fminf(fmaxf(v, -1048576.f), 1048576.f);
}

// Block-wide (lo, hi) of the rows a tile reaches -> tile_rows[slot]
__device__ __forceinline__ void pm_store_tile_rows(int lo, int hi, int2 *slot) {
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
  if (threadIdx.x == 0)
    *slot = make_int2(min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3])), max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3])));
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

// grid (ntiles, B), 256 threads = 32 pixels x 8 entry slots
template <int K>
__global__ __launch_bounds__(256) void agg_patch_table_kernel(const float *__restrict__ flow,
                                                             const float *__restrict__ attn,
                                                             PatchEntry *__restrict__ table,
                                                             int2 *__restrict__ tile_rows, int H, int W, int Hs,
                                                             int Ws) {
  constexpr int P = K + 1, E = P * P, KK = K * K;
  const int tile = blockIdx.x, ntiles = gridDim.x;
  const int64_t b = blockIdx.y;
  const int HW = H * W;
  const int pl = threadIdx.x >> 3, sub = threadIdx.x & (kPmSub - 1);
  const int p = tile * kPmTile + pl;
  PatchEntry *out = table + ((b * ntiles + tile) * kPmTile + pl) * (int64_t)E;
  int lo = 0x7fffffff, hi = -1;
  bool live = false;
  AggTaps<K> tp;
  if (p < HW) {
    const int yf = p / W, xf = p - yf * W;
    tp.init(flow[(b * 2 + 0) * HW + p], flow[(b * 2 + 1) * HW + p], xf, yf);
    live = tp.dense;
  }
  if (live) {
    const int y0 = pm_safe_int((float)tp.y0), x0 = pm_safe_int((float)tp.x0);
    lo = clampi(y0, 0, Hs - 1);
    hi = clampi(y0 + K, 0, Hs - 1);
    const float *at = attn + b * (int64_t)KK * HW + p;
    const float inv_kk = 1.f / (float)KK;
    for (int e = sub; e < E; e += kPmSub) {
      const int r = e / P, s = e - r * P;
      const Fold f = pm_fold(r, s, y0, x0, P, Hs, Ws);
      float v = 0.f;
      if (f.canonical) {
        for (int rr = f.r_lo; rr <= f.r_hi; ++rr)
          for (int ss = f.s_lo; ss <= f.s_hi; ++ss) {
            // patch entry (rr, ss) = taps (i, j) in {rr-1, rr} x {ss-1, ss}: block_extractor_kernel.cu:158-161 folded
            for (int i = max(rr - 1, 0); i <= min(rr, K - 1); ++i) {
              const float a_y = tp.ay(i), wy = rr == i ? 1.f - a_y : a_y;
              for (int j = max(ss - 1, 0); j <= min(ss, K - 1); ++j) {
                const float a_x = tp.ax(j), wx = ss == j ? 1.f - a_x : a_x;
                v += (at[(int64_t)(i * K + j) * HW] * inv_kk) * wx * wy;
              }
            }
          }
      }
      out[e] = f.canonical ? PatchEntry{(f.ty << 16) | f.tx, v} : PatchEntry{-1, 0.f};
    }
  } else {
    for (int e = sub; e < E; e += kPmSub) out[e] = PatchEntry{-1, 0.f};
  }
  pm_store_tile_rows(lo, hi, tile_rows + b * ntiles + tile);
}

// Pixels that are not a dense patch: the reference's tap-by-tap scatter with global atomics (rare: a tap within
// rounding of an integer position).  One thread per (b, p); dense pixels return at once.
template <int K>
__global__ __launch_bounds__(256) void agg_scatter_nondense_kernel(const float *__restrict__ flow,
                                                                  const float *__restrict__ attn,
                                                                  const float *__restrict__ gout,
                                                                  float *__restrict__ gsrc, int C, int H, int W,
                                                                  int Hs, int Ws) {
  constexpr int KK = K * K;
  const int HW = H * W;
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
                                                            int2 *__restrict__ tile_rows, int H, int W, int Hi, int Wi,
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
    t.template init<true>(i2[0], i2[HW], i2[2 * (int64_t)HW], x, y, Hi, Wi, 1, trunc != 0);
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
  pm_store_tile_rows(lo, hi, tile_rows + b * ntiles + tile);
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
                                                                   int accumulate) {
  constexpr int NQ = NB * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  float *wt = reinterpret_cast<float *>(gfla_smem);                   // [32][NQ]
  float *gt = wt + kPmTile * NQ;                                      // [128][33]
  unsigned short *list = reinterpret_cast<unsigned short *>(gt + kPmChannels * kPmGPitch);  // tiles reaching the band
  __shared__ int s_nlist;
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
#pragma unroll
    for (int m = 0; m < kPmMaxEntriesPerThread; ++m) {
      const int q = peq[m];
      if (q >= 0) {
        const int ty = q >> 16, tx = q & 0xffff;
        if (ty >= by0 && ty <= by1) wt[pl * NQ + (ty - by0) * Ws + tx] = pev[m];
      }
    }
    if (li + 1 < nlist)  // in flight during the MFMAs
      pm_fetch(gv, peq, pev, list[li + 1], g_row, g_ok, g_vec, gh, HW, tab, sub, E);
    __syncthreads();
    // D[c][q] += G[c][p] W[p][q]: A = G (rows = channels), B = W (columns = band positions), K = the 32 pixels
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
      for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac, bc[j], acc[j], 0, 0, 0);
    }
  }
  // C/D layout: column = lane & 31 (band position), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (channel)
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int q = 32 * j + l31;
    if (q >= nq) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (c < C) {
        float *o = dS + ((b * C + c) * (int64_t)Hs + by0) * Ws + q;
        *o = accumulate ? *o + acc[j][r] : acc[j][r];
      }
    }
  }
}

// Band height: R rows with R * Ws <= 192 positions.  Work per flow tile ~ (bands it reaches) x (column blocks of a
// band); small launches prefer more, narrower bands.
struct PmGeo {
  int R, NB, bands;
};
static PmGeo pm_geometry(int64_t B, int64_t C, int Hs, int Ws, int W, int patch) {
  PmGeo best{0, 0, 0};
  double best_cost = 1e300;
  const int forced = tuning(13);
  for (int R = 1; R <= Hs && R * Ws <= kPmMaxNB * 32; ++R) {
    if (forced > 0 && R != forced) continue;
    const int NB = (int)ceil_div((int64_t)R * Ws, 32);
    const int bands = (int)ceil_div(Hs, R);
    const double reach = (double)kPmTile / W + 1.0 + patch + 1.5;  // source rows one tile reaches (flow slack 1.5)
    double cost = (reach + R - 1) / R * NB;
    const int64_t wgs = (int64_t)bands * B * ceil_div(C, kPmChannels);
    if (wgs < 2 * kNumCU) cost *= (double)(2 * kNumCU) / (double)wgs;
    if (cost < best_cost) best_cost = cost, best = PmGeo{R, NB, bands};
  }
  return best;
}

static bool pm_supported(int64_t B, int64_t C, int64_t H, int64_t W, int64_t Hs, int64_t Ws) {
  return B > 0 && B <= 65535 && C > 0 && Hs >= 2 && Ws >= 2 && Ws <= kPmMaxNB * 32 && Hs < 32768 && Ws < 32768 &&
         H * W <= (int64_t)kPmTile * 65535 && ceil_div(C, kPmChannels) <= 65535 && tuning(14) != 1;
}

static int pm_scatter(const float *G, float *dS, const PatchEntry *table, const int2 *tile_rows, int64_t B, int64_t C,
                      int64_t H, int64_t W, int64_t Hs, int64_t Ws, int entries, int patch, int accumulate,
                      hipStream_t stream) {
  const PmGeo g = pm_geometry(B, C, (int)Hs, (int)Ws, (int)W, patch);
  if (g.R == 0) return GFLA_ERR_UNSUPPORTED;
  const int ntiles = (int)ceil_div(H * W, kPmTile);
  const dim3 grid((unsigned)g.bands, (unsigned)B, (unsigned)ceil_div(C, kPmChannels));
  const unsigned lds = (unsigned)((kPmTile * g.NB * 32 + kPmChannels * kPmGPitch) * sizeof(float) + ((ntiles + 7) & ~7) * 2);
#define GFLA_PM(NB_)                                                                                              \
  case NB_:                                                                                                       \
    patch_scatter_mfma_kernel<NB_><<<grid, 256, lds, stream>>>(G, dS, table, tile_rows, (int)C, (int)(H * W), (int)Hs, \
                                                               (int)Ws, g.R, ntiles, entries, accumulate);        \
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

// d/d source of the attention-weighted aggregation (f32).  Returns GFLA_ERR_UNSUPPORTED (nothing launched) when
// the shape is outside this path; gsrc is accumulated into (+=) when accumulate != 0, overwritten otherwise.
int agg_source_bwd_mfma(const float *flow, const float *attn, const float *gout, float *gsrc, void *workspace,
                        int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H, int64_t W, int k, int accumulate,
                        hipStream_t stream) {
  if ((k != 3 && k != 5) || !pm_supported(B, C, H, W, Hs, Ws) || !workspace) return GFLA_ERR_UNSUPPORTED;
  const int entries = (k + 1) * (k + 1);
  const PmLayout L = pm_layout(B, H * W, entries);
  PatchEntry *table = static_cast<PatchEntry *>(workspace);
  int2 *rows = reinterpret_cast<int2 *>(static_cast<unsigned char *>(workspace) + L.table_bytes);
  const dim3 tg((unsigned)L.ntiles, (unsigned)B);
  if (k == 3)
    agg_patch_table_kernel<3><<<tg, 256, 0, stream>>>(flow, attn, table, rows, (int)H, (int)W, (int)Hs, (int)Ws);
  else
    agg_patch_table_kernel<5><<<tg, 256, 0, stream>>>(flow, attn, table, rows, (int)H, (int)W, (int)Hs, (int)Ws);
  int st = launch_status();
  if (st != GFLA_OK) return st;
  st = pm_scatter(gout, gsrc, table, rows, B, C, H, W, Hs, Ws, entries, k + 1, accumulate, stream);
  if (st != GFLA_OK) return st;
  const dim3 ng((unsigned)ceil_div(H * W, 256), (unsigned)B);
  if (k == 3)
    agg_scatter_nondense_kernel<3><<<ng, 256, 0, stream>>>(flow, attn, gout, gsrc, (int)C, (int)H, (int)W, (int)Hs, (int)Ws);
  else
    agg_scatter_nondense_kernel<5><<<ng, 256, 0, stream>>>(flow, attn, gout, gsrc, (int)C, (int)H, (int)W, (int)Hs, (int)Ws);
  return launch_status();
}

// d/d input1 of resample2d (f32, dilation 1).  Same conventions.
int rs_input1_bwd_mfma(const float *in2, const float *gout, float *gin1, void *workspace, int64_t B, int64_t C,
                       int64_t Hi, int64_t Wi, int64_t H, int64_t W, int k, int trunc, int accumulate,
                       hipStream_t stream) {
  if ((k != 2 && k != 4) || !pm_supported(B, C, H, W, Hi, Wi) || !workspace) return GFLA_ERR_UNSUPPORTED;
  const int entries = k * k;
  const PmLayout L = pm_layout(B, H * W, entries);
  PatchEntry *table = static_cast<PatchEntry *>(workspace);
  int2 *rows = reinterpret_cast<int2 *>(static_cast<unsigned char *>(workspace) + L.table_bytes);
  const dim3 tg((unsigned)L.ntiles, (unsigned)B);
  if (k == 2)
    rs_patch_table_kernel<1><<<tg, 256, 0, stream>>>(in2, table, rows, (int)H, (int)W, (int)Hi, (int)Wi, trunc);
  else
    rs_patch_table_kernel<2><<<tg, 256, 0, stream>>>(in2, table, rows, (int)H, (int)W, (int)Hi, (int)Wi, trunc);
  const int st = launch_status();
  if (st != GFLA_OK) return st;
  return pm_scatter(gout, gin1, table, rows, B, C, H, W, Hi, Wi, entries, k, accumulate, stream);
}

int64_t pm_workspace_bytes(int64_t B, int64_t H, int64_t W, int entries) {
  if (B <= 0 || H <= 0 || W <= 0 || entries <= 0) return 0;
  return pm_layout(B, H * W, entries).total;
}

}  // namespace gfla

extern "C" {
/* scratch for the matrix-core scatter paths: the patch table of one op invocation */
int64_t gfla_scatter_workspace_bytes(int64_t B, int64_t H, int64_t W, int patch_entries) {
  return gfla::pm_workspace_bytes(B, H, W, patch_entries);
}
}
