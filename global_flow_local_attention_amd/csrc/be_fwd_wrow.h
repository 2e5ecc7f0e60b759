// block_extractor forward in the REFERENCE layout (B, C, K*Hf, K*Wf): lane = flow pixel for the arithmetic, a WAVE = whole
// flow rows, so that what a wave stores is one contiguous piece of the output plane (round 4).
//
// Measured on MI355X on the way here (profiles/r4_be_fwd_*.jsonl, tools/ubench/store_patterns.hip):
//   * the lane-per-pixel kernel with direct stores (be_fwd_pix.h) is bound by its write stream -- 340 us at
//     (32,128,64,44) k=5 with or without its LDS reads, 80 us with the stores compiled out -- and that stream depends on
//     how the K*Wf-float output rows line up with the L2's 128-byte lines: 5.1-5.2 TB/s when a row is a whole number of
//     lines (Wf = 32, 64, 96), 3.5-3.7 TB/s when it is not (880-byte rows at the north star's own shape);
//   * a store-only micro-benchmark reproduces it with no arithmetic at all: the same bytes written flat (a workgroup
//     streaming its planes front to back) go at 5.5-5.7 TB/s whatever the width, any per-pixel or per-64-pixel-row
//     pattern at 3.2-3.8 TB/s when Wf = 44 -- whether the lanes of an instruction are consecutive or not.  What costs is
//     a line that is completed by a DIFFERENT wave, later;
//   * routing the outputs through a workgroup-wide LDS tile (compute, barrier, flat copy, barrier) writes perfect lines
//     and runs at 2.2 TB/s: the phases serialise; a per-wave LDS row per output row (two LDS round trips per row): 3.2.
// Here a wave owns `rpw` consecutive flow rows (rpw * Wf <= 64 lanes).  Their K * rpw output rows are ONE contiguous
// piece of the plane, rpw*K*K*Wf elements, which the wave builds in its private LDS tile (laid out as the piece, same
// 16-byte phase as in global memory) and streams out itself, 16 bytes per lane, lanes consecutive: one LDS round trip per
// channel, no workgroup barrier, and only the first / last line of a piece is shared with another wave.
// The planes sit in LDS replicate-padded as in be_fwd_pix.h; arithmetic = the reference's expression and order
// (block_extractor_kernel.cu:62-84) with fused multiply-adds.
#pragma once

#include "be_fwd_pix.h"

namespace gfla {

template <typename T, int K, int ABL = 0>
__global__ __launch_bounds__(1024) void be_fwd_wrow_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, T *__restrict__ out, int C, int Hs, int Ws, int Hf, int Wf,
    int G, int ngroups, int rpw, int tile_off, int tile_stride) {
  using A = typename Num<T>::acc;
  constexpr int PAD = K;
  constexpr int V = 16 / sizeof(T);   // elements of a 16-byte store
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  A *planes = reinterpret_cast<A *>(gfla_smem);
  const int g = blockIdx.x % ngroups;
  const int b = blockIdx.x / ngroups;
  const int c0 = g * G;
  const int gc = min(G, C - c0);
  const int Wp = Ws + 2 * PAD;
  const int plane_p = Hs * Wp;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthreads >> 6;
  T *tile = reinterpret_cast<T *>(gfla_smem + tile_off + wave * tile_stride);   // this wave's piece

  // ---- stage the gc planes, interior first (coalesced), then the replicated columns -------------------------------
  {
    const T *gsrc = src + ((int64_t)b * C + c0) * ((int64_t)Hs * Ws);
    const int n = gc * Hs * Ws;
    bool vec = false;
    if constexpr (sizeof(T) == 4) {
      vec = (Ws & 3) == 0 && (reinterpret_cast<uintptr_t>(gsrc) & 15) == 0;
      if (vec) {
        const int W4 = Ws >> 2, n4 = n >> 2;
        const float4 *g4 = reinterpret_cast<const float4 *>(gsrc);
#pragma unroll 4
        for (int i = tid; i < n4; i += nthreads) {
          const float4 v = g4[i];
          const int r = i / W4, x = (i - r * W4) << 2;
          A *d = planes + r * Wp + PAD + x;
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
      }
    }
    if (!vec) {
#pragma unroll 4
      for (int i = tid; i < n; i += nthreads) {
        const int r = i / Ws, x = i - r * Ws;
        planes[r * Wp + PAD + x] = Num<T>::ld(gsrc + i);
      }
    }
    __syncthreads();
    const int rows = gc * Hs;
    for (int i = tid; i < rows * 2 * PAD; i += nthreads) {
      const int r = i / (2 * PAD), q = i - r * (2 * PAD);
      A *row = planes + r * Wp;
      if (q < PAD) row[q] = row[PAD];
      else row[Ws + q] = row[PAD + Ws - 1];
    }
    __syncthreads();
  }

  const int HW = Hf * Wf;
  const int Wo = K * Wf;
  const int64_t oplane = (int64_t)(K * Hf) * Wo;
  const T *flow_x = flow + (int64_t)(b * 2 + 0) * HW;
  const T *flow_y = flow + (int64_t)(b * 2 + 1) * HW;
  T *og = out + ((int64_t)b * C + c0) * oplane;          // output plane of the group's first channel
  // 16-byte phase of a plane's first element: planes are oplane elements apart, `out` itself may be offset
  const int phase0 = (int)((reinterpret_cast<uintptr_t>(og) / sizeof(T)) % V);
  const int plane_phase = (int)(oplane % V);
  const int yl = lane / Wf, xf = lane - yl * Wf;          // this lane's pixel inside a row group

  for (int y0 = wave * rpw; y0 < Hf; y0 += nwaves * rpw) {
    const int rows = min(rpw, Hf - y0);
    const int piece = rows * K * Wo;                      // elements of the row group's piece of one output plane
    const int64_t piece_off = (int64_t)y0 * K * Wo;       // its offset inside the plane
    // ---- per-pixel setup, once per row group (kept in registers for the G channels) --------------------------------
    const bool active = yl < rows;
    const int yf = active ? y0 + yl : y0;
    const A fx0 = Num<T>::ld(flow_x + yf * Wf + xf);
    const A fy0 = Num<T>::ld(flow_y + yf * Wf + xf);
    A ax[K], ay[K];
    int x0 = 0, yy0 = 0;
    bool dense = true;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const A dx = (fx0 + (A)(t - K / 2)) + (A)xf;  // block_extractor_kernel.cu:62-67
      const A dy = (fy0 + (A)(t - K / 2)) + (A)yf;
      const A fdx = floor_t<A>(dx), fdy = floor_t<A>(dy);
      if (t == 0) {
        x0 = (int)fdx;
        yy0 = (int)fdy;
      }
      dense &= ((int)fdx == x0 + t) & ((int)fdy == yy0 + t);
      ax[t] = dx - fdx;
      ay[t] = dy - fdy;
    }
    const int x0c = clampi(x0, -PAD, Ws - 1) + PAD;       // padded column of tap 0 (dense pixels)
    const int y0c = clampi(yy0, -(K + 1), Hs);
    const int trow = (yl * K) * Wo + K * xf;              // this pixel's first output inside the piece

    for (int cc = 0; cc < gc; ++cc) {
      // the tile starts at the 16-byte phase the piece has in global memory, so 16-byte chunks line up on both sides
      const int ph = (int)((phase0 + (int64_t)cc * plane_phase + piece_off) % V);
      T *tl = tile + ph;
      const A *pc = planes + cc * plane_p;
      if (active) {
        if (dense) {
          // Dense patch: the bilinear form separated -- every patch row interpolated ALONG x once (K values), then output
          // row i = the blend of interpolated rows i and i + 1:
          //     h_r[j] = xL_j v[r][j] + xR_j v[r][j+1]          out[i][j] = yT_i h_i[j] + yB_i h_{i+1}[j]
          // (K+1)*K*2 + K*K*2 = 110 operations per pixel and channel for K = 5 against the 200 of the reference's
          // four-term sum with its four weight products per output (:73-84) -- at this kernel's shape the vector ALUs
          // were what kept it from the write stream's rate (profiles/r4_be_fwd_wrow_ablations.jsonl).  Same value up to
          // rounding (three roundings per output either way); zero weights still reproduce the source bit for bit, and
          // the tap-by-tap branch below is the reference's expression unchanged.  be_fwd_pix.h uses the same expressions:
          // the two kernels agree bit for bit.
          auto hrow = [&](int r, A (&h)[K]) {
            const A *pr = pc + clampi(y0c + r, 0, Hs - 1) * Wp + x0c;
            A v[K + 1];
#pragma unroll
            for (int s = 0; s <= K; ++s) v[s] = (ABL & 1) ? (A)(lane + s + r) : pr[s];
#pragma unroll
            for (int j = 0; j < K; ++j) h[j] = fma_t(ax[j], v[j + 1], (1 - ax[j]) * v[j]);
          };
          A hA[K];
          hrow(0, hA);
#pragma unroll
          for (int i = 0; i < K; ++i) {
            A hB[K];
            hrow(i + 1, hB);
            const A yB_P = ay[i], yT_P = 1 - yB_P;
            T *to = tl + trow + i * Wo;
#pragma unroll
            for (int j = 0; j < K; ++j) to[j] = Num<T>::from(fma_t(yB_P, hB[j], yT_P * hA[j]));
#pragma unroll
            for (int j = 0; j < K; ++j) hA[j] = hB[j];
          }
        } else {  // a coordinate within rounding of an integer: tap by tap, as the reference does
          int xL[K], xR[K];
#pragma unroll
          for (int t = 0; t < K; ++t) {
            const A dx = (fx0 + (A)(t - K / 2)) + (A)xf;
            const A fdx = floor_t<A>(dx);
            xL[t] = clampi((int)fdx, 0, Ws - 1) + PAD;  // :69-72
            xR[t] = clampi((int)(fdx + 1), 0, Ws - 1) + PAD;
          }
#pragma unroll 1
          for (int i = 0; i < K; ++i) {
            const A dy = (fy0 + (A)(i - K / 2)) + (A)yf;
            const A fdy = floor_t<A>(dy);
            const int yT = clampi((int)fdy, 0, Hs - 1) * Wp, yB = clampi((int)(fdy + 1), 0, Hs - 1) * Wp;
            const A yB_P = dy - fdy, yT_P = 1 - yB_P;
            T *to = tl + trow + i * Wo;
#pragma unroll
            for (int j = 0; j < K; ++j) {
              const A xR_P = ax[j], xL_P = 1 - xR_P;
              A s = (xL_P * yT_P) * pc[yT + xL[j]];
              s = fma_t(xR_P * yT_P, pc[yT + xR[j]], s);
              s = fma_t(xL_P * yB_P, pc[yB + xL[j]], s);
              s = fma_t(xR_P * yB_P, pc[yB + xR[j]], s);
              to[j] = Num<T>::from(s);
            }
          }
        }
      }
      // (LDS operations of one wave execute in order: the reads below see the writes above, and the next channel's
      // writes cannot overtake them; the compiler only has to keep the program order)
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
      // ---- copy: tile -> the piece of output plane c0 + cc, whole 16-byte chunks, consecutive lanes -----------------
      {
        T *dst = og + (int64_t)cc * oplane + piece_off;
        const int head = min(piece, (V - ph) % V);      // elements before the first 16-byte boundary
        const int body = (piece - head) / V;
        const int tail0 = head + body * V;
        if (lane < head) dst[lane] = tl[lane];
        if (lane < piece - tail0) dst[tail0 + lane] = tl[tail0 + lane];
        typedef T V4 __attribute__((ext_vector_type(V)));   // naturally (16-byte) aligned: both sides are, by construction
        const V4 *t4 = reinterpret_cast<const V4 *>(tl + head);
        V4 *d4 = reinterpret_cast<V4 *>(dst + head);
#pragma unroll 2
        for (int m = lane; m < body; m += 64) {
          const V4 val = t4[m];
          if constexpr (ABL & 2) {   // timing ablation (make PROBES=1): no stores; one impossible store keeps the reads alive
            if (val[0] == (T)12345.678) d4[m] = val;
          } else {
            d4[m] = val;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
    }
  }
}

struct WrowGeo {
  int G, ngroups, rpw, threads, tile_off, tile_stride;
  unsigned lds_bytes;
};

// G planes per workgroup, `threads / 64` waves each owning rpw = floor(64 / Wf) flow rows at a time; planes + one tile per
// wave within 78 KB of LDS so that TWO workgroups share a CU (one streams while the other stages its planes: a lone
// workgroup per CU measured 10 % slower at equal wave count).  Wf > 64: not this kernel.
// Measured at (32,128,64,44) k=5 (profiles/r4_be_fwd_sweep.jsonl): 8 waves x G = 2 (2 workgroups, 16 waves per CU) 242 us,
// 4 waves x G = 4 245 us, 8 waves x G = 3 270 us -- 43 channel groups x 32 samples = 2.7 rounds of the 512 workgroup
// slots, i.e. a third round that is two-thirds empty.  So G is the value that minimises rounds(G) * G.
// tuning: key 4 = G, key 24 = threads, key 10 = LDS budget (KB).
inline WrowGeo wrow_geometry(int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, int K, int acc_bytes,
                             int elem_bytes) {
  WrowGeo g{0, 0, 0, 0, 0, 0, 0};
  if (Wf > 64 || Wf < 1) return g;
  const int64_t budget = tuning(10) >= 16 ? lds_budget() : 78 * 1024;
  const int64_t plane_bytes = Hs * (Ws + 2 * K) * acc_bytes;
  const int64_t rpw = 64 / Wf;
  const int64_t tile_stride = (rpw * K * K * Wf * elem_bytes + 16 + 15) & ~(int64_t)15;   // + room for the 16-byte phase
  const int64_t nrg = ceil_div(Hf, rpw);                     // row groups per plane
  int64_t waves = 0, G = 0;
  const int64_t slots = 2 * kNumCU;
  for (int64_t w : {(int64_t)8, (int64_t)4, (int64_t)2, (int64_t)1}) {
    int64_t cand = tuning(24) >= 64 ? tuning(24) / 64 : w;
    if (cand > 16) cand = 16;
    if (cand > nrg) cand = nrg;
    if (cand < 1) cand = 1;
    if (plane_bytes + cand * tile_stride > budget) {
      if (tuning(24) >= 64) return g;   // the forced wave count does not fit
      continue;
    }
    int64_t gmax = (budget - cand * tile_stride) / plane_bytes;
    if (gmax > C) gmax = C;
    if (gmax > 8) gmax = 8;
    int64_t best = gmax, best_cost = -1;
    for (int64_t c = gmax; c >= 1; --c) {
      const int64_t cost = ceil_div(B * ceil_div(C, c), slots) * c;
      if (best_cost < 0 || cost < best_cost) {
        best = c;
        best_cost = cost;
      }
    }
    waves = cand;
    G = best;
    // fewer waves only buy a larger G: worth it when the per-pixel setup would otherwise be paid per channel
    if (G >= 2 || w == 1) break;
  }
  if (waves < 1 || G < 1) return g;
  if (tuning(4) > 0 && tuning(4) < G) G = tuning(4);
  g.G = (int)G;
  g.ngroups = (int)ceil_div(C, G);
  g.rpw = (int)rpw;
  g.threads = (int)waves * 64;
  g.tile_off = (int)((G * plane_bytes + 15) & ~(int64_t)15);
  g.tile_stride = (int)tile_stride;
  g.lds_bytes = (unsigned)(g.tile_off + waves * tile_stride);
  return g;
}

template <typename T, int K>
static int launch_fwd_wrow(const T *src, const T *flow, T *out, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf,
                           int64_t Wf, hipStream_t stream, bool *done) {
  using A = typename Num<T>::acc;
  *done = false;
  const WrowGeo g = wrow_geometry(B, C, Hs, Ws, Hf, Wf, K, (int)sizeof(A), (int)sizeof(T));
  if (g.G <= 0) return GFLA_OK;
  const int64_t blocks = B * g.ngroups;
  if (blocks > 0x7fffffffLL || (int64_t)K * K * Hf * Wf > 0x7fffffffLL) return GFLA_OK;
#ifdef GFLA_PROBES   // timing ablations (results are garbage): key 27 bit 0 = no patch reads, bit 1 = no output stores
#define GFLA_WROW_ABL(N_)                                                                                                 \
  launch_lds(be_fwd_wrow_kernel<T, K, N_>, dim3((unsigned)blocks), dim3((unsigned)g.threads), g.lds_bytes, stream, src,    \
             flow, out, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, g.G, g.ngroups, g.rpw, g.tile_off, g.tile_stride)
  if (tuning(27) == 1) GFLA_WROW_ABL(1);
  else if (tuning(27) == 2) GFLA_WROW_ABL(2);
  else if (tuning(27) == 3) GFLA_WROW_ABL(3);
  else
#undef GFLA_WROW_ABL
#endif
  launch_lds(be_fwd_wrow_kernel<T, K>, dim3((unsigned)blocks), dim3((unsigned)g.threads), g.lds_bytes, stream, src, flow,
             out, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, g.G, g.ngroups, g.rpw, g.tile_off, g.tile_stride);
  *done = true;
  return launch_status();
}

}  // namespace gfla
