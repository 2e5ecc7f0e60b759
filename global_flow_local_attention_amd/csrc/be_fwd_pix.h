// block_extractor forward in the REFERENCE layout (B, C, K*Hf, K*Wf), lane = flow pixel, direct stores (round 4).
//
// What bounded round 1's planes-in-LDS kernel (be_fwd_lds_kernel, lane = 4 consecutive output x): its per-lane setup --
// four flow lookups, four floor/clamp/weight sets, integer divisions by a run-time k -- is paid per output QUAD and per
// channel group, ~120 vector instructions against the ~40 of the four channels it is amortised over, and every output
// costs four ds_read_b32 (block_extractor_kernel.cu:78-84 read tap by tap): 4.2-4.5 TB/s whatever the map width.
//
// Here the flow pixel owns the lane:
//   * setup once per pixel and chunk of CH channels: one flow pair, K floor/fraction pairs, the patch origin;
//   * the planes sit in LDS REPLICATE-PADDED by K columns on either side, so a dense patch row is K+1 consecutive words
//     at (row base + immediate offset): no per-tap address arithmetic and no column clamp in the channel loop;
//   * the four weights of an output (xL_P*yT_P, ... -- the reference's products, block_extractor_kernel.cu:73-84, same
//     order of accumulation, fused multiply-adds) are formed once per (i, j) and used for the CH channels of the chunk;
//   * a lane's K outputs of an output row are K consecutive floats: ONE 16-byte + one 4-byte store (k = 5), 8 + 4
//     (k = 3), at a K*4-byte lane stride.
// Measured (profiles/r4_be_fwd_pix_ablations.jsonl, r4_be_fwd_widths*.jsonl): 80 us of arithmetic at (32,128,64,44) k=5
// (stores compiled out) but 340 us with them, with or without the LDS reads: bound by the write stream, which runs at
// 5.1-5.2 TB/s when an output row is a whole number of 128-byte lines (Wf = 32, 64, 96) and 3.5-4.1 TB/s otherwise.
// So this kernel serves flow rows WIDER than 64 pixels; up to 64 the wave-per-flow-row kernel (be_fwd_wrow.h) takes
// over.  Tried on top and dropped: non-temporal stores (they skip the L2's merging: half the rate), the K outputs
// transposed through a per-wave LDS row so that the lanes of a store are consecutive (two LDS round trips per output
// row: 3.2 TB/s, and a store-only micro-benchmark shows the pattern itself is no faster, tools/ubench/store_patterns.hip).
// A pixel whose taps are not a dense patch (a coordinate within rounding of an integer) is evaluated tap by tap, as the
// reference does, on the same padded planes.
#pragma once

#include "lds_plane.h"

namespace gfla {

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T, int N>
struct RowVec {
  typedef T type __attribute__((ext_vector_type(N), aligned(sizeof(T))));
};

// K consecutive elements at p (element-aligned only): as few store instructions as the ISA allows
template <typename T, int K, bool NT>
__device__ __forceinline__ void store_row(T *p, const T (&o)[K]) {
  int j = 0;
  if constexpr (sizeof(T) == 4 || sizeof(T) == 8) {
    constexpr int W4 = 16 / sizeof(T);  // elements of a 16-byte store
    using V4 = typename RowVec<T, W4>::type;
#pragma unroll
    for (; j + W4 <= K; j += W4) {
      V4 v;
#pragma unroll
      for (int e = 0; e < W4; ++e) v[e] = o[j + e];
      if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<V4 *>(p + j));
      else *reinterpret_cast<V4 *>(p + j) = v;
    }
    if constexpr (sizeof(T) == 4) {
      if (j + 2 <= K) {
        using V2 = typename RowVec<T, 2>::type;
        V2 v;
        v[0] = o[j];
        v[1] = o[j + 1];
        if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<V2 *>(p + j));
        else *reinterpret_cast<V2 *>(p + j) = v;
        j += 2;
      }
    }
  }
#pragma unroll
  for (; j < K; ++j) {
    if constexpr (NT && sizeof(T) >= 4) __builtin_nontemporal_store(o[j], p + j);
    else p[j] = o[j];
  }
}

template <typename T, int K, int CH, bool NT, int ABL = 0>
__global__ __launch_bounds__(1024) void be_fwd_pix_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, T *__restrict__ out, int C, int Hs, int Ws, int Hf, int Wf,
    int G, int ngroups, int split) {
  using A = typename Num<T>::acc;
  constexpr int PAD = K;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  A *planes = reinterpret_cast<A *>(gfla_smem);
  int bid = blockIdx.x;
  const int sp = bid % split;
  bid /= split;
  const int g = bid % ngroups;
  const int b = bid / ngroups;
  const int c0 = g * G;
  const int gc = min(G, C - c0);
  const int Wp = Ws + 2 * PAD;
  const int plane_p = Hs * Wp;
  // the wave index as a SCALAR: everything derived from it (channel chunk, pixel block, plane and output bases) then
  // lives in SGPRs and the stores take the scalar-base form
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;

  // ---- stage the gc planes, interior first (coalesced), then the replicated columns -------------------------------
  {
    const T *gsrc = src + ((int64_t)b * C + c0) * ((int64_t)Hs * Ws);
    const int n = gc * Hs * Ws;
    if constexpr (sizeof(T) == 4) {
      if ((Ws & 3) == 0 && (reinterpret_cast<uintptr_t>(gsrc) & 15) == 0) {
        const int W4 = Ws >> 2, n4 = n >> 2;
        const float4 *g4 = reinterpret_cast<const float4 *>(gsrc);
#pragma unroll 4
        for (int i = threadIdx.x; i < n4; i += blockDim.x) {
          const float4 v = g4[i];
          const int r = i / W4, x = (i - r * W4) << 2;
          A *d = planes + r * Wp + PAD + x;
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
      } else {
#pragma unroll 4
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
          const int r = i / Ws, x = i - r * Ws;
          planes[r * Wp + PAD + x] = Num<T>::ld(gsrc + i);
        }
      }
    } else {
#pragma unroll 4
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int r = i / Ws, x = i - r * Ws;
        planes[r * Wp + PAD + x] = Num<T>::ld(gsrc + i);
      }
    }
    __syncthreads();
    const int rows = gc * Hs;
    for (int i = threadIdx.x; i < rows * 2 * PAD; i += blockDim.x) {
      const int r = i / (2 * PAD), q = i - r * (2 * PAD);
      A *row = planes + r * Wp;
      if (q < PAD) row[q] = row[PAD];
      else row[Ws + q] = row[PAD + Ws - 1];   // q - PAD + PAD + Ws
    }
    __syncthreads();
  }

  const int HW = Hf * Wf;
  const int Wo = K * Wf;
  const int64_t oplane = (int64_t)(K * Hf) * Wo;
  const int per = (HW + split - 1) / split;
  const int p0 = sp * per, p1 = min(HW, p0 + per);
  if (p0 >= p1) return;
  const int nblk = (p1 - p0 + 63) >> 6;
  const int nch = (gc + CH - 1) / CH;
  const T *flow_x = flow + (int64_t)(b * 2 + 0) * HW;
  const T *flow_y = flow + (int64_t)(b * 2 + 1) * HW;

  for (int it = wave; it < nblk * nch; it += nwaves) {
    const int ch = it / nblk, blk = it - ch * nblk;
    const int pl = p0 + (blk << 6) + lane;
    const bool active = pl < p1;
    const int p = active ? pl : p1 - 1;
    const int yf = p / Wf, xf = p - yf * Wf;
    const A fx0 = Num<T>::ld(flow_x + p);
    const A fy0 = Num<T>::ld(flow_y + p);
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
      dense &= ((int)fdx == x0 + t) & ((int)fdy == y0 + t);   // (no short circuit: a branch per tap otherwise)
      ax[t] = dx - fdx;
      ay[t] = dy - fdy;
    }
    const int cbase = ch * CH;
    const int ncc = min(CH, gc - cbase);
    const A *plc = planes + cbase * plane_p;
    T *oplane0 = out + ((int64_t)b * C + c0 + cbase) * oplane;   // output plane of the chunk's first channel
    const int ooff = (K * yf) * Wo + K * xf;                     // this lane's pixel: first output of its first row
    if (dense) {
      // padded column of tap 0 and the row bases (clamped rows; replicated columns are in the plane)
      const int x0c = clampi(x0, -PAD, Ws - 1) + PAD;
      const int y0c = clampi(y0, -(K + 1), Hs);
      // the bilinear form separated (be_fwd_wrow.h has the derivation and the operation count): patch rows interpolated
      // along x once, output row i = the blend of interpolated rows i and i + 1; identical expressions in both kernels
      auto hrow = [&](int cc, int r, A (&h)[K]) {
        const A *pc = plc + min(cc, ncc - 1) * plane_p + clampi(y0c + r, 0, Hs - 1) * Wp + x0c;
        A v[K + 1];
#pragma unroll
        for (int q = 0; q <= K; ++q) v[q] = (ABL & 1) ? (A)(lane + q + r + cc) : pc[q];
#pragma unroll
        for (int j = 0; j < K; ++j) h[j] = fma_t(ax[j], v[j + 1], (1 - ax[j]) * v[j]);
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
          T *oc = oplane0 + cc * oplane + (int64_t)i * Wo;
          if constexpr (ABL & 2) {   // timing ablation: no stores (one impossible store keeps the arithmetic alive)
            A sum = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) sum += (A)o[j];
            if (sum == (A)12345.678) oc[ooff] = Num<T>::from(sum);
          } else {
            if (active && cc < ncc) store_row<T, K, NT>(oc + ooff, o);
          }
#pragma unroll
          for (int j = 0; j < K; ++j) hA[cc][j] = hB[j];
        }
      }
    }
    if (!dense && active) {
      int xL[K], xR[K];
#pragma unroll
      for (int t = 0; t < K; ++t) {
        const A dx = (fx0 + (A)(t - K / 2)) + (A)xf;
        const A fdx = floor_t<A>(dx);
        xL[t] = clampi((int)fdx, 0, Ws - 1) + PAD;  // :69-72
        xR[t] = clampi((int)(fdx + 1), 0, Ws - 1) + PAD;
      }
      for (int cc = 0; cc < ncc; ++cc) {
        const A *pc = plc + cc * plane_p;
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
          const A dy = (fy0 + (A)(i - K / 2)) + (A)yf;
          const A fdy = floor_t<A>(dy);
          const int yT = clampi((int)fdy, 0, Hs - 1) * Wp, yB = clampi((int)(fdy + 1), 0, Hs - 1) * Wp;
          const A yB_P = dy - fdy, yT_P = 1 - yB_P;
          T o[K];
#pragma unroll
          for (int j = 0; j < K; ++j) {
            const A xR_P = ax[j], xL_P = 1 - xR_P;
            A s = (xL_P * yT_P) * pc[yT + xL[j]];
            s = fma_t(xR_P * yT_P, pc[yT + xR[j]], s);
            s = fma_t(xL_P * yB_P, pc[yB + xL[j]], s);
            s = fma_t(xR_P * yB_P, pc[yB + xR[j]], s);
            o[j] = Num<T>::from(s);
          }
          store_row<T, K, NT>(oplane0 + cc * oplane + (int64_t)i * Wo + ooff, o);
        }
      }
    }
  }
}

struct PixGeo {
  int G, ngroups, split, threads;
  unsigned lds_bytes;
};

// G channel planes (padded) per workgroup, `split` workgroups per (b, group), `threads` per workgroup.
// tuning: key 4 caps G, key 5 sets split, key 24 the threads.
template <int CH>
inline PixGeo pix_geometry(int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, int K, int acc_bytes) {
  PixGeo g{0, 0, 1, 512, 0};
  const int64_t plane_bytes = Hs * (Ws + 2 * K) * acc_bytes;
  const int64_t budget = kLdsBudget;
  if (plane_bytes > budget) return g;
  // one chunk of CH channels per workgroup unless the planes are tiny: the per-pixel setup is amortised over a chunk,
  // and small workgroups start streaming sooner (the staging of the first round is the kernel's serial prologue)
  int64_t G = budget / plane_bytes;
  if (G > C) G = C;
  if (G >= CH) G = (G / CH) * CH;
  const int64_t HW = Hf * Wf, nblk = ceil_div(HW, 64);
  if (tuning(4) > 0) G = tuning(4) < G ? tuning(4) : G;
  else if (G > CH && nblk >= 8) G = CH;
  g.G = (int)G;
  g.ngroups = (int)ceil_div(C, G);
  // waves: one pixel block of 64 per wave and pass.  Eight waves unless there are fewer items: four workgroups then share
  // a CU, and one stages its planes while the others stream -- measured at (32,256,32,22) k=3, 11 pixel blocks: 8 waves
  // 40.6 us, 11 waves (one block each) 46.4, 16 waves 46.3 (profiles/r4_be_fwd_pix_threads.txt)
  const int64_t items = nblk * ceil_div(G, CH);
  int waves = items < 8 ? (int)items : 8;
  if (tuning(24) >= 64) waves = tuning(24) / 64;
  if (waves < 1) waves = 1;
  if (waves > 16) waves = 16;
  g.threads = waves * 64;
  g.split = tuning(5) > 0 ? tuning(5) : 1;
  g.lds_bytes = (unsigned)(G * plane_bytes);
  return g;
}

template <typename T, int K>
static int launch_fwd_pix(const T *src, const T *flow, T *out, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf,
                          int64_t Wf, hipStream_t stream, bool *done) {
  using A = typename Num<T>::acc;
  constexpr int CH = sizeof(A) == 8 ? (K >= 4 ? 1 : 2) : 4;  // register budget: CH x (K+1) patch values + K x 4 weights
  *done = false;
  const PixGeo g = pix_geometry<CH>(B, C, Hs, Ws, Hf, Wf, K, (int)sizeof(A));
  if (g.G <= 0) return GFLA_OK;
  const int64_t blocks = B * g.ngroups * g.split;
  if (blocks > 0x7fffffffLL) return GFLA_OK;
  const dim3 grid((unsigned)blocks), block((unsigned)g.threads);
#define GFLA_PIX_LAUNCH(NT_, ABL_)                                                                                      \
  launch_lds(be_fwd_pix_kernel<T, K, CH, NT_, ABL_>, grid, block, g.lds_bytes, stream, src, flow, out, (int)C, (int)Hs,   \
             (int)Ws, (int)Hf, (int)Wf, g.G, g.ngroups, g.split)
  if (tuning(25) == 1) GFLA_PIX_LAUNCH(true, 0);   // non-temporal stores: measured at half the rate, kept for A/B
#ifdef GFLA_PROBES   // timing ablations (results are garbage): key 27 bit 0 = no patch reads, bit 1 = no stores
  else if (tuning(27) == 1) GFLA_PIX_LAUNCH(false, 1);
  else if (tuning(27) == 2) GFLA_PIX_LAUNCH(false, 2);
  else if (tuning(27) == 3) GFLA_PIX_LAUNCH(false, 3);
#endif
  else GFLA_PIX_LAUNCH(false, 0);
#undef GFLA_PIX_LAUNCH
  *done = true;
  return launch_status();
}

}  // namespace gfla
