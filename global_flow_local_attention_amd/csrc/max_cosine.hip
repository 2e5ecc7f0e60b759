// Best-match cosine similarity for the sampling-correctness loss, gfx950.
//
// Reference: PerceptualCorrectness.calculate_loss, external_function.py:255-268 --
//   source_norm = source / (|source|_c + eps)        [b, Ns, C]
//   target_norm = target / (|target|_c + eps)        [b, C, Nt]
//   correction  = bmm(source_norm, target_norm)      [b, Ns, Nt]   (4096^2 floats per sample at 64x64)
//   correction_max, max_indices = max(correction, dim=1)
// The reference materialises `correction` (2.1 GB at B=32, 64x64) only to reduce it.  Here it never
// leaves the accumulator registers: one workgroup owns 128 target positions of one sample, streams
// all source positions past them in 128-row tiles (f32 MFMA, v_mfma_f32_32x32x2_f32: exact f32,
// 157 TF/s peak) and keeps a running (max, argmax) per lane.  In the MFMA C/D layout a lane holds
// 16 rows of ONE column, so the max over source rows is a register reduction; lanes l / l^32 and the
// two waves stacked along the rows are merged once at the very end.  The norms are applied to the
// accumulators (dot * 1/(|s|+eps), then * 1/(|t|+eps) on the maxima; both factors are positive so
// the max commutes with them), which saves normalised copies of both feature maps.
//
// Work decomposition: a unit = (sample, 128 target columns, a RANGE of source tiles).  The source range is
// split until there are several units per workgroup slot (2 per CU), otherwise B x ceil(Nt/128) units
// quantise badly onto 512 slots (704 units = two rounds, the second 37 % full).  Units merge through one
// 64-bit atomic max per column on (ordered float bits << 32 | ~index), decoded by a small final kernel.
//
// This is the one GEMM-shaped op on the path, hence the one place MFMA is used in this library.
#include "gfla_common.h"

namespace gfla {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTM = 128;  // source positions per tile (rows)
constexpr int kTN = 128;  // target positions per unit (columns)

// (value, index) -> one orderable 64-bit key; ties prefer the lower index.  Any real key is > 0.
__device__ __forceinline__ unsigned long long pack_best(float v, int idx) {
  uint32_t u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (uint32_t)(0x7fffffff - idx);
}
__device__ __forceinline__ float unpack_best(unsigned long long key, int *idx) {
  uint32_t u = (uint32_t)(key >> 32);
  u = (u & 0x80000000u) ? (u ^ 0x80000000u) : ~u;
  *idx = 0x7fffffff - (int)(uint32_t)key;
  return __uint_as_float(u);
}

// out[b, n] = 1 / (sqrt(sum_c x[b, c, n]^2) + eps): lanes along n (coalesced), 4 channel slices.
// keys != NULL: also reset the packed maxima of these positions.
__global__ __launch_bounds__(256) void inv_norm_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                      unsigned long long *__restrict__ keys, int C, int N,
                                                      float eps) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const int64_t b = blockIdx.y;
  float s = 0.f;
  if (n < N) {
    const float *p = x + b * C * (int64_t)N + n;
    for (int c = slice; c < C; c += 4) {
      const float v = p[(int64_t)c * N];
      s = fmaf(v, v, s);
    }
  }
  part[slice][lane] = s;
  __syncthreads();
  if (slice == 0 && n < N) {
    s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    out[b * N + n] = 1.f / (sqrtf(s) + eps);
    if (keys) keys[b * N + n] = 0ull;
  }
}

__global__ __launch_bounds__(256) void max_cosine_finish_kernel(const unsigned long long *__restrict__ keys,
                                                               const float *__restrict__ rt,
                                                               float *__restrict__ out_max, int *__restrict__ out_idx,
                                                               int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int idx;
  const float v = unpack_best(keys[i], &idx);
  out_max[i] = v * rt[i];
  if (out_idx) out_idx[i] = idx;
}

// Four consecutive floats of one channel row.  FAST (rows 16-byte aligned, N % 4 == 0, C % KC == 0): one
// unconditional 16-byte load; a quad past the row end is redirected to the last quad of the row -- those
// rows/columns are masked after the MFMAs (rows -> -inf, columns never merged), so their content is free.
// Otherwise: element-wise, zero beyond the row end / channel count.
template <bool FAST>
__device__ __forceinline__ float4 load_quad(const float *__restrict__ plane, int c, int C, int N, int col) {
  if constexpr (FAST) {
    return *reinterpret_cast<const float4 *>(plane + (int64_t)c * N + min(col, N - 4));
  }
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c >= C) return r;
  const float *row = plane + (int64_t)c * N;
  if (col < N) r.x = row[col];
  if (col + 1 < N) r.y = row[col + 1];
  if (col + 2 < N) r.z = row[col + 2];
  if (col + 3 < N) r.w = row[col + 3];
  return r;
}

// KC: channels per staged chunk.  NW waves = 2 (rows) x NW/2 (columns); a wave owns 64 x (256/NW) outputs.
template <bool FAST, int KC, int NW>
__global__ __launch_bounds__(NW * 64) void max_cosine_kernel(const float *__restrict__ S, const float *__restrict__ T,
                                                        const float *__restrict__ rs,
                                                        unsigned long long *__restrict__ keys, int C, int Ns,
                                                        int Nt, int tilesN, int splitM, int total, int per_xcd) {
  constexpr int kThreads = NW * 64;
  constexpr int kQuads = KC * kTM / 4 / kThreads;  // float4 per thread, operand and chunk
  constexpr int kRowStep = kThreads / 32;          // staging rows covered by one pass of the workgroup
  constexpr int WN = 256 / NW;                     // columns per wave
  constexpr int NJ = WN / 32;                      // 32-column MFMA blocks per wave
  __shared__ float As[2][KC][kTM];
  __shared__ float Bs[2][KC][kTN];
  __shared__ float rsl[2][kTM];
  __shared__ float red_v[kTN];
  __shared__ int red_i[kTN];

  // Workgroups are dealt round-robin to the 8 XCDs: give every XCD a contiguous run of units so the
  // planes of a sample are streamed through ONE L2.  unit -> (sample, column tile, source range).
  const int v = (blockIdx.x % kNumXCD) * per_xcd + blockIdx.x / kNumXCD;
  if (v >= total) return;
  const int per_sample = tilesN * splitM;
  const int64_t b = v / per_sample;
  const int rem = v - (int)b * per_sample;
  const int n0 = (rem / splitM) * kTN;
  const int part = rem % splitM;
  const int nM_all = (Ns + kTM - 1) / kTM;
  const int mt0 = (int)((int64_t)nM_all * part / splitM);
  const int mt1 = (int)((int64_t)nM_all * (part + 1) / splitM);

  const float *Sb = S + b * C * (int64_t)Ns;
  const float *Tb = T + b * C * (int64_t)Nt;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / (NW / 2), wn = wave % (NW / 2);
  const int l31 = lane & 31, kh = lane >> 5;
  const int ld_row = t >> 5, ld_col = (t & 31) * 4;  // staging: rows ld_row + kRowStep * h; 4 floats at ld_col

  const int nK = (C + KC - 1) / KC;
  const int iters = (mt1 - mt0) * nK;

  float4 ra[kQuads], rb[kQuads];
  float rscale = 0.f;
  auto fetch = [&](int it) {
    const int mt = mt0 + it / nK, c0 = (it % nK) * KC;
#pragma unroll
    for (int h = 0; h < kQuads; ++h) {
      ra[h] = load_quad<FAST>(Sb, c0 + ld_row + kRowStep * h, C, Ns, mt * kTM + ld_col);
      rb[h] = load_quad<FAST>(Tb, c0 + ld_row + kRowStep * h, C, Nt, n0 + ld_col);
    }
    if (c0 == 0 && t < kTM) {
      const int m = mt * kTM + t;
      rscale = m < Ns ? rs[b * Ns + m] : 0.f;
    }
  };
  auto stage = [&](int it) {
    const int buf = it & 1;
#pragma unroll
    for (int h = 0; h < kQuads; ++h) {
      *reinterpret_cast<float4 *>(&As[buf][ld_row + kRowStep * h][ld_col]) = ra[h];
      *reinterpret_cast<float4 *>(&Bs[buf][ld_row + kRowStep * h][ld_col]) = rb[h];
    }
    if (it % nK == 0 && t < kTM) rsl[(it / nK) & 1][t] = rscale;
  };

  f32x16 acc[2][NJ];
  float best[NJ];
  int bidx[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    best[j] = -INFINITY;
    bidx[j] = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }

  if (iters > 0) {
    fetch(0);
    stage(0);
  }
  __syncthreads();

  for (int it = 0; it < iters; ++it) {
    if (it + 1 < iters) fetch(it + 1);

    const int buf = it & 1;
    const float *Ab = &As[buf][kh][wm * 64 + l31];
    const float *Bb = &Bs[buf][kh][wn * WN + l31];
#pragma unroll
    for (int kk = 0; kk < KC; kk += 2) {
      const float a0 = Ab[kk * kTM], a1 = Ab[kk * kTM + 32];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float bj = Bb[kk * kTN + 32 * j];
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bj, acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bj, acc[1][j], 0, 0, 0);
      }
    }

    if (it % nK == nK - 1) {
      // rows of this tile are complete: scale by 1/(|s|+eps), fold into the running column maxima.
      // C/D layout: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
      const int lt = it / nK, mt = mt0 + lt;
      const float *sc = rsl[lt & 1];
      const bool ragged = (mt + 1) * kTM > Ns;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          const int m = mt * kTM + ml;
          const float s = sc[ml];
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            float val = acc[i][j][r] * s;
            if (ragged && m >= Ns) val = -INFINITY;
            if (val > best[j]) {
              best[j] = val;
              bidx[j] = m;
            }
            acc[i][j][r] = 0.f;
          }
        }
    }

    if (it + 1 < iters) stage(it + 1);
    __syncthreads();
  }

  // merge the two row groups of a wave (lanes l, l ^ 32), then the two waves stacked along the rows,
  // then this unit into the global keys
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float ov = __shfl_xor(best[j], 32);
    const int oi = __shfl_xor(bidx[j], 32);
    if (ov > best[j] || (ov == best[j] && oi < bidx[j])) {
      best[j] = ov;
      bidx[j] = oi;
    }
  }
  if (wm == 1 && kh == 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      red_v[wn * WN + j * 32 + l31] = best[j];
      red_i[wn * WN + j * 32 + l31] = bidx[j];
    }
  }
  __syncthreads();
  if (wm == 0 && kh == 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int nl = wn * WN + j * 32 + l31;
      const float ov = red_v[nl];
      const int oi = red_i[nl];
      if (ov > best[j] || (ov == best[j] && oi < bidx[j])) {
        best[j] = ov;
        bidx[j] = oi;
      }
      const int n = n0 + nl;
      if (n < Nt && best[j] > -INFINITY) atomicMax(&keys[b * Nt + n], pack_best(best[j], bidx[j]));
    }
  }
}

static int64_t align16(int64_t v) { return (v + 15) & ~(int64_t)15; }

static int max_cosine(const float *source, const float *target, void *workspace, float *out_max, int32_t *out_idx,
                      int64_t B, int64_t C, int64_t Ns, int64_t Nt, double eps, gfla_stream_t stream_) {
  if (!source || !target || !workspace || !out_max) return GFLA_ERR_NULL_POINTER;
  if (B < 0 || C <= 0 || Ns <= 0 || Nt < 0) return GFLA_ERR_BAD_SHAPE;
  if (B == 0 || Nt == 0) return GFLA_OK;
  if (Ns > 0x7fffff00LL || Nt > 0x7fffff00LL || C > 0x7fffff00LL || B > 65535) return GFLA_ERR_UNSUPPORTED;
  const int64_t tilesN = ceil_div(Nt, kTN), nM = ceil_div(Ns, kTM);
  // several units per workgroup slot (2 per CU) so that the tail of the launch is short
  int64_t splitM = 1;
  const int64_t slots = 2 * kNumCU;
  if (tuning(5) > 0)
    splitM = tuning(5);
  else
    splitM = ceil_div(8 * slots, B * tilesN);  // workgroups start as slots free up: many short units pack best
  if (splitM > nM) splitM = nM;
  if (splitM < 1) splitM = 1;
  const int64_t total = B * tilesN * splitM;
  if (total > 0x7ffffff0LL) return GFLA_ERR_UNSUPPORTED;
  hipStream_t stream = static_cast<hipStream_t>(stream_);

  char *ws = static_cast<char *>(workspace);
  unsigned long long *keys = reinterpret_cast<unsigned long long *>(ws);
  float *inv_t = reinterpret_cast<float *>(ws + align16(8 * B * Nt));
  float *inv_s = reinterpret_cast<float *>(ws + align16(8 * B * Nt) + align16(4 * B * Nt));

  inv_norm_kernel<<<dim3((unsigned)ceil_div(Ns, 64), (unsigned)B), 256, 0, stream>>>(source, inv_s, nullptr, (int)C,
                                                                                    (int)Ns, (float)eps);
  inv_norm_kernel<<<dim3((unsigned)ceil_div(Nt, 64), (unsigned)B), 256, 0, stream>>>(target, inv_t, keys, (int)C,
                                                                                    (int)Nt, (float)eps);
  const int64_t per_xcd = ceil_div(total, kNumXCD);
  const dim3 grid((unsigned)(per_xcd * kNumXCD));
  const bool fast = C % 32 == 0 && Ns % 4 == 0 && Nt % 4 == 0 && Ns >= 4 && Nt >= 4 &&
                    ((reinterpret_cast<uintptr_t>(source) | reinterpret_cast<uintptr_t>(target)) & 15) == 0;
#define GFLA_MC_LAUNCH(FAST_, KC_, NW_)                                                                       \
  max_cosine_kernel<FAST_, KC_, NW_><<<grid, NW_ * 64, 0, stream>>>(source, target, inv_s, keys, (int)C, (int)Ns, \
                                                                    (int)Nt, (int)tilesN, (int)splitM, (int)total, \
                                                                    (int)per_xcd)
  // Measured on MI355X (profiles/r1_max_cosine_variants.txt), B=32 C=256 N=64x64: 4 waves x KC 16: 99 TF/s,
  // 4 x 32: 110, 8 x 16: 115, 8 x 32: 121 (8 waves = 64x32 outputs per wave, 4 waves per SIMD resident).
  if (fast)
    GFLA_MC_LAUNCH(true, 32, 8);
  else
    GFLA_MC_LAUNCH(false, 32, 8);
#undef GFLA_MC_LAUNCH
  max_cosine_finish_kernel<<<dim3((unsigned)ceil_div(B * Nt, 256)), 256, 0, stream>>>(keys, inv_t, out_max, out_idx,
                                                                                     B * Nt);
  return launch_status();
}

}  // namespace gfla

extern "C" {
int64_t gfla_max_cosine_workspace_bytes(int64_t B, int64_t Ns, int64_t Nt) {
  if (B < 0 || Ns < 0 || Nt < 0) return 0;
  return gfla::align16(8 * B * Nt) + gfla::align16(4 * B * Nt) + gfla::align16(4 * B * Ns);
}

int gfla_max_cosine_fwd_f32(const float *source, const float *target, void *workspace, float *out_max,
                            int32_t *out_idx, int64_t B, int64_t C, int64_t Ns, int64_t Nt, double eps,
                            gfla_stream_t stream) {
  return gfla::max_cosine(source, target, workspace, out_max, out_idx, B, C, Ns, Nt, eps, stream);
}
}
