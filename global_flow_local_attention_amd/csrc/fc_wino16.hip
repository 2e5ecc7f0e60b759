// Winograd-domain convolutions of ExtractorAttn's first FC layer with TWO-TERM f16 OPERANDS on the f16 matrix cores
// (arithmetic mode 5), gfx950.
//
// Same formulation, tiling and staging as fc_wino.hip (F(2x2,5x5) / F(4x4,3x3) on the points {0, 1, -1, 2, -1/2, inf}; reference
// base_function.py:799-807): the transforms B^T d B and A^T M A stay float32 on the vector ALUs.  What changes is the 36
// point-wise GEMMs over the channels.  fc_wino.hip runs them on v_mfma_f32_16x16x4_f32, which executes at the f32 VECTOR rate
// (157 TFLOP/s) and was 300 of that kernel's 432 us.  Here every transformed input value v and every transformed weight u is
// split into two f16 terms, v * s = hi + lo with hi = RN16(v s), lo = RN16(v s - hi) (s a power of two from the tensor's
// max |x|, so nothing over- or underflows: |v s - hi - lo| <= 2^-24 |v s|, the rounding of a float32 itself), and the product
// is formed EXACTLY from all four cross terms with f32 accumulation inside the MFMA:
//     sum_c v_c u_c = sum_c (vhi_c uhi_c + vlo_c ulo_c)  +  sum_c (vhi_c ulo_c + vlo_c uhi_c)
// With the K slots of v_mfma_f32_32x32x16_f16 filled as (hi_c, lo_c) pairs on the A side, the first sum is the MFMA against
// the weights' own (hi, lo) words and the second the MFMA against the same words with their halves swapped (one v_alignbit
// per dword): two MFMAs of 32 cycles per (32 tiles x 32 channels x 8 input channels x point) instead of eight of 32 cycles --
// the matrix-core time drops 4x and the kernel becomes bound by its LDS / vector work.  Measured error against float64 at the
// bench shapes: the same as the float32 Winograd kernels' (tools/experiments/winograd_numerics.py f16: 3.0e-6 vs 2.7e-6 of
// the largest output at k = 5, 4.7e-6 vs 4.8e-6 at k = 3 -- the f32 accumulation dominates both).
//
//   workgroup = 32 tiles x 64 output channels x 36 points, 8 waves; wave w: channel block w & 1 (32 channels), points
//   9 (w >> 1) .. +8 -> 9 accumulators of 32x32 (144 registers, two waves per SIMD);
//   per 8-channel step: A = V[point][channel quad g][tile][4 x (hi, lo)] from LDS, ONE ds_read_b128 per point (lanes 0-31 /
//   32-63 = quads 0 / 1 = the two K halves of the MFMA); B = the same layout of U from global memory, one 16-byte load per
//   point, a step ahead; the transform of the NEXT step (fc_wino.hip's, plus the split) runs in the other wave of the SIMD;
//   epilogue: every wave reduces its nine points to an m x m partial per (tile, channel); the four waves of a channel block
//   exchange partials through LDS, a quarter of the tiles each, and store 128-byte rows of the (pixel, channel) map.
#include "fc_wino_shared.h"

namespace gfla {

// ---- weights: conv0.weight (128, 2C, k, k) -> U = G w G^T * scale as (hi, lo) words in B-fragment order --------------
// U16[ntile][step = ci >> 3][point][g = (ci >> 2) & 1][n = co & 63][ci & 3]: a lane of channel block nb (n = 32 nb + lane & 31,
// g = lane >> 5) loads its 16 bytes of a (step, point) with one request; a wave's request is two runs of 512 bytes.
template <int KS>
__global__ __launch_bounds__(256) void fc_wino16_pack_w_kernel(const float *__restrict__ w0, WnPackJobs jobs, int C,
                                                               const uint32_t *__restrict__ amax_w) {
  const WnPackJob jb = jobs.j[blockIdx.y];
  const int nch = (jb.n_in + kFcChunk - 1) / kFcChunk, ntn = (jb.n_out + kWnN - 1) / kWnN;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (in channel, out channel), out fastest
  if (!jb.U || idx >= (int64_t)ntn * kWnN * nch * kFcChunk) return;
  const float su = wn16_pow2(wn16_scale_exp(*amax_w, kWn16HeadW));
  const int co = (int)(idx % (ntn * kWnN)), ci = (int)(idx / (ntn * kWnN));
  float w[KS][KS];
#pragma unroll
  for (int i = 0; i < KS; ++i)
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      float v = 0.f;
      if (ci < jb.n_in && co < jb.n_out) {
        v = jb.dgrad ? w0[(((int64_t)ci * 2 * C + jb.c_off + co) * KS + (KS - 1 - i)) * KS + (KS - 1 - j)]
                     : w0[(((int64_t)co * 2 * C + jb.c_off + ci) * KS + i) * KS + j];
      }
      w[i][j] = v * su;
    }
  float t[6][KS];  // G w: columns first
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    float col[KS], o[6];
#pragma unroll
    for (int i = 0; i < KS; ++i) col[i] = w[i][j];
    wn_g<KS>(col, o);
#pragma unroll
    for (int a = 0; a < 6; ++a) t[a][j] = o[a];
  }
  const int ntile = co / kWnN, n = co % kWnN;
  const int step = ci >> 3, g = (ci >> 2) & 1, c4 = ci & 3;
  uint32_t *dst = reinterpret_cast<uint32_t *>(jb.U) + ((((int64_t)ntile * 2 * nch + step) * kWnXi * 2 + g) * kWnN + n) * 4 + c4;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    float o[6];
    wn_g<KS>(t[a], o);
#pragma unroll
    for (int e = 0; e < 6; ++e) dst[(int64_t)(a * 6 + e) * 2 * kWnN * 4] = wn16_split(o[e]);
  }
}

// the four weight sets of one layer (same slots and sizes as fc_wino_pack_weights: fc_wino_wpack_bytes)
int fc_wino16_pack_weights(const float *w0, const uint32_t *amax_w, float *u_ft, float *u_fs, float *u_dt, float *u_ds, int C,
                           int k, hipStream_t stream) {
  WnPackJobs jobs;
  jobs.j[0] = WnPackJob{u_ft, 0, 0, C, kFcHidden};
  jobs.j[1] = WnPackJob{u_fs, C, 0, C, kFcHidden};
  jobs.j[2] = WnPackJob{u_dt, 0, 1, kFcHidden, C};
  jobs.j[3] = WnPackJob{u_ds, C, 1, kFcHidden, C};
  int64_t most = 0;
  for (int q = 0; q < 4; ++q) {
    const int64_t n = ceil_div(jobs.j[q].n_out, kWnN) * kWnN * ceil_div(jobs.j[q].n_in, kFcChunk) * kFcChunk;
    if (jobs.j[q].U && n > most) most = n;
  }
  if (most == 0) return GFLA_OK;
  const dim3 grid((unsigned)ceil_div(most, 256), 4);
  if (k == 5)
    fc_wino16_pack_w_kernel<5><<<grid, 256, 0, stream>>>(w0, jobs, C, amax_w);
  else if (k == 3)
    fc_wino16_pack_w_kernel<3><<<grid, 256, 0, stream>>>(w0, jobs, C, amax_w);
  else
    return GFLA_ERR_UNSUPPORTED;
  return launch_status();
}

// ---- the convolution ---------------------------------------------------------------------------------------------
struct Wn16KArgs {
  PackedDesc X;
  const uint32_t *U;
  const uint32_t *amax_x;
  float *out;
  int64_t out_bs;
  int ldo, n_valid, Ho, Wv, Wp;
  WnGeo geo;
  int ntn;
  int64_t total_groups, S;
};

// A^T (m x 6) as a table (the epilogue folds it at compile time)
template <int M>
struct WnAT;
template <>
struct WnAT<2> {
  static constexpr float v[2][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 2.f, -0.5f, 1.f}};
};
template <>
struct WnAT<4> {
  static constexpr float v[4][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f},
                                    {0.f, 1.f, -1.f, 2.f, -0.5f, 0.f},
                                    {0.f, 1.f, 1.f, 4.f, 0.25f, 0.f},
                                    {0.f, 1.f, -1.f, 8.f, -0.125f, 1.f}};
};

template <int PG_>
struct PgTag { static constexpr int value = PG_; };

// DBG (timing ablations, `make PROBES=1` builds only, tuning key 20; results are garbage): 1 no transform, 2 no MFMAs / A reads,
// 4 no B reloads, 8 no raw staging, 32 transform without the f16 split (hi only), 64 no epilogue
extern unsigned long long *g_wino_stamps;   // fc_wino.hip (DBG & 16: per-wave phase times, s_memtime; tools/probe_wino_phases.py)
template <int KS, bool DB = true, int DBG = 0>
__global__ __launch_bounds__(kWnThreads, 2) void fc_wino16_conv_kernel(Wn16KArgs a0, Wn16KArgs a1, unsigned n0, int nch,
                                                                      const uint32_t *__restrict__ amax_w,
                                                                      unsigned long long *stamps) {
  unsigned long long tk0 = 0, t_first = 0, t_second = 0, t_bar = 0, t_pro = 0, t_epi = 0;
  if constexpr (DBG & 16) tk0 = __builtin_amdgcn_s_memtime();
  auto stamp = [&](unsigned long long &slot) {
    if constexpr (DBG & 16) {
      __builtin_amdgcn_s_waitcnt(0);
      const unsigned long long n = __builtin_amdgcn_s_memtime();
      slot += n - tk0;
      tk0 = n;
    }
  };
  constexpr int M = Wn<KS>::M, PITCH = Wn<KS>::PITCH, NP = 9;   // points per wave
  const bool second = blockIdx.x >= n0;
#define GFLA_PICK(f) (second ? a1.f : a0.f)
  PackedDesc X;
  X.base = GFLA_PICK(X.base), X.split_stride = 0, X.batch_stride = GFLA_PICK(X.batch_stride);
  X.chunk_stride = GFLA_PICK(X.chunk_stride), X.pix_stride = GFLA_PICK(X.pix_stride);
  const uint32_t *__restrict__ U = GFLA_PICK(U);
  const uint32_t *__restrict__ amax_x = GFLA_PICK(amax_x);
  float *__restrict__ out = GFLA_PICK(out);
  const int64_t out_bs = GFLA_PICK(out_bs), total_groups = GFLA_PICK(total_groups), S = GFLA_PICK(S);
  const int ldo = GFLA_PICK(ldo), n_valid = GFLA_PICK(n_valid), Ho = GFLA_PICK(Ho), Wv = GFLA_PICK(Wv), Wp = GFLA_PICK(Wp);
  const int ntn = GFLA_PICK(ntn);
  WnGeo geo;
  geo.TH = GFLA_PICK(geo.TH), geo.TW = GFLA_PICK(geo.TW), geo.ngroups = GFLA_PICK(geo.ngroups), geo.span = GFLA_PICK(geo.span);
  geo.tpg = GFLA_PICK(geo.tpg);
#undef GFLA_PICK
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  uint32_t *vbuf = reinterpret_cast<uint32_t *>(gfla_smem);   // [2][36 points][2 quads][32 tiles][4 channels] (hi, lo) words
  unsigned char *raw = gfla_smem + 2 * kWnVFloats * 4;        // [1 or 2][span][PITCH] float32, scaled
  const int raw_bytes = (geo.span * PITCH + 15) & ~15;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nblk = wave & 1, pg = (wave >> 1) & 3, xh = wave >> 2;
  const int g = lane >> 5, l31 = lane & 31;
  // scales: input (this job's tensor), weights; the epilogue multiplies by their inverses (two exact power-of-two factors)
  const int ex = wn16_scale_exp(*amax_x, kWn16HeadX), ew = wn16_scale_exp(*amax_w, kWn16HeadW);
  const float sx = wn16_pow2(ex), inv_x = wn16_pow2(254 - ex), inv_w = wn16_pow2(254 - ew);
  // workgroup -> (group of tiles, output-channel tile): ids x and x + 8 run on the same XCD (fc_wino.hip)
  const int64_t x = blockIdx.x - (second ? n0 : 0u);
  const int xcd = (int)(x & 7);
  const int64_t slot = x >> 3;
  const int ntile = (int)(slot % ntn);
  const int64_t glin = (slot / ntn) * 8 + xcd;
  if (glin >= total_groups) return;
  const int64_t b = glin / geo.ngroups;
  const int grp = (int)(glin - b * geo.ngroups);
  const int ntiles = geo.TH * geo.TW;
  const int tile0 = grp * geo.tpg;
  const int ty_first = tile0 / geo.TW;
  const int p0 = M * ty_first * Wp;
  const int64_t avail = S - p0;

  // transform item of this thread: (tile, channel of the 8-channel step), rows 3*xh .. 3*xh + 2 of the point grid
  const int tl = (t & 255) >> 3, c8 = t & 7;
  int toff;
  {
    const int tau = min(tile0 + min(tl, geo.tpg - 1), ntiles - 1);   // (slots behind the group's tiles repeat its last one)
    const int ty = tau / geo.TW, tx = tau - ty * geo.TW;
    toff = ((M * ty * Wp + M * tx) - p0) * PITCH + c8 * 4;
  }
  // word offset of V[first point of this half][quad c8 >> 2][tile][c8 & 3]
  const int vpos = xh * 18 * 256 + (c8 >> 2) * 128 + tl * 4 + (c8 & 3);

  const unsigned char *xg = X.base + b * X.batch_stride + (int64_t)p0 * X.pix_stride;

  f32x16 acc[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;

  // raw span of one chunk: global -> registers -> LDS (scaled), as in fc_wino.hip
  const int npieces = geo.span * 4;
  u32x4w pf[kWnPF];
  auto piece_off = [&](int q) -> unsigned {
    const int pix = q >> 2;
    return (unsigned)min((int64_t)pix, avail - 1) * (unsigned)X.pix_stride + (unsigned)(q & 3) * 16u;
  };
  auto piece_store = [&](int q, u32x4w v, int cc) {
    const int pix = q >> 2;
    float f[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = pix >= avail ? 0.f : __uint_as_float(v[e]) * sx;
    float2 *d = reinterpret_cast<float2 *>(raw + (DB ? (cc & 1) * raw_bytes : 0) + pix * PITCH + (q & 3) * 16);
    d[0] = make_float2(f[0], f[1]);
    d[1] = make_float2(f[2], f[3]);
  };
  auto prefetch = [&](int cc) {
    const unsigned char *base = xg + (int64_t)cc * X.chunk_stride;
#pragma unroll
    for (int i = 0; i < kWnPF; ++i) pf[i] = *reinterpret_cast<const u32x4w *>(base + piece_off(min(t + kWnThreads * i, npieces - 1)));
  };
  auto commit = [&](int cc) {
#pragma unroll
    for (int i = 0; i < kWnPF; ++i) {
      // UNCONDITIONAL (threads behind the span rewrite its last piece with the same data, as they loaded it): with the store
      // under `if (q < npieces)` the consumer of pf[i] sat in a divergent branch, hipcc kept the register "pending" on the
      // skipped path and the NEXT prefetch -- which reuses pf[i]'s registers for its addresses right behind the multiply half
      // -- opened with s_waitcnt vmcnt(4) .. vmcnt(0): a wait for the B words requested a moment earlier (seen in the ISA,
      // round 6; the float32 kernel had carried it since round 3)
      piece_store(min(t + kWnThreads * i, npieces - 1), pf[i], cc);
    }
    const unsigned char *base = xg + (int64_t)cc * X.chunk_stride;
    for (int q = t + kWnThreads * kWnPF; q < npieces; q += kWnThreads)
      piece_store(q, *reinterpret_cast<const u32x4w *>(base + piece_off(q)), cc);
  };

  // this lane's B words: U16[ntile][step][point][g][n][4]
  const int nsteps = 2 * nch;
  const unsigned ub_wave = __builtin_amdgcn_readfirstlane((unsigned)((((unsigned)ntile * nsteps) * kWnXi + NP * pg) * 2 * kWnN + nblk * 32));
  const u32x4w *ub = reinterpret_cast<const u32x4w *>(U) + ub_wave + g * kWnN + l31;
  u32x4w bf[NP];
  auto load_b = [&](int step, int p) { return ub[((unsigned)step * kWnXi + p) * 2 * kWnN]; };

  // transform of step `step` (fc_wino.hip's, then the split): raw -> V[step & 1][point rows 3*HALF..][quad][tile][channel]
  auto transform = [&](auto half_tag, int step) {
    constexpr int HALF = decltype(half_tag)::value;
    const unsigned char *src = raw + (DB ? ((step >> 1) & 1) * raw_bytes : 0) + toff + (step & 1) * 32;
    uint32_t *dst = vbuf + (step & 1) * kWnVFloats + vpos;
    __builtin_amdgcn_s_setprio(3);
    float tm[3][6];
#pragma unroll
    for (int jp = 0; jp < 3; ++jp) {
      f32x2v d[6], o[3];
#pragma unroll
      for (int i = 0; i < 6; ++i)
        d[i] = f32x2v{*reinterpret_cast<const float *>(src + (i * Wp + 2 * jp) * PITCH),
                      *reinterpret_cast<const float *>(src + (i * Wp + 2 * jp + 1) * PITCH)};
      wn_bt3<HALF, f32x2v>(d, o);
#pragma unroll
      for (int r = 0; r < 3; ++r) tm[r][2 * jp] = o[r][0], tm[r][2 * jp + 1] = o[r][1];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float o[6];
      wn_bt_pk(tm[r], o);
#pragma unroll
      for (int e = 0; e < 6; ++e) dst[(r * 6 + e) * 256] = (DBG & 32) ? (uint32_t)__float_as_uint(o[e]) : wn16_split(o[e]);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // the wave's 18 MFMAs of step s: 9 points x (words, swapped words), taken in pairs of points so that the two MFMAs of an
  // accumulator are an issue slot apart; A words run one pair ahead of their MFMAs
  auto multiply = [&](int s, int sn) {
    const u32x4w *va = reinterpret_cast<const u32x4w *>(vbuf + (s & 1) * kWnVFloats + (NP * pg) * 256 + g * 128 + l31 * 4);
    u32x4w ra[2][2];
    auto read_pair = [&](int p, int sl) {
      ra[sl][0] = va[p * 64];
      if (p + 1 < NP) ra[sl][1] = va[(p + 1) * 64];
    };
    auto swapped = [](u32x4w w) {
      u32x4w r;
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = __builtin_amdgcn_alignbit(w[e], w[e], 16);
      return r;
    };
    read_pair(0, 0);
#pragma unroll
    for (int p = 0; p < NP; p += 2) {
      const int sl = (p >> 1) & 1;
      if (p + 2 < NP) read_pair(p + 2, sl ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 a0v = __builtin_bit_cast(f16x8, ra[sl][0]);
      const f16x8 b0v = __builtin_bit_cast(f16x8, bf[p]), b0s = __builtin_bit_cast(f16x8, swapped(bf[p]));
      if (p + 1 < NP) {
        const f16x8 a1v = __builtin_bit_cast(f16x8, ra[sl][1]);
        const f16x8 b1v = __builtin_bit_cast(f16x8, bf[p + 1]), b1s = __builtin_bit_cast(f16x8, swapped(bf[p + 1]));
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0v, b0v, acc[p], 0, 0, 0);
        acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1v, b1v, acc[p + 1], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0v, b0s, acc[p], 0, 0, 0);
        acc[p + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1v, b1s, acc[p + 1], 0, 0, 0);
        if constexpr (!(DBG & 4)) {
          bf[p] = load_b(sn, p);
          bf[p + 1] = load_b(sn, p + 1);
        }
      } else {
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0v, b0v, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0v, b0s, acc[p], 0, 0, 0);
        if constexpr (!(DBG & 4)) bf[p] = load_b(sn, p);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  prefetch(0);
  commit(0);
#pragma unroll
  for (int p = 0; p < NP; ++p) bf[p] = load_b(0, p);
  __syncthreads();
  if (xh == 0) transform(Half0{}, 0);
  else transform(Half1{}, 0);
  __syncthreads();

  stamp(t_pro);
  for (int s = 0; s < nsteps; ++s) {
    const int cc = s >> 1;
    const int sn = min(s + 1, nsteps - 1);
    const bool stage_next = !(s & 1) && cc + 1 < nch;
    // The request for the next chunk's pixels, the transform and the write of those pixels sit in ONE branch: as two separate
    // `if (stage_next)` around a shared transform hipcc cannot see that the write always follows the request, keeps the staging
    // registers "pending" at the loop header and opens the next request with s_waitcnt vmcnt(4) .. vmcnt(0) -- a wait for the
    // B words the multiply half requested a moment earlier (seen in the ISA, round 6; fc_wino.hip carried it since round 3).
    constexpr bool kT = !(DBG & 1), kM = !(DBG & 2), kS = !(DBG & 8);
    if (xh == 0) {
      if constexpr (kM) multiply(s, sn);
      __builtin_amdgcn_sched_barrier(0);
      stamp(t_first);
      if (kS && stage_next) {
        prefetch(cc + 1);
        if constexpr (kT) transform(Half0{}, s + 1);
        if constexpr (DB) commit(cc + 1);
      } else {
        if constexpr (kT) transform(Half0{}, s + 1);
      }
      stamp(t_second);
    } else {
      if (kS && stage_next) {
        prefetch(cc + 1);
        if constexpr (kT) transform(Half1{}, s + 1);
        if constexpr (DB) commit(cc + 1);
      } else {
        if constexpr (kT) transform(Half1{}, s + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      stamp(t_first);
      if constexpr (kM) multiply(s, sn);
      stamp(t_second);
    }
    __syncthreads();
    stamp(t_bar);
    if constexpr (!DB) {
      if (stage_next) {  // single raw buffer: written between two barriers (large maps only)
        commit(cc + 1);
        __syncthreads();
      }
    }
  }

  // epilogue: Y = A^T M A = sum over the points (a, e) of A^T[i][a] A^T[j][e] M[a][e].  A wave holds nine points -- row
  // a = (9 pg) / 6 from column (9 pg) % 6 on and what follows -- of 32 tiles x 32 channels: C/D layout of the 32x32 MFMA,
  // acc[p][i] = (tile 8 (i >> 2) + 4 g + (i & 3), channel lane & 31).  The tiles are finished a quarter at a time: every wave
  // reduces its points to the m x m partial of the quarter's four accumulator rows, three waves of a channel block park
  // theirs in LDS, the fourth (pg == quarter) adds them to its own and stores.
  float4 *xch = reinterpret_cast<float4 *>(gfla_smem);   // [3 writers][2 channel blocks][4 rows][m*m / 4][64 lanes]
  constexpr int MM4 = M * M / 4;
  const int col = ntile * kWnN + nblk * 32 + l31;
  float *ob = out + b * out_bs + col;
  auto partial = [&](auto pg_tag, int i, float (&part)[M * M]) {
    constexpr int PG = decltype(pg_tag)::value;
#pragma unroll
    for (int e = 0; e < M * M; ++e) part[e] = 0.f;
    // the wave's points, row by row of the point grid
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      constexpr int first = NP * PG, last = NP * PG + NP - 1;
      const int e0 = a * 6 > first ? 0 : first - a * 6, e1 = a * 6 + 5 < last ? 5 : last - a * 6;   // columns of row a in the set
      if (a * 6 + 5 < first || a * 6 > last) continue;
      float tj[M];
#pragma unroll
      for (int j = 0; j < M; ++j) {
        float sacc = 0.f;
#pragma unroll
        for (int e = 0; e < 6; ++e)
          if (e >= e0 && e <= e1 && WnAT<M>::v[j][e] != 0.f) sacc = fmaf(WnAT<M>::v[j][e], acc[a * 6 + e - first][i], sacc);
        tj[j] = sacc;
      }
#pragma unroll
      for (int ii = 0; ii < M; ++ii)
        if (WnAT<M>::v[ii][a] != 0.f) {
#pragma unroll
          for (int j = 0; j < M; ++j) part[ii * M + j] = fmaf(WnAT<M>::v[ii][a], tj[j], part[ii * M + j]);
        }
    }
  };
  // quarters per exchange round: all four at m = 2 (one round: two barriers per workgroup -- four rounds of two measured
  // 70 us of the k = 5 forward's 490), one at m = 4 (16 values per tile and channel: a round of one quarter fills 96 KB)
  constexpr int QPR = M == 2 ? 4 : 1;
  auto finish = [&](auto pg_tag) {
    constexpr int PG = decltype(pg_tag)::value;
#pragma unroll
    for (int q0 = 0; q0 < 4; q0 += QPR) {
      float part[QPR][4][M * M];
#pragma unroll
      for (int qq = 0; qq < QPR; ++qq)
#pragma unroll
        for (int r = 0; r < 4; ++r) partial(pg_tag, 4 * (q0 + qq) + r, part[qq][r]);
      if constexpr (!(DBG & 256)) __syncthreads();   // the main loop's LDS (or the previous round's partials) is dead
#pragma unroll
      for (int qq = 0; qq < QPR; ++qq) {
        const int qd = q0 + qq;
        if (PG != qd && !(DBG & 256)) {
          const int w3 = (PG - qd - 1) & 3;   // 0..2
          float4 *dst = xch + (((qq * 3 + w3) * 2 + nblk) * 4 * MM4) * 64 + lane;
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c4 = 0; c4 < MM4; ++c4)
              dst[(r * MM4 + c4) * 64] = make_float4(part[qq][r][4 * c4], part[qq][r][4 * c4 + 1], part[qq][r][4 * c4 + 2],
                                                     part[qq][r][4 * c4 + 3]);
        }
      }
      if constexpr (!(DBG & 256)) __syncthreads();
#pragma unroll
      for (int qq = 0; qq < QPR; ++qq) {
        const int qd = q0 + qq;
        if (PG != qd) continue;
#pragma unroll
        for (int w3 = 0; w3 < ((DBG & 256) ? 0 : 3); ++w3) {
          const float4 *srcp = xch + (((qq * 3 + w3) * 2 + nblk) * 4 * MM4) * 64 + lane;
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c4 = 0; c4 < MM4; ++c4) {
              const float4 v = srcp[(r * MM4 + c4) * 64];
              part[qq][r][4 * c4] += v.x, part[qq][r][4 * c4 + 1] += v.y, part[qq][r][4 * c4 + 2] += v.z, part[qq][r][4 * c4 + 3] += v.w;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int slot_ = 8 * qd + 4 * g + r, tau = tile0 + slot_;
          if (slot_ >= geo.tpg || tau >= ntiles || col >= n_valid) continue;
          const int ty = tau / geo.TW, tx = tau - ty * geo.TW;
#pragma unroll
          for (int i = 0; i < M; ++i) {
            const int yo = M * ty + i;
            if (yo >= Ho) continue;
#pragma unroll
            for (int j = 0; j < M; ++j) {
              const int xo = M * tx + j;
              if (xo < Wv && (!(DBG & 128) || part[qq][r][i * M + j] == 1.2345f))
                ob[(int64_t)(yo * Wv + xo) * ldo] = (part[qq][r][i * M + j] * inv_x) * inv_w;
            }
          }
        }
      }
    }
  };
  if constexpr (DBG & 64) {
    float keep = 0.f;   // (every accumulator stays live: the MFMAs must not be eliminated with the epilogue)
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int i = 0; i < 16; ++i) keep += acc[p][i];
    if (keep == 1.2345f) ob[0] = keep * inv_x * inv_w;
    return;
  }
  if (pg == 0) finish(PgTag<0>{});
  else if (pg == 1) finish(PgTag<1>{});
  else if (pg == 2) finish(PgTag<2>{});
  else finish(PgTag<3>{});
  if constexpr (DBG & 16) {
    stamp(t_epi);
    if (stamps && lane == 0) {
      unsigned long long *o = stamps + ((int64_t)blockIdx.x * 8 + wave) * 6;
      o[0] = t_pro, o[1] = t_first, o[2] = t_second, o[3] = t_bar, o[4] = t_epi, o[5] = (unsigned long long)xh;
    }
  }
}

static unsigned wn16_lds_bytes(int k, const WnGeo &g, bool double_raw) {
  const int m = k == 5 ? 2 : 4;
  const unsigned raw = k == 5 ? wn_raw_bytes<5>(g) : wn_raw_bytes<3>(g);
  const unsigned main_loop = (unsigned)(2 * kWnVFloats * 4) + (double_raw ? 2u : 1u) * raw;
  const unsigned exchange = (unsigned)(3 * 2 * 4 * m * m * 64 * 4) * (m == 2 ? 4u : 1u);   // (quarters per round: QPR)
  return main_loop > exchange ? main_loop : exchange;
}

template <int K_>
static int wn16_launch(const Wn16ConvJob *jobs, int njobs, int64_t B, int nch, const uint32_t *amax_w, hipStream_t stream) {
  Wn16KArgs a[2];
  int64_t wgs[2] = {0, 0};
  bool db = tuning(21) != 1;
  unsigned lds = 0;
  for (int j = 0; j < njobs; ++j) db = db && wn16_lds_bytes(K_, wn_geometry<K_>(jobs[j].M, jobs[j].Wv, jobs[j].Wp), true) <= kWnLdsLimit;
  for (int j = 0; j < 2; ++j) {
    const Wn16ConvJob &J = jobs[j < njobs ? j : 0];
    const WnGeo g = wn_geometry<K_>(J.M, J.Wv, J.Wp);
    const int ntn = (int)ceil_div(J.n_valid, kWnN);
    const int64_t groups = B * g.ngroups;
    a[j] = Wn16KArgs{J.X, J.U, J.amax_x, J.out, J.out_bs, J.ldo, J.n_valid, J.M / J.Wv, J.Wv, J.Wp, g, ntn, groups, J.S};
    if (j < njobs) {
      wgs[j] = ceil_div(groups, 8) * 8 * ntn;
      lds = std::max(lds, wn16_lds_bytes(K_, g, db));
    }
  }
  if (wgs[0] + wgs[1] > 0x7fffffffLL || lds > kWnLdsLimit) return GFLA_ERR_UNSUPPORTED;
#define GFLA_W16_LAUNCH(D_)                                                                                              \
  {                                                                                                                     \
    auto kern = fc_wino16_conv_kernel<K_, true, D_>;                                                                    \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    kern<<<dim3((unsigned)(wgs[0] + wgs[1])), kWnThreads, lds, stream>>>(a[0], a[1], (unsigned)wgs[0], nch, amax_w, g_wino_stamps); \
  }
  if (db) {
#ifdef GFLA_PROBES
    switch (K_ == 5 ? tuning(20) : 0) {
      case 1: GFLA_W16_LAUNCH(1) break;
      case 2: GFLA_W16_LAUNCH(2) break;
      case 3: GFLA_W16_LAUNCH(3) break;
      case 4: GFLA_W16_LAUNCH(4) break;
      case 8: GFLA_W16_LAUNCH(8) break;
      case 16: GFLA_W16_LAUNCH(16) break;
      case 32: GFLA_W16_LAUNCH(32) break;
      case 64: GFLA_W16_LAUNCH(64) break;
      case 128: GFLA_W16_LAUNCH(128) break;
      case 256: GFLA_W16_LAUNCH(256) break;
      case 384: GFLA_W16_LAUNCH(384) break;
      case 67: GFLA_W16_LAUNCH(67) break;
      case 79: GFLA_W16_LAUNCH(79) break;
      default: GFLA_W16_LAUNCH(0) break;
    }
#else
    GFLA_W16_LAUNCH(0)
#endif
#undef GFLA_W16_LAUNCH
  } else {
    auto kern = fc_wino16_conv_kernel<K_, false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    kern<<<dim3((unsigned)(wgs[0] + wgs[1])), kWnThreads, lds, stream>>>(a[0], a[1], (unsigned)wgs[0], nch, amax_w, nullptr);
  }
  return launch_status();
}

bool fc_wino16_fits(int M, int Wv, int Wp, int k) {
  if (!fc_wino_fits(M, Wv, Wp, k)) return false;
  const WnGeo g = k == 5 ? wn_geometry<5>(M, Wv, Wp) : wn_geometry<3>(M, Wv, Wp);
  return wn16_lds_bytes(k, g, false) <= kWnLdsLimit;
}

// one or two convolutions (same B, input chunks nch, k) in one launch; the contract of fc_wino_conv_jobs with the weights of
// fc_wino16_pack_weights and the max |x| slot of every job's input
int fc_wino16_conv_jobs(const Wn16ConvJob *jobs, int njobs, int64_t B, int nch, int k, const uint32_t *amax_w, hipStream_t stream) {
  if (B <= 0 || njobs <= 0) return GFLA_OK;
  if (njobs > 2 || !amax_w) return GFLA_ERR_UNSUPPORTED;
  for (int j = 0; j < njobs; ++j)
    if (!jobs[j].amax_x || !fc_wino16_fits(jobs[j].M, jobs[j].Wv, jobs[j].Wp, k)) return GFLA_ERR_UNSUPPORTED;
  if (njobs == 2 && tuning(21) == 2) {
    const int st = fc_wino16_conv_jobs(jobs, 1, B, nch, k, amax_w, stream);
    return st != GFLA_OK ? st : fc_wino16_conv_jobs(jobs + 1, 1, B, nch, k, amax_w, stream);
  }
  return k == 5 ? wn16_launch<5>(jobs, njobs, B, nch, amax_w, stream) : wn16_launch<3>(jobs, njobs, B, nch, amax_w, stream);
}

}  // namespace gfla
