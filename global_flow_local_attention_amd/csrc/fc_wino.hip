// Winograd-domain f32 MFMA convolutions for ExtractorAttn's first FC layer, gfx950 (arithmetic mode 4).
//
// The stride-1 k x k convolutions of the "sample the convolved map" formulation (fc_gemm.hip; reference
// base_function.py:799-807) are evaluated as F(2x2, 5x5) (k = 5) / F(4x4, 3x3) (k = 3) with the SAME 6 interpolation
// points {0, 1, -1, 2, -1/2, inf}:   Y = A^T [ (G w G^T) .* (B^T d B) ] A   per 6 x 6 input tile d and channel pair,
// so a tile's m x m outputs cost 36 multiplies per (input channel, output channel) instead of 100 (k = 5, m = 2) or
// 144 (k = 3, m = 4): 2.78x / 4x fewer MFMA flops, all arithmetic float32 (transforms on the vector ALUs, the 36
// point-wise products as 36 small GEMMs over the channels on v_mfma_f32_16x16x4_f32).  Measured error of the
// formulation in float32 against float64 (tests/test_fc_wino_*.py): forward 2.4e-6 / 5.6e-6 of the largest output,
// weight gradient 5e-6 -- the direct f32 form gives 4e-7 / 2.6e-6; the reference's own cuDNN / MIOpen convolutions
// are Winograd kernels of the same family.
//
// One fused kernel per convolution (forward of either half, and the data gradient = the same kernel on the Z-layout
// gradient map with flipped / transposed weights): no transformed tensor ever exists in HBM.
//   workgroup = 32 tiles (two 16-row MFMA blocks) x 64 output channels x ALL 36 points; 8 waves, wave w = output
//   channels 16(w&3)..+15 and 18 of the 36 points (three rows of the point grid): 18 x 2 accumulators of 16x16 (144
//   registers, two waves per SIMD);
//   per 8-channel step:  A = transformed input V[point][tile][8 ch] from LDS (double buffered: the NEXT step's
//   transform -- a thread takes one (tile, channel) item, B^T d in registers, then its wave's three rows of
//   (B^T d) B -- runs on the vector ALUs while the other wave of the SIMD runs this step's MFMAs: the two waves of a
//   SIMD take the two halves of a step in opposite order), B = the wave's slice of U = G w G^T, global -> registers
//   directly in fragment layout, each register reloaded for the next step right after its last use;
//   raw input pixels of a 16-channel chunk: one contiguous span of the linearised map (tap (i,j) = pixel offset
//   i*Wp + j, as in fc_conv_impl.h), prefetched into registers one step ahead, LDS pitch 80 / 72 bytes so that the
//   transform's reads (8 tiles x 8 channels per wave) are spread over all banks;
//   epilogue: A^T M A: each wave reduces its three point rows to an m x m partial per (tile, channel) in registers, the
//   wave pairs swap partials through LDS, stores go to the same (pixel, channel) f32 map fc_conv writes.
#include "fc_wino_shared.h"

namespace gfla {


// ---- weights: conv0.weight (128, 2C, k, k) -> U = G w G^T in MFMA B-fragment order -------------------------------
// U[ntile][chunk][half][point pair][nblock][lane][point & 1][2]: lane (kq = lane >> 4, n = lane & 15) holds input channels
// 16*chunk + 8*half + {kq, 4 + kq} of output channel 64*ntile + 16*nblock + n, for two neighbouring points: ONE 16-byte
// load per point pair and step.
// forward:        in = conv0 input channel c_off + ci, out = hidden n, taps as stored;
// data gradient:  in = hidden n, out = conv0 input channel c_off + co, taps flipped (the transposed convolution).

// grid (blocks, 4 jobs): forward / data-gradient sets of the target / source half in ONE launch.  Threads run along the
// OUTPUT channel: the 36 stores of 16 neighbouring threads fill consecutive fragment slots.
template <int KS>
__global__ __launch_bounds__(256) void fc_wino_pack_w_kernel(const float *__restrict__ w0, WnPackJobs jobs, int C) {
  const WnPackJob jb = jobs.j[blockIdx.y];
  const int nch = (jb.n_in + kFcChunk - 1) / kFcChunk, ntn = (jb.n_out + kWnN - 1) / kWnN;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (in channel, out channel), out fastest
  if (!jb.U || idx >= (int64_t)ntn * kWnN * nch * kFcChunk) return;
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
      w[i][j] = v;
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
  const int ntile = co / kWnN, nb = (co % kWnN) >> 4, n = co & 15;
  const int cc = ci >> 4, half = (ci >> 3) & 1, ks = (ci >> 2) & 1, kq = ci & 3;
  float *dst = jb.U + ((((((int64_t)ntile * nch + cc) * 2 + half) * (kWnXi / 2)) * 4 + nb) * 64 + kq * 16 + n) * 4 + ks;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    float o[6];
    wn_g<KS>(t[a], o);
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const int q = a * 6 + e;
      dst[(int64_t)(q >> 1) * 4 * 64 * 4 + (q & 1) * 2] = o[e];
    }
  }
}

int64_t fc_wino_wpack_bytes(int n_in, int n_out) {
  return (int64_t)ceil_div(n_out, kWnN) * ceil_div(n_in, kFcChunk) * 2 * kWnXi * 4 * 64 * 2 * 4;
}

// the four weight sets of one layer: forward (C -> 128) and data gradient (128 -> C) of the target / source half
int fc_wino_pack_weights(const float *w0, float *u_ft, float *u_fs, float *u_dt, float *u_ds, int C, int k,
                         hipStream_t stream) {
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
    fc_wino_pack_w_kernel<5><<<grid, 256, 0, stream>>>(w0, jobs, C);
  else if (k == 3)
    fc_wino_pack_w_kernel<3><<<grid, 256, 0, stream>>>(w0, jobs, C);
  else
    return GFLA_ERR_UNSUPPORTED;
  return launch_status();
}

// ---- the convolution ---------------------------------------------------------------------------------------------

// 8 waves.  Wave w: output channels 16*(w & 3) .. +15 of the workgroup's 64, points 18*(w >> 2) .. +17 (three rows of the
// 6 x 6 point grid), both 16-tile blocks: 18 x 2 accumulators of 16x16 = 144 registers, two waves per SIMD.  Waves w and
// w + 4 sit on the same SIMD and run the two halves of a step in OPPOSITE order -- one multiplies (matrix cores) while the
// other transforms the next step's input (vector ALUs, LDS).
//
// One launch carries up to TWO independent convolutions (the target and the source half of a layer: same weights' shape,
// different maps): workgroups [0, n0) belong to job 0, the rest to job 1.  A workgroup lives for ~1/6 of a launch, so a
// launch of 5.5 or 6.4 rounds of 256 workgroups spends its last round half empty; two jobs in one grid share that tail
// (L2, k = 5: 6 + 7 and 7 + 8 rounds become 12 and 14).
struct WnKArgs {
  PackedDesc X;
  const float *U;
  float *out;
  int64_t out_bs;
  int ldo, n_valid, Ho, Wv, Wp;
  WnGeo geo;
  int ntn;
  int64_t total_groups, S;
};
template <int KS, int DBG = 0, bool DB = true>
__global__ __launch_bounds__(kWnThreads, 2) void fc_wino_conv_kernel(WnKArgs a0, WnKArgs a1, unsigned n0, int nch,
                                                                    unsigned long long *stamps) {
  constexpr int M = Wn<KS>::M, PITCH = Wn<KS>::PITCH, NX = kWnXi / 2;
  // the job's parameters: workgroup-uniform selects (scalar registers)
  const bool second = blockIdx.x >= n0;
#define GFLA_PICK(f) (second ? a1.f : a0.f)
  PackedDesc X;
  X.base = GFLA_PICK(X.base), X.split_stride = 0, X.batch_stride = GFLA_PICK(X.batch_stride);
  X.chunk_stride = GFLA_PICK(X.chunk_stride), X.pix_stride = GFLA_PICK(X.pix_stride);
  const float *__restrict__ U = GFLA_PICK(U);
  float *__restrict__ out = GFLA_PICK(out);
  const int64_t out_bs = GFLA_PICK(out_bs), total_groups = GFLA_PICK(total_groups), S = GFLA_PICK(S);
  const int ldo = GFLA_PICK(ldo), n_valid = GFLA_PICK(n_valid), Ho = GFLA_PICK(Ho), Wv = GFLA_PICK(Wv), Wp = GFLA_PICK(Wp);
  const int ntn = GFLA_PICK(ntn);
  WnGeo geo;
  geo.TH = GFLA_PICK(geo.TH), geo.TW = GFLA_PICK(geo.TW), geo.ngroups = GFLA_PICK(geo.ngroups), geo.span = GFLA_PICK(geo.span);
  geo.tpg = GFLA_PICK(geo.tpg);
#undef GFLA_PICK
  // DBG & 16: per-wave phase timing (s_memtime) summed over the steps -> stamps[workgroup][wave][6]
  unsigned long long tk0 = 0, t_first = 0, t_second = 0, t_bar = 0, t_pro = 0, t_epi = 0;
  if constexpr (DBG & 16) tk0 = __builtin_amdgcn_s_memtime();
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  float *vbuf = reinterpret_cast<float *>(gfla_smem);      // [2][36][32][8]
  unsigned char *raw = gfla_smem + 2 * kWnVFloats * 4;     // [1 or 2][span][PITCH]
  const int raw_bytes = (geo.span * PITCH + 15) & ~15;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nb = wave & 3, xh = wave >> 2;
  // workgroup -> (group of tiles, output-channel tile).  Ids x and x + 8 run on the same XCD: the workgroups that
  // share one group's input pixels (different channel tiles) are neighbours in that XCD's queue (shared L2).
  const int64_t x = blockIdx.x - (second ? n0 : 0u);   // n0 is a multiple of 8: id & 7 is still the XCD
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
  const int p0 = M * ty_first * Wp;                         // first pixel of the staged span
  const int64_t avail = S - p0;                             // pixels of this sample behind p0 (the rest reads as zero)

  // transform item of this thread: (tile, channel of the 8-channel step), rows 3*xh .. 3*xh + 2 of the point grid
  const int tl = (t & 255) >> 3, c8 = t & 7;
  int toff;
  {
    const int tau = min(tile0 + min(tl, geo.tpg - 1), ntiles - 1);   // (slots behind the group's tiles repeat its last one)
    const int ty = tau / geo.TW, tx = tau - ty * geo.TW;
    toff = ((M * ty * Wp + M * tx) - p0) * PITCH + c8 * 4;
  }
  // float offset of V[first point][16-tile block][channel pair kq >> 1][tile][kq & 1][k step]: the A fragments of a
  // half-wave (kq = 0, 1 x 16 tiles, 8 bytes each) are then 256 contiguous bytes -- no bank conflicts on the b64 reads
  const int vpos = xh * NX * kWnTiles * 8 + (tl >> 4) * 128 + (((c8 & 3) >> 1) * 16 + (tl & 15)) * 4 + (c8 & 1) * 2 + (c8 >> 2);

  const unsigned char *xg = X.base + b * X.batch_stride + (int64_t)p0 * X.pix_stride;  // workgroup-uniform

  f32x4v acc[NX][2];
#pragma unroll
  for (int q = 0; q < NX; ++q)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc[q][mb] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // raw span of one chunk: 16-byte pieces t, t + 512, ... go global -> registers -> LDS (LDS-DMA was measured and dropped:
  // with a DMA in flight hipcc turns the counted vmcnt waits of the B-fragment stream into vmcnt(0)).  Addresses = a
  // uniform base + a 32-bit per-lane offset; pixels behind the end of the sample read its last pixel and are stored as
  // zeros.  Spans beyond kWnPF pieces per thread are loaded at the commit (large maps only).
  const int npieces = geo.span * 4;
  u32x4v pf[kWnPF];
  auto piece_off = [&](int q) -> unsigned {
    const int pix = q >> 2;
    return (unsigned)min((int64_t)pix, avail - 1) * (unsigned)X.pix_stride + (unsigned)(q & 3) * 16u;
  };
  auto piece_store = [&](int q, u32x4v v, int cc) {
    const int pix = q >> 2;
    if (pix >= avail) v = u32x4v{0u, 0u, 0u, 0u};
    uint2 *d = reinterpret_cast<uint2 *>(raw + (DB ? (cc & 1) * raw_bytes : 0) + pix * PITCH + (q & 3) * 16);
    d[0] = make_uint2(v[0], v[1]);
    d[1] = make_uint2(v[2], v[3]);
  };
  auto prefetch = [&](int cc) {
    const unsigned char *base = xg + (int64_t)cc * X.chunk_stride;
#pragma unroll
    for (int i = 0; i < kWnPF; ++i) pf[i] = *reinterpret_cast<const u32x4v *>(base + piece_off(min(t + kWnThreads * i, npieces - 1)));
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
      piece_store(q, *reinterpret_cast<const u32x4v *>(base + piece_off(q)), cc);
  };

  // this lane's B fragments of the current step: U[ntile][cc][half][point][nb][lane][2]
  // (the wave's base offset goes through readfirstlane: a scalar base + one per-lane offset register, no per-load
  // vector address arithmetic in the MFMA stream)
  const unsigned ub_wave = __builtin_amdgcn_readfirstlane((unsigned)((((unsigned)ntile * nch * 2 * (kWnXi / 2) + xh * (NX / 2)) * 4 + nb) * 64));
  const f32x4v *ub = reinterpret_cast<const f32x4v *>(U) + ub_wave;
  f32x4v bf[NX / 2];   // [point pair]: (point 0: k step 0, 1; point 1: k step 0, 1)
  auto load_b = [&](int step, int qp) { return (ub + ((unsigned)step * (kWnXi / 2) + qp) * 4 * 64)[lane]; };

  const int arow = ((lane >> 5) * 16 + (lane & 15)) * 4 + ((lane >> 4) & 1) * 2;  // this lane's A fragment inside V[point][block]

  // ---- the two halves of a step ---------------------------------------------------------------------------------------
  // f32-input MFMAs execute on the SIMD's f32 ALUs (157 TFLOP/s = the vector rate): vector instructions are NOT hidden
  // behind them, they add (measured: every ablation of this kernel is additive).  So the instruction streams are kept
  // lean, and the two waves of a SIMD run the two halves in opposite order so that the LDS / global latencies of one sit
  // under the arithmetic of the other.
  // transform of step `step`: raw[(tile pixel + i*Wp + j)][channel] -> V[step & 1][point rows 3*HALF..][tile][channel]
  // (only the three rows this wave group owns: 18 live values, half the column-pass arithmetic)
  auto transform = [&](auto half_tag, int step) {
    constexpr int HALF = decltype(half_tag)::value;
    const unsigned char *src = raw + (DB ? ((step >> 1) & 1) * raw_bytes : 0) + toff + (step & 1) * 32;
    float *dst = vbuf + (step & 1) * kWnVFloats + vpos;
    __builtin_amdgcn_s_setprio(3);  // the short phase goes first whenever both waves of the SIMD can issue
    // column pass on PAIRS of columns: the same fma chain for columns j, j + 1 is one v_pk_fma_f32 / v_pk_add_f32 each
    // (vector instructions add to the MFMA time here, so half as many of them is worth having)
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
      for (int e = 0; e < 6; ++e) dst[(r * 6 + e) * kWnTiles * 8] = o[e];
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // the wave's 72 MFMAs of step s: 18 points x 2 tile blocks x 2 k steps, taken in 9 pairs of points -- eight MFMAs on
  // four accumulators, the two MFMAs of an accumulator four issue slots apart (a 16x16x4 f32 MFMA has a 40-cycle
  // dependent latency against a 32-cycle issue).  A fragments run one pair ahead of their MFMAs; the sched_barriers keep
  // hipcc from sinking the reads next to their use, where every point would expose a full LDS round trip.
  auto multiply = [&](int s, int sn) {
    const float *va = vbuf + (s & 1) * kWnVFloats + xh * NX * kWnTiles * 8 + arow;
    float2 ra[2][2][2];  // [pair parity][point of the pair][tile block]
    auto read_pair = [&](int q, int slot) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ra[slot][u][0] = *reinterpret_cast<const float2 *>(va + (q + u) * kWnTiles * 8);
        ra[slot][u][1] = *reinterpret_cast<const float2 *>(va + (q + u) * kWnTiles * 8 + 16 * 8);
      }
    };
    read_pair(0, 0);
#pragma unroll
    for (int q = 0; q < NX; q += 2) {
      const int slot = (q >> 1) & 1;
      if (q + 2 < NX) read_pair(q + 2, slot ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      const f32x4v bq = bf[q >> 1];
      acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[slot][0][0].x, bq[0], acc[q][0], 0, 0, 0);
      acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[slot][0][1].x, bq[0], acc[q][1], 0, 0, 0);
      acc[q + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[slot][1][0].x, bq[2], acc[q + 1][0], 0, 0, 0);
      acc[q + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[slot][1][1].x, bq[2], acc[q + 1][1], 0, 0, 0);
      acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[slot][0][0].y, bq[1], acc[q][0], 0, 0, 0);
      acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[slot][0][1].y, bq[1], acc[q][1], 0, 0, 0);
      acc[q + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[slot][1][0].y, bq[3], acc[q + 1][0], 0, 0, 0);
      acc[q + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[slot][1][1].y, bq[3], acc[q + 1][1], 0, 0, 0);
      if constexpr (!(DBG & 4)) bf[q >> 1] = load_b(sn, q >> 1);  // the register is free again: next step's slice, a step ahead
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int nsteps = 2 * nch;
  prefetch(0);
  commit(0);
#pragma unroll
  for (int q = 0; q < NX / 2; ++q) bf[q] = load_b(0, q);
  __syncthreads();
  if (xh == 0) transform(Half0{}, 0);
  else transform(Half1{}, 0);
  __syncthreads();

  if constexpr (DBG & 16) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t_pro = n - tk0; tk0 = n; }
  for (int s = 0; s < nsteps; ++s) {
    const int cc = s >> 1;
    const int sn = min(s + 1, nsteps - 1);
    constexpr bool kS = !(DBG & 8);
    auto stamp = [&](unsigned long long &slot) {
      if constexpr (DBG & 16) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long n = __builtin_amdgcn_s_memtime();
        slot += n - tk0;
        tk0 = n;
      }
    };
    // raw staging: the next chunk is requested and written inside the even step, around the transform (its registers are
    // live across the transform only; the loads have its duration to land).  Two raw buffers: no extra barrier.
    const bool stage_next = !(s & 1) && cc + 1 < nch;
    constexpr bool kT = !(DBG & 1), kM = !(DBG & 2);
    // (the transform of the step after the last one reads a stale raw buffer into the unused V buffer: harmless)
    // request, transform and write of the next chunk's pixels in ONE branch: as two separate `if (stage_next)` around a shared
    // transform hipcc cannot see that the write always follows the request, keeps the staging registers "pending" at the loop
    // header and opens the next request with s_waitcnt vmcnt(4) .. vmcnt(0) -- a wait for the B fragments the multiply half
    // requested a moment earlier (seen in the ISA in round 6; the kernel had carried it since round 3)
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
    if constexpr (kS && !DB) {
      if (stage_next) {  // single raw buffer: written between two barriers (large maps only)
        commit(cc + 1);
        __syncthreads();
      }
    }
    stamp(t_bar);
  }

  // epilogue: Y = A^T M A = sum over the point rows a of A^T[:, a] (x) (A^T M[a, :]).  A wave holds three of the six rows:
  // it reduces them to an m x m partial per (tile, channel); wave pairs (w, w + 4) swap partials through LDS -- wave w
  // finishes tile block 0, wave w + 4 block 1.  C/D layout of the 16x16 MFMA: column (channel) = lane & 15,
  // row (tile) = 4*(lane >> 4) + r.
  float *xch = reinterpret_cast<float *>(gfla_smem);  // [mb][nb][lane][4 r][m*m], written by the wave that does NOT own mb
  const int col = ntile * kWnN + nb * 16 + (lane & 15);
  float *ob = out + b * out_bs + col;
  auto finish = [&](auto half_tag) {
    constexpr int HALF = decltype(half_tag)::value;   // this wave's point rows 3*HALF.., and the tile block it finishes
    float part[2][4][M * M];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float qv[3][M];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          float v[6], y[M];
#pragma unroll
          for (int e = 0; e < 6; ++e) v[e] = acc[a * 6 + e][mb][r];
          wn_at<M>(v, y);
#pragma unroll
          for (int j = 0; j < M; ++j) qv[a][j] = y[j];
        }
#pragma unroll
        for (int j = 0; j < M; ++j) {
          float v[6], y[M];
#pragma unroll
          for (int a = 0; a < 6; ++a) v[a] = (a >= 3 * HALF && a < 3 * HALF + 3) ? qv[a - 3 * HALF][j] : 0.f;
          wn_at<M>(v, y);
#pragma unroll
          for (int i = 0; i < M; ++i) part[mb][r][i * M + j] = y[i];
        }
      }
    __syncthreads();  // the main loop's LDS is dead
    {
      float *dst = xch + (((1 - HALF) * 4 + nb) * 64 + lane) * 4 * M * M;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < M * M; ++e) dst[r * M * M + e] = part[1 - HALF][r][e];
    }
    __syncthreads();
    const float *srcp = xch + ((HALF * 4 + nb) * 64 + lane) * 4 * M * M;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int slot_ = HALF * 16 + 4 * (lane >> 4) + r, tau = tile0 + slot_;
      if (slot_ >= geo.tpg || tau >= ntiles || col >= n_valid) continue;
      const int ty = tau / geo.TW, tx = tau - ty * geo.TW;
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const int yo = M * ty + i;
        if (yo >= Ho) continue;
#pragma unroll
        for (int j = 0; j < M; ++j) {
          const int xo = M * tx + j;
          if (xo < Wv) ob[(int64_t)(yo * Wv + xo) * ldo] = part[HALF][r][i * M + j] + srcp[r * M * M + i * M + j];
        }
      }
    }
  };
  if (xh == 0) finish(Half0{});
  else finish(Half1{});
  if constexpr (DBG & 16) {
    t_epi = __builtin_amdgcn_s_memtime() - tk0;
    if (stamps && lane == 0) {
      unsigned long long *o = stamps + ((int64_t)blockIdx.x * 8 + wave) * 6;
      o[0] = t_pro, o[1] = t_first, o[2] = t_second, o[3] = t_bar, o[4] = t_epi, o[5] = (unsigned long long)xh;
    }
  }
}

bool fc_wino_fits(int M, int Wv, int Wp, int k) {
  if (k != 3 && k != 5) return false;
  if (Wv <= 0 || Wv > Wp || M <= 0 || M % Wv) return false;
  const WnGeo g = k == 5 ? wn_geometry<5>(M, Wv, Wp) : wn_geometry<3>(M, Wv, Wp);
  const unsigned lds = k == 5 ? wn_lds_bytes<5>(g, false) : wn_lds_bytes<3>(g, false);
  return lds <= kWnLdsLimit;
}

// out[b][r][n] = sum_{chunk, tap, c} X[b][chunk][pix(r) + tap][c] * w[...]  -- the contract of fc_conv (fc_conv_impl.h),
// with the weights given as the transformed U of fc_wino_pack_weights.  S = pixels per sample X may be read for.
unsigned long long *g_wino_stamps = nullptr;  // timing probe buffer (gfla_fc_wino_debug_buffer; tools only; also fc_wino16.hip)

template <int K_>
static int wn_launch(const WnConvJob *jobs, int njobs, int64_t B, int nch, hipStream_t stream) {
  unsigned long long *stamps = g_wino_stamps;
  WnKArgs a[2];
  int64_t wgs[2] = {0, 0};
  bool db = tuning(21) != 1;
  unsigned lds = 0;
  for (int j = 0; j < njobs; ++j) db = db && wn_lds_bytes<K_>(wn_geometry<K_>(jobs[j].M, jobs[j].Wv, jobs[j].Wp), true) <= kWnLdsLimit;
  for (int j = 0; j < 2; ++j) {
    const WnConvJob &J = jobs[j < njobs ? j : 0];
    const WnGeo g = wn_geometry<K_>(J.M, J.Wv, J.Wp);
    const int ntn = (int)ceil_div(J.n_valid, kWnN);
    const int64_t groups = B * g.ngroups;
    a[j] = WnKArgs{J.X, J.U, J.out, J.out_bs, J.ldo, J.n_valid, J.M / J.Wv, J.Wv, J.Wp, g, ntn, groups, J.S};
    if (j < njobs) {
      wgs[j] = ceil_div(groups, 8) * 8 * ntn;
      lds = std::max(lds, wn_lds_bytes<K_>(g, db));
    }
  }
  if (wgs[0] + wgs[1] > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
#define GFLA_WINO_LAUNCH(D_, DB_)                                                                                       \
  {                                                                                                                    \
    auto kern = fc_wino_conv_kernel<K_, D_, DB_>;                                                                      \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    kern<<<dim3((unsigned)(wgs[0] + wgs[1])), kWnThreads, lds, stream>>>(a[0], a[1], (unsigned)wgs[0], nch, stamps);     \
  }
  if (!db) GFLA_WINO_LAUNCH(0, false)
#ifdef GFLA_PROBES  // `make PROBES=1`: timing ablations of the k = 5 kernel (tuning key 20; their results are garbage, so
                    // a default build does not contain them and a stray key 20 cannot corrupt a forward / backward)
  else switch (K_ == 5 ? tuning(20) : 0) {
    case 1: GFLA_WINO_LAUNCH(1, true) break;
    case 2: GFLA_WINO_LAUNCH(2, true) break;
    case 4: GFLA_WINO_LAUNCH(4, true) break;
    case 8: GFLA_WINO_LAUNCH(8, true) break;
    case 3: GFLA_WINO_LAUNCH(3, true) break;
    case 16: GFLA_WINO_LAUNCH(16, true) break;
    case 5: GFLA_WINO_LAUNCH(5, true) break;
    case 13: GFLA_WINO_LAUNCH(13, true) break;
    default: GFLA_WINO_LAUNCH(0, true) break;
  }
#else
  else GFLA_WINO_LAUNCH(0, true)
#endif
#undef GFLA_WINO_LAUNCH
  return launch_status();
}

// one or two convolutions (same B, input chunks nch, k) in one launch; tuning key 21 = 2: one launch per job
int fc_wino_conv_jobs(const WnConvJob *jobs, int njobs, int64_t B, int nch, int k, hipStream_t stream) {
  if (B <= 0 || njobs <= 0) return GFLA_OK;
  if (njobs > 2) return GFLA_ERR_UNSUPPORTED;
  for (int j = 0; j < njobs; ++j)
    if (!fc_wino_fits(jobs[j].M, jobs[j].Wv, jobs[j].Wp, k)) return GFLA_ERR_UNSUPPORTED;
  if (njobs == 2 && tuning(21) == 2) {
    const int st = fc_wino_conv_jobs(jobs, 1, B, nch, k, stream);
    return st != GFLA_OK ? st : fc_wino_conv_jobs(jobs + 1, 1, B, nch, k, stream);
  }
  return k == 5 ? wn_launch<5>(jobs, njobs, B, nch, stream) : wn_launch<3>(jobs, njobs, B, nch, stream);
}

int fc_wino_conv(const PackedDesc &X, const float *U, float *out, int64_t out_bs, int ldo, int n_valid, int64_t B, int nch,
                 int M, int Wv, int Wp, int64_t S, int k, hipStream_t stream) {
  const WnConvJob job{X, U, out, out_bs, ldo, n_valid, M, Wv, Wp, S};
  return fc_wino_conv_jobs(&job, 1, B, nch, k, stream);
}

// =====================================================================================================================
// Weight gradient in the Winograd domain (arithmetic mode 4).
//
//   dU[point][c][n] = sum over tiles  V[tile][c][point] * Zh[tile][n][point],   V = B^T d B (the forward's input transform),
//   Zh = A dY A^T (the m x m output-gradient tile lifted to the 6 x 6 points),   dW[n][c] = G^T dU[.][c][n] G  (k x k),
// i.e. the adjoint of Y = A^T [U .* V] A: 36 multiplies per (tile, c, n) instead of 100 (k = 5) / 144 (k = 3).
//
// Workgroup (8 waves) = one 16-channel chunk of the input x ALL 36 points x the 128 hidden channels (wave w: columns
// 16w..16w+15: 36 accumulators of 16x16) x a contiguous range of "units" (= up to SEG tiles of ONE tile row of one sample;
// equal ranges per split, one round of workgroups).  The reduction runs over tiles, four per v_mfma_f32_16x16x4_f32:
//   A = V[point][tile][c] from LDS, produced 16 tiles at a time by the conv kernel's transform (a thread = one (tile,
//       channel) item, three of the six point rows; two V buffers), read back as one ds_read_b128 per point and step;
//   B = Zh: every lane lifts its own (tile, hidden channel) dY values -- 4 (k = 5) or 16 (k = 3) floats straight from the
//       gradient map in global memory, one k step ahead -- to the 36 points IN REGISTERS (about 32 vector ops per k step);
//   the two halves of a step (multiply / transform the next 16 tiles) run in opposite order on the two waves of a SIMD.
// Partial sums per split leave as plain coalesced stores; fc_wino_wgrad_reduce adds the splits, applies G^T . G and writes
// conv0.weight.grad's layout.
constexpr int kWwPitch = 72;   // LDS bytes per raw pixel: 16 tiles' stride (8 / 16 pixels) lands on the other half of the banks

template <int KS>
struct Ww {
  static constexpr int M = KS == 5 ? 2 : 4;
  static constexpr int SEG = KS == 5 ? 32 : 16;          // tiles per unit
  static constexpr int L = M * SEG + 6 - M;              // raw pixels per row of a unit
  static constexpr int RAW = ((6 * L * kWwPitch + 15) & ~15);
  // multi-row units (narrow maps): up to SEGM tiles = two 16-tile steps, so that the second step's transform runs under
  // the first one's MFMAs; the unit's raw rows (its own pitch) must fit kWwRawMax bytes and PFM pieces per thread
  static constexpr int SEGM = 32;
  static constexpr int PFM = KS == 5 ? 4 : 5;
};
constexpr int kWwRawMax = 43 * 1024;   // per raw buffer: 2 V buffers (72 KB) + 2 x 43 KB = the 160 KB of a CU
constexpr int kWwVFloats = 4 * 4 * 16 * kWnXi;           // one V buffer: [tile >> 2][tile & 3][channel][point]: a lane's 36
                                                         // A values of a k step are contiguous (nine ds_read_b128)

struct WwGeo {
  int TH, TW, nseg;
  int R, ups;   // tile rows per unit (> 1: narrow maps, see ww_geometry), units per sample
};

// A dY A^T for one (tile, channel): dy[i][j] (m x m) -> zh[36]
template <int M>
__device__ __forceinline__ void ww_lift(const float (&dy)[M][M], float (&zh)[kWnXi]) {
  float t[6][M];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    if constexpr (M == 2) {
      const float a = dy[0][j], b = dy[1][j];
      t[0][j] = a, t[1][j] = a + b, t[2][j] = a - b, t[3][j] = a + 2.f * b, t[4][j] = a - 0.5f * b, t[5][j] = b;
    } else {
      const float a = dy[0][j], b = dy[1][j], c = dy[2][j], d = dy[3][j];
      t[0][j] = a;
      t[1][j] = a + b + c + d;
      t[2][j] = a - b + c - d;
      t[3][j] = a + 2.f * b + 4.f * c + 8.f * d;
      t[4][j] = a - 0.5f * b + 0.25f * c - 0.125f * d;
      t[5][j] = d;
    }
  }
#pragma unroll
  for (int a6 = 0; a6 < 6; ++a6) {
    if constexpr (M == 2) {
      const float a = t[a6][0], b = t[a6][1];
      zh[a6 * 6 + 0] = a, zh[a6 * 6 + 1] = a + b, zh[a6 * 6 + 2] = a - b, zh[a6 * 6 + 3] = a + 2.f * b;
      zh[a6 * 6 + 4] = a - 0.5f * b, zh[a6 * 6 + 5] = b;
    } else {
      const float a = t[a6][0], b = t[a6][1], c = t[a6][2], d = t[a6][3];
      zh[a6 * 6 + 0] = a;
      zh[a6 * 6 + 1] = a + b + c + d;
      zh[a6 * 6 + 2] = a - b + c - d;
      zh[a6 * 6 + 3] = a + 2.f * b + 4.f * c + 8.f * d;
      zh[a6 * 6 + 4] = a - 0.5f * b + 0.25f * c - 0.125f * d;
      zh[a6 * 6 + 5] = d;
    }
  }
}

struct WwUnit {
  int64_t b;
  int ty, tx0, ntx;   // first tile row, first tile column, tiles per row
  int nt;             // tiles of the unit = ntx * rows (rows > 1 only in the multi-row instantiation)
};

// DBG (timing ablations, tools only; results are garbage): 1 no input transform, 2 no MFMAs / A reads, 4 no dY loads,
// 8 no lift of dY to the 36 points, 16 no raw staging
struct WwKArgs {
  PackedDesc X;
  const float *Z;
  float *part;
  int64_t z_bs, z_lead, total_units, SX;
  int Wp, Wo, nsplit;
  WwGeo geo;
};
// MR (multi-row units): on a map whose tile rows are at most half a unit (TW <= SEG / 2: the k = 3 layer at 32x22 has 6
// tiles of 4x4 per row against units of 16) a unit of ONE tile row left the k steps mostly empty -- 6 of 16 tiles, one
// exposed transform per 6 tiles -- and the direct kernel won (135 vs 172 us).  With MR a unit is R = SEG / TW whole tile
// rows: the raw rows are staged with the map's own pitch instead of the unit's maximum, tile t of the unit is (t / TW,
// t % TW).  The single-row instantiation keeps its compile-time pitch (every LDS offset of the transform an immediate).
template <int KS, int DBG = 0, bool MR = false>
__global__ __launch_bounds__(kWnThreads, 2) void fc_wino_wgrad_kernel(WwKArgs a0, WwKArgs a1, int nsplit0, int cpad,
                                                                     int raw_stride) {
  // One launch carries up to TWO weight gradients (the source and the target half of a layer): split indices
  // [0, nsplit0) belong to job 0, the rest to job 1 -- each job is one round of workgroups, and in one grid the second
  // round starts on a CU the moment the first one's workgroup there retires.  Workgroup-uniform selects (scalar registers).
  const bool second = (int)blockIdx.y >= nsplit0;
#define GFLA_PICK(f) (second ? a1.f : a0.f)
  PackedDesc X;
  X.base = GFLA_PICK(X.base), X.split_stride = 0, X.batch_stride = GFLA_PICK(X.batch_stride);
  X.chunk_stride = GFLA_PICK(X.chunk_stride), X.pix_stride = GFLA_PICK(X.pix_stride);
  const float *__restrict__ Z = GFLA_PICK(Z);
  float *__restrict__ part = GFLA_PICK(part);
  const int64_t z_bs = GFLA_PICK(z_bs), z_lead = GFLA_PICK(z_lead), total_units = GFLA_PICK(total_units), SX = GFLA_PICK(SX);
  const int Wp = GFLA_PICK(Wp), Wo = GFLA_PICK(Wo), nsplit = GFLA_PICK(nsplit);
  WwGeo geo;
  geo.TH = GFLA_PICK(geo.TH), geo.TW = GFLA_PICK(geo.TW), geo.nseg = GFLA_PICK(geo.nseg);
  geo.R = GFLA_PICK(geo.R), geo.ups = GFLA_PICK(geo.ups);
#undef GFLA_PICK
  constexpr int M = Ww<KS>::M, SEG = Ww<KS>::SEG, L = Ww<KS>::L, PITCH = kWwPitch;
  const int RAW = MR ? raw_stride : Ww<KS>::RAW;   // bytes between the two raw buffers
  // 16-byte pieces of a unit's raw rows per thread
  constexpr int PF1 = (6 * L * 4 + kWnThreads - 1) / kWnThreads;
  constexpr int PF = MR ? (PF1 > Ww<KS>::PFM ? PF1 : Ww<KS>::PFM) : PF1;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  float *vbuf = reinterpret_cast<float *>(gfla_smem);            // [2][kWwVFloats]
  unsigned char *raw = gfla_smem + 2 * kWwVFloats * 4;           // [2][6][L][PITCH]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, xh = wave >> 2;
  const int cc = blockIdx.x, sp = (int)blockIdx.y - (second ? nsplit0 : 0);
  const int64_t u0 = total_units * sp / nsplit, u1 = total_units * (sp + 1) / nsplit;
  const int per_sample = geo.ups;
  // raw row pitch in pixels and raw rows of a unit: the unit's maximum (compile time) or, multi-row, the map's own
  const int Lr = (MR && geo.R > 1) ? M * geo.TW + 6 - M : L;
  const int raw_rows = (MR && geo.R > 1) ? M * geo.R + 6 - M : 6;
  const int npieces = raw_rows * Lr * 4;
  const unsigned inv_ntx = (65536u + (unsigned)geo.TW - 1u) / (unsigned)geo.TW;   // t / TW = (t * inv) >> 16 for t < 2^8

  f32x4v acc[kWnXi];
#pragma unroll
  for (int q = 0; q < kWnXi; ++q) acc[q] = f32x4v{0.f, 0.f, 0.f, 0.f};

  auto unit_of = [&](int64_t u) {
    WwUnit un;
    un.b = u / per_sample;
    const int r = (int)(u - un.b * per_sample);
    if (MR && geo.R > 1) {   // R whole tile rows
      un.ty = r * geo.R;
      un.tx0 = 0;
      un.ntx = geo.TW;
      un.nt = geo.TW * min(geo.R, geo.TH - un.ty);
    } else {
      un.ty = r / geo.nseg;
      un.tx0 = (r - un.ty * geo.nseg) * SEG;
      un.ntx = min(SEG, geo.TW - un.tx0);
      un.nt = un.ntx;
    }
    return un;
  };
  // tile t of a unit -> (tile row inside the unit, tile column)
  auto tile_rc = [&](int t_, int &tr, int &tcol) {
    if (MR && geo.R > 1) {
      tr = (int)(((unsigned)t_ * inv_ntx) >> 16);
      tcol = t_ - tr * geo.TW;
    } else {
      tr = 0;
      tcol = t_;
    }
  };

  // raw rows of a unit: piece q -> (row = q / (4 L), pixel, part); global -> registers -> LDS (pitch 72: two b64 stores)
  u32x4v pf[PF];
  auto piece_addr = [&](const WwUnit &un, int q, int &ldso) -> const unsigned char * {
    const int row = q / (4 * Lr), rem = q - row * (4 * Lr), px = rem >> 2, prt = rem & 3;
    ldso = (row * Lr + px) * PITCH + prt * 16;
    const int64_t pix = (int64_t)(M * un.ty + row) * Wp + M * un.tx0 + px;
    return X.base + un.b * X.batch_stride + (int64_t)cc * X.chunk_stride + (pix < SX ? pix : SX - 1) * X.pix_stride + prt * 16;
  };
  auto prefetch = [&](const WwUnit &un) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      int ldso;
      pf[i] = *reinterpret_cast<const u32x4v *>(piece_addr(un, min(t + kWnThreads * i, npieces - 1), ldso));
    }
  };
  auto commit = [&](const WwUnit &un, int buf) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      // unconditional, like the convolution kernel's commit (threads behind the unit's pieces rewrite the last one)
      const int q = min(t + kWnThreads * i, npieces - 1);
      int ldso;
      (void)piece_addr(un, q, ldso);
      uint2 *d = reinterpret_cast<uint2 *>(raw + buf * RAW + ldso);
      d[0] = make_uint2(pf[i][0], pf[i][1]);
      d[1] = make_uint2(pf[i][2], pf[i][3]);
    }
  };

  // transform item: tile (wave & 3) + 4 * (lane >> 4) of the step's 16, channel lane & 15, point rows 3*xh..
  const int tq = wave & 3, tks = lane >> 4, tc = lane & 15;
  const int tl = tq + 4 * tks;
  const int vpos = ((tks * 4 + tq) * 16 + tc) * kWnXi + xh * 18;   // float offset of V[ks][tq][c][first point of this half]
  auto transform = [&](auto half_tag, const WwUnit &un, int h, int rbuf, int vb) {
    constexpr int HALF = decltype(half_tag)::value;
    const int tile = min(h * 16 + tl, un.nt - 1);
    int tr, tcol;
    tile_rc(tile, tr, tcol);
    const unsigned char *src = raw + rbuf * RAW + ((M * tr) * Lr + M * tcol) * PITCH + tc * 4;
    float *dst = vbuf + vb * kWwVFloats + vpos;
    __builtin_amdgcn_s_setprio(3);
    float tm[3][6];
#pragma unroll
    for (int jp = 0; jp < 3; ++jp) {   // column pass on pairs of columns (packed f32 instructions, as in the convolution kernel)
      f32x2v d[6], o[3];
#pragma unroll
      for (int i = 0; i < 6; ++i)
        d[i] = f32x2v{*reinterpret_cast<const float *>(src + (i * Lr + 2 * jp) * PITCH),
                      *reinterpret_cast<const float *>(src + (i * Lr + 2 * jp + 1) * PITCH)};
      wn_bt3<HALF, f32x2v>(d, o);
#pragma unroll
      for (int r = 0; r < 3; ++r) tm[r][2 * jp] = o[r][0], tm[r][2 * jp + 1] = o[r][1];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float o[6];
      wn_bt_pk(tm[r], o);
      float2 *d2 = reinterpret_cast<float2 *>(dst + r * 6);   // 8-byte aligned: 144-byte records, halves at +72, rows at +24
      d2[0] = make_float2(o[0], o[1]);
      d2[1] = make_float2(o[2], o[3]);
      d2[2] = make_float2(o[4], o[5]);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // multiply: the step's k steps (4 tiles each); this lane's B operand = Zh of tile 4*ks + (lane >> 4), channel 16*wave + (lane & 15)
  const int kq = lane >> 4, n = wave * 16 + (lane & 15);
  auto load_dy = [&](const WwUnit &un, int h, int ks, float (&dy)[M][M]) {
    const int tile = h * 16 + 4 * ks + kq;
    const bool live = tile < un.nt;
    int tr, tcol;
    tile_rc(live ? tile : 0, tr, tcol);
    const int xo0 = M * (un.tx0 + tcol);
    const float *zp = Z + un.b * z_bs + (z_lead + (int64_t)(M * (un.ty + tr)) * Wp + xo0) * kFcHidden + n;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j < M; ++j) {
        dy[i][j] = (DBG & 4) ? 1.f : zp[(int64_t)(i * Wp + j) * kFcHidden];   // RAW: masked at use (mask_dy)
      }
  };
  // The masks are applied where the values are consumed, not where they are loaded: a select right behind the load made
  // every load_dy wait for its own round trip -- the "two k steps ahead" never happened (80 of the kernel's 385 us).
  auto mask_dy = [&](const WwUnit &un, int h, int ks, float (&dy)[M][M]) {
    const int tile = h * 16 + 4 * ks + kq;
    const bool live = tile < un.nt;
    int tr, tcol;
    tile_rc(live ? tile : 0, tr, tcol);
    const int xo0 = M * (un.tx0 + tcol);
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j < M; ++j) {
        // columns Wo .. Wp-1 and the rows behind Ho are zero in the Z layout, but a partial tile of a 4 x 4 tiling can reach
        // column Wp = the next row's first output: masked
        dy[i][j] = (live && (M == 2 || xo0 + j < Wo)) ? dy[i][j] : 0.f;
      }
  };
  auto multiply = [&](const WwUnit &un, int h, int vb) {
    const int nks = min(4, (un.nt - h * 16 + 3) >> 2);
    const float *va = vbuf + vb * kWwVFloats + (kq * 16 + (lane & 15)) * kWnXi;
    // the dY values run TWO k steps ahead of their use (global loads: an L2 round trip is about one k step of MFMAs)
    constexpr int AHEAD = M == 2 ? 2 : 1;
    float dy[AHEAD + 1][M][M];
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
      if (a < nks) load_dy(un, h, a, dy[a]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < nks) {  // wave-uniform
        if (ks + AHEAD < nks) load_dy(un, h, ks + AHEAD, dy[(ks + AHEAD) % (AHEAD + 1)]);
        float zh[kWnXi];
        mask_dy(un, h, ks, dy[ks % (AHEAD + 1)]);
        if constexpr (DBG & 8) {
#pragma unroll
          for (int q = 0; q < kWnXi; ++q) zh[q] = dy[ks % (AHEAD + 1)][q & 1][(q >> 1) & 1];
        } else {
          ww_lift<M>(dy[ks % (AHEAD + 1)], zh);
        }
        const f32x4v *vp = reinterpret_cast<const f32x4v *>(va + ks * 4 * 16 * kWnXi);
        if constexpr (DBG & 2) {
#pragma unroll
          for (int q = 0; q < kWnXi; ++q) acc[q][0] += zh[q];
        } else {
#pragma unroll
          for (int q4 = 0; q4 < kWnXi / 4; ++q4) {
            const f32x4v a4 = vp[q4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[q4 * 4 + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], zh[q4 * 4 + e], acc[q4 * 4 + e], 0, 0, 0);
          }
        }
      }
    }
  };

  if (u0 < u1) {
    // prologue: raw of the first unit, V of its first step
    WwUnit cur = unit_of(u0);
    prefetch(cur);
    commit(cur, 0);
    __syncthreads();
    if (xh == 0) transform(Half0{}, cur, 0, 0, 0);
    else transform(Half1{}, cur, 0, 0, 0);
    __syncthreads();
    int vb = 0, rbuf = 0;
    for (int64_t u = u0; u < u1; ++u) {
      const int nh = (cur.nt + 15) >> 4;
      const bool has_next = u + 1 < u1;
      const WwUnit nxt = has_next ? unit_of(u + 1) : cur;
      for (int h = 0; h < nh; ++h) {
        const bool last_h = h + 1 == nh;
        // the transform half of this step prepares the unit's next 16 tiles, or (last step of a two-step unit) the next
        // unit's first 16 -- whose raw rows were written during the unit's first step
        const bool t_same = !(DBG & 1) && !last_h, t_next = !(DBG & 1) && last_h && has_next && nh > 1;
        const bool stage = !(DBG & 16) && h == 0 && has_next;   // the next unit's raw rows: requested / written around the transform
        if (xh == 0) {
          multiply(cur, h, vb);
          __builtin_amdgcn_sched_barrier(0);
          if (stage) prefetch(nxt);
          if (t_same) transform(Half0{}, cur, h + 1, rbuf, vb ^ 1);
          else if (t_next) transform(Half0{}, nxt, 0, rbuf ^ 1, vb ^ 1);
          if (stage) commit(nxt, rbuf ^ 1);
        } else {
          if (stage) prefetch(nxt);
          if (t_same) transform(Half1{}, cur, h + 1, rbuf, vb ^ 1);
          else if (t_next) transform(Half1{}, nxt, 0, rbuf ^ 1, vb ^ 1);
          if (stage) commit(nxt, rbuf ^ 1);
          __builtin_amdgcn_sched_barrier(0);
          multiply(cur, h, vb);
        }
        __syncthreads();
        if (last_h && has_next && nh == 1) {
          // a unit of ONE step: the next unit's raw rows were written during this very step, so its first transform runs
          // here, between two barriers (k = 3 layers and narrow maps: one exposed transform per unit)
          if constexpr (!(DBG & 1)) {
            if (xh == 0) transform(Half0{}, nxt, 0, rbuf ^ 1, vb ^ 1);
            else transform(Half1{}, nxt, 0, rbuf ^ 1, vb ^ 1);
          }
          __syncthreads();
        }
        vb ^= 1;
      }
      cur = nxt;
      rbuf ^= 1;
    }
  }

  // C/D layout of the 16x16 MFMA: column (hidden channel) = lane & 15, row (input channel) = 4*(lane >> 4) + r: a lane holds
  // ALL 36 points of its four (c, n) pairs, so it applies dW = G^T dU G itself and the split's partial leaves as k*k values
  // per pair instead of 36 -- in the direct kernel's [split][tap][c][n] layout, which fc_wgrad_reduce sums straight into
  // conv0.weight.grad (k = 3: a quarter of the partial traffic, k = 5: 70 %, and no separate transform pass; the transform
  // is linear, so doing it per split changes rounding only).
  // G (6 x k): G[a][i] = p_a^i / f_a for a < 5, G[5][k-1] = 1 (wn_g)
  constexpr float inv_f[5] = {1.f, -1.f / 3.f, 1.f / 3.f, 1.f / 15.f, -16.f / 15.f};
  constexpr float pt[5] = {0.f, 1.f, -1.f, 2.f, -0.5f};
  float G[6][KS];
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    float pw = 1.f;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      G[a][i] = pw * inv_f[a];
      pw *= pt[a];
    }
  }
#pragma unroll
  for (int i = 0; i < KS; ++i) G[5][i] = i == KS - 1 ? 1.f : 0.f;
  float *o = part + (((int64_t)sp * KS * KS) * cpad + cc * kFcChunk) * kFcHidden + wave * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float tmp[KS][6];
#pragma unroll
    for (int i = 0; i < KS; ++i)
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        float sum = 0.f;
#pragma unroll
        for (int a = 0; a < 6; ++a) sum += G[a][i] * acc[a * 6 + e][r];
        tmp[i][e] = sum;
      }
#pragma unroll
    for (int i = 0; i < KS; ++i)
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 6; ++e) sum += tmp[i][e] * G[e][j];
        o[((int64_t)(i * KS + j) * cpad + 4 * kq + r) * kFcHidden] = sum;
      }
  }
}


// =====================================================================================================================
// The k = 5 weight gradient with TWO-TERM f16 OPERANDS (arithmetic mode 5, round 6).
//
// fc_wino_wgrad_kernel spends 4 608 of a step's cycles per wave in v_mfma_f32_16x16x4_f32 (36 points x 4 k steps x 32 cycles),
// and the float32 matrix instructions run at the f32 VECTOR rate: 0.54-0.56 of that pipe is all the kernel ever reached.  Same
// formulation, units, staging and epilogue here, but both operands of the 36 point-wise products are split into two f16 terms
// (fc_wino16.hip: hi = RN16(v s), lo = RN16(v s - hi), s a power of two from the tensor's max |x|) and a step's 16 tiles are ONE
// K = 32 reduction of v_mfma_f32_16x16x32_f16 -- K slots of a lane = its four tiles as (hi, hi, lo, lo | hi, hi, lo, lo) --
// issued twice per point: against the lifted gradient's words as they are (hi hi + lo lo) and with the words of each pair
// exchanged, which is a RENAMING of registers (hi lo + lo hi): 72 MFMAs of 16 cycles per step instead of 144 of 32.
//   A = V[point][tile quad][c][(hi, hi, lo, lo) x 2] from LDS: the transform threads store their two halves of a value as two
//       16-bit words (a lane = one (tile, channel, half of the points) item as before, but lanes now run over the four tiles of a
//       quad first: 16-byte records fill up from four lanes);
//   B = Zh: a lane lifts the dY values of ITS FOUR tiles (tile quad = lane >> 4) of a step, one point row at a time, and splits
//       PAIRS of tiles: v_cvt_pk_f16_f32 (both hi), two v_fma_mix_f32 (the exact remainders, hi read as f16 from either half),
//       v_cvt_pk_f16_f32 (both lo) -- two instructions per value, no half swaps;
//   dY of the NEXT step (16 floats per lane) is requested while this step multiplies.
// Error: the same as the convolutions' (every product is formed from all four cross terms with f32 accumulation inside the
// MFMA); the accumulation over the tiles is f32 in both kernels.
template <int V_>
struct PgTag6 { static constexpr int value = V_; };
constexpr int kWw16Pitch = 80;      // LDS bytes per raw pixel: the four tiles of a quad are 2 pixels = 40 words = 8 banks apart
constexpr int kWn16HeadZ = 4;       // A dY A^T grows a gradient by at most 9
constexpr int kWw16VBytes = kWnXi * 4 * 16 * 16;   // one V buffer: [point][tile quad][channel][16 bytes]

__device__ __forceinline__ void wn16_split_halves(float v, _Float16 &h, _Float16 &l) {
  asm("" : "+v"(v));   // (opaque: see wn16_split)
  h = (_Float16)v;
  float rem;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(rem) : "v"(v), "v"(h));
  l = (_Float16)rem;
}
// (hi0, hi1) and (lo0, lo1) words of two values (fc_gemm.h)
__device__ __forceinline__ void wn16_split_pair(float v0, float v1, uint32_t &hi, uint32_t &lo) { fc_split_pair(v0, v1, hi, lo); }
// row `A6` of A (6 x 2) applied to (a, b): the lift of one output-gradient pair to a point
template <int A6>
__device__ __forceinline__ float ww_lift2(float a, float b) {
  if constexpr (A6 == 0) return a;
  else if constexpr (A6 == 1) return a + b;
  else if constexpr (A6 == 2) return a - b;
  else if constexpr (A6 == 3) return fmaf(2.f, b, a);
  else if constexpr (A6 == 4) return fmaf(-0.5f, b, a);
  else return b;
}

template <int A6>
__device__ __forceinline__ f32x2v ww_lift2v(f32x2v a, f32x2v b) {
  if constexpr (A6 == 0) return a;
  else if constexpr (A6 == 1) return a + b;
  else if constexpr (A6 == 2) return a - b;
  else if constexpr (A6 == 3) return __builtin_elementwise_fma(f32x2v{2.f, 2.f}, b, a);
  else if constexpr (A6 == 4) return __builtin_elementwise_fma(f32x2v{-0.5f, -0.5f}, b, a);
  else return b;
}

// DBG (timing ablations, `make PROBES=1` builds only, tuning key 20 = 64 + bits; results are garbage): 1 no input transform,
// 2 no MFMAs / A reads, 4 no lift / split of dY (constant B words), 8 no dY loads
template <bool MR, int DBG = 0>
__global__ __launch_bounds__(kWnThreads, 2) void fc_wino16_wgrad_kernel(WwKArgs a0, WwKArgs a1, int nsplit0, int cpad,
                                                                       int raw_stride, const uint32_t *__restrict__ amax_x0,
                                                                       const uint32_t *__restrict__ amax_x1,
                                                                       const uint32_t *__restrict__ amax_z0,
                                                                       const uint32_t *__restrict__ amax_z1) {
  constexpr int KS = 5;
  const bool second = (int)blockIdx.y >= nsplit0;
#define GFLA_PICK(f) (second ? a1.f : a0.f)
  PackedDesc X;
  X.base = GFLA_PICK(X.base), X.split_stride = 0, X.batch_stride = GFLA_PICK(X.batch_stride);
  X.chunk_stride = GFLA_PICK(X.chunk_stride), X.pix_stride = GFLA_PICK(X.pix_stride);
  const float *__restrict__ Z = GFLA_PICK(Z);
  float *__restrict__ part = GFLA_PICK(part);
  const int64_t z_bs = GFLA_PICK(z_bs), z_lead = GFLA_PICK(z_lead), total_units = GFLA_PICK(total_units), SX = GFLA_PICK(SX);
  const int Wp = GFLA_PICK(Wp), nsplit = GFLA_PICK(nsplit);
  WwGeo geo;
  geo.TH = GFLA_PICK(geo.TH), geo.TW = GFLA_PICK(geo.TW), geo.nseg = GFLA_PICK(geo.nseg);
  geo.R = GFLA_PICK(geo.R), geo.ups = GFLA_PICK(geo.ups);
#undef GFLA_PICK
  const int ex = wn16_scale_exp(second ? *amax_x1 : *amax_x0, kWn16HeadX), ez = wn16_scale_exp(second ? *amax_z1 : *amax_z0, kWn16HeadZ);
  const float sx = wn16_pow2(ex), sz = wn16_pow2(ez), inv_x = wn16_pow2(254 - ex), inv_z = wn16_pow2(254 - ez);
  constexpr int M = 2, SEG = Ww<KS>::SEG, L = Ww<KS>::L, PITCH = kWw16Pitch;
  constexpr int RAW1 = (6 * L * PITCH + 15) & ~15;
  const int RAW = MR ? raw_stride : RAW1;   // bytes between the two raw buffers
  constexpr int PF1 = (6 * L * 4 + kWnThreads - 1) / kWnThreads;
  constexpr int PF = MR ? (PF1 > Ww<KS>::PFM ? PF1 : Ww<KS>::PFM) : PF1;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  unsigned char *vbuf = gfla_smem;                                  // [2][kWw16VBytes]
  unsigned char *raw = gfla_smem + 2 * kWw16VBytes;                 // [2][rows][L][PITCH] float32, scaled
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, xh = wave >> 2;
  const int cc = blockIdx.x, sp = (int)blockIdx.y - (second ? nsplit0 : 0);
  const int64_t u0 = total_units * sp / nsplit, u1 = total_units * (sp + 1) / nsplit;
  const int per_sample = geo.ups;
  const int Lr = (MR && geo.R > 1) ? M * geo.TW + 6 - M : L;
  const int raw_rows = (MR && geo.R > 1) ? M * geo.R + 6 - M : 6;
  const int npieces = raw_rows * Lr * 4;
  const unsigned inv_ntx = (65536u + (unsigned)geo.TW - 1u) / (unsigned)geo.TW;

  f32x4v acc[kWnXi];
#pragma unroll
  for (int q = 0; q < kWnXi; ++q) acc[q] = f32x4v{0.f, 0.f, 0.f, 0.f};

  auto unit_of = [&](int64_t u) {
    WwUnit un;
    un.b = u / per_sample;
    const int r = (int)(u - un.b * per_sample);
    if (MR && geo.R > 1) {
      un.ty = r * geo.R;
      un.tx0 = 0;
      un.ntx = geo.TW;
      un.nt = geo.TW * min(geo.R, geo.TH - un.ty);
    } else {
      un.ty = r / geo.nseg;
      un.tx0 = (r - un.ty * geo.nseg) * SEG;
      un.ntx = min(SEG, geo.TW - un.tx0);
      un.nt = un.ntx;
    }
    return un;
  };
  auto tile_rc = [&](int t_, int &tr, int &tcol) {
    if (MR && geo.R > 1) {
      tr = (int)(((unsigned)t_ * inv_ntx) >> 16);
      tcol = t_ - tr * geo.TW;
    } else {
      tr = 0;
      tcol = t_;
    }
  };

  // raw rows of a unit, scaled: global -> registers -> LDS
  u32x4v pf[PF];
  auto piece_addr = [&](const WwUnit &un, int q, int &ldso) -> const unsigned char * {
    const int row = q / (4 * Lr), rem = q - row * (4 * Lr), px = rem >> 2, prt = rem & 3;
    ldso = (row * Lr + px) * PITCH + prt * 16;
    const int64_t pix = (int64_t)(M * un.ty + row) * Wp + M * un.tx0 + px;
    return X.base + un.b * X.batch_stride + (int64_t)cc * X.chunk_stride + (pix < SX ? pix : SX - 1) * X.pix_stride + prt * 16;
  };
  auto prefetch = [&](const WwUnit &un) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      int ldso;
      pf[i] = *reinterpret_cast<const u32x4v *>(piece_addr(un, min(t + kWnThreads * i, npieces - 1), ldso));
    }
  };
  auto commit = [&](const WwUnit &un, int buf) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int q = min(t + kWnThreads * i, npieces - 1);
      int ldso;
      (void)piece_addr(un, q, ldso);
      float2 *d = reinterpret_cast<float2 *>(raw + buf * RAW + ldso);
      d[0] = make_float2(__uint_as_float(pf[i][0]) * sx, __uint_as_float(pf[i][1]) * sx);
      d[1] = make_float2(__uint_as_float(pf[i][2]) * sx, __uint_as_float(pf[i][3]) * sx);
    }
  };

  // transform item: tile 4 * (wave & 3) + (lane & 3) of the step's 16, channel lane >> 2, point rows 3*xh..
  const int tj = lane & 3, tc = lane >> 2, tkg = wave & 3;
  const int tl = 4 * tkg + tj;
  // byte offset of this item's hi half inside a (point, quad, channel) record: (hi0, hi1, lo0, lo1, hi2, hi3, lo2, lo3)
  const int vpos = (tkg * 16 + tc) * 16 + (tj >> 1) * 8 + (tj & 1) * 2;
  auto transform = [&](auto half_tag, const WwUnit &un, int h, int rbuf, int vb) {
    constexpr int HALF = decltype(half_tag)::value;
    const int tile = min(h * 16 + tl, un.nt - 1);
    int tr, tcol;
    tile_rc(tile, tr, tcol);
    const unsigned char *src = raw + rbuf * RAW + ((M * tr) * Lr + M * tcol) * PITCH + tc * 4;
    unsigned char *dst = vbuf + vb * kWw16VBytes + vpos;
    __builtin_amdgcn_s_setprio(3);
    float tm[3][6];
#pragma unroll
    for (int jp = 0; jp < 3; ++jp) {
      f32x2v d[6], o[3];
#pragma unroll
      for (int i = 0; i < 6; ++i)
        d[i] = f32x2v{*reinterpret_cast<const float *>(src + (i * Lr + 2 * jp) * PITCH),
                      *reinterpret_cast<const float *>(src + (i * Lr + 2 * jp + 1) * PITCH)};
      wn_bt3<HALF, f32x2v>(d, o);
#pragma unroll
      for (int r = 0; r < 3; ++r) tm[r][2 * jp] = o[r][0], tm[r][2 * jp + 1] = o[r][1];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      float o[6];
      wn_bt_pk(tm[r], o);
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        _Float16 hh, ll;
        wn16_split_halves(o[e], hh, ll);
        unsigned char *rec = dst + ((HALF * 3 + r) * 6 + e) * (4 * 16 * 16);
        *reinterpret_cast<_Float16 *>(rec) = hh;
        *reinterpret_cast<_Float16 *>(rec + 4) = ll;
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // multiply: this lane's B operand = Zh of the step's tiles 4 kq .. 4 kq + 3, hidden channel 16 wave + (lane & 15)
  const int kq = lane >> 4, n = wave * 16 + (lane & 15);
  float dy[4][M][M];   // raw dY values of the step about to be multiplied (masked and scaled at the top of multiply)
  auto load_dy = [&](const WwUnit &un, int h) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tile = h * 16 + 4 * kq + j;
      int tr, tcol;
      tile_rc(tile < un.nt ? tile : 0, tr, tcol);
      const float *zp = Z + un.b * z_bs + (z_lead + (int64_t)(M * (un.ty + tr)) * Wp + M * (un.tx0 + tcol)) * kFcHidden + n;
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int jj = 0; jj < M; ++jj) dy[j][i][jj] = (DBG & 8) ? 1.f : zp[(int64_t)(i * Wp + jj) * kFcHidden];
    }
  };
  typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));
  auto multiply = [&](const WwUnit &un, int h, int vb, const WwUnit &un_next, int h_next, bool any_next) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool live = h * 16 + 4 * kq + j < un.nt;
#pragma unroll
      for (int i = 0; i < M; ++i)
#pragma unroll
        for (int jj = 0; jj < M; ++jj) dy[j][i][jj] = live ? dy[j][i][jj] * sz : 0.f;
    }
    const u32x4w *va = reinterpret_cast<const u32x4w *>(vbuf + vb * kWw16VBytes) + kq * 16 + (lane & 15);
    // A words run two points ahead of their MFMAs (an LDS round trip is longer than a point's eight split instructions)
    u32x4w a_q[4];
    a_q[0] = va[0];
    a_q[1] = va[64];
    auto row = [&](auto a6_tag) {
      constexpr int A6 = decltype(a6_tag)::value;
      // the row's lift of the four tiles, as PAIRS of tiles (packed f32 adds; a pair is what one split consumes)
      f32x2v ta01, ta23, tb01, tb23;
      ta01 = ww_lift2v<A6>(f32x2v{dy[0][0][0], dy[1][0][0]}, f32x2v{dy[0][1][0], dy[1][1][0]});
      ta23 = ww_lift2v<A6>(f32x2v{dy[2][0][0], dy[3][0][0]}, f32x2v{dy[2][1][0], dy[3][1][0]});
      tb01 = ww_lift2v<A6>(f32x2v{dy[0][0][1], dy[1][0][1]}, f32x2v{dy[0][1][1], dy[1][1][1]});
      tb23 = ww_lift2v<A6>(f32x2v{dy[2][0][1], dy[3][0][1]}, f32x2v{dy[2][1][1], dy[3][1][1]});
      // two points at a time: the second product of a point depends on its first -- the other point's MFMA sits between them
      auto points = [&](auto e_tag) {
        constexpr int E = decltype(e_tag)::value;
        constexpr int q = A6 * 6 + E;
        u32x4w bw[2], bx[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (q + u + 2 < kWnXi && !(DBG & 2)) a_q[(q + u + 2) % 4] = va[(q + u + 2) * 64];
          f32x2v z01, z23;
          if (u == 0) z01 = ww_lift2v<E>(ta01, tb01), z23 = ww_lift2v<E>(ta23, tb23);
          else z01 = ww_lift2v<E + 1>(ta01, tb01), z23 = ww_lift2v<E + 1>(ta23, tb23);
          uint32_t h01, l01, h23, l23;
          if constexpr (DBG & 4) {
            h01 = __float_as_uint(dy[0][0][0]), l01 = __float_as_uint(dy[1][0][0]), h23 = __float_as_uint(dy[2][0][0]), l23 = __float_as_uint(dy[3][0][0]);
          } else {
            wn16_split_pair(z01[0], z01[1], h01, l01);
            wn16_split_pair(z23[0], z23[1], h23, l23);
          }
          const u32x2w p01 = u32x2w{h01, l01}, p23 = u32x2w{h23, l23};
          // (lo, hi) of each pair for the cross terms: one v_pk_mov_b32 per pair (the MFMA wants four consecutive registers).
          // EARLY-CLOBBER outputs + s_nop: hipcc's hazard recognizer does not look inside inline asm.  Allocated in place the
          // move landed right behind the first MFMA, which was still reading those registers (wrong sums, measured); a vector
          // write also needs wait states before an MFMA reads the register (NaNs without the s_nop, measured).
          u32x2w x01, x23;
          asm("v_pk_mov_b32 %0, %2, %2 op_sel:[1,0]\n\tv_pk_mov_b32 %1, %3, %3 op_sel:[1,0]\n\ts_nop 3"
              : "=&v"(x01), "=&v"(x23)
              : "v"(p01), "v"(p23));
          bw[u] = u32x4w{p01[0], p01[1], p23[0], p23[1]}, bx[u] = u32x4w{x01[0], x01[1], x23[0], x23[1]};
        }
        const f16x8 av0 = __builtin_bit_cast(f16x8, a_q[q % 4]), av1 = __builtin_bit_cast(f16x8, a_q[(q + 1) % 4]);
        if constexpr (DBG & 2) {
          acc[q][0] += __uint_as_float(bw[0][0] ^ bx[0][1]), acc[q + 1][0] += __uint_as_float(bw[1][2] ^ bx[1][3]);
        } else {
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av0, __builtin_bit_cast(f16x8, bw[0]), acc[q], 0, 0, 0);
          acc[q + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av1, __builtin_bit_cast(f16x8, bw[1]), acc[q + 1], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av0, __builtin_bit_cast(f16x8, bx[0]), acc[q], 0, 0, 0);
          acc[q + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av1, __builtin_bit_cast(f16x8, bx[1]), acc[q + 1], 0, 0, 0);
        }
      };
      points(PgTag6<0>{}), points(PgTag6<2>{}), points(PgTag6<4>{});
    };
    row(PgTag6<0>{}), row(PgTag6<1>{}), row(PgTag6<2>{}), row(PgTag6<3>{}), row(PgTag6<4>{}), row(PgTag6<5>{});
    // the NEXT step's dY values: requested now, into the registers this step is done with; they fly through the barrier and
    // the other half of the next step
    if (any_next) load_dy(un_next, h_next);
  };

  if (u0 < u1) {
    WwUnit cur = unit_of(u0);
    prefetch(cur);
    load_dy(cur, 0);
    commit(cur, 0);
    __syncthreads();
    if (xh == 0) transform(Half0{}, cur, 0, 0, 0);
    else transform(Half1{}, cur, 0, 0, 0);
    __syncthreads();
    int vb = 0, rbuf = 0;
    for (int64_t u = u0; u < u1; ++u) {
      const int nh = (cur.nt + 15) >> 4;
      const bool has_next = u + 1 < u1;
      const WwUnit nxt = has_next ? unit_of(u + 1) : cur;
      for (int h = 0; h < nh; ++h) {
        const bool last_h = h + 1 == nh;
        const bool t_same = !last_h, t_next = last_h && has_next && nh > 1;
        const bool stage = h == 0 && has_next;
        // the step whose dY values this step's multiply half requests
        const bool any_next = !last_h || has_next;
        const WwUnit &dn = last_h ? nxt : cur;
        const int hn = last_h ? 0 : h + 1;
        if (xh == 0) {
          multiply(cur, h, vb, dn, hn, any_next);
          __builtin_amdgcn_sched_barrier(0);
          if (stage) prefetch(nxt);
          if constexpr (!(DBG & 1)) {
            if (t_same) transform(Half0{}, cur, h + 1, rbuf, vb ^ 1);
            else if (t_next) transform(Half0{}, nxt, 0, rbuf ^ 1, vb ^ 1);
          }
          if (stage) commit(nxt, rbuf ^ 1);
        } else {
          if (stage) prefetch(nxt);
          if constexpr (!(DBG & 1)) {
            if (t_same) transform(Half1{}, cur, h + 1, rbuf, vb ^ 1);
            else if (t_next) transform(Half1{}, nxt, 0, rbuf ^ 1, vb ^ 1);
          }
          if (stage) commit(nxt, rbuf ^ 1);
          __builtin_amdgcn_sched_barrier(0);
          multiply(cur, h, vb, dn, hn, any_next);
        }
        __syncthreads();
        if (last_h && has_next && nh == 1) {
          if (xh == 0) transform(Half0{}, nxt, 0, rbuf ^ 1, vb ^ 1);
          else transform(Half1{}, nxt, 0, rbuf ^ 1, vb ^ 1);
          __syncthreads();
        }
        vb ^= 1;
      }
      cur = nxt;
      rbuf ^= 1;
    }
  }

  // epilogue: dW = G^T dU G per (c, n) pair, as in fc_wino_wgrad_kernel, times the two inverse scales
  constexpr float inv_f[5] = {1.f, -1.f / 3.f, 1.f / 3.f, 1.f / 15.f, -16.f / 15.f};
  constexpr float pt[5] = {0.f, 1.f, -1.f, 2.f, -0.5f};
  float G[6][KS];
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    float pw = 1.f;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      G[a][i] = pw * inv_f[a];
      pw *= pt[a];
    }
  }
#pragma unroll
  for (int i = 0; i < KS; ++i) G[5][i] = i == KS - 1 ? 1.f : 0.f;
  float *o = part + (((int64_t)sp * KS * KS) * cpad + cc * kFcChunk) * kFcHidden + wave * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float tmp[KS][6];
#pragma unroll
    for (int i = 0; i < KS; ++i)
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        float sum = 0.f;
#pragma unroll
        for (int a = 0; a < 6; ++a) sum += G[a][i] * acc[a * 6 + e][r];
        tmp[i][e] = sum;
      }
#pragma unroll
    for (int i = 0; i < KS; ++i)
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 6; ++e) sum += tmp[i][e] * G[e][j];
        o[((int64_t)(i * KS + j) * cpad + 4 * kq + r) * kFcHidden] = (sum * inv_x) * inv_z;
      }
  }
}

static WwGeo ww_geometry(int Ho, int Wo, int k) {
  const int m = k == 5 ? 2 : 4, seg = k == 5 ? 32 : 16;
  WwGeo g;
  g.TH = (Ho + m - 1) / m;
  g.TW = (Wo + m - 1) / m;
  g.nseg = (g.TW + seg - 1) / seg;
  // Units of R whole tile rows (the MR instantiation) wherever that needs fewer 16-tile steps per sample than one row per
  // unit: the k = 3 layer at 32x22 has 6 tiles per row (5 rows = 30 of 32), the k = 5 layer at 64x44 has 22 / 24 (one row:
  // 16 + 6..8 of 32; two rows: 44..48 of 48 -- a quarter fewer steps).  Bounds: the unit's raw rows (the map's own pitch)
  // within kWwRawMax bytes and the staging registers of the instantiation.  Tuning key 29: 1 = one row per unit (round 3),
  // 2 = at most 16 tiles per unit.
  g.R = 1;
  if (g.nseg == 1 && tuning(29) != 1) {
    const int pfm = (k == 5 ? 4 : 5) * kWnThreads;
    const int cap = tuning(29) == 2 ? 16 : 1 << 20;
    auto steps = [&](int R) { return (g.TH / R) * ((R * g.TW + 15) / 16) + (g.TH % R ? ((g.TH % R) * g.TW + 15) / 16 : 0); };
    int best = 1, best_steps = steps(1);
    for (int R = 2; R <= g.TH && R * g.TW <= cap; ++R) {
      const int px = (m * R + 6 - m) * (m * g.TW + 6 - m);
      if (px * kWwPitch > kWwRawMax || px * 4 > pfm) break;
      if (steps(R) <= best_steps) best = R, best_steps = steps(R);
    }
    g.R = best;
  }
  g.ups = g.R > 1 ? (g.TH + g.R - 1) / g.R : g.TH * g.nseg;
  return g;
}

int fc_wino_wgrad_splits(int64_t B, int Ho, int Wo, int cpad, int k) {
  const WwGeo g = ww_geometry(Ho, Wo, k);
  const int64_t units = B * g.ups;
  // 8 waves per workgroup, two per SIMD: ONE workgroup per CU; one round of 256 workgroups
  int64_t s = tuning(12) > 0 ? tuning(12) : kNumCU / (cpad / kFcChunk);
  if (s < 1) s = 1;
  return (int)(s > units ? units : s);
}

// part: fc_wino_wgrad_splits(...) * 36 * cpad * 128 floats.  X: packed f32 records; Z: the f32 (B, Sz, 128) Z-layout map.
static int ww_launch(const WwJob *jobs, int njobs, int cpad, int64_t B, int k, hipStream_t stream) {
  if (k != 3 && k != 5) return GFLA_ERR_UNSUPPORTED;
  if (B <= 0 || njobs <= 0) return GFLA_OK;
  WwKArgs a[2];
  int ns[2] = {0, 0};
  bool multirow = false;
  int raw_stride = 0;   // bytes of one raw buffer: the larger of the jobs' needs
  for (int j = 0; j < 2; ++j) {
    const WwJob &J = jobs[j < njobs ? j : 0];
    if (J.X.pix_stride != 64) return GFLA_ERR_UNSUPPORTED;
    const WwGeo g = ww_geometry(J.Ho, J.Wo, k);
    const int nsplit = fc_wino_wgrad_splits(B, J.Ho, J.Wo, cpad, k);
    a[j] = WwKArgs{J.X, J.Z, J.part, J.z_bs, J.z_lead, B * g.ups, J.SX, J.Wp, J.Wo, nsplit, g};
    if (j < njobs) {
      ns[j] = nsplit;
      multirow = multirow || g.R > 1;
      const int m = k == 5 ? 2 : 4;
      const int need = g.R > 1 ? (((m * g.R + 6 - m) * (m * g.TW + 6 - m) * kWwPitch + 15) & ~15) : (k == 5 ? Ww<5>::RAW : Ww<3>::RAW);
      if (need > raw_stride) raw_stride = need;
    }
  }
  const dim3 grid((unsigned)(cpad / kFcChunk), (unsigned)(ns[0] + ns[1]));
#define GFLA_WW(K_, D_)                                                                                                \
  {                                                                                                                    \
    const unsigned lds = (unsigned)(2 * kWwVFloats * 4 + 2 * raw_stride);                                              \
    auto kern = multirow ? fc_wino_wgrad_kernel<K_, D_, true> : fc_wino_wgrad_kernel<K_, D_, false>;                   \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    kern<<<grid, kWnThreads, lds, stream>>>(a[0], a[1], ns[0], cpad, raw_stride);                                       \
  }
  if (k == 5) {
#ifdef GFLA_PROBES  // timing ablations (tuning key 20 = 32 + bits; results are garbage): `make PROBES=1` builds only
    switch (tuning(20) >= 32 ? tuning(20) - 32 : 0) {
      case 1: GFLA_WW(5, 1) break;
      case 2: GFLA_WW(5, 2) break;
      case 4: GFLA_WW(5, 4) break;
      case 8: GFLA_WW(5, 8) break;
      case 16: GFLA_WW(5, 16) break;
      case 12: GFLA_WW(5, 12) break;
      case 29: GFLA_WW(5, 29) break;
      default: GFLA_WW(5, 0) break;
    }
#else
    GFLA_WW(5, 0)
#endif
  } else {
    GFLA_WW(3, 0)
  }
#undef GFLA_WW
  return launch_status();
}

int fc_wino_wgrad(const PackedDesc &X, const float *Z, int64_t z_bs, int64_t z_lead, float *part, int cpad, int64_t B, int Ho,
                  int Wo, int Wp, int64_t SX, int k, hipStream_t stream) {
  const WwJob job{X, Z, part, z_bs, z_lead, SX, Ho, Wo, Wp};
  return ww_launch(&job, 1, cpad, B, k, stream);
}

// two weight gradients (same B, cpad, k) in one launch
int fc_wino_wgrad_jobs(const WwJob *jobs, int njobs, int cpad, int64_t B, int k, hipStream_t stream) {
  if (njobs > 2) return GFLA_ERR_UNSUPPORTED;
  return ww_launch(jobs, njobs, cpad, B, k, stream);
}

// the k = 5 weight gradients of both halves with two-term f16 operands: fc_wino_wgrad_jobs' contract plus the max |x| slots of
// every job's activations (amax_x) and gradient map (amax_z)
int fc_wino16_wgrad_jobs(const WwJob *jobs, int njobs, int cpad, int64_t B, int k, const uint32_t *const *amax_x,
                         const uint32_t *const *amax_z, hipStream_t stream) {
  if (k != 5 || njobs > 2) return GFLA_ERR_UNSUPPORTED;
  if (B <= 0 || njobs <= 0) return GFLA_OK;
  WwKArgs a[2];
  int ns[2] = {0, 0};
  bool multirow = false;
  int raw_stride = 0;
  for (int j = 0; j < 2; ++j) {
    const int jj = j < njobs ? j : 0;
    const WwJob &J = jobs[jj];
    if (J.X.pix_stride != 64 || !amax_x[jj] || !amax_z[jj]) return GFLA_ERR_UNSUPPORTED;
    const WwGeo g = ww_geometry(J.Ho, J.Wo, k);
    const int nsplit = fc_wino_wgrad_splits(B, J.Ho, J.Wo, cpad, k);
    a[j] = WwKArgs{J.X, J.Z, J.part, J.z_bs, J.z_lead, B * g.ups, J.SX, J.Wp, J.Wo, nsplit, g};
    if (j < njobs) {
      ns[j] = nsplit;
      multirow = multirow || g.R > 1;
      const int need = g.R > 1 ? (((2 * g.R + 4) * (2 * g.TW + 4) * kWw16Pitch + 15) & ~15) : ((6 * Ww<5>::L * kWw16Pitch + 15) & ~15);
      if (need > raw_stride) raw_stride = need;
    }
  }
  const unsigned lds = (unsigned)(2 * kWw16VBytes + 2 * raw_stride);
  if (lds > kWnLdsLimit) return GFLA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)(cpad / kFcChunk), (unsigned)(ns[0] + ns[1]));
  auto kern = multirow ? fc_wino16_wgrad_kernel<true> : fc_wino16_wgrad_kernel<false>;
#ifdef GFLA_PROBES
  switch (tuning(20) >= 64 ? tuning(20) - 64 : 0) {
    case 1: kern = fc_wino16_wgrad_kernel<true, 1>; break;
    case 2: kern = fc_wino16_wgrad_kernel<true, 2>; break;
    case 3: kern = fc_wino16_wgrad_kernel<true, 3>; break;
    case 4: kern = fc_wino16_wgrad_kernel<true, 4>; break;
    case 6: kern = fc_wino16_wgrad_kernel<true, 6>; break;
    case 7: kern = fc_wino16_wgrad_kernel<true, 7>; break;
    case 8: kern = fc_wino16_wgrad_kernel<true, 8>; break;
    case 15: kern = fc_wino16_wgrad_kernel<true, 15>; break;
    default: break;
  }
#endif
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kern<<<grid, kWnThreads, lds, stream>>>(a[0], a[1], ns[0], cpad, raw_stride, amax_x[0], amax_x[njobs > 1 ? 1 : 0], amax_z[0],
                                          amax_z[njobs > 1 ? 1 : 0]);
  return launch_status();
}


// `part` holds nsplit slabs of k*k * cpad * 128 floats (the kernel's epilogue has applied G^T . G): the direct kernel's layout
int fc_wino_wgrad_reduce(float *part, int nsplit, float *grad_w0, int C, int c_off, int cpad, int k, hipStream_t stream) {
  if (k != 3 && k != 5) return GFLA_ERR_UNSUPPORTED;
  return fc_wgrad_reduce(part, nsplit, grad_w0, C, c_off, cpad, k, stream);
}

// source half (conv0 input channels C..2C-1) and target half (0..C-1) in one launch
int fc_wino_wgrad_reduce2(float *part_s, int nsplit_s, float *part_t, int nsplit_t, float *grad_w0, int C, int cpad, int k,
                          hipStream_t stream) {
  if (k != 3 && k != 5) return GFLA_ERR_UNSUPPORTED;
  return fc_wgrad_reduce2(part_s, nsplit_s, part_t, nsplit_t, grad_w0, C, cpad, k, stream);
}

}  // namespace gfla

extern "C" {
/* tools only: device buffer (workgroups x 8 waves x 6 uint64) that the DBG=16 instantiation of the Winograd kernel (tuning
 * key 20 = 16) fills with per-wave phase times in shader cycles; NULL switches it off */
int gfla_fc_wino_debug_buffer(void *buffer) {
#ifdef GFLA_PROBES
  gfla::g_wino_stamps = static_cast<unsigned long long *>(buffer);
  return GFLA_OK;
#else
  (void)buffer;   // the probe instantiations are not part of a default build (make PROBES=1)
  return GFLA_ERR_UNSUPPORTED;
#endif
}
}
