// The stride-1 k x k convolutions of the FC layer as implicit GEMMs on the matrix cores, gfx950 (see fc_gemm.hip for
// the formulation).  Included by fc_conv_m{0,2,3}.hip, one translation unit per arithmetic mode.
//
// out[b][r][n] = inv_scale * sum_{chunk, tap=(i,j), c} X[b][chunk][pix(r) + i*Wp + j][c] * Wk[ntile][chunk][tap][n][c]
// with pix(r) = (r / Wv) * Wp + r % Wv: the output rows are the VALID positions only (Wv = valid width of a row of
// the linearised map); Wv = Wp makes pix the identity -- the data-gradient convolution, where every position of the
// padded domain is an output.  A lane owns one row of each 32-row MFMA block, so the row -> pixel map is one
// per-lane LDS offset per block: the k-1 wrap-around columns of the linearised map cost no MFMA work.
//
// grid (row tiles, B, channel tiles); 256 threads = 4 waves.  Wave w owns output channels 32w..32w+31 of the
// 128-channel tile and ALL 32*NMB rows of the row tile (NMB = 2..8 row blocks, a template parameter chosen per launch
// so that the last round of workgroups is full, see pick_row_blocks):
//   * A operand: the input pixels of the row tile (+ tap halo) of one 16-channel chunk, staged once per chunk in LDS
//     and reused by all k*k taps and all four waves; fragments are read one row block ahead of the MFMAs that use
//     them (an f32 MFMA occupies the pipe for 64 cycles, the LDS answers in about as many);
//   * B operand: the wave's 32 x 16 slice of the (chunk, tap) weight tile goes global -> registers directly in MFMA
//     fragment layout (a wave reads 1-2 KB contiguous), prefetched one tap ahead: no LDS, no barrier per tap.
#pragma once

#include "fc_mma.h"

namespace gfla {

constexpr int kFcMinRowBlocks = 2, kFcMaxRowBlocks = 8;
constexpr int kFcPrefetch = 8;  // 16-byte input-tile pieces per thread held in registers across a chunk (mode 0)

template <int MODE>
__device__ __forceinline__ void load_b_frags(Frag<MODE> (&fb)[Fc<MODE>::KB], const unsigned char *p, int64_t split_stride) {
  if constexpr (MODE == 0) {
    fb[0].v = *reinterpret_cast<const float4 *>(p);
    fb[1].v = *reinterpret_cast<const float4 *>(p + 32);
  } else {
#pragma unroll
    for (int sp = 0; sp < Fc<MODE>::NS; ++sp) fb[0].s[sp] = *reinterpret_cast<const f16x8 *>(p + sp * split_stride);
  }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // a native vector: stays in registers as an array element

// Mode 0 (one term, 4 pieces per 64-byte pixel record): thread t owns piece t & 3 of pixels (t >> 2) + 64*q, so pass q of
// the input-tile staging is one base address + q times a uniform stride on both sides.
__device__ __forceinline__ void fc_prefetch_pieces(u32x4 (&pf)[kFcPrefetch], const unsigned char *src, int x_ps, int pix0,
                                                   int tmh) {
#pragma unroll
  for (int q = 0; q < kFcPrefetch; ++q)  // pixel clamped: the load is always legal, the LDS store is predicated
    pf[q] = *reinterpret_cast<const u32x4 *>(src + (int64_t)min(pix0 + 64 * q, tmh - 1) * x_ps);
}

// Row tiling of one launch.  Every sample's ceil(M / 32) row blocks are cut into T tiles: `rem` "heavy" ones of
// base + 1 blocks and T - rem "light" ones of base blocks, so that B * T workgroups fill the 2 x 256 workgroup slots
// of the chip in ONE round with at most one block of imbalance.  Heavy tiles take the low workgroup ids; when the
// launch fits one round, ids >= 256 are mirrored (id -> N - 1 - (id - 256)) so that the workgroups a CU gets in the
// first and second dispatch wave (ids w and w + 256 land on the same CU in practice) are a large and a small one.
// (Placement is an optimisation only: any dispatch order computes the same result.)
struct ConvTiling {
  int T, base, rem, heavy, per_ntile;  // heavy = B * rem tiles of base + 1 blocks; per_ntile = B * T
};

// SRC32 (mode 2 only): the input map is FLOAT32 -- packed 64-byte records or a (B, S, C) map in place, mode 0's descriptor --
// and becomes the two f16 term planes when a chunk's pixels are written to LDS (scale from the tensor's max |x| slot, pair
// splits of fc_gemm.h: 12 vector instructions per 16-byte piece, once per chunk against k*k taps of MFMAs); the pieces are held
// in registers across the previous chunk like mode 0's.  This is what lets arithmetic mode 5 run its k = 5 convolutions on the
// direct kernels without a second packed copy of anything (round 6, fc_block.hip).
template <int MODE, int KS, int NMB, bool SRC32 = false>
__global__ __launch_bounds__(256, 2) void fc_conv_kernel(PackedDesc X, const unsigned char *__restrict__ Wk,
                                                        int64_t w_split_stride, float *__restrict__ out,
                                                        int64_t out_bs, int ldo, int n_valid, int M, int Wv, int Wp,
                                                        int nch, int tmh, ConvTiling tl,
                                                        const uint32_t *__restrict__ amax_x,
                                                        const uint32_t *__restrict__ amax_w) {
  using F = Fc<MODE>;
  constexpr int KK = KS * KS, PITCH = F::PITCH, STEPS = F::KB * NMB;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  unsigned char *xs = gfla_smem;  // [NS][tmh][PITCH]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, kh = lane >> 5;
  // workgroup -> (channel tile, sample, first row block, row blocks): see ConvTiling
  int wid = blockIdx.x;
  const int ntile = wid / tl.per_ntile;
  wid -= ntile * tl.per_ntile;
  if (tl.per_ntile <= 2 * kNumCU && wid >= kNumCU) wid = tl.per_ntile - 1 - (wid - kNumCU);  // pair large with small
  else if (tl.per_ntile <= kNumCU && (ntile & 1)) wid = tl.per_ntile - 1 - wid;            // (across channel tiles too)
  int64_t b;
  int first, nblk;
  if (wid < tl.heavy) {
    b = wid / tl.rem;
    first = (wid - (int)b * tl.rem) * (tl.base + 1);
    nblk = tl.base + 1;
  } else {
    const int u = wid - tl.heavy, per = tl.T - tl.rem;
    b = u / per;
    first = tl.rem * (tl.base + 1) + (u - (int)b * per) * tl.base;
    nblk = tl.base;
  }
  const int m0 = first * 32;
  const int y0 = m0 / Wv, p0 = y0 * Wp + (m0 - y0 * Wv);  // first input pixel of the row tile
  const int64_t x_ss = X.split_stride, x_cs = X.chunk_stride;
  const int x_ps = X.pix_stride;
  const unsigned char *xg = X.base + b * X.batch_stride + (int64_t)p0 * x_ps;
  constexpr int64_t kWTile = (int64_t)kFcTN * F::REC;  // bytes of one (chunk, tap) weight tile of one term
  // this lane's slice of a weight tile: row n = 32*wave + l31, K half kh
  const unsigned char *wg = Wk + (int64_t)ntile * nch * KK * kWTile + (wave * 32 + l31) * F::REC + kh * 16;

  int a_off[NMB];  // LDS byte offset of this lane's row in each 32-row block
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb) {
    const int r = min(m0 + mb * 32 + l31, M - 1);
    const int y = r / Wv;
    a_off[mb] = (y * Wp + (r - y * Wv) - p0) * PITCH;
  }

  f32x16 acc[NMB];
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

  static_assert(!SRC32 || MODE == 2, "float32 source: mode 2 arithmetic only");
  const int per = tmh * (SRC32 ? 4 : F::PIECES);  // 16-byte pieces of one term of the input tile (SRC32: of the float32 tile)
  const int x_total = SRC32 ? per : per * F::NS;
  const int xplane = tmh * PITCH;

  // Input tile staging.  Mode 0: the first 64*kFcPrefetch pixels of a chunk's tile are loaded into registers while
  // the PREVIOUS chunk is being multiplied and only written to LDS at the chunk boundary, so the boundary costs two
  // barriers and the LDS writes, not a global-memory round trip.  (Modes 2/3, and tile pixels beyond the register
  // budget: loaded at the boundary.)
  constexpr int PF = kFcPrefetch;
  constexpr bool kPrefetch = MODE == 0 || SRC32;
  const float sx32 = SRC32 ? fc_scale(amax_x) : 1.f;
  // SRC32: a float32 piece (4 channels of a pixel) -> 8 bytes in each term plane
  auto store_split = [&](unsigned char *dst8, u32x4 v) {
    uint32_t h01, l01, h23, l23;
    fc_split_pair(__uint_as_float(v[0]) * sx32, __uint_as_float(v[1]) * sx32, h01, l01);
    fc_split_pair(__uint_as_float(v[2]) * sx32, __uint_as_float(v[3]) * sx32, h23, l23);
    *reinterpret_cast<uint2 *>(dst8) = make_uint2(h01, h23);
    *reinterpret_cast<uint2 *>(dst8 + xplane) = make_uint2(l01, l23);
  };
  u32x4 pf[PF];
  const int pix0 = t >> 2;
  const unsigned char *pf_src = xg + (t & 3) * 16;
  unsigned char *pf_dst = xs + (size_t)pix0 * PITCH + (t & 3) * (SRC32 ? 8 : 16);
  const int x_first = kPrefetch ? min(x_total, 256 * PF) : 0;  // pieces covered by the register path

  Frag<MODE> cur[F::KB], nxt[F::KB];
  load_b_frags<MODE>(cur, wg, w_split_stride);
  if constexpr (kPrefetch) fc_prefetch_pieces(pf, pf_src, x_ps, pix0, tmh);
  for (int cc = 0; cc < nch; ++cc) {
    __syncthreads();  // every wave is done with the previous chunk's tile
    if constexpr (kPrefetch) {
#pragma unroll
      for (int q = 0; q < PF; ++q)
        if (pix0 + 64 * q < tmh) {
          if constexpr (SRC32) store_split(pf_dst + (size_t)q * 64 * PITCH, pf[q]);
          else *reinterpret_cast<u32x4 *>(pf_dst + (size_t)q * 64 * PITCH) = pf[q];
        }
    }
    if constexpr (SRC32) {   // tile pixels beyond the register budget: loaded at the boundary
      const unsigned char *src = xg + (int64_t)cc * x_cs + (t & 3) * 16;
      for (int pix = pix0 + 64 * PF; pix < tmh; pix += 64)
        store_split(xs + (size_t)pix * PITCH + (t & 3) * 8, *reinterpret_cast<const u32x4 *>(src + (int64_t)pix * x_ps));
    } else if (x_first < x_total) {
      const unsigned char *src = xg + (int64_t)cc * x_cs;
      for (int base = x_first + t; base < x_total; base += 256 * 4) {
        uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0, v2 = v0, v3 = v0;
#define GFLA_X_SP(idx_) (((idx_) >= per) + ((idx_) >= 2 * per))
#define GFLA_X_REM(idx_) ((idx_)-GFLA_X_SP(idx_) * per)
#define GFLA_X_SRC(idx_) \
  (src + GFLA_X_SP(idx_) * x_ss + (int64_t)(GFLA_X_REM(idx_) / F::PIECES) * x_ps + (GFLA_X_REM(idx_) % F::PIECES) * 16)
#define GFLA_X_DST(idx_) \
  (xs + ((size_t)GFLA_X_SP(idx_) * tmh + GFLA_X_REM(idx_) / F::PIECES) * PITCH + (GFLA_X_REM(idx_) % F::PIECES) * 16)
        if (base < x_total) v0 = *reinterpret_cast<const uint4 *>(GFLA_X_SRC(base));
        if (base + 256 < x_total) v1 = *reinterpret_cast<const uint4 *>(GFLA_X_SRC(base + 256));
        if (base + 512 < x_total) v2 = *reinterpret_cast<const uint4 *>(GFLA_X_SRC(base + 512));
        if (base + 768 < x_total) v3 = *reinterpret_cast<const uint4 *>(GFLA_X_SRC(base + 768));
        if (base < x_total) *reinterpret_cast<uint4 *>(GFLA_X_DST(base)) = v0;
        if (base + 256 < x_total) *reinterpret_cast<uint4 *>(GFLA_X_DST(base + 256)) = v1;
        if (base + 512 < x_total) *reinterpret_cast<uint4 *>(GFLA_X_DST(base + 512)) = v2;
        if (base + 768 < x_total) *reinterpret_cast<uint4 *>(GFLA_X_DST(base + 768)) = v3;
#undef GFLA_X_SRC
#undef GFLA_X_DST
#undef GFLA_X_SP
#undef GFLA_X_REM
      }
    }
    __syncthreads();
    Frag<MODE> fa = load_frag<MODE>(xs + a_off[0], xplane, 0, kh);  // first fragment of tap 0
#pragma unroll 1
    for (int tap = 0; tap < KK; ++tap) {
      {  // next (chunk, tap) weight slice: in flight during this tap's MFMAs
        const int lin = cc * KK + tap + 1;
#pragma unroll
        for (int q = 0; q < F::KB; ++q) nxt[q] = cur[q];
        if (lin < nch * KK) load_b_frags<MODE>(nxt, wg + (int64_t)lin * kWTile, w_split_stride);
        if constexpr (kPrefetch)
          if (tap == 0 && cc + 1 < nch)  // behind the weight slice: its wait leaves these in flight
            fc_prefetch_pieces(pf, pf_src + (int64_t)(cc + 1) * x_cs, x_ps, pix0, tmh);
      }
      const int i = tap / KS, j = tap - i * KS;
      const unsigned char *xa = xs + (size_t)(i * Wp + j) * PITCH;
      const int tn = tap + 1 < KK ? tap + 1 : 0;  // after the last tap: a harmless re-read, the next chunk starts over
      const int in = tn / KS, jn = tn - in * KS;
      const unsigned char *xa_next = xs + (size_t)(in * Wp + jn) * PITCH;
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        const int kb = st / NMB, mb = st % NMB;
        const Frag<MODE> fc = fa;
        if (st + 1 < STEPS)
          fa = load_frag<MODE>(xa + a_off[(st + 1) % NMB], xplane, (st + 1) / NMB, kh);
        else
          fa = load_frag<MODE>(xa_next + a_off[0], xplane, 0, kh);
        __builtin_amdgcn_sched_barrier(0);  // keep the LDS read ahead of the MFMAs it overlaps with
        if (mb + 1 < NMB || nblk == NMB)    // a "light" tile has one row block fewer (wave-uniform)
          acc[mb] = mma<MODE>(fc, cur[kb], acc[mb]);
      }
#pragma unroll
      for (int q = 0; q < F::KB; ++q) cur[q] = nxt[q];
    }
  }

  // C/D layout of the 32x32 MFMAs: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const float inv = MODE == 0 ? 1.f : fc_inv_scale(amax_x) * fc_inv_scale(amax_w);
  float *ob = out + b * out_bs;
  const int col = ntile * kFcTN + wave * 32 + l31;
  if (col < n_valid) {
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
      if (mb >= nblk) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (m < M) ob[(int64_t)m * ldo + col] = acc[mb][r] * inv;
      }
    }
  }
}



// pixels of the LDS input tile for 32*mb output rows: the rows' own span + the tap halo
inline int fc_conv_tile_pixels(int mb, int Wv, int Wp, int k) {
  const int tm = 32 * mb;
  return tm + ((tm - 1) / Wv + 1) * (Wp - Wv) + (k - 1) * (Wp + 1);
}

// Tiles per sample: as many as fill the chip's workgroup slots once (2 per CU when two tiles' LDS fit), never more
// than kFcMaxRowBlocks row blocks per tile.  Returns the template's NMB (= the heavy tile's blocks), 0 = does not fit.
inline int pick_conv_tiling(int M, int64_t B, int ntiles_n, int Wv, int Wp, int k, int mode, ConvTiling *tl) {
  const int ns = fc_nsplit(mode), pitch = mode ? 48 : 80;
  const int nb = (int)ceil_div(M, 32);
  const int forced = tuning(11);
  int64_t T = (2 * kNumCU) / (B * ntiles_n);
  if (T < 1) T = 1;
  if (T > nb) T = nb;
  if (ceil_div(nb, T) > kFcMaxRowBlocks) T = ceil_div(nb, kFcMaxRowBlocks);
  if (forced >= kFcMinRowBlocks && forced <= kFcMaxRowBlocks) T = ceil_div(nb, forced);
  // LDS: shrink tiles until one fits; two per CU preferred
  for (;; ++T) {
    const int nmb = (int)ceil_div(nb, T);
    const int64_t lds = (int64_t)ns * fc_conv_tile_pixels(nmb < kFcMinRowBlocks ? kFcMinRowBlocks : nmb, Wv, Wp, k) * pitch;
    if (lds <= 156 * 1024) break;
    if (nmb <= kFcMinRowBlocks) return 0;
  }
  tl->T = (int)T;
  tl->base = nb / (int)T;
  tl->rem = nb % (int)T;
  tl->heavy = (int)B * tl->rem;
  tl->per_ntile = (int)(B * T);
  int nmb = tl->base + (tl->rem ? 1 : 0);
  if (nmb < kFcMinRowBlocks) {  // tiny problems: the smallest instantiation, its spare blocks masked off
    nmb = kFcMinRowBlocks;
  }
  return nmb;
}

template <int MODE, int KS, int NMB, bool SRC32 = false>
static int launch_conv(const PackedDesc &X, const void *wk, int64_t w_split_stride, float *out, int64_t out_bs,
                       int ldo, int n_valid, int64_t B, int nch, int M, int Wv, int Wp, const ConvTiling &tl,
                       const uint32_t *amax_x, const uint32_t *amax_w, hipStream_t stream) {
  using F = Fc<MODE>;
  const int tmh = fc_conv_tile_pixels(NMB, Wv, Wp, KS);
  const unsigned lds = (unsigned)(F::NS * tmh * F::PITCH);
  const int64_t wgs = (int64_t)tl.per_ntile * ceil_div(n_valid, kFcTN);
  if (wgs > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  auto kern = fc_conv_kernel<MODE, KS, NMB, SRC32>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kern<<<dim3((unsigned)wgs), 256, lds, stream>>>(X, static_cast<const unsigned char *>(wk), w_split_stride, out, out_bs,
                                                  ldo, n_valid, M, Wv, Wp, nch, tmh, tl, amax_x, amax_w);
  return launch_status();
}

template <int MODE, int KS, bool SRC32 = false>
static int dispatch_conv(int nmb, const PackedDesc &X, const void *wk, int64_t wss, float *out, int64_t out_bs, int ldo,
                         int n_valid, int64_t B, int nch, int M, int Wv, int Wp, const ConvTiling &tl, const uint32_t *ax,
                         const uint32_t *aw, hipStream_t s) {
  switch (nmb) {
#define GFLA_CASE(N_) \
  case N_: return launch_conv<MODE, KS, N_, SRC32>(X, wk, wss, out, out_bs, ldo, n_valid, B, nch, M, Wv, Wp, tl, ax, aw, s)
    GFLA_CASE(2);
    GFLA_CASE(3);
    GFLA_CASE(4);
    GFLA_CASE(5);
    GFLA_CASE(6);
    GFLA_CASE(7);
    GFLA_CASE(8);
#undef GFLA_CASE
  }
  return GFLA_ERR_UNSUPPORTED;
}

// one arithmetic mode, both kernel sizes (defined in fc_conv_m<MODE>.hip)
template <int MODE>
int fc_conv_mode(const PackedDesc &X, const void *wk, int64_t w_split_stride, float *out, int64_t out_bs, int ldo,
                 int n_valid, int64_t B, int nch, int M, int Wv, int Wp, int k, const uint32_t *amax_x,
                 const uint32_t *amax_w, hipStream_t stream);

#define GFLA_DEFINE_FC_CONV_MODE(MODE_)                                                                               \
  template <>                                                                                                         \
  int fc_conv_mode<MODE_>(const PackedDesc &X, const void *wk, int64_t w_split_stride, float *out, int64_t out_bs,    \
                          int ldo, int n_valid, int64_t B, int nch, int M, int Wv, int Wp, int k,                     \
                          const uint32_t *amax_x, const uint32_t *amax_w, hipStream_t stream) {                       \
    ConvTiling tl;                                                                                                    \
    const int nmb = pick_conv_tiling(M, B, (int)ceil_div(n_valid, kFcTN), Wv, Wp, k, MODE_, &tl);                      \
    if (nmb == 0) return GFLA_ERR_UNSUPPORTED;                                                                        \
    if (k == 3)                                                                                                       \
      return dispatch_conv<MODE_, 3>(nmb, X, wk, w_split_stride, out, out_bs, ldo, n_valid, B, nch, M, Wv, Wp, tl,     \
                                     amax_x, amax_w, stream);                                                         \
    return dispatch_conv<MODE_, 5>(nmb, X, wk, w_split_stride, out, out_bs, ldo, n_valid, B, nch, M, Wv, Wp, tl,       \
                                   amax_x, amax_w, stream);                                                           \
  }

}  // namespace gfla
