// Shared pieces of the Winograd-domain kernels (fc_wino.hip: float32 operands on v_mfma_f32_16x16x4_f32; fc_wino16.hip: the same
// domain with two-term f16 operands on v_mfma_f32_32x32x16_f16): the three transforms on the points {0, 1, -1, 2, -1/2, inf}
// and the geometry of a workgroup's tile group.  See fc_wino.hip for the formulation.
#pragma once

#include "fc_gemm.h"
#include <algorithm>

namespace gfla {

constexpr int kWnXi = 36;       // 6 x 6 points
constexpr int kWnTiles = 32;    // tiles per workgroup
constexpr int kWnN = 64;        // output channels per workgroup
constexpr int kWnVFloats = kWnXi * kWnTiles * 8;  // one V buffer: [point][tile][8 channels]
constexpr int kWnThreads = 512;
constexpr unsigned kWnLdsLimit = 160 * 1024;
constexpr int kWnPF = 5;        // 16-byte pieces of the raw span a thread holds in registers across half a step

typedef float f32x2v __attribute__((ext_vector_type(2)));

template <int KS>
struct Wn {
  static constexpr int M = KS == 5 ? 2 : 4;        // output tile edge
  // LDS bytes per raw pixel (16 channels + pad).  k = 5: tiles are 2 pixels = 40 words apart (banks 8 t + channel: two tiles
  // per bank among the 8 a wave reads).  k = 3: tiles are 4 pixels apart -- 80 words = 16 mod 32 with an 80-byte pitch (four
  // tiles per bank), 72 words = 8 mod 32 with 72 bytes -- and the smaller pitch is what lets the 32x22 layer's span fit TWO
  // raw buffers next to the V buffers (the single-buffer staging costs a barrier and an exposed copy per chunk).
  static constexpr int PITCH = KS == 5 ? 80 : 72;
};

// ---- the three transforms (points 0, 1, -1, 2, -1/2, inf) -----------------------------------------------------
// B^T (6 x 6)
__device__ __forceinline__ void wn_bt(const float (&d)[6], float (&o)[6]) {
  o[0] = d[0] + 1.5f * d[1] - 2.f * d[2] - 1.5f * d[3] + d[4];
  o[1] = -d[1] - 2.5f * d[2] - 0.5f * d[3] + d[4];
  o[2] = d[1] + 0.5f * d[2] - 2.5f * d[3] + d[4];
  o[3] = -0.5f * d[1] - d[2] + 0.5f * d[3] + d[4];
  o[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
  o[5] = d[1] + 1.5f * d[2] - 2.f * d[3] - 1.5f * d[4] + d[5];
}
// The same six outputs as three PAIRS -- (1, 2), (3, 4), (0, 5) -- of packed-f32 fma chains: coefficient pairs are scalar
// constants, the inputs are broadcast by op_sel, so a row costs ~10 v_pk_fma_f32 (+ a few moves) instead of ~22 scalar ops
__device__ __forceinline__ void wn_bt_pk(const float (&d)[6], float (&o)[6]) {
  const f32x2v s1{d[1], d[1]}, s2{d[2], d[2]}, s3{d[3], d[3]}, s4{d[4], d[4]};
  const f32x2v p12 = s4 + f32x2v{-1.f, 1.f} * s1 + f32x2v{-2.5f, 0.5f} * s2 + f32x2v{-0.5f, -2.5f} * s3;
  const f32x2v p34 = s4 + f32x2v{-0.5f, 2.f} * s1 + f32x2v{-1.f, -1.f} * s2 + f32x2v{0.5f, -2.f} * s3;
  const f32x2v p05 = f32x2v{d[0], d[5]} + f32x2v{1.5f, 1.f} * s1 + f32x2v{-2.f, 1.5f} * s2 + f32x2v{-1.5f, -2.f} * s3 +
                     f32x2v{1.f, -1.5f} * s4;
  o[0] = p05[0], o[5] = p05[1], o[1] = p12[0], o[2] = p12[1], o[3] = p34[0], o[4] = p34[1];
}
// rows 3*HALF .. 3*HALF + 2 of B^T d
template <int HALF, typename T = float>
__device__ __forceinline__ void wn_bt3(const T (&d)[6], T (&o)[3]) {
  if constexpr (HALF == 0) {
    o[0] = d[0] + 1.5f * d[1] - 2.f * d[2] - 1.5f * d[3] + d[4];
    o[1] = -d[1] - 2.5f * d[2] - 0.5f * d[3] + d[4];
    o[2] = d[1] + 0.5f * d[2] - 2.5f * d[3] + d[4];
  } else {
    o[0] = -0.5f * d[1] - d[2] + 0.5f * d[3] + d[4];
    o[1] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
    o[2] = d[1] + 1.5f * d[2] - 2.f * d[3] - 1.5f * d[4] + d[5];
  }
}
// A^T (m x 6)
template <int M>
__device__ __forceinline__ void wn_at(const float (&v)[6], float (&y)[M]) {
  y[0] = v[0] + v[1] + v[2] + v[3] + v[4];
  if constexpr (M == 2) {
    y[1] = v[1] - v[2] + 2.f * v[3] - 0.5f * v[4] + v[5];
  } else {
    y[1] = v[1] - v[2] + 2.f * v[3] - 0.5f * v[4];
    y[2] = v[1] + v[2] + 4.f * v[3] + 0.25f * v[4];
    y[3] = v[1] - v[2] + 8.f * v[3] - 0.125f * v[4] + v[5];
  }
}
// G (6 x r): G[i][j] = p_i^j / prod_{l != i} (p_i - p_l), last row = e_{r-1}
template <int KS>
__device__ __forceinline__ void wn_g(const float (&w)[KS], float (&o)[6]) {
  o[0] = w[0];
  o[5] = w[KS - 1];
  if constexpr (KS == 5) {
    o[1] = -(w[0] + w[1] + w[2] + w[3] + w[4]) * (1.f / 3.f);
    o[2] = (w[0] - w[1] + w[2] - w[3] + w[4]) * (1.f / 3.f);
    o[3] = (w[0] + 2.f * w[1] + 4.f * w[2] + 8.f * w[3] + 16.f * w[4]) * (1.f / 15.f);
    o[4] = (-16.f * w[0] + 8.f * w[1] - 4.f * w[2] + 2.f * w[3] - w[4]) * (1.f / 15.f);
  } else {
    o[1] = -(w[0] + w[1] + w[2]) * (1.f / 3.f);
    o[2] = (w[0] - w[1] + w[2]) * (1.f / 3.f);
    o[3] = (w[0] + 2.f * w[1] + 4.f * w[2]) * (1.f / 15.f);
    o[4] = (-16.f * w[0] + 8.f * w[1] - 4.f * w[2]) * (1.f / 15.f);
  }
}

// one weight set of fc_wino_pack_weights / fc_wino16_pack_weights (forward: in = conv0 input channel c_off + ci, out = hidden n;
// data gradient: in = hidden n, out = conv0 input channel c_off + co, taps flipped)
struct WnPackJob {
  float *U;
  int c_off, dgrad, n_in, n_out;
};
struct WnPackJobs {
  WnPackJob j[4];
};

struct WnGeo {
  int TH, TW, ngroups, span;  // tile grid, groups of up to 32 tiles per sample, raw pixels a group stages per chunk
  int tpg;                    // tiles per group: 32, or whole tile rows on narrow maps (below)
};

// span of the groups of `tpg` consecutive tiles: from the first pixel of a group's first tile row to the last pixel of its last
// tile's 6 x 6 window (groups start at multiples of tpg tiles, so few of them are the worst case)
template <int KS>
inline int wn_span(const WnGeo &g, int tpg, int Wp) {
  constexpr int m = Wn<KS>::M;
  const int ntiles = g.TH * g.TW, ngroups = (ntiles + tpg - 1) / tpg;
  int exact = 0;
  for (int grp = 0; grp < ngroups; ++grp) {
    const int t0 = grp * tpg, t1 = std::min(t0 + tpg, ntiles) - 1;
    const int r0 = t0 / g.TW, r1 = t1 / g.TW;
    int need = 0;
    for (int r = std::max(r0, r1 - 1); r <= r1; ++r) {   // the last pixel is the last tile's, or the previous row's last tile's
      const int c = r == r1 ? t1 - r1 * g.TW : g.TW - 1;
      need = std::max(need, (m * r + 5) * Wp + m * c + 5 + 1 - m * r0 * Wp);
    }
    exact = std::max(exact, need);
  }
  return exact;
}

template <int KS>
inline WnGeo wn_geometry(int M, int Wv, int Wp) {
  constexpr int m = Wn<KS>::M;
  WnGeo g;
  const int Ho = M / Wv;
  g.TH = (Ho + m - 1) / m;
  g.TW = (Wv + m - 1) / m;
  const int ntiles = g.TH * g.TW;
  g.tpg = kWnTiles;
  g.ngroups = (ntiles + kWnTiles - 1) / kWnTiles;
  g.span = wn_span<KS>(g, kWnTiles, Wp);
  // Narrow maps (the k = 3 layer at 32x22: 6 tiles of 4 x 4 per row): groups of WHOLE tile rows -- 30 of the 32 slots -- reach
  // one tile row less than 32 consecutive tiles that start mid-row, and that is what lets the span fit TWO raw buffers next
  // to the V buffers (610 -> 534 pixels at 32x22: 161.5 KB -> 151 KB; the single-buffer staging costs a barrier and an exposed
  // copy per chunk).  Taken when it wastes at most 4 slots and does not add a group.  Tuning key 44 = 1: always 32.
  const int whole = g.TW < kWnTiles ? (kWnTiles / g.TW) * g.TW : kWnTiles;
  if (whole != kWnTiles && whole >= kWnTiles - 4 && (ntiles + whole - 1) / whole == g.ngroups && tuning(44) != 1) {
    const int sp = wn_span<KS>(g, whole, Wp);
    if (sp < g.span) g.tpg = whole, g.span = sp;
  }
  return g;
}

template <int KS>
inline unsigned wn_raw_bytes(const WnGeo &g) { return (unsigned)((g.span * Wn<KS>::PITCH + 15) & ~15); }

// double_raw: two raw buffers (the next chunk's pixels land while this chunk is transformed: no extra barrier)
template <int KS>
inline unsigned wn_lds_bytes(const WnGeo &g, bool double_raw) {
  constexpr int m = Wn<KS>::M;
  const unsigned main_loop = (unsigned)(2 * kWnVFloats * 4) + (double_raw ? 2u : 1u) * wn_raw_bytes<KS>(g);
  const unsigned exchange = (unsigned)(kWnThreads * 4 * m * m * 4);  // epilogue: partial outputs of the wave pairs
  return main_loop > exchange ? main_loop : exchange;
}

// ---- two-term f16 operands (arithmetic mode 5: fc_wino16.hip, the f16 weight gradient in fc_wino.hip) ------------
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
// power-of-two scale of a tensor that gets split into f16 terms, with `headroom` bits left for what the transform adds:
// B^T d B grows an input by at most 49 (6 bits), G w G^T a weight by at most 4.3 (3 bits)
__host__ __device__ __forceinline__ int wn16_scale_exp(uint32_t amax_bits, int headroom) {
  int se = fc_scale_exp(amax_bits) - headroom;
  return se < 2 ? 2 : se;
}
__device__ __forceinline__ float wn16_pow2(int biased) { return __uint_as_float((uint32_t)biased << 23); }
constexpr int kWn16HeadX = 6, kWn16HeadW = 3;

// (hi, lo) word of a value in three instructions: v_cvt_f16_f32, v_fma_mix_f32 (v * 1 - hi with hi read as f16: the exact
// remainder, no convert back), v_cvt_pk_f16_f32 of (v, remainder) -- RN16(v) again in the low half, RN16(remainder) in the
// high one.  (The plain C form compiles to five: two converts, a convert back, a subtract and an or.)
typedef _Float16 f16x2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t wn16_split(float v) {
  // (v made opaque: with the multiply that produced it in sight hipcc fuses it into the convert -- v_fma_mixlo_f16 of the
  // unrounded product -- and `h` is no longer the half the packed convert below stores)
  asm("" : "+v"(v));
  const _Float16 h = (_Float16)v;
  float rem;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(rem) : "v"(v), "v"(h));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2v{v, rem}, f16x2w));
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
struct Half0 { static constexpr int value = 0; };
struct Half1 { static constexpr int value = 1; };
typedef Half1 Yes;
typedef Half0 No;

}  // namespace gfla
