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
//   workgroup = 32 tiles (two 16-row MFMA blocks) x 64 output channels x ALL 36 points; 4 waves, wave w = output
//   channels 16w..16w+15, 36 x 2 accumulators of 16x16 (288 registers; one wave per SIMD, 512-register budget);
//   per 8-channel step:  A = transformed input V[point][tile][8 ch] from LDS (double buffered: the NEXT step's
//   transform -- 256 (tile, channel) items, one per thread, B^T d B in registers -- runs on the vector ALUs while
//   this step's 144 MFMAs run on the matrix cores), B = the wave's slice of U = G w G^T, global -> registers directly
//   in fragment layout, each register reloaded for the next step right after its last use;
//   raw input pixels of a 16-channel chunk: one contiguous span of the linearised map (tap (i,j) = pixel offset
//   i*Wp + j, as in fc_conv_impl.h), prefetched into registers one step ahead, LDS pitch 80 / 72 bytes so that the
//   transform's reads (8 tiles x 8 channels per wave) are spread over all banks;
//   epilogue: A^T M A per lane in registers (a lane holds all 36 points of its (tile, channel) pairs), stores to the
//   same (pixel, channel) f32 map fc_conv writes.
#include "fc_gemm.h"

namespace gfla {

constexpr int kWnXi = 36;       // 6 x 6 points
constexpr int kWnTiles = 32;    // tiles per workgroup
constexpr int kWnN = 64;        // output channels per workgroup
constexpr int kWnVFloats = kWnXi * kWnTiles * 8;  // one V buffer: [point][tile][8 channels]
constexpr int kWnPF = 10;       // 16-byte pieces of the raw span a thread holds in registers across a step

template <int KS>
struct Wn {
  static constexpr int M = KS == 5 ? 2 : 4;        // output tile edge
  static constexpr int PITCH = KS == 5 ? 80 : 72;  // LDS bytes per raw pixel (16 channels + pad): tile stride = 8 banks
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

// ---- weights: conv0.weight (128, 2C, k, k) -> U = G w G^T in MFMA B-fragment order -------------------------------
// U[ntile][chunk][half][point][nblock][lane][2]: lane (kq = lane >> 4, n = lane & 15) holds input channels
// 16*chunk + 8*half + {kq, 4 + kq} of output channel 64*ntile + 16*nblock + n.
// forward:        in = conv0 input channel c_off + ci, out = hidden n, taps as stored;
// data gradient:  in = hidden n, out = conv0 input channel c_off + co, taps flipped (the transposed convolution).
template <int KS>
__global__ __launch_bounds__(256) void fc_wino_pack_w_kernel(const float *__restrict__ w0, float *__restrict__ U, int C,
                                                            int c_off, int dgrad, int n_in, int n_out, int nch) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (out channel, in channel)
  const int ntn = (n_out + kWnN - 1) / kWnN;
  if (idx >= (int64_t)ntn * kWnN * nch * kFcChunk) return;
  const int ci = (int)(idx % (nch * kFcChunk)), co = (int)(idx / (nch * kFcChunk));
  float w[KS][KS];
#pragma unroll
  for (int i = 0; i < KS; ++i)
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      float v = 0.f;
      if (ci < n_in && co < n_out) {
        v = dgrad ? w0[(((int64_t)ci * 2 * C + c_off + co) * KS + (KS - 1 - i)) * KS + (KS - 1 - j)]
                  : w0[(((int64_t)co * 2 * C + c_off + ci) * KS + i) * KS + j];
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
  float *dst = U + ((((((int64_t)ntile * nch + cc) * 2 + half) * kWnXi) * 4 + nb) * 64 + kq * 16 + n) * 2 + ks;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    float o[6];
    wn_g<KS>(t[a], o);
#pragma unroll
    for (int e = 0; e < 6; ++e) dst[(int64_t)(a * 6 + e) * 4 * 64 * 2] = o[e];
  }
}

int64_t fc_wino_wpack_bytes(int n_in, int n_out) {
  return (int64_t)ceil_div(n_out, kWnN) * ceil_div(n_in, kFcChunk) * 2 * kWnXi * 4 * 64 * 2 * 4;
}

int fc_wino_pack_weights(const float *w0, float *U, int C, int c_off, int dgrad, int k, hipStream_t stream) {
  const int n_in = dgrad ? kFcHidden : C, n_out = dgrad ? C : kFcHidden;
  const int nch = (int)ceil_div(n_in, kFcChunk);
  const int64_t total = ceil_div(n_out, kWnN) * kWnN * (int64_t)nch * kFcChunk;
  const dim3 grid((unsigned)ceil_div(total, 256));
  if (k == 5)
    fc_wino_pack_w_kernel<5><<<grid, 256, 0, stream>>>(w0, U, C, c_off, dgrad, n_in, n_out, nch);
  else if (k == 3)
    fc_wino_pack_w_kernel<3><<<grid, 256, 0, stream>>>(w0, U, C, c_off, dgrad, n_in, n_out, nch);
  else
    return GFLA_ERR_UNSUPPORTED;
  return launch_status();
}

// ---- the convolution ---------------------------------------------------------------------------------------------
struct WnGeo {
  int TH, TW, ngroups, span;  // tile grid, groups of 32 tiles per sample, raw pixels a group stages per chunk
};

template <int KS>
static WnGeo wn_geometry(int M, int Wv, int Wp) {
  constexpr int m = Wn<KS>::M;
  WnGeo g;
  const int Ho = M / Wv;
  g.TH = (Ho + m - 1) / m;
  g.TW = (Wv + m - 1) / m;
  g.ngroups = (g.TH * g.TW + kWnTiles - 1) / kWnTiles;
  // tile rows a group of 32 consecutive tiles can touch
  int rows = g.TW >= kWnTiles ? 2 : (kWnTiles + g.TW - 2) / g.TW + 1;
  if (rows > g.TH) rows = g.TH;
  g.span = ((rows - 1) * m + 6) * Wp + 6;
  return g;
}

template <int KS>
static unsigned wn_lds_bytes(const WnGeo &g) {
  return (unsigned)(2 * kWnVFloats * 4 + ((g.span * Wn<KS>::PITCH + 15) & ~15));
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

template <int KS>
__global__ __launch_bounds__(256, 1) void fc_wino_conv_kernel(PackedDesc X, const float *__restrict__ U,
                                                             float *__restrict__ out, int64_t out_bs, int ldo,
                                                             int n_valid, int Ho, int Wv, int Wp, int nch, WnGeo geo,
                                                             int ntn, int64_t total_groups, int64_t S) {
  constexpr int M = Wn<KS>::M, PITCH = Wn<KS>::PITCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  float *vbuf = reinterpret_cast<float *>(gfla_smem);      // [2][36][32][8]
  unsigned char *raw = gfla_smem + 2 * kWnVFloats * 4;     // [span][PITCH]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // workgroup -> (group of tiles, output-channel tile).  Ids x and x + 8 run on the same XCD: the workgroups that
  // share one group's input pixels (different channel tiles) are neighbours in that XCD's queue (shared L2).
  const int64_t x = blockIdx.x;
  const int xcd = (int)(x & 7);
  const int64_t slot = x >> 3;
  const int ntile = (int)(slot % ntn);
  const int64_t glin = (slot / ntn) * 8 + xcd;
  if (glin >= total_groups) return;
  const int64_t b = glin / geo.ngroups;
  const int grp = (int)(glin - b * geo.ngroups);
  const int ntiles = geo.TH * geo.TW;
  const int tile0 = grp * kWnTiles;
  const int ty_first = tile0 / geo.TW;
  const int p0 = M * ty_first * Wp;                         // first pixel of the staged span
  const int64_t avail = S - p0;                             // pixels of this sample behind p0 (the rest reads as zero)

  // transform item of this thread: (tile, channel of the 8-channel step)
  const int tl = t >> 3, c8 = t & 7;
  int toff;
  {
    const int tau = min(tile0 + tl, ntiles - 1);
    const int ty = tau / geo.TW, tx = tau - ty * geo.TW;
    toff = ((M * ty * Wp + M * tx) - p0) * PITCH + c8 * 4;
  }
  const int vpos = tl * 8 + (c8 & 3) * 2 + (c8 >> 2);       // float offset inside V[point]

  const unsigned char *xg = X.base + b * X.batch_stride + (int64_t)p0 * X.pix_stride;
  const int npieces = geo.span * 4;

  f32x4v acc[kWnXi][2];
#pragma unroll
  for (int q = 0; q < kWnXi; ++q)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc[q][mb] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // raw span of one chunk: pieces t, t + 256, ... ; the first kWnPF go through registers (prefetched a step ahead)
  u32x4v pf[kWnPF];
  auto piece_src = [&](int q, int cc) -> const unsigned char * {
    const int pix = q >> 2;
    return xg + (int64_t)cc * X.chunk_stride + (int64_t)min((int64_t)pix, avail - 1) * X.pix_stride + (q & 3) * 16;
  };
  auto piece_store = [&](int q, u32x4v v) {
    const int pix = q >> 2;
    if (pix >= avail) v = u32x4v{0u, 0u, 0u, 0u};
    uint2 *d = reinterpret_cast<uint2 *>(raw + pix * PITCH + (q & 3) * 16);
    d[0] = make_uint2(v[0], v[1]);
    d[1] = make_uint2(v[2], v[3]);
  };
  auto prefetch = [&](int cc) {
#pragma unroll
    for (int i = 0; i < kWnPF; ++i) {
      const int q = min(t + 256 * i, npieces - 1);
      pf[i] = *reinterpret_cast<const u32x4v *>(piece_src(q, cc));
    }
  };
  auto commit = [&](int cc) {
#pragma unroll
    for (int i = 0; i < kWnPF; ++i) {
      const int q = t + 256 * i;
      if (q < npieces) piece_store(q, pf[i]);
    }
    for (int q = t + 256 * kWnPF; q < npieces; q += 256)   // spans beyond the register budget: loaded here
      piece_store(q, *reinterpret_cast<const u32x4v *>(piece_src(q, cc)));
  };

  // this lane's B fragments of the current step: U[ntile][cc][half][point][wave][lane][2]
  const float2 *ub = reinterpret_cast<const float2 *>(U) + ((int64_t)ntile * nch * 2 * kWnXi * 4 + wave) * 64 + lane;
  float2 bf[kWnXi];
  auto load_b = [&](int step, int q) { return ub[((int64_t)step * kWnXi + q) * 4 * 64]; };

  // transform of one step: raw[(tile pixel + i*Wp + j)][channel] -> V[buf][point][tile][channel]
  auto transform = [&](int step, int buf) {
    const unsigned char *src = raw + toff + (step & 1) * 32;
    float *dst = vbuf + buf * kWnVFloats + vpos;
    float tm[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float d[6], o[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) d[i] = *reinterpret_cast<const float *>(src + (i * Wp + j) * PITCH);
      wn_bt(d, o);
#pragma unroll
      for (int a = 0; a < 6; ++a) tm[a][j] = o[a];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      float o[6];
      wn_bt(tm[a], o);
#pragma unroll
      for (int e = 0; e < 6; ++e) dst[(a * 6 + e) * kWnTiles * 8] = o[e];
    }
  };

  const int nsteps = 2 * nch;
  prefetch(0);
  commit(0);
#pragma unroll
  for (int q = 0; q < kWnXi; ++q) bf[q] = load_b(0, q);
  __syncthreads();
  transform(0, 0);
  __syncthreads();

  const int arow = (lane & 15) * 8 + (lane >> 4) * 2;  // float offset of this lane's A fragment inside V[point][block]
  for (int s = 0; s < nsteps; ++s) {
    const int cc = s >> 1;
    const bool stage_next = !(s & 1) && cc + 1 < nch;
    if (stage_next) prefetch(cc + 1);
    const float *va = vbuf + (s & 1) * kWnVFloats + arow;
    const int sn = min(s + 1, nsteps - 1);
#pragma unroll
    for (int q = 0; q < kWnXi; ++q) {
      const float2 a0 = *reinterpret_cast<const float2 *>(va + q * kWnTiles * 8);
      const float2 a1 = *reinterpret_cast<const float2 *>(va + q * kWnTiles * 8 + 16 * 8);
      const float2 bq = bf[q];
      acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bq.x, acc[q][0], 0, 0, 0);
      acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bq.x, acc[q][1], 0, 0, 0);
      acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bq.y, acc[q][0], 0, 0, 0);
      acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bq.y, acc[q][1], 0, 0, 0);
      bf[q] = load_b(sn, q);  // the register is free again: next step's slice, one whole step ahead of its use
    }
    if (s + 1 < nsteps) transform(s + 1, (s + 1) & 1);
    __syncthreads();
    if (stage_next) {
      commit(cc + 1);
      __syncthreads();
    }
  }

  // epilogue: Y = A^T M A.  C/D layout of the 16x16 MFMA: column (channel) = lane & 15, row (tile) = 4*(lane >> 4) + r
  const int col = ntile * kWnN + wave * 16 + (lane & 15);
  float *ob = out + b * out_bs + col;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tau = tile0 + mb * 16 + 4 * (lane >> 4) + r;
      float qv[6][M];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        float v[6], y[M];
#pragma unroll
        for (int e = 0; e < 6; ++e) v[e] = acc[a * 6 + e][mb][r];
        wn_at<M>(v, y);
#pragma unroll
        for (int j = 0; j < M; ++j) qv[a][j] = y[j];
      }
      if (tau >= ntiles || col >= n_valid) continue;
      const int ty = tau / geo.TW, tx = tau - ty * geo.TW;
#pragma unroll
      for (int j = 0; j < M; ++j) {
        float v[6], y[M];
#pragma unroll
        for (int a = 0; a < 6; ++a) v[a] = qv[a][j];
        wn_at<M>(v, y);
        const int xo = M * tx + j;
        if (xo >= Wv) continue;
#pragma unroll
        for (int i = 0; i < M; ++i) {
          const int yo = M * ty + i;
          if (yo < Ho) ob[(int64_t)(yo * Wv + xo) * ldo] = y[i];
        }
      }
    }
  }
}

bool fc_wino_fits(int M, int Wv, int Wp, int k) {
  if (k != 3 && k != 5) return false;
  if (Wv <= 0 || Wv > Wp || M <= 0 || M % Wv) return false;
  const WnGeo g = k == 5 ? wn_geometry<5>(M, Wv, Wp) : wn_geometry<3>(M, Wv, Wp);
  const unsigned lds = k == 5 ? wn_lds_bytes<5>(g) : wn_lds_bytes<3>(g);
  return lds <= 160 * 1024;
}

// out[b][r][n] = sum_{chunk, tap, c} X[b][chunk][pix(r) + tap][c] * w[...]  -- the contract of fc_conv (fc_conv_impl.h),
// with the weights given as the transformed U of fc_wino_pack_weights.  S = pixels per sample X may be read for.
int fc_wino_conv(const PackedDesc &X, const float *U, float *out, int64_t out_bs, int ldo, int n_valid, int64_t B, int nch,
                 int M, int Wv, int Wp, int64_t S, int k, hipStream_t stream) {
  if (B <= 0) return GFLA_OK;
  if (!fc_wino_fits(M, Wv, Wp, k)) return GFLA_ERR_UNSUPPORTED;
  const int ntn = (int)ceil_div(n_valid, kWnN);
#define GFLA_WINO(K_)                                                                                                  \
  {                                                                                                                    \
    const WnGeo g = wn_geometry<K_>(M, Wv, Wp);                                                                        \
    const unsigned lds = wn_lds_bytes<K_>(g);                                                                          \
    const int64_t groups = B * g.ngroups;                                                                              \
    const int64_t wgs = ceil_div(groups, 8) * 8 * ntn;                                                                 \
    if (wgs > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;                                                               \
    auto kern = fc_wino_conv_kernel<K_>;                                                                               \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    kern<<<dim3((unsigned)wgs), 256, lds, stream>>>(X, U, out, out_bs, ldo, n_valid, M / Wv, Wv, Wp, nch, g, ntn, groups, S); \
  }
  if (k == 5) GFLA_WINO(5) else GFLA_WINO(3)
#undef GFLA_WINO
  return launch_status();
}

}  // namespace gfla
