// MFMA kernels for the first FC layer of ExtractorAttn, gfx950.
//
// Reference (model/networks/base_function.py:799-807):
//   fully_connect_layer[0] = Conv2d(2C, 128, k, stride k) applied to cat(block_target, block_source), where
//   block_source = BlockExtractor(source, flow), block_target = BlockExtractor(target, 0): 2*C*k*k inputs per
//   output position, 147 + 27 GFLOP per forward at the PoseGenerator shapes -- the one real contraction on the
//   hot path.  The reference materialises both (B,C,kH,kW) block tensors and runs a strided convolution.
//
// Here no block tensor (and no k^2-times-larger GEMM operand) exists in either direction:
//   * all k*k taps of one position share ONE fractional offset, and integer shifts commute with a convolution, so
//       FC0_source(p) = bilinear_sample( conv_kxk(replicate_extend(source, k-1), W[:, C:]), p + flow(p) )
//     exactly, border rule (clamped index, unclamped weight; block_extractor_kernel.cu:69-76) included;
//     FC0_target(p) = conv_kxk(replicate_pad(target), W[:, :C])(p).  Both halves are PLAIN stride-1 convolutions of
//     small maps; the flow enters only through a 4-tap sampling of a 128-channel map (fc_sample.hip).
//   * backward: the hidden gradient is scattered (4 taps) into the convolved map's gradient, then the data
//     gradient is the transposed convolution (the same kernel with flipped weights) and the weight gradient a
//     pixel-reduction GEMM.
//
// The convolutions are implicit GEMMs on the matrix cores:
//   * operands are "linearised": a map is stored pixel-linear with its padded width as row pitch, 16-channel
//     records per pixel, so tap (i,j) is the constant pixel offset i*Wp + j and an input tile is ONE contiguous
//     range (outputs in the k-1 wrap-around columns of a row are computed and ignored: 8 % of the work);
//   * gradient maps use the same pitch with the k-1 trailing columns of each row zero ("Z layout") and
//     (k-1)*(Wp+1) leading zeros: the wrap-around of a tap then lands on zeros, so the transposed convolution
//     needs no border logic at all;
//   * a workgroup (4 waves) owns 128 output pixels x 128 output channels: the input tile (+ tap halo) of a
//     16-channel chunk is staged once and reused by all k*k taps, the 128x16 weight tile of a tap is streamed
//     through a double buffer; a wave accumulates 64x64 outputs in 4 MFMA 32x32 tiles;
//   * the weight gradient reduces over pixels, which are the STRIDED dimension of the pixel-major records:
//     ds_read_b64_tr_b16 (transposing LDS read) delivers 4 consecutive pixels of one channel per lane.
// Arithmetic modes (fc_gemm.h): exact f32 MFMA, or f16-split operands with f32 accumulation.
#include "fc_mma.h"

namespace gfla {

// ------------------------------------------------------------------------------------------ helpers
template <int NS>
__device__ __forceinline__ void split_f16(float v, _Float16 (&o)[NS]) {
  o[0] = (_Float16)v;
  if constexpr (NS >= 2) {
    const float r = v - (float)o[0];
    o[1] = (_Float16)r;
    if constexpr (NS >= 3) o[2] = (_Float16)(r - (float)o[1]);
  }
}

// 8 consecutive channels -> one 16-byte piece per f16 term (or two pieces of f32), scaled by s
template <int MODE>
__device__ __forceinline__ void store_pieces(const float (&v)[8], float s, unsigned char *dst, int64_t split_stride) {
  using F = Fc<MODE>;
  if constexpr (MODE == 0) {
    float4 *d = reinterpret_cast<float4 *>(dst);
    d[0] = make_float4(v[0], v[1], v[2], v[3]);
    d[1] = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    f16x8 out[F::NS];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      _Float16 parts[F::NS];
      split_f16<F::NS>(v[e] * s, parts);
#pragma unroll
      for (int sp = 0; sp < F::NS; ++sp) out[sp][e] = parts[sp];
    }
#pragma unroll
    for (int sp = 0; sp < F::NS; ++sp) *reinterpret_cast<f16x8 *>(dst + sp * split_stride) = out[sp];
  }
}

PackedDesc fc_desc_packed(const void *base, int64_t B, int nch, int64_t S, int mode) {
  PackedDesc d;
  d.base = static_cast<const unsigned char *>(base);
  d.pix_stride = kFcChunk * fc_esz(mode);
  d.chunk_stride = S * d.pix_stride;
  d.batch_stride = (int64_t)nch * d.chunk_stride;
  d.split_stride = B * d.batch_stride;
  return d;
}
PackedDesc fc_desc_nhwc(const float *base, int64_t S, int Cz) {  // f32 (B, S, Cz): mode 0 reads it in place
  PackedDesc d;
  d.base = reinterpret_cast<const unsigned char *>(base);
  d.pix_stride = Cz * 4;
  d.chunk_stride = kFcChunk * 4;
  d.batch_stride = S * (int64_t)Cz * 4;
  d.split_stride = 0;
  return d;
}
int64_t fc_packed_bytes(int64_t B, int nch, int64_t S, int mode) {
  return (int64_t)fc_nsplit(mode) * B * nch * S * kFcChunk * fc_esz(mode);
}
int64_t fc_wpack_bytes(int ntiles, int nch, int k, int mode) {
  return (int64_t)fc_nsplit(mode) * ntiles * nch * k * k * kFcTN * kFcChunk * fc_esz(mode);
}

// ------------------------------------------------------------------------------------------ max |x|
// blockIdx.y = job: up to three tensors per launch (the f16-split modes need max |x| of both activations and the weights
// before anything else can start, and of both gradient maps in the backward pass)
struct MaxAbsJob {
  const float *x;
  int64_t n;
  uint32_t *slot;
};
struct MaxAbsJobs {
  MaxAbsJob j[3];
};
__global__ __launch_bounds__(256) void fc_maxabs_kernel(MaxAbsJobs jobs) {
  const MaxAbsJob &J = jobs.j[blockIdx.y];
  const float *__restrict__ x = J.x;
  const int64_t n = J.n;
  uint32_t m = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n >> 2 : 0;
  const float4 *x4 = reinterpret_cast<const float4 *>(x);
  // four 16-byte requests in flight per thread: with one the pass ran at 2.8 TB/s (82 us per step in arithmetic mode 5, which
  // reads every operand tensor once more than mode 4: one-stream trace, round 6)
  auto fold4 = [&](const float4 &v) {
    m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
    m = max(max(m, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
  };
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * stride < n4; i += 8 * stride) {   // eight requests in flight
    const float4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
    const float4 v4 = x4[i + 4 * stride], v5 = x4[i + 5 * stride], v6 = x4[i + 6 * stride], v7 = x4[i + 7 * stride];
    fold4(v0), fold4(v1), fold4(v2), fold4(v3), fold4(v4), fold4(v5), fold4(v6), fold4(v7);
  }
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
    fold4(v0), fold4(v1), fold4(v2), fold4(v3);
  }
  for (; i < n4; i += stride) fold4(x4[i]);
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, s));
  // ONE atomic per workgroup: thousands of atomics on a single address serialise at ~12 ns each
  __shared__ uint32_t s_m[4];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
    if (m) atomicMax(J.slot, m);
  }
}

int fc_maxabs_multi(const float *x0, int64_t n0, uint32_t *slot0, const float *x1, int64_t n1, uint32_t *slot1,
                    const float *x2, int64_t n2, uint32_t *slot2, hipStream_t stream) {
  MaxAbsJobs jobs;
  int nj = 0;
  int64_t most = 0;
  const float *x[3] = {x0, x1, x2};
  const int64_t n[3] = {n0, n1, n2};
  uint32_t *slot[3] = {slot0, slot1, slot2};
  for (int i = 0; i < 3; ++i)
    if (x[i] && n[i] > 0) {
      jobs.j[nj++] = MaxAbsJob{x[i], n[i], slot[i]};
      if (n[i] > most) most = n[i];
    }
  if (nj == 0) return GFLA_OK;
  for (int i = nj; i < 3; ++i) jobs.j[i] = jobs.j[0];
  int64_t blocks = ceil_div(most, 256 * 16);
  // about one workgroup per CU over all jobs: every workgroup ends in an atomic on its job's slot, and with four workgroups per
  // CU and job (rounds 2-5) those were most of the kernel -- 40 -> 16 us per launch at the bench shapes (round 6)
  const int64_t cap = std::max<int64_t>(64, kNumCU / nj);
  if (blocks > cap) blocks = cap;
  fc_maxabs_kernel<<<dim3((unsigned)blocks, (unsigned)nj), 256, 0, stream>>>(jobs);
  return launch_status();
}

int fc_maxabs(const float *x, int64_t n, uint32_t *slot, hipStream_t stream) {
  return fc_maxabs_multi(x, n, slot, nullptr, 0, nullptr, nullptr, 0, nullptr, stream);
}

// ------------------------------------------------------------------------------ pack: NCHW f32 -> records
// out[term][b][chunk][m = yp*Wp + xp][16] = src[b][16*chunk + ch][clamp(yp - pad_t)][clamp(xp - pad_l)] * scale for
// m < Hp*Wp, zero for the read slack behind it (m < S) and for channels >= C (C not a multiple of 16).
// A thread produces one 8-channel piece: its 8 reads are coalesced along x across the lanes, the stores contiguous.
// One launch carries up to TWO packs (the source and the target half of a layer): blockIdx.z < nb0 is job 0.
struct PackJob {
  const float *src;
  const uint32_t *amax;
  unsigned char *out;
  int Hp, Wp, pad_t, pad_l;
  int64_t S, split_stride;
};
struct PackJobs {
  PackJob j[2];
};
template <int MODE>
__global__ __launch_bounds__(256) void fc_pack_act_kernel(PackJobs jobs, int nb0, int C, int H, int W, int nch) {
  using F = Fc<MODE>;
  const bool second = (int)blockIdx.z >= nb0;
  const PackJob &J = jobs.j[second ? 1 : 0];
  const float *__restrict__ src = J.src;
  const int Hp = J.Hp, Wp = J.Wp, pad_t = J.pad_t, pad_l = J.pad_l;
  const int64_t S = J.S;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (m, half)
  const int64_t m = idx >> 1;
  const int half = (int)(idx & 1);
  if (m >= S) return;
  const int cc = blockIdx.y;
  const int64_t b = (int)blockIdx.z - (second ? nb0 : 0);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (m < (int64_t)Hp * Wp) {
    const int yp = (int)(m / Wp), xp = (int)(m - (int64_t)yp * Wp);
    const int y = clampi(yp - pad_t, 0, H - 1), x = clampi(xp - pad_l, 0, W - 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cc * kFcChunk + half * 8 + e;
      if (c < C) v[e] = src[((b * C + c) * H + y) * (int64_t)W + x];
    }
  }
  const float s = MODE == 0 ? 1.f : fc_scale(J.amax);
  unsigned char *dst = J.out + (((b * nch + cc) * S + m) * kFcChunk + half * 8) * F::ESZ;
  store_pieces<MODE>(v, s, dst, J.split_stride);
}

static int fc_pack_act_launch(const PackJobs &jobs, int njobs, int64_t B, int C, int H, int W, int mode, hipStream_t stream) {
  if (B <= 0 || njobs <= 0) return GFLA_OK;
  const int nch = (int)ceil_div(C, kFcChunk);
  if (nch > 65535 || njobs * B > 65535) return GFLA_ERR_UNSUPPORTED;
  int64_t smax = jobs.j[0].S;
  if (njobs > 1 && jobs.j[1].S > smax) smax = jobs.j[1].S;
  const dim3 grid((unsigned)ceil_div(2 * smax, 256), (unsigned)nch, (unsigned)(njobs * B));
#define GFLA_LAUNCH(M_) fc_pack_act_kernel<M_><<<grid, 256, 0, stream>>>(jobs, (int)B, C, H, W, nch)
  if (mode == 0) GFLA_LAUNCH(0);
  else if (mode == 1) GFLA_LAUNCH(1);
  else if (mode == 2) GFLA_LAUNCH(2);
  else GFLA_LAUNCH(3);
#undef GFLA_LAUNCH
  return launch_status();
}

static PackJob pack_job(const float *src, const uint32_t *amax, void *out, int64_t B, int C, const FcHalf &g, int mode) {
  const int nch = (int)ceil_div(C, kFcChunk);
  const PackedDesc d = fc_desc_packed(out, B, nch, g.Sx, mode);
  return PackJob{src, amax, static_cast<unsigned char *>(out), g.Hp, g.Wp, g.pad_t, g.pad_l, g.Sx, d.split_stride};
}

int fc_pack_act(const float *src, const uint32_t *amax, void *out, int64_t B, int C, int H, int W, const FcHalf &g,
                int mode, hipStream_t stream) {
  PackJobs jobs;
  jobs.j[0] = pack_job(src, amax, out, B, C, g, mode);
  jobs.j[1] = jobs.j[0];
  return fc_pack_act_launch(jobs, 1, B, C, H, W, mode, stream);
}

// both halves of a layer in one launch
int fc_pack_act2(const float *src_s, const uint32_t *amax_s, void *out_s, const FcHalf &gs, const float *src_t,
                 const uint32_t *amax_t, void *out_t, const FcHalf &gt, int64_t B, int C, int H, int W, int mode,
                 hipStream_t stream) {
  PackJobs jobs;
  jobs.j[0] = pack_job(src_s, amax_s, out_s, B, C, gs, mode);
  jobs.j[1] = pack_job(src_t, amax_t, out_t, B, C, gt, mode);
  return fc_pack_act_launch(jobs, 2, B, C, H, W, mode, stream);
}

// ------------------------------------------------------------- pack: f32 (B, S, Cz) pixel-major -> f16 records
// (mode 2/3 only; mode 0 reads the f32 map in place).  A thread owns one pixel and walks its chunks: the reads of
// a wave hit every 64-byte line of the map exactly once (through L2), the stores are contiguous per chunk.
struct PackZJob {
  const float *z;
  const uint32_t *amax;
  unsigned char *out;
  int64_t S, split_stride;
};
struct PackZJobs {
  PackZJob j[2];
};
template <int MODE>
__global__ __launch_bounds__(256) void fc_pack_z_kernel(PackZJobs jobs, int nb0, int Cz) {
  using F = Fc<MODE>;
  const bool second = (int)blockIdx.y >= nb0;
  const PackZJob &J = jobs.j[second ? 1 : 0];
  const int64_t S = J.S;
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= S) return;
  const int64_t b = (int)blockIdx.y - (second ? nb0 : 0);
  const int nch = Cz / kFcChunk;
  const float s = fc_scale(J.amax);
  const float4 *zp = reinterpret_cast<const float4 *>(J.z + (b * S + m) * Cz);
  for (int cc = 0; cc < nch; ++cc) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const float4 a = zp[cc * 4 + half * 2], c = zp[cc * 4 + half * 2 + 1];
      const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
      unsigned char *dst = J.out + (((b * nch + cc) * S + m) * kFcChunk + half * 8) * F::ESZ;
      store_pieces<MODE>(v, s, dst, J.split_stride);
    }
  }
}

static int fc_pack_z_launch(const PackZJobs &jobs, int njobs, int64_t B, int Cz, int mode, hipStream_t stream) {
  if (mode == 0 || B <= 0 || njobs <= 0) return GFLA_OK;
  int64_t smax = jobs.j[0].S;
  if (njobs > 1 && jobs.j[1].S > smax) smax = jobs.j[1].S;
  if (njobs * B > 65535) return GFLA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)ceil_div(smax, 256), (unsigned)(njobs * B));
  if (mode == 1)
    fc_pack_z_kernel<1><<<grid, 256, 0, stream>>>(jobs, (int)B, Cz);
  else if (mode == 2)
    fc_pack_z_kernel<2><<<grid, 256, 0, stream>>>(jobs, (int)B, Cz);
  else
    fc_pack_z_kernel<3><<<grid, 256, 0, stream>>>(jobs, (int)B, Cz);
  return launch_status();
}

int fc_pack_z(const float *z, const uint32_t *amax, void *out, int64_t B, int64_t S, int Cz, int mode,
              hipStream_t stream) {
  if (mode == 0) return GFLA_OK;
  PackZJobs jobs;
  jobs.j[0] = jobs.j[1] =
      PackZJob{z, amax, static_cast<unsigned char *>(out), S, fc_desc_packed(out, B, Cz / kFcChunk, S, mode).split_stride};
  return fc_pack_z_launch(jobs, 1, B, Cz, mode, stream);
}

// both gradient maps of a layer in one launch
int fc_pack_z2(const float *z_s, const uint32_t *amax_s, void *out_s, int64_t S_s, const float *z_t, const uint32_t *amax_t,
               void *out_t, int64_t S_t, int64_t B, int Cz, int mode, hipStream_t stream) {
  if (mode == 0) return GFLA_OK;
  PackZJobs jobs;
  jobs.j[0] = PackZJob{z_s, amax_s, static_cast<unsigned char *>(out_s), S_s,
                       fc_desc_packed(out_s, B, Cz / kFcChunk, S_s, mode).split_stride};
  jobs.j[1] = PackZJob{z_t, amax_t, static_cast<unsigned char *>(out_t), S_t,
                       fc_desc_packed(out_t, B, Cz / kFcChunk, S_t, mode).split_stride};
  return fc_pack_z_launch(jobs, 2, B, Cz, mode, stream);
}

// ------------------------------------------------------------------- unpack: one-f16-term records -> f32 records
// The packed activation layout is the same for every arithmetic mode up to the element type (records of 16 channels,
// pixel-linear, chunk-major): out32[i] = (float)in16[i] / scale.  Exact: a mode-1 operand IS the (scaled) input value
// whenever that value has <= 11 significant bits -- a bf16 feature has 8.  Lets the float32 Winograd-domain weight
// gradient (fc_wino.hip) read the activations the mode-1 forward packed (fc_block.hip: bf16 features, k = 5).
__global__ __launch_bounds__(256) void fc_unpack_act_kernel(const _Float16 *__restrict__ x16, const uint32_t *__restrict__ amax,
                                                           float *__restrict__ x32, int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const float inv = fc_inv_scale(amax);
  const f16x8 v = reinterpret_cast<const f16x8 *>(x16)[i];
  float4 a, b;
  a.x = (float)v[0] * inv; a.y = (float)v[1] * inv; a.z = (float)v[2] * inv; a.w = (float)v[3] * inv;
  b.x = (float)v[4] * inv; b.y = (float)v[5] * inv; b.z = (float)v[6] * inv; b.w = (float)v[7] * inv;
  reinterpret_cast<float4 *>(x32)[2 * i] = a;
  reinterpret_cast<float4 *>(x32)[2 * i + 1] = b;
}
int fc_unpack_act(const void *x16, const uint32_t *amax, float *x32, int64_t B, int nch, int64_t S, hipStream_t stream) {
  const int64_t n8 = B * nch * S * kFcChunk / 8;
  if (n8 <= 0) return GFLA_OK;
  if (ceil_div(n8, 256) > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  fc_unpack_act_kernel<<<dim3((unsigned)ceil_div(n8, 256)), 256, 0, stream>>>(static_cast<const _Float16 *>(x16), amax, x32, n8);
  return launch_status();
}

// ------------------------------------------------------------------------------------ pack: weights
// conv0.weight (128, 2C, k, k), channel order (target, source) (base_function.py:805: cat((block_target, block_source))).
// Forward tiles   wf[term][0][chunk][tap][n = 0..127][16 c]      = W[n][off + 16*chunk + c][i][j]
// data-grad tiles wd[term][ntile][chunk][tap'][c_out = 0..127][16 n] = W[16*chunk + n][off + 128*ntile + c_out][k-1-i'][k-1-j']
// (the transposed convolution as a convolution with flipped taps and swapped channel roles).
struct PackWJob {
  unsigned char *dst;
  int c_off, dgrad, nch;
  int64_t total_pieces, split_stride;
};
struct PackWJobs {
  PackWJob j[4];
};

// grid (blocks, 4 jobs): forward / data-gradient tiles of the target / source half in ONE launch
template <int MODE>
__global__ __launch_bounds__(256) void fc_pack_w_kernel(const float *__restrict__ w0, const uint32_t *__restrict__ amax,
                                                       PackWJobs jobs, int C, int k) {
  using F = Fc<MODE>;
  const PackWJob jb = jobs.j[blockIdx.y];
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (ntile, chunk, tap, row, half)
  if (!jb.dst || idx >= jb.total_pieces) return;
  const int half = (int)(idx & 1);
  const int row = (int)((idx >> 1) & 127);
  int64_t rest = idx >> 8;
  const int KK = k * k;
  const int tap = (int)(rest % KK);
  rest /= KK;
  const int cc = (int)(rest % jb.nch);
  const int ntile = (int)(rest / jb.nch);
  const int i = tap / k, j = tap - i * k;
  const float s = MODE == 0 ? 1.f : fc_scale(amax);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int kc = cc * kFcChunk + half * 8 + e;
    float val = 0.f;
    if (!jb.dgrad) {
      const int n = ntile * kFcTN + row;
      if (n < kFcHidden && kc < C) val = w0[(((int64_t)n * 2 * C + jb.c_off + kc) * k + i) * k + j];
    } else {
      const int co = ntile * kFcTN + row;
      if (co < C && kc < kFcHidden) val = w0[(((int64_t)kc * 2 * C + jb.c_off + co) * k + (k - 1 - i)) * k + (k - 1 - j)];
    }
    v[e] = val;
  }
  store_pieces<MODE>(v, s, jb.dst + (idx >> 1) * (int64_t)F::REC + half * 8 * F::ESZ, jb.split_stride);
}

int fc_pack_weights(const float *w0, const uint32_t *amax, void *wf_t, void *wf_s, void *wd_t, void *wd_s, int C,
                    int k, int mode, hipStream_t stream) {
  const int nch_c = (int)ceil_div(C, kFcChunk), nch_h = kFcHidden / kFcChunk, nt_d = (int)ceil_div(C, kFcTN);
  struct Spec {
    void *dst;
    int c_off, dgrad, ntiles, nch;
  } specs[4] = {{wf_t, 0, 0, 1, nch_c}, {wf_s, C, 0, 1, nch_c}, {wd_t, 0, 1, nt_d, nch_h}, {wd_s, C, 1, nt_d, nch_h}};
  PackWJobs jobs;
  int64_t most = 0;
  for (int q = 0; q < 4; ++q) {
    const int64_t pieces = (int64_t)specs[q].ntiles * specs[q].nch * k * k * kFcTN * 2;
    jobs.j[q] = PackWJob{static_cast<unsigned char *>(specs[q].dst), specs[q].c_off, specs[q].dgrad, specs[q].nch, pieces,
                         pieces / 2 * kFcChunk * fc_esz(mode)};
    if (specs[q].dst && pieces > most) most = pieces;
  }
  if (most == 0) return GFLA_OK;
  const dim3 grid((unsigned)ceil_div(most, 256), 4);
  if (mode == 0)
    fc_pack_w_kernel<0><<<grid, 256, 0, stream>>>(w0, amax, jobs, C, k);
  else if (mode == 1)
    fc_pack_w_kernel<1><<<grid, 256, 0, stream>>>(w0, amax, jobs, C, k);
  else if (mode == 2)
    fc_pack_w_kernel<2><<<grid, 256, 0, stream>>>(w0, amax, jobs, C, k);
  else
    fc_pack_w_kernel<3><<<grid, 256, 0, stream>>>(w0, amax, jobs, C, k);
  return launch_status();
}

// -------------------------------------------------------------------------------- weight gradient
// dwacc[tap][16*chunk + c][n] += sum_{b, m} X[b][chunk][m + i*Wp + j][c] * Y[b][n/16][lead + m][n%16]
// (scaled by both operands' scales; fc_unpack_wgrad undoes that).  The reduction runs over pixels, the strided
// dimension of both operands, so fragments come from transposing LDS reads (mode 2/3) or plain 4-byte reads
// (mode 0).  A workgroup owns one 16-channel chunk of X, a group of 2*NB taps and a range of samples; the MFMA rows
// are (tap parity, channel), the columns the 128 hidden channels (wave w: columns 32w..32w+31), so results leave
// as 128-byte coalesced atomics.
template <int MODE>
struct WgTile {
  static constexpr int KC = (MODE == 2 || MODE == 1) ? 64 : 32;  // pixels per staged K step
  static constexpr int PX = MODE == 0 ? 64 : 32;      // LDS pitch of an X pixel record
  static constexpr int PY = MODE == 0 ? 512 : 320;    // LDS pitch of a Y pixel (128 channels; 256 B + 64: tr-read banks)
};

// Two ds_read_b64_tr_b16: within each group of 16 lanes the hardware reads 8 bytes (4 halves) at every lane's own
// address and hands lane i, element j the element (i & 3) read by lane 4*j + (i >> 2) of its group -- a 4 x 16 block
// of halves (row = 4 consecutive supplier lanes, here one pixel) delivered column-wise: lane i gets channel i of 4
// consecutive pixels.  GFLA_TR_EMULATE builds the same thing from plain reads + shuffles (debug A/B).
__device__ __forceinline__ s16x4 tr_read4(const unsigned char *p) {
#ifdef GFLA_TR_EMULATE
  const s16x4 mine = *reinterpret_cast<const s16x4 *>(p);
  const int lane = threadIdx.x & 63, i = lane & 15, gbase = lane & 48;
  s16x4 out;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int src = gbase + 4 * j + (i >> 2);
    const int e0 = __shfl((int)mine[0], src), e1 = __shfl((int)mine[1], src);
    const int e2 = __shfl((int)mine[2], src), e3 = __shfl((int)mine[3], src);
    const int sel = i & 3;
    out[j] = (short)(sel == 0 ? e0 : sel == 1 ? e1 : sel == 2 ? e2 : e3);
  }
  return out;
#else
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
#endif
}
__device__ __forceinline__ f16x8 tr_read8(const unsigned char *p, int second) {
  const s16x4 lo = tr_read4(p), hi = tr_read4(p + second);
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(f16x8, v);
}

// debug/test entry: raw semantics of the transposing read on an arbitrary LDS image and per-lane byte offsets
__global__ void fc_tr_probe_kernel(const short *__restrict__ image, int n_halves, const int *__restrict__ offsets,
                                   short *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  short *img = reinterpret_cast<short *>(gfla_smem);
  for (int i = threadIdx.x; i < n_halves; i += 64) img[i] = image[i];
  __syncthreads();
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4 *)(gfla_smem + offsets[threadIdx.x]));
#pragma unroll
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int fc_tr_probe(const short *image, int n_halves, const int *offsets, short *out, hipStream_t stream) {
  if (n_halves <= 0 || n_halves > 16384) return GFLA_ERR_BAD_SHAPE;
  fc_tr_probe_kernel<<<1, 64, n_halves * 2, stream>>>(image, n_halves, offsets, out);
  return launch_status();
}

template <int MODE, int KS, int NB>
__global__ __launch_bounds__(256, 2) void fc_wgrad_kernel(PackedDesc X, PackedDesc Y, int64_t y_lead,
                                                         float *__restrict__ dwacc, int cpad, int Mk, int Wp,
                                                         int spw, int B) {
  using F = Fc<MODE>;
  using T = WgTile<MODE>;
  constexpr int KK = KS * KS, KC = T::KC, PX = T::PX, PY = T::PY;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  const int xh = KC + (KS - 1) * (Wp + 1);               // X pixels per K step (tile + tap halo)
  unsigned char *xs = gfla_smem;                         // [NS][xh][PX]
  unsigned char *ys = gfla_smem + (size_t)F::NS * xh * PX;  // [NS][KC][PY]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cc = blockIdx.x, tg = blockIdx.z;
  const int b_begin = blockIdx.y * spw, b_end = min(B, b_begin + spw);
  const int kh = lane >> 5;

  // per-lane LDS offsets of the fragments
  int xa_off, yb_off, cb;
  if constexpr (MODE == 0) {
    cb = (lane >> 4) & 1;                           // MFMA row = (tap parity, channel)
    xa_off = kh * PX + (lane & 15) * 4;
    yb_off = kh * PY + (wave * 32 + (lane & 31)) * 4;
  } else {
    const int i = lane & 15;
    cb = (lane >> 4) & 1;
    xa_off = ((i >> 2) + 8 * kh) * PX + (i & 3) * 8;
    yb_off = ((i >> 2) + 8 * kh) * PY + (wave * 32 + cb * 16 + (i & 3) * 4) * 2;
  }
  int xsh[NB];  // tap shift of this lane's MFMA rows, per row block
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int tap = tg * 2 * NB + 2 * nb + cb;
    const int i = tap / KS, j = tap - i * KS;
    xsh[nb] = tap < KK ? (i * Wp + j) * PX : 0;
  }

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  const int xper = xh * F::PIECES;                 // pieces of one term of the X tile
  constexpr int yper = KC * (kFcHidden / kFcChunk) * F::PIECES;  // pieces of one term of the Y tile
  for (int b = b_begin; b < b_end; ++b) {
    const unsigned char *xg = X.base + (int64_t)b * X.batch_stride + (int64_t)cc * X.chunk_stride;
    const unsigned char *yg = Y.base + (int64_t)b * Y.batch_stride + y_lead * Y.pix_stride;
    const int64_t x_ss = X.split_stride, y_ss = Y.split_stride, y_cs = Y.chunk_stride;
    const int x_ps = X.pix_stride, y_ps = Y.pix_stride;
    for (int mc = 0; mc < Mk; mc += KC) {
      __syncthreads();
      for (int idx = t; idx < xper * F::NS; idx += 256) {
        const int sp = (idx >= xper) + (idx >= 2 * xper);
        const int rem = idx - sp * xper;
        const int r = rem / F::PIECES, piece = rem % F::PIECES;
        *reinterpret_cast<uint4 *>(xs + ((size_t)sp * xh + r) * PX + piece * 16) = *reinterpret_cast<const uint4 *>(
            xg + sp * x_ss + (int64_t)(mc + r) * x_ps + piece * 16);
      }
      for (int idx = t; idx < yper * F::NS; idx += 256) {
        const int sp = idx / yper, rem = idx % yper;
        const int piece = rem % F::PIECES, cn = (rem / F::PIECES) % (kFcHidden / kFcChunk);
        const int r = rem / (F::PIECES * (kFcHidden / kFcChunk));
        *reinterpret_cast<uint4 *>(ys + ((size_t)sp * KC + r) * PY + cn * F::REC + piece * 16) =
            *reinterpret_cast<const uint4 *>(yg + sp * y_ss + (int64_t)cn * y_cs + (int64_t)(mc + r) * y_ps + piece * 16);
      }
      __syncthreads();
      if constexpr (MODE == 0) {
#pragma unroll 4
        for (int s = 0; s < KC / 2; ++s) {
          const float bv = *reinterpret_cast<const float *>(ys + yb_off + 2 * s * PY);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const float av = *reinterpret_cast<const float *>(xs + xa_off + xsh[nb] + 2 * s * PX);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nb], 0, 0, 0);
          }
        }
      } else {
#pragma unroll 2
        for (int ks = 0; ks < KC / 16; ++ks) {
          Frag<MODE> fb;
#pragma unroll
          for (int sp = 0; sp < F::NS; ++sp)
            fb.s[sp] = tr_read8(ys + (size_t)sp * KC * PY + yb_off + ks * 16 * PY, 4 * PY);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            Frag<MODE> fa;
#pragma unroll
            for (int sp = 0; sp < F::NS; ++sp)
              fa.s[sp] = tr_read8(xs + (size_t)sp * xh * PX + xa_off + xsh[nb] + ks * 16 * PX, 4 * PX);
            acc[nb] = mma<MODE>(fa, fb, acc[nb]);
          }
        }
      }
    }
  }
  // rows = (tap parity, channel), columns = hidden channel 32*wave + (lane & 31)
  const int n = wave * 32 + (lane & 31);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
      const int tap = tg * 2 * NB + 2 * nb + (row >> 4);
      if (tap < KK)
        atomic_add(dwacc + ((int64_t)tap * cpad + cc * kFcChunk + (row & 15)) * kFcHidden + n, acc[nb][r]);
    }
}

template <int MODE, int KS, int NB>
static void launch_wgrad(const PackedDesc &X, const PackedDesc &Y, int64_t y_lead, float *dwacc, int cpad, int64_t B,
                         int Mk, int Wp, hipStream_t stream) {
  using F = Fc<MODE>;
  using T = WgTile<MODE>;
  constexpr int KK = KS * KS;
  const int xh = T::KC + (KS - 1) * (Wp + 1);
  const unsigned lds = (unsigned)(F::NS * (xh * T::PX + T::KC * T::PY));
  const int ngroups = (int)ceil_div(ceil_div(KK, 2), NB);
  const int nch = cpad / kFcChunk;
  // samples per workgroup: keep >= 2 workgroups per CU when the batch allows, fewer atomics otherwise
  int spw = 1;
  while ((int64_t)nch * ngroups * ceil_div(B, spw * 2) >= 2 * kNumCU && spw < 8) spw *= 2;
  const dim3 grid((unsigned)nch, (unsigned)ceil_div(B, spw), (unsigned)ngroups);
  auto kern = fc_wgrad_kernel<MODE, KS, NB>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kern<<<grid, 256, lds, stream>>>(X, Y, y_lead, dwacc, cpad, Mk, Wp, spw, (int)B);
}

int fc_wgrad(const PackedDesc &X, const PackedDesc &Y, int64_t y_lead, float *dwacc, int cpad, int64_t B, int Mk,
             int Wp, int k, int mode, hipStream_t stream) {
  if (!fc_mode_ok(mode) || (k != 3 && k != 5)) return GFLA_ERR_UNSUPPORTED;
#define GFLA_WG(M_, K_, NB_) launch_wgrad<M_, K_, NB_>(X, Y, y_lead, dwacc, cpad, B, Mk, Wp, stream)
  if (k == 3) {
    if (mode == 0) GFLA_WG(0, 3, 5);
    else if (mode == 1) GFLA_WG(1, 3, 5);
    else if (mode == 2) GFLA_WG(2, 3, 5);
    else GFLA_WG(3, 3, 5);
  } else {
    if (mode == 0) GFLA_WG(0, 5, 7);
    else if (mode == 1) GFLA_WG(1, 5, 7);
    else if (mode == 2) GFLA_WG(2, 5, 7);
    else GFLA_WG(3, 5, 7);
  }
#undef GFLA_WG
  return launch_status();
}

// ------------------------------------------------------------------- weight gradient, exact f32 (mode 0)
// part[s][tap][16*chunk + c][n] = sum over the (sample, pixel block)s of split s of X[b][chunk][m + i*Wp + j][c] *
// Z[b][lead + m][n] on v_mfma_f32_16x16x4_f32 (same FLOP rate as the 32x32x2 form, but 16-row blocks: one 16-channel
// chunk per block, so ALL k*k taps fit one wave's accumulators and no MFMA row is padding).
// A workgroup (8 waves) = one chunk x all taps x the 128 hidden channels (wave w: columns 16w..16w+15) x a contiguous
// range of 32-pixel blocks of the flattened (sample, pixel) space -- the ranges are equal, so one round of workgroups
// is full whatever the batch.  Per 4-pixel K step a wave issues k*k MFMAs against k*k + 1 four-byte LDS reads, the
// reads of a tap row running one row ahead of its MFMAs.  Blocks are double-buffered in LDS and arrive by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, the next block lands while this one is multiplied; both the
// 16-channel X records and the 128-channel Z pixels are contiguous in memory, so the LDS image is linear).  Partial
// sums leave as plain coalesced stores, fc_wgrad_reduce adds the splits and writes conv0.weight.grad's layout.
constexpr int kWgKC = 32;  // pixels per staged block

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma16(const unsigned char *gsrc, unsigned char *lds_wave_base) {
  // 64 lanes x 16 bytes: lane l's 16 bytes land at lds_wave_base + 16*l (the LDS address is wave-uniform + lane*16)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int KS>
__global__ __launch_bounds__(512, 4) void fc_wgrad_f32_kernel(PackedDesc X, PackedDesc Y, int64_t y_lead,
                                                             float *__restrict__ part, int cpad, int Wp, int nblk,
                                                             int64_t total_blocks, int nsplit) {
  constexpr int KK = KS * KS, KC = kWgKC, YB = KC * kFcHidden * 4;  // bytes of one Z block
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  const int xh = KC + (KS - 1) * (Wp + 1);
  const int nx = (xh * 4 + 63) >> 6;      // wave-instructions (64 x 16 B) of one X block, rounded up (the input has slack)
  constexpr int ny = YB >> 10;
  const int xbytes = nx << 10;
  unsigned char *xs = gfla_smem;                // [2][xbytes]: pixel records of 64 B
  unsigned char *ys = gfla_smem + 2 * xbytes;   // [2][KC][512 B]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int cc = blockIdx.x, sp = blockIdx.y;
  const int64_t blk0 = total_blocks * sp / nsplit, blk1 = total_blocks * (sp + 1) / nsplit;
  const int l15 = lane & 15, kq = lane >> 4;
  const int xa = kq * 64 + l15 * 4;
  const int yb = kq * 512 + (wave * 16 + l15) * 4;
  const int row_pitch = Wp * 64;

  f32x4 acc[KK];
#pragma unroll
  for (int tap = 0; tap < KK; ++tap)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[tap][r] = 0.f;

  auto issue = [&](int64_t blk, int buf) {
    const int64_t b = blk / nblk;
    const int m = (int)(blk - b * nblk) * KC;
    const unsigned char *xg = X.base + b * X.batch_stride + (int64_t)cc * X.chunk_stride + (int64_t)m * 64 + lane * 16;
    const unsigned char *yg = Y.base + b * Y.batch_stride + (y_lead + m) * 512 + lane * 16;
    for (int w = wave; w < nx + ny; w += 8) {
      if (w < nx)
        lds_dma16(xg + ((int64_t)w << 10), xs + buf * xbytes + (w << 10));
      else
        lds_dma16(yg + ((int64_t)(w - nx) << 10), ys + buf * YB + ((w - nx) << 10));
    }
  };

  if (blk0 < blk1) issue(blk0, 0);
  int buf = 0;
  for (int64_t blk = blk0; blk < blk1; ++blk, buf ^= 1) {
    // the compiler drains this wave's DMA (vmcnt 0) ahead of the barrier: past it the block has landed for every
    // wave, and every wave is done with the other buffer
    __syncthreads();
    if (blk + 1 < blk1) issue(blk + 1, buf ^ 1);
    const unsigned char *xb = xs + buf * xbytes + xa;
    const unsigned char *ybp = ys + buf * YB + yb;
    // Rows of work = (K step s4, tap row i), KS MFMAs each.  Fully unrolled with two explicit register sets: the
    // LDS reads of row r + 1 are issued ahead of the MFMAs of row r and land in the OTHER set, so the wait the
    // compiler puts in front of a row's MFMAs only covers reads issued a whole row (KS MFMAs) earlier.
    constexpr int ROWS = (KC / 4) * KS;
    float abuf[2][KS], bbuf[2];
#pragma unroll
    for (int j = 0; j < KS; ++j) abuf[0][j] = *reinterpret_cast<const float *>(xb + j * 64);
    bbuf[0] = *reinterpret_cast<const float *>(ybp);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const int s4 = r / KS, i = r - s4 * KS, cur = r & 1;
      if (r + 1 < ROWS) {
        const int s4n = (r + 1) / KS, in = (r + 1) - s4n * KS;
        const unsigned char *nx_row = xb + s4n * 256 + in * row_pitch;
#pragma unroll
        for (int j = 0; j < KS; ++j) abuf[cur ^ 1][j] = *reinterpret_cast<const float *>(nx_row + j * 64);
        if (in == 0) bbuf[s4n & 1] = *reinterpret_cast<const float *>(ybp + s4n * 2048);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs they overlap with
#pragma unroll
      for (int j = 0; j < KS; ++j)
        acc[i * KS + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(abuf[cur][j], bbuf[s4 & 1], acc[i * KS + j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // C/D layout of the 16x16 MFMA: column = lane & 15, row = 4 * (lane >> 4) + r
  float *o = part + (((int64_t)sp * KK) * cpad + cc * kFcChunk) * kFcHidden + wave * 16 + l15;
#pragma unroll
  for (int tap = 0; tap < KK; ++tap)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[((int64_t)tap * cpad + 4 * kq + r) * kFcHidden] = acc[tap][r];
}

int fc_wgrad_splits(int64_t B, int Mk, int cpad) {
  const int nch = cpad / kFcChunk;
  const int64_t total = B * ceil_div(Mk, kWgKC);
  int64_t s = tuning(12) > 0 ? tuning(12) : (2 * kNumCU) / nch;
  if (s < 1) s = 1;
  return (int)(s > total ? total : s);
}

// part: fc_wgrad_splits(...) * k*k * cpad * 128 floats.  X: packed f32 records (mode 0), Y: the f32 (B, Sz, 128) map
int fc_wgrad_f32(const PackedDesc &X, const PackedDesc &Y, int64_t y_lead, float *part, int cpad, int64_t B, int Mk,
                 int Wp, int k, hipStream_t stream) {
  if (k != 3 && k != 5) return GFLA_ERR_UNSUPPORTED;
  if (X.pix_stride != 64 || Y.pix_stride != kFcHidden * 4 || Y.chunk_stride != 64) return GFLA_ERR_UNSUPPORTED;
  if (B <= 0) return GFLA_OK;
  const int nblk = (int)ceil_div(Mk, kWgKC);
  const int nsplit = fc_wgrad_splits(B, Mk, cpad);
  const int xh = kWgKC + (k - 1) * (Wp + 1);
  const unsigned lds = (unsigned)(2 * ((((xh * 4 + 63) >> 6) << 10) + kWgKC * kFcHidden * 4));
  if (lds > 156 * 1024) return GFLA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)(cpad / kFcChunk), (unsigned)nsplit);
#define GFLA_WGF(K_)                                                                                                 \
  {                                                                                                                  \
    auto kern = fc_wgrad_f32_kernel<K_>;                                                                             \
    if (lds > 64 * 1024)                                                                                             \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    kern<<<grid, 512, lds, stream>>>(X, Y, y_lead, part, cpad, Wp, nblk, B * nblk, nsplit);                           \
  }
  if (k == 3) GFLA_WGF(3) else GFLA_WGF(5)
#undef GFLA_WGF
  return launch_status();
}

// conv0.weight.grad[n][c_off + c][i][j] = sum_s part[s][tap][c][n]  (one half of the 2C input channels).
// blockIdx.y = job: both halves of a layer in one launch (fc_wgrad_reduce2).
struct WgRedJob {
  const float *part;
  int nsplit, c_off;
};
struct WgRedJobs {
  WgRedJob j[2];
};
__global__ __launch_bounds__(256) void fc_wgrad_reduce_kernel(WgRedJobs jobs, float *__restrict__ gw, int C, int cpad, int KK) {
  const WgRedJob &J = jobs.j[blockIdx.y];
  const float *__restrict__ part = J.part;
  const int nsplit = J.nsplit, c_off = J.c_off;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (tap, c, n), n fastest: coalesced reads
  const int64_t per = (int64_t)KK * cpad * kFcHidden;
  if (idx >= per) return;
  const int n = (int)(idx & (kFcHidden - 1));
  const int c = (int)((idx >> 7) % cpad);
  const int tap = (int)(idx / ((int64_t)cpad * kFcHidden));
  if (c >= C) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int sp = 0;
  for (; sp + 4 <= nsplit; sp += 4) {
    s0 += part[(sp + 0) * per + idx];
    s1 += part[(sp + 1) * per + idx];
    s2 += part[(sp + 2) * per + idx];
    s3 += part[(sp + 3) * per + idx];
  }
  for (; sp < nsplit; ++sp) s0 += part[sp * per + idx];
  gw[((int64_t)n * 2 * C + c_off + c) * KK + tap] = (s0 + s1) + (s2 + s3);
}

int fc_wgrad_reduce(const float *part, int nsplit, float *grad_w0, int C, int c_off, int cpad, int k,
                    hipStream_t stream) {
  const int64_t per = (int64_t)k * k * cpad * kFcHidden;
  WgRedJobs jobs;
  jobs.j[0] = jobs.j[1] = WgRedJob{part, nsplit, c_off};
  fc_wgrad_reduce_kernel<<<dim3((unsigned)ceil_div(per, 256), 1), 256, 0, stream>>>(jobs, grad_w0, C, cpad, k * k);
  return launch_status();
}

// source half (channels C..2C-1) and target half (0..C-1) in one launch
int fc_wgrad_reduce2(const float *part_s, int nsplit_s, const float *part_t, int nsplit_t, float *grad_w0, int C, int cpad,
                     int k, hipStream_t stream) {
  const int64_t per = (int64_t)k * k * cpad * kFcHidden;
  WgRedJobs jobs;
  jobs.j[0] = WgRedJob{part_s, nsplit_s, C};
  jobs.j[1] = WgRedJob{part_t, nsplit_t, 0};
  fc_wgrad_reduce_kernel<<<dim3((unsigned)ceil_div(per, 256), 2), 256, 0, stream>>>(jobs, grad_w0, C, cpad, k * k);
  return launch_status();
}

// conv0.weight.grad (128, 2C, k, k) from the two accumulators [tap][cpad][128], undoing the operand scales
__global__ __launch_bounds__(256) void fc_unpack_wgrad_kernel(const float *__restrict__ dw_t,
                                                             const float *__restrict__ dw_s,
                                                             const uint32_t *amax_xt, const uint32_t *amax_xs,
                                                             const uint32_t *amax_zt, const uint32_t *amax_zs,
                                                             float *__restrict__ gw, int C, int cpad, int k) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int KK = k * k;
  const int64_t total = (int64_t)kFcHidden * 2 * C * KK;
  if (idx >= total) return;
  const int tap = (int)(idx % KK);
  const int c2 = (int)((idx / KK) % (2 * C));
  const int n = (int)(idx / ((int64_t)KK * 2 * C));
  const bool src = c2 >= C;
  const int c = src ? c2 - C : c2;
  const float *acc = src ? dw_s : dw_t;
  const float inv = src ? fc_inv_scale(amax_xs) * fc_inv_scale(amax_zs) : fc_inv_scale(amax_xt) * fc_inv_scale(amax_zt);
  gw[idx] = acc[((int64_t)tap * cpad + c) * kFcHidden + n] * inv;
}

int fc_unpack_wgrad(const float *dw_t, const float *dw_s, const uint32_t *amax_xt, const uint32_t *amax_xs,
                    const uint32_t *amax_zt, const uint32_t *amax_zs, float *grad_w0, int C, int cpad, int k,
                    hipStream_t stream) {
  const int64_t total = (int64_t)kFcHidden * 2 * C * k * k;
  fc_unpack_wgrad_kernel<<<dim3((unsigned)ceil_div(total, 256)), 256, 0, stream>>>(dw_t, dw_s, amax_xt, amax_xs,
                                                                                   amax_zt, amax_zs, grad_w0, C, cpad, k);
  return launch_status();
}

}  // namespace gfla
