// The non-GEMM part of ExtractorAttn's fully_connect_layer in the "sample the convolved map" formulation
// (fc_gemm.hip), gfx950.
//
// Reference (base_function.py:799-807): hidden = Conv2d(2C,128,k,stride k)(cat(block_target, block_source)),
// logits = Conv2d(128, k*k, 1)(nonlinearity(hidden)).  With Gs = conv_kxk(extended source, W[:, C:]) and
// Gt = conv_kxk(padded target, W[:, :C]) (both (pixel, 128) f32 maps out of fc_conv):
//   forward : hidden[p, n] = b0[n] + Gt[p, n] + sum_{4 corners} w_corner(p) * Gs[q(p) + corner, n]
//             with q = floor(p + flow(p)), the reference's corner weights / clamped indices
//             (block_extractor_kernel.cu:66-76, applied to the convolved map instead of to every tap), then
//             LeakyReLU and the 1x1 convolution -> logits (B, k*k, H, W).
//   backward: d hidden = (W1^T d logits) * lrelu'(hidden); it IS the gradient of Gt (written into the zero-bordered
//             "Z layout" the transposed convolution wants), is scattered with the 4 corner weights into the gradient
//             of Gs (coalesced 128-channel atomics), and d flow = sum_n d hidden[n] * d/d(x,y) of the bilinear mix.
// Work decomposition: a workgroup = 64 positions of one sample.  Sampling / scattering phases put the 128 hidden
// channels on the lanes (a tap is 512 contiguous bytes), the 1x1 convolution puts the positions on the lanes and
// deals the channels to the 4 waves (as fc_tail.hip); a (64 x 128) tile in LDS turns one into the other.
#include "fc_gemm.h"
#include "lds_plane.h"

namespace gfla {

constexpr int kSmpPix = 64;
constexpr int kSmpPitch = kFcHidden + 1;  // floats; odd pitch: column reads (lanes = positions) are conflict-free

struct Corner {
  int i00, i01, i10, i11;   // indices into the convolved map (row pitch wps)
  int z00, z01, z10, z11;   // the same four positions in the Z-layout gradient map (row pitch wpz)
  float xl, xr, yt, yb;     // the reference's xL_P, xR_P, yT_P, yB_P
};

// block_extractor_kernel.cu:58-70 for the centre tap; the convolved map lives on [-hi, H-1+lo] x [-hi, W-1+lo]
template <int KS>
__device__ __forceinline__ Corner corners(float fx, float fy, int x, int y, int H, int W, int wps, int wpz = 0) {
  constexpr int LO = KS / 2, HI = KS - 1 - LO;
  const float dx = fx + (float)x, dy = fy + (float)y;
  const float fdx = floorf(dx), fdy = floorf(dy);
  Corner c;
  c.xr = dx - fdx;
  c.xl = 1.f - c.xr;
  c.yb = dy - fdy;
  c.yt = 1.f - c.yb;
  // float clamp first: keeps the int conversion defined for huge / non-finite flows
  const float cx = fminf(fmaxf(fdx, -(float)(HI + 1)), (float)(W + LO));
  const float cy = fminf(fmaxf(fdy, -(float)(HI + 1)), (float)(H + LO));
  const int qx = (int)cx, qy = (int)cy;
  const int gx0 = clampi(qx, -HI, W - 1 + LO) + HI, gx1 = clampi(qx + 1, -HI, W - 1 + LO) + HI;
  const int gy0 = clampi(qy, -HI, H - 1 + LO) + HI, gy1 = clampi(qy + 1, -HI, H - 1 + LO) + HI;
  c.i00 = gy0 * wps + gx0;
  c.i01 = gy0 * wps + gx1;
  c.i10 = gy1 * wps + gx0;
  c.i11 = gy1 * wps + gx1;
  c.z00 = gy0 * wpz + gx0;
  c.z01 = gy0 * wpz + gx1;
  c.z10 = gy1 * wpz + gx0;
  c.z11 = gy1 * wpz + gx1;
  return c;
}

__device__ __forceinline__ float lrelu_f(float v, float slope) { return v > 0.f ? v : v * slope; }

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
// The same sum, valid in LANE 63 ONLY, as six data-parallel-primitive adds: quad swaps, row mirrors, then the row totals
// handed down the wave (row_bcast:15 / :31).  __shfl_xor compiles to ds_bpermute_b32 -- an LDS round trip per step, twelve
// dependent ones per walked position in fc_tail_bwd_kernel.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = dpp_add<0xB1>(v);        // quad_perm:[1,0,3,2]
  v = dpp_add<0x4E>(v);        // quad_perm:[2,3,0,1]
  v = dpp_add<0x141>(v);       // row_half_mirror
  v = dpp_add<0x140>(v);       // row_mirror: every lane of a row holds the row's sum
  v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

template <int KS>
__global__ __launch_bounds__(256) void fc_tail_fwd_kernel(const float *__restrict__ gs, const float *__restrict__ gt,
                                                         const float *__restrict__ flow, const float *__restrict__ b0,
                                                         const float *__restrict__ w1, const float *__restrict__ b1,
                                                         float *__restrict__ hid, float *__restrict__ logits, int H,
                                                         int W, int64_t gs_bs, int64_t gt_bs, int wps, int wpt,
                                                         float slope) {
  constexpr int KK = KS * KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  float *tile = reinterpret_cast<float *>(gfla_smem);  // [64][129] hidden pre-activations
  float *w_s = tile + kSmpPix * kSmpPitch;             // [128][KK]
  float *red = tile;                                   // [4][KK][64] partial logits, once the tile is dead
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int HW = H * W;
  const int64_t b = blockIdx.y;
  const int p0 = blockIdx.x * kSmpPix;
  for (int i = t; i < kFcHidden * KK; i += 256) {
    const int o = i / KK, q = i - o * KK;
    w_s[i] = w1[q * kFcHidden + o];
  }
  // phase A: lanes = hidden channels (lane, lane + 64); a wave walks 16 positions
  const float *gsb = gs + b * gs_bs, *gtb = gt + b * gt_bs;
  const float bias0 = b0 ? b0[lane] : 0.f, bias1 = b0 ? b0[lane + 64] : 0.f;
#pragma unroll 4
  for (int it = 0; it < kSmpPix / 4; ++it) {
    const int pp = wave + 4 * it, p = p0 + pp;
    float h0 = 0.f, h1 = 0.f;
    if (p < HW) {
      const int y = p / W, x = p - y * W;
      const float fx = flow[(b * 2 + 0) * HW + p], fy = flow[(b * 2 + 1) * HW + p];
      const Corner c = corners<KS>(fx, fy, x, y, H, W, wps);
      const float *g00 = gsb + (int64_t)c.i00 * kFcHidden + lane, *g01 = gsb + (int64_t)c.i01 * kFcHidden + lane;
      const float *g10 = gsb + (int64_t)c.i10 * kFcHidden + lane, *g11 = gsb + (int64_t)c.i11 * kFcHidden + lane;
      const float *tp = gtb + (int64_t)(y * wpt + x) * kFcHidden + lane;
      const float wa = c.xl * c.yt, wb = c.xr * c.yt, wc = c.xl * c.yb, wd = c.xr * c.yb;
      h0 = bias0 + tp[0] + (wa * g00[0] + wb * g01[0] + wc * g10[0] + wd * g11[0]);
      h1 = bias1 + tp[64] + (wa * g00[64] + wb * g01[64] + wc * g10[64] + wd * g11[64]);
      float *hp = hid + (b * HW + p) * kFcHidden + lane;
      hp[0] = h0;
      hp[64] = h1;
    }
    tile[pp * kSmpPitch + lane] = h0;
    tile[pp * kSmpPitch + lane + 64] = h1;
  }
  __syncthreads();
  // phase B: lanes = positions, wave s takes hidden channels s, s+4, ...
  float acc[KK];
#pragma unroll
  for (int q = 0; q < KK; ++q) acc[q] = 0.f;
#pragma unroll 4
  for (int o = wave; o < kFcHidden; o += 4) {
    const float a = lrelu_f(tile[lane * kSmpPitch + o], slope);
    const float *w = w_s + o * KK;
#pragma unroll
    for (int q = 0; q < KK; ++q) acc[q] = fmaf(w[q], a, acc[q]);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < KK; ++q) red[(wave * KK + q) * kSmpPix + lane] = acc[q];
  __syncthreads();
  float *lg = logits + b * (int64_t)KK * HW;
  for (int i = t; i < KK * kSmpPix; i += 256) {
    const int q = i / kSmpPix, l = i - q * kSmpPix;
    const int pq = p0 + l;
    if (pq >= HW) continue;
    float v = b1 ? b1[q] : 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) v += red[(s * KK + q) * kSmpPix + l];
    lg[(int64_t)q * HW + pq] = v;
  }
}

// b0_partials: one row of 128 per workgroup (sum of d hidden over its positions); the caller adds the rows up.
template <int KS>
__global__ __launch_bounds__(256) void fc_tail_bwd_kernel(
    const float *__restrict__ gs, const float *__restrict__ flow, const float *__restrict__ hid,
    const float *__restrict__ w1, const float *__restrict__ g_logits, float *__restrict__ dzs,
    float *__restrict__ dzt, float *__restrict__ gflow, float *__restrict__ b0_partials, int H, int W, int64_t gs_bs,
    int wps, int wpz, int wpt, int64_t zs_bs, int64_t zt_bs, int lead_s, int lead_t, float slope, int acc_flow,
    uint32_t *__restrict__ amax_d) {
  constexpr int KK = KS * KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  float *tile = reinterpret_cast<float *>(gfla_smem);  // [64][129]: hidden pre-activations, then their gradient
  float *w_s = tile + kSmpPix * kSmpPitch;             // [128][KK]
  float *bsum = w_s + kFcHidden * KK;                  // [4][128]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int HW = H * W;
  const int64_t b = blockIdx.y;
  const int p0 = blockIdx.x * kSmpPix;
  for (int i = t; i < kFcHidden * KK; i += 256) {
    const int o = i / KK, q = i - o * KK;
    w_s[i] = w1[q * kFcHidden + o];
  }
  for (int i = t; i < kSmpPix * kFcHidden; i += 256) {
    const int pp = i >> 7, n = i & 127;
    tile[pp * kSmpPitch + n] = p0 + pp < HW ? hid[(b * HW + p0 + pp) * kFcHidden + n] : 0.f;
  }
  __syncthreads();
  {  // phase 1: lanes = positions: d hidden = (W1^T d logits) * lrelu'(hidden)
    const int p = p0 + lane;
    const bool live = p < HW;
    float gl[KK];
    const float *glp = g_logits + b * (int64_t)KK * HW + (live ? p : 0);
#pragma unroll
    for (int q = 0; q < KK; ++q) gl[q] = live ? glp[(int64_t)q * HW] : 0.f;
    uint32_t mx = 0;   // max |d hidden| of this lane (bit pattern: a NaN compares above every finite value)
#pragma unroll 4
    for (int o = wave; o < kFcHidden; o += 4) {
      const float pre = tile[lane * kSmpPitch + o];
      const float *w = w_s + o * KK;
      float ga = 0.f;
#pragma unroll
      for (int q = 0; q < KK; ++q) ga = fmaf(w[q], gl[q], ga);
      const float gp = pre > 0.f ? ga : ga * slope;
      tile[lane * kSmpPitch + o] = gp;
      mx = max(mx, __float_as_uint(gp) & 0x7fffffffu);
    }
    if (amax_d) {   // the scale of the gradient map's fixed-point scatter (fc_scatter_own_kernel) and of its two-term f16 split
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, m));
      // one atomic per wave, and only when it would raise the slot (a stale read costs an atomic, never a wrong maximum)
      if (lane == 0 && mx > __hip_atomic_load(amax_d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax_d, mx);
    }
  }
  __syncthreads();
  // phase 2: lanes = hidden channels (lane, lane + 64); a wave walks 16 CONSECUTIVE positions.  The right-hand corners
  // of position p are the left-hand corners of p + 1 whenever the flow moves both by the same whole step (the common
  // case for a smooth flow), so the wave keeps the right column's two sums pending and folds the next position's left
  // column into them: 4 atomics per position and channel half instead of 8.  (Indices are wave-uniform: no divergence.)
  const float *gsb = gs + b * gs_bs;
  float *zsb = dzs ? dzs + b * zs_bs : nullptr;
  float *ztb = dzt ? dzt + b * zt_bs : nullptr;
  float s0 = 0.f, s1 = 0.f;
  int pend_t = -1, pend_b = -1;                // Z-layout indices of the pending column (-1: none)
  float pt0 = 0.f, pt1 = 0.f, pb0 = 0.f, pb1 = 0.f;  // its sums: (top, bottom) x (channel lane, lane + 64)
  // The walk is a chain of dependent global accesses per position (flow -> corner addresses -> the four corners of the
  // convolved map); it is software-pipelined: the flows of the wave's 16 positions are fetched up front (lane i holds
  // position i's), and the corner values of position it+1 are in flight while position it is processed.
  constexpr int kWalk = kSmpPix / 4;
  const int pw0 = p0 + wave * kWalk;
  const int nwalk = max(0, min(kWalk, HW - pw0));
  float fx_l = 0.f, fy_l = 0.f;
  if ((lane & (kWalk - 1)) < nwalk) {
    fx_l = flow[(b * 2 + 0) * HW + pw0 + (lane & (kWalk - 1))];
    fy_l = flow[(b * 2 + 1) * HW + pw0 + (lane & (kWalk - 1))];
  }
  Corner cn;
  float gn[8];
  auto issue = [&](int it) {
    const int p = pw0 + it;
    const int y = p / W, x = p - y * W;
    const float fx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fx_l), it));
    const float fy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fy_l), it));
    cn = corners<KS>(fx, fy, x, y, H, W, wps, wpz);
    if (gflow) {
      const float *g00 = gsb + (int64_t)cn.i00 * kFcHidden + lane, *g01 = gsb + (int64_t)cn.i01 * kFcHidden + lane;
      const float *g10 = gsb + (int64_t)cn.i10 * kFcHidden + lane, *g11 = gsb + (int64_t)cn.i11 * kFcHidden + lane;
      gn[0] = g00[0], gn[1] = g01[0], gn[2] = g10[0], gn[3] = g11[0];
      gn[4] = g00[64], gn[5] = g01[64], gn[6] = g10[64], gn[7] = g11[64];
    }
  };
  if (!zsb) {
    // The gradient map of Gs is somebody else's (fc_scatter_own_kernel): nothing is carried from position to position, so the
    // corner values of FOUR positions are in flight at a time, and the flow gradients of the wave's 16 positions are
    // collected in lanes 0-15 and leave through one store each (the read-modify-write of an accumulated gradient used to sit
    // in lane 63 of every iteration: a dependent global round trip per position, 16 in a row)
    float old_x = 0.f, old_y = 0.f, res_x = 0.f, res_y = 0.f;
    if (gflow && acc_flow && lane < nwalk) {
      old_x = gflow[(b * 2 + 0) * HW + pw0 + lane];
      old_y = gflow[(b * 2 + 1) * HW + pw0 + lane];
    }
    for (int it0 = 0; it0 < nwalk; it0 += 4) {
      Corner c4[4];
      float g4[4][8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        issue(min(it0 + j, nwalk - 1));
        c4[j] = cn;
#pragma unroll
        for (int i = 0; i < 8; ++i) g4[j][i] = gn[i];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int it = it0 + j;
        if (it >= nwalk) break;
        const int pp = wave * kWalk + it, p = p0 + pp;
        const Corner &c = c4[j];
        const float *g = g4[j];
        const float d0 = tile[pp * kSmpPitch + lane], d1 = tile[pp * kSmpPitch + lane + 64];
        s0 += d0;
        s1 += d1;
        if (ztb) {
          const int y = p / W, x = p - y * W;
          float *zp = ztb + (int64_t)(lead_t + y * wpt + x) * kFcHidden + lane;
          zp[0] = d0;
          zp[64] = d1;
        }
        if (gflow) {
          float gx = d0 * (c.yt * (g[1] - g[0]) + c.yb * (g[3] - g[2])) + d1 * (c.yt * (g[5] - g[4]) + c.yb * (g[7] - g[6]));
          float gy = d0 * (c.xl * (g[2] - g[0]) + c.xr * (g[3] - g[1])) + d1 * (c.xl * (g[6] - g[4]) + c.xr * (g[7] - g[5]));
          gx = wave_sum_lane63(gx);
          gy = wave_sum_lane63(gy);
          const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gx), 63));
          const float sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gy), 63));
          if (lane == it) res_x = sx, res_y = sy;
        }
      }
    }
    if (gflow && lane < nwalk) {  // this workgroup is the only writer of its pixels
      gflow[(b * 2 + 0) * HW + pw0 + lane] = acc_flow ? old_x + res_x : res_x;
      gflow[(b * 2 + 1) * HW + pw0 + lane] = acc_flow ? old_y + res_y : res_y;
    }
  }
  if (zsb && nwalk > 0) issue(0);
  for (int it = 0; zsb && it < nwalk; ++it) {
    const int pp = wave * kWalk + it, p = p0 + pp;
    const Corner c = cn;
    float g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = gn[i];
    if (it + 1 < nwalk) issue(it + 1);
    const float d0 = tile[pp * kSmpPitch + lane], d1 = tile[pp * kSmpPitch + lane + 64];
    s0 += d0;
    s1 += d1;
    if (ztb) {
      const int y = p / W, x = p - y * W;
      float *zp = ztb + (int64_t)(lead_t + y * wpt + x) * kFcHidden + lane;
      zp[0] = d0;
      zp[64] = d1;
    }
    if (zsb) {
      const float wa = c.xl * c.yt, wb = c.xr * c.yt, wc = c.xl * c.yb, wd = c.xr * c.yb;
      float lt0 = wa * d0, lt1 = wa * d1, lb0 = wc * d0, lb1 = wc * d1;  // this position's left column
      if (pend_t == c.z00 && pend_b == c.z10) {
        lt0 += pt0, lt1 += pt1, lb0 += pb0, lb1 += pb1;
      } else if (pend_t >= 0) {
        float *zt = zsb + (int64_t)(lead_s + pend_t) * kFcHidden + lane, *zb = zsb + (int64_t)(lead_s + pend_b) * kFcHidden + lane;
        atomic_add(zt, pt0); atomic_add(zt + 64, pt1);
        atomic_add(zb, pb0); atomic_add(zb + 64, pb1);
      }
      float *z00 = zsb + (int64_t)(lead_s + c.z00) * kFcHidden + lane, *z10 = zsb + (int64_t)(lead_s + c.z10) * kFcHidden + lane;
      atomic_add(z00, lt0); atomic_add(z00 + 64, lt1);
      atomic_add(z10, lb0); atomic_add(z10 + 64, lb1);
      pend_t = c.z01, pend_b = c.z11;
      pt0 = wb * d0, pt1 = wb * d1, pb0 = wd * d0, pb1 = wd * d1;
    }
    if (gflow) {
      // block_extractor_kernel.cu:160-161 with the convolved map in place of the source plane
      float gx = d0 * (c.yt * (g[1] - g[0]) + c.yb * (g[3] - g[2])) + d1 * (c.yt * (g[5] - g[4]) + c.yb * (g[7] - g[6]));
      float gy = d0 * (c.xl * (g[2] - g[0]) + c.xr * (g[3] - g[1])) + d1 * (c.xl * (g[6] - g[4]) + c.xr * (g[7] - g[5]));
      gx = wave_sum_lane63(gx);
      gy = wave_sum_lane63(gy);
      if (lane == 63) {  // this workgroup is the only writer of its pixels
        float *fxp = gflow + (b * 2 + 0) * HW + p, *fyp = gflow + (b * 2 + 1) * HW + p;
        *fxp = acc_flow ? *fxp + gx : gx;
        *fyp = acc_flow ? *fyp + gy : gy;
      }
    }
  }
  if (zsb && pend_t >= 0) {
    float *zt = zsb + (int64_t)(lead_s + pend_t) * kFcHidden + lane, *zb = zsb + (int64_t)(lead_s + pend_b) * kFcHidden + lane;
    atomic_add(zt, pt0); atomic_add(zt + 64, pt1);
    atomic_add(zb, pb0); atomic_add(zb + 64, pb1);
  }
  if (b0_partials) {
    bsum[wave * kFcHidden + lane] = s0;
    bsum[wave * kFcHidden + lane + 64] = s1;
    __syncthreads();
    if (t < kFcHidden)
      b0_partials[(b * gridDim.x + blockIdx.x) * kFcHidden + t] =
          (bsum[t] + bsum[kFcHidden + t]) + (bsum[2 * kFcHidden + t] + bsum[3 * kFcHidden + t]);
  }
}

static int smp_check(int64_t B, int H, int W, int k) {
  if (B < 0 || H <= 0 || W <= 0) return GFLA_ERR_BAD_SHAPE;
  if (k != 3 && k != 5) return GFLA_ERR_UNSUPPORTED;
  if (B > 65535 || (int64_t)H * W > 0x3fffffffLL) return GFLA_ERR_UNSUPPORTED;
  return GFLA_OK;
}

int fc_sample_tail_fwd(const float *gs, const float *gt, const float *flow, const float *b0, const float *w1,
                       const float *b1, float *hid, float *logits, int64_t B, int H, int W, int k, int64_t gs_bs,
                       int64_t gt_bs, int wps, int wpt, float slope, hipStream_t stream) {
  if (!gs || !gt || !flow || !w1 || !hid || !logits) return GFLA_ERR_NULL_POINTER;
  if (int rc = smp_check(B, H, W, k)) return rc;
  if (B == 0) return GFLA_OK;
  const dim3 grid((unsigned)ceil_div((int64_t)H * W, kSmpPix), (unsigned)B);
  const unsigned lds = (unsigned)((kSmpPix * kSmpPitch + kFcHidden * k * k) * sizeof(float));
  if (k == 3)
    fc_tail_fwd_kernel<3><<<grid, 256, lds, stream>>>(gs, gt, flow, b0, w1, b1, hid, logits, H, W, gs_bs, gt_bs, wps, wpt, slope);
  else
    fc_tail_fwd_kernel<5><<<grid, 256, lds, stream>>>(gs, gt, flow, b0, w1, b1, hid, logits, H, W, gs_bs, gt_bs, wps, wpt, slope);
  return launch_status();
}

int fc_sample_tail_bwd(const float *gs, const float *flow, const float *hid, const float *w1, const float *g_logits,
                       float *dzs, float *dzt, float *gflow, float *b0_partials, int64_t B, int H, int W, int k,
                       int64_t gs_bs, int wps, int wpz, int wpt, int64_t zs_bs, int64_t zt_bs, int lead_s, int lead_t,
                       float slope, int acc_flow, hipStream_t stream, uint32_t *amax_d) {
  if (!gs || !flow || !hid || !w1 || !g_logits) return GFLA_ERR_NULL_POINTER;
  if (int rc = smp_check(B, H, W, k)) return rc;
  if (B == 0) return GFLA_OK;
  const dim3 grid((unsigned)ceil_div((int64_t)H * W, kSmpPix), (unsigned)B);
  const unsigned lds = (unsigned)((kSmpPix * kSmpPitch + kFcHidden * k * k + 4 * kFcHidden) * sizeof(float));
  if (k == 3)
    fc_tail_bwd_kernel<3><<<grid, 256, lds, stream>>>(gs, flow, hid, w1, g_logits, dzs, dzt, gflow, b0_partials, H, W,
                                                       gs_bs, wps, wpz, wpt, zs_bs, zt_bs, lead_s, lead_t, slope, acc_flow, amax_d);
  else
    fc_tail_bwd_kernel<5><<<grid, 256, lds, stream>>>(gs, flow, hid, w1, g_logits, dzs, dzt, gflow, b0_partials, H, W,
                                                       gs_bs, wps, wpz, wpt, zs_bs, zt_bs, lead_s, lead_t, slope, acc_flow, amax_d);
  return launch_status();
}

// ------------------------------------------------------------------ d Gs without global atomics (round 6)
// fc_tail_bwd_kernel scatters d hidden into the gradient of the convolved source map with 4 coalesced 128-channel float
// atomics per position: 46 M lane-atomics at B = 32, 64x44 -- and global float atomics run at 250 G lane-ops/s whatever
// their scope or contention (profiles/r5_ubench_global_atomics_scope_and_xcc_id.txt): 184 of the kernel's 195 us.  Here the
// OUTPUT is dealt out instead: a workgroup owns R whole rows of one sample's gradient map (R x Wo cells x 128 channels), finds
// the positions whose corners land in its rows -- every workgroup of a sample walks all H W positions of the flow field, 512
// at a time: corner geometry is ~40 instructions per position, and an arbitrary flow needs no special case, only more list
// entries --, reads their d hidden rows (the target half's gradient map: d hidden IS d Gt, written by fc_tail_bwd_kernel
// a moment earlier) with the 128 channels on the lanes, and adds the weighted rows into its cells in LDS: 64-bit FIXED POINT
// (lds_plane.h: ds_add_u64 at 5-8 lanes/clk/CU = 3-5 T lane-ops/s per chip, against 0.25 T for the global float atomics;
// scale = the power of two that puts max |d hidden| -- fc_tail_bwd_kernel's by-product -- at 2^40; integer sums are exactly
// associative: the map comes out bit-reproducible run to run, which float atomics never were).  The rows leave through ONE
// coalesced plain store, zero border of the "Z layout" included: the map needs no memset any more, and the maximum of its
// entries (the scale of the data-gradient convolution's two-term f16 split) falls out of the flush: no max |x| pass either.
constexpr int kOwnThreads = 512;
constexpr int kOwnCap = 1024;    // list entries (32 KB): all of a workgroup's positions in ONE round unless the flow is wild
constexpr int kOwnPD = 8;        // list entries whose d hidden rows a wave keeps in flight
constexpr int kOwnPre = 6;       // positions per thread whose flow values are requested up front (maps beyond 6 x 512: on demand)
struct OwnEntry {
  int zt;          // pixel index of the position in the target half's gradient map
  int row_t, row_b;  // word offset / 128 of the corner rows inside the tile (row * Wo), -1: not this workgroup's
  int gx;          // gx0 | gx1 << 16
  float xl, xr, yt, yb;
};

// 64-bit fixed point -> double by the magic-number add run backwards (|x| < 2^51), the plain conversion otherwise
__device__ __forceinline__ double fix_to_double(lds_fix_t x) {
  if (__builtin_expect(((unsigned long long)(x + (1ll << 51)) >> 52) != 0ull, 0)) return (double)x;
  return __longlong_as_double(x + 0x4338000000000000ll) - kFixMagic;
}

template <int KS>
__global__ __launch_bounds__(kOwnThreads) void fc_scatter_own_kernel(
    const float *__restrict__ flow, const float *__restrict__ dzt, float *__restrict__ dzs,
    const uint32_t *__restrict__ amax_d, uint32_t *__restrict__ amax_out, int H, int W, int Ho, int Wo, int wpz, int wpt,
    int64_t zs_bs, int64_t zt_bs, int lead_s, int lead_t, int64_t Sz, int R) {
  constexpr int LO = KS / 2, HI = KS - 1 - LO;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  lds_fix_t *tile = reinterpret_cast<lds_fix_t *>(gfla_smem);                              // [R][Wo][128]
  OwnEntry *list = reinterpret_cast<OwnEntry *>(tile + (size_t)R * Wo * kFcHidden);         // [kOwnCap]
  __shared__ int s_count, s_total;
  __shared__ unsigned s_max;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t b = blockIdx.y;
  const int r0 = blockIdx.x * R, rows = min(R, Ho - r0);
  const int HW = H * W;
  const float *fxp = flow + (b * 2 + 0) * HW, *fyp = flow + (b * 2 + 1) * HW;
  const float *dzb = dzt + b * zt_bs + lane;
  // every workgroup of a sample walks the whole flow field: its values are requested once, up front
  float fxq[kOwnPre], fyq[kOwnPre];
#pragma unroll
  for (int i = 0; i < kOwnPre; ++i) {
    const int p = min(i * kOwnThreads + t, HW - 1);
    fxq[i] = fxp[p], fyq[i] = fyp[p];
  }
  {
    longlong2 *t2 = reinterpret_cast<longlong2 *>(tile);
    for (int i = t; i < rows * Wo * (kFcHidden / 2); i += kOwnThreads) t2[i] = longlong2{0, 0};
  }
  if (t == 0) s_max = 0, s_total = 0, s_count = 0;
  const FixScale fs = fix_scale(*amax_d);
  const int npos = (HW + kOwnThreads - 1) / kOwnThreads;   // positions per thread: p = i * 512 + t
  // corner geometry of position i * 512 + t: block_extractor_kernel.cu:58-70 for the centre tap on the convolved map's
  // domain (as corners<KS>() above); false: none of its corners lands in this workgroup's rows
  auto evaluate = [&](int i, OwnEntry &e) -> bool {
    const int p = i * kOwnThreads + t;
    if (p >= HW) return false;
    float fx = 0.f, fy = 0.f;
    if (i < kOwnPre) {
#pragma unroll
      for (int q = 0; q < kOwnPre; ++q)
        if (q == i) fx = fxq[q], fy = fyq[q];
    } else {
      fx = fxp[p], fy = fyp[p];
    }
    const int y = p / W, x = p - y * W;
    const float dx = fx + (float)x, dy = fy + (float)y;
    const float fdx = floorf(dx), fdy = floorf(dy);
    const float cx = fminf(fmaxf(fdx, -(float)(HI + 1)), (float)(W + LO));
    const float cy = fminf(fmaxf(fdy, -(float)(HI + 1)), (float)(H + LO));
    const int qx = (int)cx, qy = (int)cy;
    const int gx0 = clampi(qx, -HI, W - 1 + LO) + HI, gx1 = clampi(qx + 1, -HI, W - 1 + LO) + HI;
    const int gy0 = clampi(qy, -HI, H - 1 + LO) + HI, gy1 = clampi(qy + 1, -HI, H - 1 + LO) + HI;
    const bool top = gy0 >= r0 && gy0 < r0 + rows, bot = gy1 >= r0 && gy1 < r0 + rows;
    e.zt = lead_t + y * wpt + x;
    e.row_t = top ? (gy0 - r0) * Wo : -1;
    e.row_b = bot ? (gy1 - r0) * Wo : -1;
    e.gx = gx0 | (gx1 << 16);
    e.xr = dx - fdx;
    e.xl = 1.f - e.xr;
    e.yb = dy - fdy;
    e.yt = 1.f - e.yb;
    return top || bot;
  };
  // how many positions reach this workgroup's rows?  A smooth flow: about (R + 1) rows of positions -- one list round;
  // otherwise rounds of kOwnCap / 512 positions per thread
  {
    int mine = 0;
    for (int i = 0; i < npos; ++i) {
      OwnEntry e;
      mine += evaluate(i, e) ? 1 : 0;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mine += __shfl_xor(mine, m);
    if (lane == 0 && mine) atomicAdd(&s_total, mine);
  }
  __syncthreads();   // (the tile is clear, the total is known)
  const int per_round = s_total <= kOwnCap ? npos : kOwnCap / kOwnThreads;
  for (int i0 = 0; i0 < npos; i0 += per_round) {
    for (int i = i0; i < min(npos, i0 + per_round); ++i) {
      OwnEntry e;
      const bool in = evaluate(i, e);
      // list slots: one LDS atomic per wave (returning atomics on ONE address run at half a lane per clock)
      const unsigned long long mask = __ballot(in);
      int base = 0;
      if (lane == 0 && mask) base = atomicAdd(&s_count, __popcll(mask));
      base = __builtin_amdgcn_readfirstlane(base);
      if (in) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = e;
    }
    __syncthreads();
    const int n = s_count;
    for (int e0 = wave; e0 < n; e0 += (kOwnThreads / 64) * kOwnPD) {
      OwnEntry en[kOwnPD];
      float d0[kOwnPD], d1[kOwnPD];
#pragma unroll
      for (int j = 0; j < kOwnPD; ++j) {
        en[j] = list[min(e0 + (kOwnThreads / 64) * j, n - 1)];   // (one address per wave: a broadcast read)
        const float *dp = dzb + (int64_t)__builtin_amdgcn_readfirstlane(en[j].zt) * kFcHidden;
        d0[j] = dp[0];
        d1[j] = dp[64];
      }
#pragma unroll
      for (int j = 0; j < kOwnPD; ++j) {
        if (e0 + (kOwnThreads / 64) * j >= n) break;
        const int row_t = __builtin_amdgcn_readfirstlane(en[j].row_t), row_b = __builtin_amdgcn_readfirstlane(en[j].row_b);
        const int gx = __builtin_amdgcn_readfirstlane(en[j].gx);
        const int gx0 = gx & 0xffff, gx1 = gx >> 16;
        const float s0 = d0[j] * fs.up, s1 = d1[j] * fs.up;   // (a power of two: the products below round as the float ones)
        if (row_t >= 0) {
          const float wl = en[j].xl * en[j].yt, wr = en[j].xr * en[j].yt;
          lds_fix_t *cl = tile + (size_t)(row_t + gx0) * kFcHidden + lane, *cr = tile + (size_t)(row_t + gx1) * kFcHidden + lane;
          lds_add_fix(cl, wl * s0), lds_add_fix(cl + 64, wl * s1);
          lds_add_fix(cr, wr * s0), lds_add_fix(cr + 64, wr * s1);
        }
        if (row_b >= 0) {
          const float wl = en[j].xl * en[j].yb, wr = en[j].xr * en[j].yb;
          lds_fix_t *cl = tile + (size_t)(row_b + gx0) * kFcHidden + lane, *cr = tile + (size_t)(row_b + gx1) * kFcHidden + lane;
          lds_add_fix(cl, wl * s0), lds_add_fix(cl + 64, wl * s1);
          lds_add_fix(cr, wr * s0), lds_add_fix(cr + 64, wr * s1);
        }
      }
    }
    __syncthreads();   // every wave is done with the list
    if (t == 0) s_count = 0;
    __syncthreads();
  }
  // flush: the workgroup's rows of the Z layout, pitch wpz, columns >= Wo zero, four channels per lane (16-byte stores); the
  // first / last workgroup of a sample also write the zero pixels ahead of / behind the map
  float *zb = dzs + b * zs_bs;
  const float nanv = __uint_as_float(0x7fc00000u);
  uint32_t mx = 0;
  float4 *out4 = reinterpret_cast<float4 *>(zb + (int64_t)(lead_s + r0 * wpz) * kFcHidden);
  for (int i = t; i < rows * wpz * (kFcHidden / 4); i += kOwnThreads) {
    int cell = i >> 5, r = 0;
    const int c4 = i & 31;
    while (cell >= wpz) cell -= wpz, ++r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cell < Wo) {
      const longlong2 *src = reinterpret_cast<const longlong2 *>(tile + (size_t)(r * Wo + cell) * kFcHidden + c4 * 4);
      const longlong2 a = src[0], c = src[1];
      if (fs.finite) {
        v.x = (float)(fix_to_double(a.x) * fs.down), v.y = (float)(fix_to_double(a.y) * fs.down);
        v.z = (float)(fix_to_double(c.x) * fs.down), v.w = (float)(fix_to_double(c.y) * fs.down);
      } else {
        v = make_float4(nanv, nanv, nanv, nanv);
      }
    }
    out4[i] = v;
    mx = max(max(mx, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
    mx = max(max(mx, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r0 == 0)
    for (int64_t i = t; i < (int64_t)lead_s * (kFcHidden / 4); i += kOwnThreads) reinterpret_cast<float4 *>(zb)[i] = z4;
  if (r0 + rows >= Ho)
    for (int64_t i = (int64_t)(lead_s + Ho * wpz) * (kFcHidden / 4) + t; i < Sz * (kFcHidden / 4); i += kOwnThreads)
      reinterpret_cast<float4 *>(zb)[i] = z4;
  if (amax_out) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, m));
    if (lane == 0) atomicMax(&s_max, mx);
    __syncthreads();
    if (t == 0 && s_max > __hip_atomic_load(amax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax_out, s_max);
  }
}

// rows of the gradient map one workgroup owns (0: a single row does not fit the LDS -- the caller keeps the atomics)
int fc_scatter_own_rows(int64_t B, int Ho, int Wo) {
  const int64_t budget = 160 * 1024 - 512 - (int64_t)kOwnCap * (int64_t)sizeof(OwnEntry);
  const int64_t row_bytes = (int64_t)Wo * kFcHidden * (int64_t)sizeof(lds_fix_t);
  int R = (int)std::min<int64_t>(budget / row_bytes, Ho);
  while (R > 1 && ceil_div(Ho, R) * B < 2 * kNumCU) --R;   // enough workgroups to fill the chip twice
  return R;
}

int fc_sample_scatter_own(const float *flow, const float *dzt, float *dzs, const uint32_t *amax_d, uint32_t *amax_out,
                          int64_t B, int H, int W, int k, int Ho, int Wo, int wpz, int wpt, int64_t zs_bs, int64_t zt_bs,
                          int lead_s, int lead_t, int64_t Sz, hipStream_t stream) {
  if (!flow || !dzt || !dzs || !amax_d) return GFLA_ERR_NULL_POINTER;
  if (int rc = smp_check(B, H, W, k)) return rc;
  if (B == 0) return GFLA_OK;
  const int R = fc_scatter_own_rows(B, Ho, Wo);
  if (R < 1 || Wo > 0x7fff) return GFLA_ERR_UNSUPPORTED;
  const unsigned lds = (unsigned)((size_t)R * Wo * kFcHidden * sizeof(lds_fix_t) + (size_t)kOwnCap * sizeof(OwnEntry));
  const dim3 grid((unsigned)ceil_div(Ho, R), (unsigned)B);
#define GFLA_OWN(K_)                                                                                                         \
  {                                                                                                                           \
    auto kern = fc_scatter_own_kernel<K_>;                                                                                    \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
    kern<<<grid, kOwnThreads, lds, stream>>>(flow, dzt, dzs, amax_d, amax_out, H, W, Ho, Wo, wpz, wpt, zs_bs, zt_bs, lead_s,  \
                                             lead_t, Sz, R);                                                                  \
  }
  if (k == 3) GFLA_OWN(3) else GFLA_OWN(5)
#undef GFLA_OWN
  return launch_status();
}

// ------------------------------------------------------------------ d W1 (k*k, 128) and d b1 (k*k)
// dW1[q][n] = sum_{b,p} g_logits[b,q,p] * lrelu(hidden[b,p,n]): a (32 x 128 x pixels) product on the f32 matrix
// cores (rows q >= k*k are zero).  A workgroup reduces a range of positions of one sample and writes one partial
// row of 32*128 + 32 floats (the last 32: sum_p g_logits[q]); fc_reduce_rows adds the rows.
constexpr int kDw1Row = 32 * kFcHidden + 32;

__global__ __launch_bounds__(256) void fc_dw1_kernel(const float *__restrict__ hid, const float *__restrict__ g_logits,
                                                    float *__restrict__ partials, int HW, int KK, int per, float slope) {
  __shared__ float gls[32][65];
  __shared__ float hs[64][kFcHidden];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, kk = lane >> 5;
  const int64_t b = blockIdx.y;
  const int p_begin = blockIdx.x * per, p_end = min(HW, p_begin + per);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  // the next 64-pixel step's operands are fetched into registers while the current step multiplies (one global round
  // trip per step was exposed before: 37 us for 55 MB)
  float gq[8], hq[32];
  auto fetch = [&](int pc) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = t + 256 * j, q = i >> 6, pp = i & 63;
      // RAW values: masks and the activation are applied when the registers are written to LDS (a select or a multiply
      // right behind a load makes the load synchronous, and this fetch is meant to fly under the MFMAs)
      gq[j] = g_logits[(b * KK + min(q, KK - 1)) * (int64_t)HW + min(pc + pp, HW - 1)];
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int i = t + 256 * j, pp = i >> 7, n = i & 127;
      hq[j] = hid[(b * HW + min(pc + pp, HW - 1)) * (int64_t)kFcHidden + n];
    }
  };
  if (p_begin < p_end) fetch(p_begin);
  for (int pc = p_begin; pc < p_end; pc += 64) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = t + 256 * j, q = i >> 6, pp = i & 63;
      gls[q][pp] = (q < KK && pc + pp < p_end) ? gq[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int i = t + 256 * j, pp = i >> 7;
      hs[pp][i & 127] = pc + pp < p_end ? lrelu_f(hq[j], slope) : 0.f;
    }
    __syncthreads();
    if (pc + 64 < p_end) fetch(pc + 64);
#pragma unroll 8
    for (int s = 0; s < 32; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gls[l31][2 * s + kk], hs[2 * s + kk][wave * 32 + l31], acc, 0, 0, 0);
    if (t < 32) {
      float v = 0.f;
#pragma unroll 8
      for (int pp = 0; pp < 64; ++pp) v += gls[t][pp];
      bsum += v;
    }
  }
  float *row = partials + (b * gridDim.x + blockIdx.x) * (int64_t)kDw1Row;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = (r & 3) + 8 * (r >> 2) + 4 * kk;
    row[q * kFcHidden + wave * 32 + l31] = acc[r];
  }
  if (t < 32) row[32 * kFcHidden + t] = bsum;
}

int fc_dw1(const float *hid, const float *g_logits, float *partials, int64_t B, int HW, int KK, int tiles_per_sample,
           float slope, hipStream_t stream) {
  if (!hid || !g_logits || !partials) return GFLA_ERR_NULL_POINTER;
  if (KK > 32 || KK <= 0 || tiles_per_sample <= 0) return GFLA_ERR_UNSUPPORTED;
  if (B <= 0) return GFLA_OK;
  const int per = (int)round_up(ceil_div(HW, tiles_per_sample), 64);
  fc_dw1_kernel<<<dim3((unsigned)tiles_per_sample, (unsigned)B), 256, 0, stream>>>(hid, g_logits, partials, HW, KK, per,
                                                                                   slope);
  return launch_status();
}

// out[c] = scale * sum_r partials[r][c].  Two passes when there are many rows: kRedSplits row ranges are summed by
// separate workgroups into tmp (kRedSplits x cols floats), then added up (a single pass over 1408 rows x 128
// columns ran on 2 workgroups: 40 us).
constexpr int kRedSplits = 32;

__global__ __launch_bounds__(256) void fc_reduce_rows_kernel(const float *__restrict__ partials, float *__restrict__ out,
                                                            int64_t rows, int cols, float scale) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int64_t r0 = rows * blockIdx.y / gridDim.y, r1 = rows * (blockIdx.y + 1) / gridDim.y;
  float s = 0.f;
  if (c < cols)
    for (int64_t r = r0 + slice; r < r1; r += 4) s += partials[r * cols + c];
  red[slice][lane] = s;
  __syncthreads();
  if (slice == 0 && c < cols)
    out[(int64_t)blockIdx.y * cols + c] = scale * ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}

int fc_reduce_rows(const float *partials, float *out, int64_t rows, int cols, float scale, float *tmp,
                   hipStream_t stream) {
  if (cols <= 0) return GFLA_OK;
  const unsigned gx = (unsigned)ceil_div(cols, 64);
  if (tmp && rows >= 4 * kRedSplits) {
    fc_reduce_rows_kernel<<<dim3(gx, kRedSplits), 256, 0, stream>>>(partials, tmp, rows, cols, 1.f);
    fc_reduce_rows_kernel<<<dim3(gx, 1), 256, 0, stream>>>(tmp, out, kRedSplits, cols, scale);
  } else {
    fc_reduce_rows_kernel<<<dim3(gx, 1), 256, 0, stream>>>(partials, out, rows, cols, scale);
  }
  return launch_status();
}

// Both row reductions of a backward pass -- d b0 (rows of 128) and d W1 | d b1 (rows of 32*128 + 32) -- as ONE two-pass
// launch pair that writes the three gradients in place: the step spent 8 reduce launches + 4 device copies of ~5 us (+ a
// ~4 us dependent-launch gap) each on them (profiles/r4_final_steady_state_steps.txt).
// job j: out segment A = columns [0, nA) -> dstA, segment B = columns [offB, offB + nB) -> dstB (other columns dropped).
struct RedJob {
  const float *partials;
  float *dstA, *dstB;
  int64_t rows;
  int cols, nA, offB, nB, gx;   // gx = column blocks of 64
};
struct RedJobs {
  RedJob j[2];
};
__global__ __launch_bounds__(256) void fc_reduce_jobs_kernel(RedJobs jobs, float *__restrict__ tmp, int pass) {
  __shared__ float red[4][64];
  const bool second = blockIdx.x >= (unsigned)jobs.j[0].gx;
  const RedJob &J = jobs.j[second ? 1 : 0];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int c = (int)(blockIdx.x - (second ? jobs.j[0].gx : 0)) * 64 + lane;
  float *t = tmp + (second ? (int64_t)kRedSplits * jobs.j[0].cols : 0);   // [split][cols] of this job
  const float *src = pass == 0 ? J.partials : t;
  const int64_t rows = pass == 0 ? J.rows : kRedSplits;
  const int64_t r0 = rows * blockIdx.y / gridDim.y, r1 = rows * (blockIdx.y + 1) / gridDim.y;
  float s = 0.f;
  if (c < J.cols)
    for (int64_t r = r0 + slice; r < r1; r += 4) s += src[r * J.cols + c];
  red[slice][lane] = s;
  __syncthreads();
  if (slice != 0 || c >= J.cols) return;
  const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  if (pass == 0) {
    t[(int64_t)blockIdx.y * J.cols + c] = v;
  } else {
    if (J.dstA && c < J.nA) J.dstA[c] = v;
    if (J.dstB && c >= J.offB && c < J.offB + J.nB) J.dstB[c - J.offB] = v;
  }
}

// g_b0[128] from b0_partials (rows_b0 x 128), g_w1[KK*128] and g_b1[KK] from dw1 partials (rows_w1 x (32*128 + 32)); a
// job with no destination is skipped.  tmp: kFcRedTmpFloats floats.
int fc_reduce_bias_w1(const float *b0_partials, int64_t rows_b0, float *g_b0, const float *dw1_partials, int64_t rows_w1,
                      float *g_w1, float *g_b1, int KK, float *tmp, hipStream_t stream) {
  RedJobs jobs;
  int n = 0;
  if (g_b0 && b0_partials && rows_b0 > 0)
    jobs.j[n++] = RedJob{b0_partials, g_b0, nullptr, rows_b0, kFcHidden, kFcHidden, 0, 0, kFcHidden / 64};
  if ((g_w1 || g_b1) && dw1_partials && rows_w1 > 0)
    jobs.j[n++] = RedJob{dw1_partials, g_w1, g_b1, rows_w1, kDw1Row, KK * kFcHidden, 32 * kFcHidden, KK, (kDw1Row + 63) / 64};
  if (n == 0) return GFLA_OK;
  if (!tmp) return GFLA_ERR_NULL_POINTER;
  if (n == 1) jobs.j[1] = RedJob{nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0};
  const unsigned gx = (unsigned)(jobs.j[0].gx + jobs.j[1].gx);
  fc_reduce_jobs_kernel<<<dim3(gx, kRedSplits), 256, 0, stream>>>(jobs, tmp, 0);
  fc_reduce_jobs_kernel<<<dim3(gx, 1), 256, 0, stream>>>(jobs, tmp, 1);
  return launch_status();
}

// ------------------------------------------------------------------ replicate-pad gradient + (pixel, C) -> NCHW
// grad[b,c,y,x] (+)= sum of dxpad[b, (yy, xx), c] over the padded positions that clamp onto (y, x).
// One launch carries up to TWO folds (the source and the target half of a layer: same B, C, H, W, different padding):
// blockIdx.z < B is job 0, the rest job 1 -- one dependent launch less per backward pass, and the second half's workgroups
// fill the first one's tail.  When accumulating, the old values are requested together with the gradient rows (one global
// round trip per workgroup instead of two).
struct FoldJob {
  const float *dxpad;
  float *grad;
  int64_t dx_bs;
  int Hp, Wp, pad_t, pad_l, accumulate;
};
struct FoldJobs {
  FoldJob j[2];
};
constexpr int kFoldOld = 16;   // old values a thread keeps in flight (maps up to 64 columns); wider maps load them late

__global__ __launch_bounds__(256) void fc_fold_kernel(FoldJobs jobs, int nb0, int C, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  float *tile = reinterpret_cast<float *>(gfla_smem);  // [64][W + 1]
  const bool second = (int)blockIdx.z >= nb0;
  const FoldJob &J = jobs.j[second ? 1 : 0];
  const float *__restrict__ dxpad = J.dxpad;
  float *__restrict__ grad = J.grad;
  const int Hp = J.Hp, Wp = J.Wp, pad_t = J.pad_t, pad_l = J.pad_l, accumulate = J.accumulate;
  const int y = blockIdx.x, c0 = blockIdx.y * 64;
  const int64_t b = (int)blockIdx.z - (second ? nb0 : 0);
  const int c = threadIdx.x & 63, xq = threadIdx.x >> 6;
  const int y0 = y == 0 ? 0 : y + pad_t, y1 = y == H - 1 ? Hp - 1 : y + pad_t;
  const float *src = dxpad + b * J.dx_bs + c0 + c;
  // the values this workgroup will add to (element i of its 64 x W block <-> thread i % 256): requested first
  const int n = min(64, C - c0) * W;
  const bool early = accumulate && n <= kFoldOld * 256;
  float old[kFoldOld];
  if (early) {
#pragma unroll
    for (int u = 0; u < kFoldOld; ++u) {
      const int i = min((int)threadIdx.x + 256 * u, n - 1);
      const int cl = i / W, x = i - cl * W;
      old[u] = grad[((b * C + c0 + cl) * H + y) * (int64_t)W + x];
    }
  }
  if (c0 + c < C) {
    constexpr int XU = 8;  // positions per thread and batch: their centre loads are issued together (one round trip)
    for (int xb = xq; xb < W; xb += 4 * XU) {
      float acc[XU];
#pragma unroll
      for (int i = 0; i < XU; ++i) {
        const int x = min(xb + 4 * i, W - 1);
        acc[i] = src[(int64_t)((y + pad_t) * Wp + x + pad_l) * C];
      }
#pragma unroll
      for (int i = 0; i < XU; ++i) {
        const int x = xb + 4 * i;
        if (x >= W) break;
        if (y == 0 || y == H - 1 || x == 0 || x == W - 1) {  // border: the padded positions that clamp onto (y, x)
          const int x0 = x == 0 ? 0 : x + pad_l, x1 = x == W - 1 ? Wp - 1 : x + pad_l;
          float extra = 0.f;
          for (int yy = y0; yy <= y1; ++yy)
            for (int xx = x0; xx <= x1; ++xx)
              if (yy != y + pad_t || xx != x + pad_l) extra += src[(int64_t)(yy * Wp + xx) * C];
          acc[i] += extra;
        }
        tile[c * (W + 1) + x] = acc[i];
      }
    }
  }
  __syncthreads();
  if (early || !accumulate) {
#pragma unroll
    for (int u = 0; u < kFoldOld; ++u) {
      const int i = (int)threadIdx.x + 256 * u;
      if (i >= n) break;
      const int cl = i / W, x = i - cl * W;
      const float v = tile[cl * (W + 1) + x];
      grad[((b * C + c0 + cl) * H + y) * (int64_t)W + x] = early ? old[u] + v : v;
    }
    for (int i = (int)threadIdx.x + 256 * kFoldOld; i < n; i += 256) {   // (not accumulating, wide maps)
      const int cl = i / W, x = i - cl * W;
      grad[((b * C + c0 + cl) * H + y) * (int64_t)W + x] = tile[cl * (W + 1) + x];
    }
    return;
  }
  // accumulating into a wide map: four elements per thread and pass, their four loads one round trip
  for (int i0 = threadIdx.x; i0 < n; i0 += 4 * 256) {
    float *g[4];
    float v[4], o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + 256 * u, n - 1);
      const int cl = i / W, x = i - cl * W;
      g[u] = grad + ((b * C + c0 + cl) * H + y) * (int64_t)W + x;
      v[u] = tile[cl * (W + 1) + x];
      o[u] = *g[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + 256 * u < n) *g[u] = o[u] + v[u];
  }
}

static int fc_fold_launch(const FoldJobs &jobs, int njobs, int64_t B, int C, int H, int W, hipStream_t stream) {
  if (B <= 0 || njobs <= 0) return GFLA_OK;
  if (njobs * B > 65535 || ceil_div(C, 64) > 65535 || (int64_t)64 * (W + 1) * 4 > 64 * 1024) return GFLA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)H, (unsigned)ceil_div(C, 64), (unsigned)(njobs * B));
  fc_fold_kernel<<<grid, 256, (unsigned)(64 * (W + 1) * sizeof(float)), stream>>>(jobs, (int)B, C, H, W);
  return launch_status();
}

int fc_fold(const float *dxpad, float *grad, int64_t B, int C, int H, int W, const FcHalf &g, int64_t dx_bs,
            int accumulate, hipStream_t stream) {
  if (!dxpad || !grad) return GFLA_ERR_NULL_POINTER;
  FoldJobs jobs;
  jobs.j[0] = FoldJob{dxpad, grad, dx_bs, g.Hp, g.Wp, g.pad_t, g.pad_l, accumulate};
  jobs.j[1] = jobs.j[0];
  return fc_fold_launch(jobs, 1, B, C, H, W, stream);
}

// both halves of a layer in one launch (either may be absent: dxpad == NULL)
int fc_fold2(const float *dx_s, float *grad_s, const FcHalf &gs, int64_t dxs_bs, int acc_s, const float *dx_t, float *grad_t,
             const FcHalf &gt, int64_t dxt_bs, int acc_t, int64_t B, int C, int H, int W, hipStream_t stream) {
  FoldJobs jobs;
  int n = 0;
  if (dx_s && grad_s) jobs.j[n++] = FoldJob{dx_s, grad_s, dxs_bs, gs.Hp, gs.Wp, gs.pad_t, gs.pad_l, acc_s};
  if (dx_t && grad_t) jobs.j[n++] = FoldJob{dx_t, grad_t, dxt_bs, gt.Hp, gt.Wp, gt.pad_t, gt.pad_l, acc_t};
  if (n == 0) return GFLA_OK;
  if (n == 1) jobs.j[1] = jobs.j[0];
  return fc_fold_launch(jobs, n, B, C, H, W, stream);
}

}  // namespace gfla
