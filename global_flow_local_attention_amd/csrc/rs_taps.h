// Per-pixel tap geometry and Gaussian weights of resample2d (resample2d_kernel.cu:42-93), shared by resample2d.hip
// and patch_mfma.hip.
#pragma once

#include "gfla_common.h"
#include <type_traits>

namespace gfla {

constexpr double kEps = 1e-8;  // resample2d_kernel.cu:14

// SAFE_DIV(a,b) with a,b in the arithmetic type, value in double (resample2d_kernel.cu:15).
template <typename A>
__device__ __forceinline__ double safe_div(A a, A b) {
  return (b == 0) ? ((double)a / kEps) : (double)(a / b);
}

template <typename A>
__device__ __forceinline__ A gauss(A dist, A sigma) {
  return (A)exp(safe_div<A>(-dist * dist, 2 * sigma * sigma));  // :75-78
}
// Single-precision exp for the LDS-plane kernels, where the weights are recomputed once per channel
// GROUP and eight double-precision exps per lane would dominate the instruction stream.  expf is
// within 1 ulp of the reference's (float)exp((double)q): a 6e-8 relative change of a weight.
template <typename A>
__device__ __forceinline__ A gauss_fast(A dist, A sigma) {
  return gauss<A>(dist, sigma);
}
template <>
__device__ __forceinline__ float gauss_fast<float>(float dist, float sigma) {
  const float num = -dist * dist, den = 2 * sigma * sigma;
  return (den == 0) ? (float)exp((double)num / kEps) : expf(num / den);
}

// Per-pixel state shared by all three kernels: tap row/column offsets and 1-D weights.
template <typename A, int KH>
struct Taps {
  int xL[KH], xR[KH], yT[KH], yB[KH];  // yT/yB pre-multiplied by the row pitch
  A xLd[KH], xRd[KH], yTd[KH], yBd[KH];  // distances
  A xLp[KH], xRp[KH], yTp[KH], yBp[KH];  // Gaussian weights
  A sum;
  A sigma;
  int ix0, iy0;  // (int)floor(x + dx), (int)floor(y + dy)

  // The 2*KH tap rows / columns as flat lists ordered by position when dilation == 1:
  // index r < KH -> the "top"/"left" tap KH-1-r (above/left of the sample), r >= KH -> "bottom"/"right" tap r-KH.
  __device__ __forceinline__ int row_off(int r) const { return r < KH ? yT[KH - 1 - r] : yB[r - KH]; }
  __device__ __forceinline__ A row_w(int r) const { return r < KH ? yTp[KH - 1 - r] : yBp[r - KH]; }
  __device__ __forceinline__ A row_d(int r) const { return r < KH ? yTd[KH - 1 - r] : yBd[r - KH]; }
  __device__ __forceinline__ int col_off(int q) const { return q < KH ? xL[KH - 1 - q] : xR[q - KH]; }
  __device__ __forceinline__ A col_w(int q) const { return q < KH ? xLp[KH - 1 - q] : xRp[q - KH]; }
  __device__ __forceinline__ A col_d(int q) const { return q < KH ? xLd[KH - 1 - q] : xRd[q - KH]; }

  // floor_alpha: fractional part from floor (forward, input2 gradient) or from int() truncation
  // (the reference's input1 gradient, resample2d_kernel.cu:137-138).
  // pitch: what a row index is multiplied by in yT / yB (the plane's row pitch Wi by default; the tile kernels pass 1 and
  // place the rows in their own window)
  // FAST: 0 = the reference's evaluation (double exp of the float quotient), 1 = expf of the float quotient (the planes-in-LDS
  // kernels: within an ulp of the weight), 2 = round 5's tile kernels, float only: ONE division 1 / (2 sigma^2), the eight
  // exponentials as v_exp_f32 of q * log2(e) (__expf), the normalisation as (sum of row weights) x (sum of column weights)
  // instead of the sixteen products -- every weight within ~2e-7 relative of the reference's, far inside the 2e-6 bar, and
  // ~150 instructions per pixel instead of ~1000 (the setup was 6 of the forward's 21 us at (1,64,256,176),
  // profiles/r5_rs_fwd_tile_ablations.txt).  sigma == 0 takes the reference's SAFE_DIV path in every mode.
  template <int FAST = 0>
  __device__ __forceinline__ void init(A dx, A dy, A sg, int x, int y, int Hi, int Wi, int dil,
                                       bool trunc_alpha, int pitch = -1) {
    if (pitch < 0) pitch = Wi;
    sigma = sg;
    const A xf = (A)x + dx, yf = (A)y + dy;  // :52-53
    const A fxf = floor_t<A>(xf), fyf = floor_t<A>(yf);
    ix0 = (int)fxf;
    iy0 = (int)fyf;
    const A alpha = trunc_alpha ? xf - (A)(int)xf : xf - fxf;
    const A beta = trunc_alpha ? yf - (A)(int)yf : yf - fyf;
    sum = 0;
#pragma unroll
    for (int f = 0; f < KH; ++f) {
      yT[f] = clampi((int)(fyf - (A)(f * dil)), 0, Hi - 1) * pitch;      // :62-63
      yB[f] = clampi((int)(fyf + (A)((f + 1) * dil)), 0, Hi - 1) * pitch;
      xL[f] = clampi((int)(fxf - (A)(f * dil)), 0, Wi - 1);              // :66-67
      xR[f] = clampi((int)(fxf + (A)((f + 1) * dil)), 0, Wi - 1);
      xLd[f] = (A)(f * dil) + alpha;                                     // :70-73
      xRd[f] = (A)((1. + f) * dil) - alpha;
      yTd[f] = (A)(f * dil) + beta;
      yBd[f] = (A)((1. + f) * dil) - beta;
      if constexpr (FAST != 2 || !std::is_same<A, float>::value) {
        xLp[f] = FAST ? gauss_fast<A>(xLd[f], sg) : gauss<A>(xLd[f], sg);
        xRp[f] = FAST ? gauss_fast<A>(xRd[f], sg) : gauss<A>(xRd[f], sg);
        yTp[f] = FAST ? gauss_fast<A>(yTd[f], sg) : gauss<A>(yTd[f], sg);
        yBp[f] = FAST ? gauss_fast<A>(yBd[f], sg) : gauss<A>(yBd[f], sg);
      }
    }
    if constexpr (FAST == 2 && std::is_same<A, float>::value) {
      const float den = 2 * sg * sg;
      // den == 0: the reference's SAFE_DIV; a denormal den has no finite -log2(e) / den (an integer flow would give
      // 0 * -inf = NaN): both take the division form of the exponent
      if (!(den >= 1.1754944e-38f)) {
#pragma unroll
        for (int f = 0; f < KH; ++f) {
          xLp[f] = gauss_fast<A>(xLd[f], sg), xRp[f] = gauss_fast<A>(xRd[f], sg);
          yTp[f] = gauss_fast<A>(yTd[f], sg), yBp[f] = gauss_fast<A>(yBd[f], sg);
        }
      } else {
        const float s2 = -1.4426950408889634f / den;   // exp(-d^2 / den) = 2^(d^2 * s2)
        float sx = 0, sy = 0;
#pragma unroll
        for (int f = 0; f < KH; ++f) {
          xLp[f] = __builtin_amdgcn_exp2f(xLd[f] * xLd[f] * s2), xRp[f] = __builtin_amdgcn_exp2f(xRd[f] * xRd[f] * s2);
          yTp[f] = __builtin_amdgcn_exp2f(yTd[f] * yTd[f] * s2), yBp[f] = __builtin_amdgcn_exp2f(yBd[f] * yBd[f] * s2);
          sx += xLp[f] + xRp[f];
          sy += yTp[f] + yBp[f];
        }
        sum = sx * sy;
        return;
      }
    }
#pragma unroll
    for (int fy = 0; fy < KH; ++fy)
#pragma unroll
      for (int fx = 0; fx < KH; ++fx)  // :89
        sum += (yTp[fy] * xLp[fx] + yTp[fy] * xRp[fx] + yBp[fy] * xLp[fx] + yBp[fy] * xRp[fx]);
  }
};

// "All of these values are needed HERE": an empty asm that takes the taps of a channel as read-write register operands.
// hipcc otherwise sinks each LDS / global load down to its first use -- the channel loop of the forward then reads
// ds_read -> s_waitcnt lgkmcnt(0) -> fma, sixteen round trips per channel (seen in the ISA, round 5); with the pin the
// sixteen requests are issued back to back and waited for once.  kernel_size 4 / 5 (the reference's production
// configuration) only: an asm statement takes at most 30 operands.
template <typename A, int N>
__device__ __forceinline__ void pin_taps(A (&v)[N][N]) {
  if constexpr (N == 4) {
    asm volatile("" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[0][3]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[1][2]),
                      "+v"(v[1][3]), "+v"(v[2][0]), "+v"(v[2][1]), "+v"(v[2][2]), "+v"(v[2][3]), "+v"(v[3][0]), "+v"(v[3][1]),
                      "+v"(v[3][2]), "+v"(v[3][3]));
  } else if constexpr (N == 2) {
    asm volatile("" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[1][0]), "+v"(v[1][1]));
  }
}

// resample2d d/d input2 (kernel_size 4, dilation 1, f32 / bf16 storage) on the aggregation's streaming machinery
// (local_attn_aggregate.hip); GFLA_ERR_UNSUPPORTED where the shape does not fit it
template <typename T>
int rs_bwd2_stream(const T *in1, const T *in2, const T *gout, float *gin2, int64_t B, int64_t C, int64_t Hi, int64_t Wi,
                   int64_t H, int64_t W, hipStream_t stream);

// d/d (dx, dy, sigma) of one pixel from the channel sums of its taps (resample2d_kernel.cu:273-328):
//   Racc[r] = sum_c g_c sum_q w_x[q] v_c[r][q],   Cacc[q] = sum_c g_c sum_r w_y[r] v_c[r][q]
// (rows / columns in position order, Taps::row_w / col_w).  Shared by rs_lds_kernel<MODE 2> and the streaming kernel.
template <typename A, int KH>
__device__ __forceinline__ void rs_bwd2_finish(const Taps<A, KH> &t, const A (&Racc)[2 * KH], const A (&Cacc)[2 * KH],
                                               A &rx, A &ry, A &rs) {
  constexpr int N = 2 * KH;
  const A sg = t.sigma;
  // 1/(-sigma^2) and 1/sigma^3 with the SAFE_DIV zero rule (resample2d_kernel.cu:273-292)
  const A d2 = -sg * sg, d3 = sg * sg * sg;
  const A inv2 = (d2 == 0) ? (A)(1.0 / kEps) : (A)1 / d2;
  const A inv3 = (d3 == 0) ? (A)(1.0 / kEps) : (A)1 / d3;
  A wy[N], wx[N];
#pragma unroll
  for (int r = 0; r < N; ++r) {
    wy[r] = t.row_w(r);
    wx[r] = t.col_w(r);
  }
  // fold the row / column weights back in and apply the derivative coefficients (:273-292);
  // "L"/"T" taps (index < KH) enter d/dx, d/dy with +, "R"/"B" taps with -
  A S = 0, g1x = 0, g1y = 0, g1s = 0, Wy = 0, Wx = 0, sx1 = 0, sy1 = 0, ssx = 0, ssy = 0;
#pragma unroll
  for (int r = 0; r < N; ++r) {
    const A R = wy[r] * Racc[r], Cq = wx[r] * Cacc[r];
    const A yd = t.row_d(r), xd = t.col_d(r);
    const A ay = (r < KH ? yd : -yd) * inv2, ax = (r < KH ? xd : -xd) * inv2;
    S += R;
    g1y += ay * R;
    g1x += ax * Cq;
    g1s += (yd * yd * inv3) * R + (xd * xd * inv3) * Cq;
    Wy += wy[r];
    Wx += wx[r];
    sy1 += ay * wy[r];
    sx1 += ax * wx[r];
    ssy += (yd * yd * inv3) * wy[r];
    ssx += (xd * xd * inv3) * wx[r];
  }
  const A sgx = sx1 * Wy, sgy = sy1 * Wx, sgs = ssy * Wx + ssx * Wy;  // "sumgrad", counted once (:277,318)
  // :328  grad1/sum - grad2/sum^2 with grad2 = sumgrad * S
  const A sum = t.sum, sum2 = t.sum * t.sum;
  const A is = (sum == 0) ? (A)(1.0 / kEps) : (A)1 / sum;
  const A is2 = (sum2 == 0) ? (A)(1.0 / kEps) : (A)1 / sum2;
  rx = g1x * is - (sgx * S) * is2;
  ry = g1y * is - (sgy * S) * is2;
  rs = g1s * is - (sgs * S) * is2;
}

}  // namespace gfla
