// resample2d for gfx950: normalised Gaussian-weighted k x k flow warping with a per-pixel sigma.
//
// Semantics follow resample2d_kernel.cu:20-95 (forward), :98-202 (d/d input1) and :204-330
// (d/d input2 = d/d(dx,dy,sigma)): taps at floor(p) - f*d and floor(p) + (f+1)*d for
// f in [0,k/2), clamped; weight exp(-dist^2 / (2 sigma^2)) evaluated in double (the reference's
// SAFE_DIV macro promotes the argument, :15,75-78); output = sum(w*v)/sum(w).
//
// What is different from the reference: the Gaussian weights depend only on (b,y,x), so one lane
// computes them ONCE for its pixel and then walks a chunk of channels (the reference recomputes
// 4*(k/2)^2 double-precision exps for every channel of every pixel); the input2 gradient
// produces dx, dy and sigma in one pass over the channels (the reference runs three threads per
// pixel, each looping over all channels twice).
#include "gfla_common.h"
#include <type_traits>

#include "lds_plane.h"
#include "rs_taps.h"
#include "patch_mfma.h"
#include "tile_map.h"

namespace gfla {

__device__ __forceinline__ bool decode(int sp_blocks, int ncg, int HW, int W, int &b, int &cg, int &y,
                                       int &x) {
  const int sp_blk = blockIdx.x % sp_blocks;
  const int bc = blockIdx.x / sp_blocks;
  const int p = sp_blk * kBlock + threadIdx.x;
  if (p >= HW) return false;
  b = bc / ncg;
  cg = bc - b * ncg;
  y = p / W;
  x = p - y * W;
  return true;
}

// Where a plane lives decides how it is added to: device-scope atomics in global memory, ds_add in LDS.
struct GlobalPlane {
  template <typename P, typename A>
  static __device__ __forceinline__ void add(P *p, A v) { atomic_add(p, (P)v); }
};
struct LdsPlane {
  template <typename P, typename A>
  static __device__ __forceinline__ void add(P *p, A v) { lds_add(p, (P)v); }
};
struct LdsFixPlane {  // fixed-point planes (lds_plane.h): the caller pre-scales the gradient
  template <typename A>
  static __device__ __forceinline__ void add(lds_fix_t *p, A v) { lds_add_fix(p, (float)v); }
};

// ---- per-pixel bodies, shared by the global-memory and the LDS-plane kernels ------------------
// The tap weights are an outer product w[r][q] = wy[r] * wx[q] of 2*KH row and 2*KH column weights
// (the reference's (fy,fx) x {T,B} x {L,R} loop enumerates exactly these pairs), so every sum over
// the taps is evaluated separably: 2*KH row/column weights in registers instead of (2*KH)^2 products.

// forward: `nch` channels, planes `plane_sz` apart starting at `plane`, outputs `ostride` apart.
// Round 5: the N x N taps of a channel are REQUESTED TOGETHER, then combined (written as a running sum over loads hipcc
// emitted ds_read -> s_waitcnt lgkmcnt(0) -> fma sixteen times per channel: the loop was one LDS round trip per tap); where
// no lane of the wave has a tap column clamped at the border the taps of a row are base + 0..N-1 (immediate offsets, read
// in pairs); SAFE_DIV(val, sum) (:93) is a multiplication by the reciprocal formed once per pixel (the float AND the double
// division of the macro sat in the channel loop) -- an ulp of the result.
template <typename T, typename PT, int KH, typename A>
__device__ __forceinline__ void rs_fwd_pixel(const Taps<A, KH> &t, const PT *__restrict__ plane, int64_t plane_sz,
                                             T *__restrict__ o, int64_t ostride, int nch) {
  constexpr int N = 2 * KH;
  int ro[N], co[N];
  A wy[N], wx[N];
  bool consecutive = true;
#pragma unroll
  for (int r = 0; r < N; ++r) {
    ro[r] = t.row_off(r);
    co[r] = t.col_off(r);
    wy[r] = t.row_w(r);
    wx[r] = t.col_w(r);
    consecutive = consecutive && co[r] == co[0] + r;
  }
  // a float32 weight sum below the normal range (sigma ~ 0.05 with fractions near 0.5) has no float32 reciprocal: those
  // pixels divide, as the reference does (resample2d_kernel.cu:93); everything else multiplies by 1 / sum
  const bool tiny = t.sum != 0 && t.sum < (A)1.1754944e-38;
  const A inv = (t.sum == 0) ? (A)(1.0 / kEps) : (tiny ? (A)1 : (A)1 / t.sum);
  auto combine = [&](const A (&v)[N][N]) {
    A val = 0;  // resample2d_kernel.cu:85-88: sum_r wy[r] * sum_q wx[q] * v[r][q]
#pragma unroll
    for (int r = 0; r < N; ++r) {
      A rowacc = 0;
#pragma unroll
      for (int q = 0; q < N; ++q) rowacc += wx[q] * v[r][q];
      val += wy[r] * rowacc;
    }
    return tiny ? val / t.sum : val * inv;
  };
  if constexpr (std::is_same<A, float>::value && std::is_same<PT, float>::value && N == 4) {
    // kernel_size 4 / 5 in float, no clamped column in the wave: the sixteen normalised weights wy[r] wx[q] / sum once per
    // pixel, then per channel 8 reads of pairs + 8 packed multiply-adds (the channel loop was ~60 instructions, 9 of the
    // forward's 21 us at (1,64,256,176))
    if (__all(consecutive)) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      v2f w2[N][2];
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const float qy = tiny ? wy[r] / t.sum : wy[r] * inv;
        w2[r][0] = v2f{qy * wx[0], qy * wx[1]};
        w2[r][1] = v2f{qy * wx[2], qy * wx[3]};
      }
      const PT *base = plane + co[0];
      auto request = [&](const PT *b, A (&v)[N][N]) {
#pragma unroll
        for (int r = 0; r < N; ++r) {
          const PT *rp = b + ro[r];
#pragma unroll
          for (int q = 0; q < N; ++q) v[r][q] = rp[q];
        }
      };
      auto combine2 = [&](const A (&v)[N][N]) {   // two independent chains of packed multiply-adds
        v2f a0 = w2[0][0] * v2f{v[0][0], v[0][1]}, a1 = w2[0][1] * v2f{v[0][2], v[0][3]};
#pragma unroll
        for (int r = 1; r < N; ++r) {
          a0 = w2[r][0] * v2f{v[r][0], v[r][1]} + a0;
          a1 = w2[r][1] * v2f{v[r][2], v[r][3]} + a1;
        }
        a0 += a1;
        return a0.x + a0.y;
      };
      // (A software pipeline over the channels -- channel c + 1's taps requested before channel c's are combined -- was
      // measured and dropped: 19.9 -> 25.6 us at (1,64,256,176); the second register set costs more waves than the overlap
      // buys.  profiles/r5_config2_sweeps.txt)
      for (int c = 0; c < nch; ++c) {
        A v[N][N];
        request(base, v);
        pin_taps<A, N>(v);
        *o = Num<T>::from(combine2(v));
        base += plane_sz;
        o += ostride;
      }
      return;
    }
  }
  if (__all(consecutive)) {
    const PT *base = plane + co[0];
    for (int c = 0; c < nch; ++c) {
      A v[N][N];
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const PT *rp = base + ro[r];
#pragma unroll
        for (int q = 0; q < N; ++q) v[r][q] = Num<PT>::ld(rp + q);
      }
      pin_taps<A, N>(v);   // every request above, every use below
      *o = Num<T>::from(combine(v));
      base += plane_sz;
      o += ostride;
    }
  } else {
    for (int c = 0; c < nch; ++c) {
      A v[N][N];
#pragma unroll
      for (int r = 0; r < N; ++r)
#pragma unroll
        for (int q = 0; q < N; ++q) v[r][q] = Num<PT>::ld(plane + ro[r] + co[q]);
      pin_taps<A, N>(v);
      *o = Num<T>::from(combine(v));
      plane += plane_sz;
      o += ostride;
    }
  }
}

// d/d input1: scatter SAFE_DIV(w, sum) * grad_out into the gradient planes (:195-198).
// ro / co: row (x row pitch) and column offsets of the N x N taps in position order, qy = w_y / sum, wx = w_x.
template <typename T, typename PT, int N, typename A, typename Where>
__device__ __forceinline__ void rs_bwd1_apply(const int (&ro)[N], const int (&co)[N], const A (&qy)[N], const A (&wx)[N],
                                              const T *__restrict__ g, int64_t gstride, PT *__restrict__ gplane,
                                              int64_t plane_sz, int nch, A gscale) {
  A gnext = Num<T>::ld(g);  // the next channel's gradient is requested before this channel's adds are issued
  for (int c = 0; c < nch; ++c) {
    const A go = gnext * gscale;
    g += gstride;
    if (c + 1 < nch) gnext = Num<T>::ld(g);
#pragma unroll
    for (int r = 0; r < N; ++r) {
      const A gr_ = go * qy[r];
#pragma unroll
      for (int q = 0; q < N; ++q) Where::add(gplane + ro[r] + co[q], gr_ * wx[q]);
    }
    gplane += plane_sz;
  }
}
// The same scatter into fixed-point planes (lds_plane.h), written for the instruction count: the scatter kernel's time is
// its instruction stream (VALU and LDS issue add up; removing the atomics alone gains 20 %).  Per contribution: ONE
// v_fma_f64 (row gradient x column weight, exact in double, + the 1.5 * 2^52 magic number), the high-word correction and
// the ds_add_u64 -- the address is an immediate offset from the row's first tap when every lane of the wave has
// consecutive columns (no lane clamped at a border), else one more add.  Six instructions before, three now.
template <typename T, int N>
__device__ __forceinline__ void rs_bwd1_apply_fix(const int (&ro)[N], const int (&co)[N], const float (&qy)[N],
                                                  const float (&wx)[N], float g_first, const T *__restrict__ g,
                                                  int64_t gstride, lds_fix_t *__restrict__ gplane, int64_t plane_sz, int nch,
                                                  float gscale) {
  // g_first = the first channel's gradient at g (the caller requested it an iteration ahead)
  // The product (gradient x row weight / sum) is formed in DOUBLE: qy = w_y / sum alone is not bounded by 1 (only qy * wx
  // is), and with a sigma small enough for the column weights to underflow go * qy overflows float while the contribution
  // itself is <= |go| (advisor finding, round 3).
  double wxd[N], qyd[N];
  bool consecutive = true;
#pragma unroll
  for (int q = 0; q < N; ++q) {
    wxd[q] = (double)wx[q];
    qyd[q] = (double)qy[q];
    consecutive = consecutive && co[q] == co[0] + q;
  }
  float gnext = g_first;
  if (__all(consecutive)) {
    for (int c = 0; c < nch; ++c) {
      const float go = gnext * gscale;
      g += gstride;
      if (c + 1 < nch) gnext = Num<T>::ld(g);
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const double gr_ = (double)go * qyd[r];
        lds_fix_t *row = gplane + ro[r] + co[0];
#pragma unroll
        for (int q = 0; q < N; ++q) lds_add_fix_biased(row + q, __builtin_fma(gr_, wxd[q], kFixMagic));
      }
      gplane += plane_sz;
    }
  } else {
    for (int c = 0; c < nch; ++c) {
      const float go = gnext * gscale;
      g += gstride;
      if (c + 1 < nch) gnext = Num<T>::ld(g);
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const double gr_ = (double)go * qyd[r];
#pragma unroll
        for (int q = 0; q < N; ++q) lds_add_fix_biased(gplane + ro[r] + co[q], __builtin_fma(gr_, wxd[q], kFixMagic));
      }
      gplane += plane_sz;
    }
  }
}
template <typename T, typename PT, int KH, typename A, typename Where>
__device__ __forceinline__ void rs_bwd1_pixel(const Taps<A, KH> &t, const T *__restrict__ g, int64_t gstride,
                                              PT *__restrict__ gplane, int64_t plane_sz, int nch, A gscale = 1) {
  constexpr int N = 2 * KH;
  int ro[N], co[N];
  A qy[N], wx[N];
#pragma unroll
  for (int r = 0; r < N; ++r) {
    ro[r] = t.row_off(r);
    co[r] = t.col_off(r);
    qy[r] = (A)safe_div<A>(t.row_w(r), t.sum);
    wx[r] = t.col_w(r);
  }
  if constexpr (std::is_same<Where, LdsFixPlane>::value)
    rs_bwd1_apply_fix<T, N>(ro, co, qy, wx, Num<T>::ld(g), g, gstride, gplane, plane_sz, nch, gscale);
  else
    rs_bwd1_apply<T, PT, N, A, Where>(ro, co, qy, wx, g, gstride, gplane, plane_sz, nch, gscale);
}

// Tap records (kernel_size 4 / 5, float): everything rs_bwd1_pixel derives from (dx, dy, sigma) -- eight Gaussians with
// the reference's exact divisions, their normalisation -- is ~450 instructions per pixel, and a workgroup of the scatter
// kernel repeats it for every pixel while it owns only G = 2..4 of the C channels: at 64x44 C256 that setup was 70 % of
// the instruction stream.  With scratch memory the setup runs ONCE per pixel (rs_tap_table_kernel) and the scatter
// kernel reads 48-byte records (L2-resident: 48 B x B H W) one pixel ahead.  Same expressions, same rounding.
struct alignas(16) RsTapRec {
  uint32_t rows[2], cols[2];  // 4 + 4 clamped row / column INDICES, 16 bits each, position order
  float qy[4], wx[4];
};
static_assert(sizeof(RsTapRec) == kRsTapRecBytes, "record size is part of the workspace layout (patch_mfma.h)");
__global__ __launch_bounds__(256) void rs_tap_table_kernel(const float *__restrict__ in2, RsTapRec *__restrict__ tab,
                                                          int HW, int W, int Hi, int Wi, int dil, int trunc,
                                                          const unsigned *__restrict__ skip_stat, unsigned skip_limit) {
  if (skip_stat) {
    unsigned tot = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) tot += skip_stat[i];
    if (tot <= skip_limit) return;
  }
  const int p = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (p >= HW) return;
  const int y = p / W, x = p - y * W;
  const float *i2 = in2 + (int64_t)b * 3 * HW + p;
  Taps<float, 2> t;
  t.template init<2>(i2[0], i2[HW], i2[2 * HW], x, y, Hi, Wi, dil, (trunc & 1) != 0);
  RsTapRec r;
  unsigned ri[4], ci[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ri[j] = (unsigned)(t.row_off(j) / Wi);
    ci[j] = (unsigned)t.col_off(j);
    r.qy[j] = safe_div<float>(t.row_w(j), t.sum);
    r.wx[j] = t.col_w(j);
  }
  r.rows[0] = ri[0] | ri[1] << 16, r.rows[1] = ri[2] | ri[3] << 16;
  r.cols[0] = ci[0] | ci[1] << 16, r.cols[1] = ci[2] | ci[3] << 16;
  tab[(int64_t)b * HW + p] = r;
}

// d/d input2 = d/d(dx, dy, sigma) for `nch` channels; returns the three partial results (linear in
// the channel sums, so channel chunks combine by addition).  With T_rq = wy[r] wx[q] g v[r][q]:
//   S = sum T;  d/dx = sum_q ax[q] C_q,  d/dy = sum_r ay[r] R_r,  d/dsigma = sum_r bs_y[r] R_r + sum_q bs_x[q] C_q
// where R_r / C_q are the row / column sums of T accumulated over the channels (8 accumulators for k=4).
template <typename T, typename PT, int KH, typename A>
__device__ __forceinline__ void rs_bwd2_pixel(const Taps<A, KH> &t, const PT *__restrict__ plane, int64_t plane_sz,
                                              const T *__restrict__ g, int64_t gstride, int nch, A &rx, A &ry,
                                              A &rs) {
  constexpr int N = 2 * KH;
  int ro[N], co[N];
  A wy[N], wx[N];
#pragma unroll
  for (int r = 0; r < N; ++r) {
    ro[r] = t.row_off(r);
    co[r] = t.col_off(r);
    wy[r] = t.row_w(r);
    wx[r] = t.col_w(r);
  }
  A Racc[N], Cacc[N];
  bool consecutive = true;
#pragma unroll
  for (int r = 0; r < N; ++r) {
    Racc[r] = Cacc[r] = 0;
    consecutive = consecutive && co[r] == co[0] + r;
  }
  const bool fast = __all(consecutive);   // no lane's tap columns clamped: row taps are base + 0..N-1
  // (round 5: a channel's taps and its gradient are requested together, see rs_fwd_pixel)
  for (int c = 0; c < nch; ++c) {
    const A go = Num<T>::ld(g);
    A v[N][N];
    if (fast) {
#pragma unroll
      for (int r = 0; r < N; ++r) {
        const PT *rp = plane + ro[r] + co[0];
#pragma unroll
        for (int q = 0; q < N; ++q) v[r][q] = Num<PT>::ld(rp + q);
      }
    } else {
#pragma unroll
      for (int r = 0; r < N; ++r)
#pragma unroll
        for (int q = 0; q < N; ++q) v[r][q] = Num<PT>::ld(plane + ro[r] + co[q]);
    }
    pin_taps<A, N>(v);   // every request above, every use below
    A cs[N];
#pragma unroll
    for (int q = 0; q < N; ++q) cs[q] = 0;
#pragma unroll
    for (int r = 0; r < N; ++r) {
      A rsum = 0;
#pragma unroll
      for (int q = 0; q < N; ++q) {
        rsum += wx[q] * v[r][q];
        cs[q] += wy[r] * v[r][q];
      }
      Racc[r] += go * rsum;
    }
#pragma unroll
    for (int q = 0; q < N; ++q) Cacc[q] += go * cs[q];
    plane += plane_sz;
    g += gstride;
  }
  rs_bwd2_finish<A, KH>(t, Racc, Cacc, rx, ry, rs);
}

// ---- global-memory kernels: thread <-> (b, channel chunk, pixel) ---------------------------------
template <typename T, int KH>
__global__ __launch_bounds__(kBlock) void rs_fwd_kernel(const T *__restrict__ in1,
                                                       const T *__restrict__ in2, T *__restrict__ out,
                                                       int C, int Hi, int Wi, int H, int W, int dil,
                                                       int cpt, int ncg, int sp_blocks) {
  using A = typename Num<T>::acc;
  int b, cg, y, x;
  if (!decode(sp_blocks, ncg, H * W, W, b, cg, y, x)) return;
  const int64_t HW = (int64_t)H * W;
  const T *i2 = in2 + (int64_t)b * 3 * HW + (int64_t)y * W + x;
  Taps<A, KH> t;
  t.init(Num<T>::ld(i2), Num<T>::ld(i2 + HW), Num<T>::ld(i2 + 2 * HW), x, y, Hi, Wi, dil, false);
  const int c0 = cg * cpt, c1 = min(C, c0 + cpt);
  const int64_t plane_sz = (int64_t)Hi * Wi;
  rs_fwd_pixel<T, T, KH, A>(t, in1 + ((int64_t)b * C + c0) * plane_sz, plane_sz,
                            out + ((int64_t)b * C + c0) * HW + (int64_t)y * W + x, HW, c1 - c0);
}

template <typename T, int KH>
__global__ __launch_bounds__(kBlock) void rs_bwd1_kernel(const T *__restrict__ in2,
                                                        const T *__restrict__ gout, T *__restrict__ gin1,
                                                        int C, int Hi, int Wi, int H, int W, int dil,
                                                        int trunc, int cpt, int ncg, int sp_blocks) {
  using A = typename Num<T>::acc;
  int b, cg, y, x;
  if (!decode(sp_blocks, ncg, H * W, W, b, cg, y, x)) return;
  const int64_t HW = (int64_t)H * W;
  const T *i2 = in2 + (int64_t)b * 3 * HW + (int64_t)y * W + x;
  Taps<A, KH> t;
  t.init(Num<T>::ld(i2), Num<T>::ld(i2 + HW), Num<T>::ld(i2 + 2 * HW), x, y, Hi, Wi, dil, (trunc & 1) != 0);
  const int c0 = cg * cpt, c1 = min(C, c0 + cpt);
  const int64_t plane_sz = (int64_t)Hi * Wi;
  rs_bwd1_pixel<T, T, KH, A, GlobalPlane>(t, gout + ((int64_t)b * C + c0) * HW + (int64_t)y * W + x, HW,
                                          gin1 + ((int64_t)b * C + c0) * plane_sz, plane_sz, c1 - c0);
}

template <typename T, int KH>
__global__ __launch_bounds__(kBlock) void rs_bwd2_kernel(const T *__restrict__ in1,
                                                        const T *__restrict__ in2,
                                                        const T *__restrict__ gout, T *__restrict__ gin2,
                                                        int C, int Hi, int Wi, int H, int W, int dil,
                                                        int cpt, int ncg, int sp_blocks) {
  using A = typename Num<T>::acc;
  int b, cg, y, x;
  if (!decode(sp_blocks, ncg, H * W, W, b, cg, y, x)) return;
  const int64_t HW = (int64_t)H * W;
  const T *i2 = in2 + (int64_t)b * 3 * HW + (int64_t)y * W + x;
  Taps<A, KH> t;
  t.init(Num<T>::ld(i2), Num<T>::ld(i2 + HW), Num<T>::ld(i2 + 2 * HW), x, y, Hi, Wi, dil, false);
  const int c0 = cg * cpt, c1 = min(C, c0 + cpt);
  const int64_t plane_sz = (int64_t)Hi * Wi;
  A rx, ry, rs;
  rs_bwd2_pixel<T, T, KH, A>(t, in1 + ((int64_t)b * C + c0) * plane_sz, plane_sz,
                             gout + ((int64_t)b * C + c0) * HW + (int64_t)y * W + x, HW, c1 - c0, rx, ry, rs);
  T *o = gin2 + (int64_t)b * 3 * HW + (int64_t)y * W + x;
  if (ncg == 1) {
    o[0] = Num<T>::from(Num<T>::ld(o) + rx);
    o[HW] = Num<T>::from(Num<T>::ld(o + HW) + ry);
    o[2 * HW] = Num<T>::from(Num<T>::ld(o + 2 * HW) + rs);
  } else {
    atomic_add(o, (T)rx);
    atomic_add(o + HW, (T)ry);
    atomic_add(o + 2 * HW, (T)rs);
  }
}

// ---- LDS-plane kernels: workgroup <-> (b, group of G channels[, 1/split of the pixels]) ----------
// MODE 0 forward (source planes staged), 1 d/d input1 (gradient planes accumulated in LDS, flushed
// once), 2 d/d input2 (source planes staged, three atomics per lane and group).
// element type of what a mode writes: the warped map / the input1 gradient in the storage type; the (dx, dy, sigma)
// gradient -- a reduction over channel groups, accumulated with atomics -- in the arithmetic type (float for bf16)
template <typename T, int MODE>
struct RsOut {
  using type = typename std::conditional<MODE == 2, typename Num<T>::acc, T>::type;
};

// FIX (MODE 1 only): the scatter planes are 64-bit fixed point (lds_plane.h) instead of double
// TAB (MODE 1, KH 2, float): in2 points at the tap records of rs_tap_table_kernel instead of (dx, dy, sigma)
template <typename T, int KH, int MODE, bool WIN, bool FIX = false, bool TAB = false>
__global__ __launch_bounds__(kLdsThreads) void rs_lds_kernel(const T *__restrict__ in1, const T *__restrict__ in2,
                                                            const T *__restrict__ gout,
                                                            typename RsOut<T, MODE>::type *__restrict__ outp,
                                                            int C, int Hi, int Wi, int H, int W, int dil,
                                                            int trunc, int G, int ngroups, int split, int per,
                                                            int margin, const unsigned *__restrict__ skip_stat,
                                                            unsigned skip_limit) {
  using A = typename Num<T>::acc;
  using PT = typename std::conditional<MODE == 1, typename std::conditional<FIX, lds_fix_t, lds_acc_t>::type, A>::type;
  static_assert(!FIX || (MODE == 1 && std::is_same<A, float>::value), "fixed-point planes: float scatter only");
  static_assert(!TAB || (MODE == 1 && KH == 2 && std::is_same<T, float>::value), "tap records: float scatter, kernel_size 4 / 5");
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  __shared__ unsigned s_amax;
  // adaptive dispatch (patch_mfma.hip): the matrix-core path took this launch when its statistic is within the limit
  if (skip_stat) {  // the statistic is a sum over 32 counters (patch_mfma.hip: kPmStatSlots)
    unsigned tot = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) tot += skip_stat[i];
    if (tot <= skip_limit) return;
  }
  PT *planes = reinterpret_cast<PT *>(gfla_smem);
  int bid = blockIdx.x;
  const int sp = bid % split;
  bid /= split;
  const int g = bid % ngroups;
  const int b = bid / ngroups;
  const int c0 = g * G;
  const int gc = min(G, C - c0);
  const int plane_sz = Hi * Wi;
  const int HW = H * W;
  const int p_begin = sp * per;
  const int p_end = min(HW, p_begin + per);
  if (p_begin >= p_end) return;
  // rows of the input1 plane resident in LDS: all (margin < 0) or the window this band of pixel rows
  // reaches with |dy| <= margin
  const Window win = make_window(p_begin / W, (p_end - 1) / W, (KH - 1) * dil, KH * dil, WIN ? margin : -1, Hi);
  const int win_sz = win.rows * Wi;
  const T *in1_0 = in1 + ((int64_t)b * C + c0) * plane_sz;
  if constexpr (MODE == 1) {
    zero_planes<PT>(planes, gc * win_sz);
    if (FIX && threadIdx.x == 0) s_amax = 0;
  } else {
    for (int c = 0; c < gc; ++c)
      stage_planes<T, PT>(in1_0 + (int64_t)c * plane_sz + win.lo * Wi, planes + (size_t)c * win_sz, win_sz);
  }
  __syncthreads();
  PT *planes0 = planes - win.lo * Wi;  // plane-relative offsets index the window
  const int lo_off = win.lo * Wi, hi_off = (win.lo + win.rows) * Wi;
  FixScale fix{1.f, 1.0, true};
  if constexpr (FIX) {  // every contribution is a gradient of this range times weights <= 1: its maximum sets the scale
    unsigned m = 0;
    for (int c = 0; c < gc; ++c) {
      const T *go = gout + ((int64_t)b * C + c0 + c) * HW;
      for (int p = p_begin + threadIdx.x; p < p_end; p += blockDim.x) m = max(m, __float_as_uint(fabsf(Num<T>::ld(go + p))));
    }
    fix = fix_scale(block_umax(m, &s_amax));
  }
  if constexpr (TAB) {
    // records one pixel ahead (a workgroup makes only a handful of passes of this loop)
    const uint4 *rec = reinterpret_cast<const uint4 *>(in2) + (int64_t)b * HW * 3;
    const T *gout0 = gout + ((int64_t)b * C + c0) * HW;
    uint4 na{}, nb{}, nc{};
    A ng = 0;  // and the first channel's gradient of the next pixel
    if (p_begin + (int)threadIdx.x < p_end) {
      const uint4 *r = rec + (int64_t)(p_begin + threadIdx.x) * 3;
      na = r[0], nb = r[1], nc = r[2];
      ng = Num<T>::ld(gout0 + p_begin + threadIdx.x);
    }
    for (int p = p_begin + threadIdx.x; p < p_end; p += blockDim.x) {
      const uint4 ra = na, rb = nb, rc = nc;
      const A g_first = ng;
      if (p + (int)blockDim.x < p_end) {
        const uint4 *r = rec + (int64_t)(p + blockDim.x) * 3;
        na = r[0], nb = r[1], nc = r[2];
        ng = Num<T>::ld(gout0 + p + blockDim.x);
      }
      const int ro[4] = {(int)(ra.x & 0xffff) * Wi, (int)(ra.x >> 16) * Wi, (int)(ra.y & 0xffff) * Wi, (int)(ra.y >> 16) * Wi};
      const int co[4] = {(int)(ra.z & 0xffff), (int)(ra.z >> 16), (int)(ra.w & 0xffff), (int)(ra.w >> 16)};
      const A qy[4] = {__uint_as_float(rb.x), __uint_as_float(rb.y), __uint_as_float(rb.z), __uint_as_float(rb.w)};
      const A wx[4] = {__uint_as_float(rc.x), __uint_as_float(rc.y), __uint_as_float(rc.z), __uint_as_float(rc.w)};
      const bool inside = !WIN || (ro[0] >= lo_off && ro[3] < hi_off);
      const T *go = gout0 + p;
      if (inside) {
        if constexpr (FIX)
          rs_bwd1_apply_fix<T, 4>(ro, co, qy, wx, g_first, go, HW, planes0, win_sz, gc, fix.up);
        else
          rs_bwd1_apply<T, PT, 4, A, LdsPlane>(ro, co, qy, wx, go, HW, planes0, win_sz, gc, (A)1);
      } else {
        rs_bwd1_apply<T, T, 4, A, GlobalPlane>(ro, co, qy, wx, go, HW, outp + ((int64_t)b * C + c0) * plane_sz, plane_sz, gc, (A)1);
      }
    }
  } else
  for (int p = p_begin + threadIdx.x; p < p_end; p += blockDim.x) {
    const int y = p / W, x = p - y * W;
    const T *i2 = in2 + (int64_t)b * 3 * HW + p;
    Taps<A, KH> t;
    t.template init<2>(Num<T>::ld(i2), Num<T>::ld(i2 + HW), Num<T>::ld(i2 + 2 * HW), x, y, Hi, Wi, dil,
                          MODE == 1 && (trunc & 1) != 0);
    // outermost taps bound the rows this pixel touches; beyond the window it uses global memory
    const bool inside = !WIN || (t.yT[KH - 1] >= lo_off && t.yB[KH - 1] < hi_off);
    if constexpr (MODE == 0) {
      T *o = outp + ((int64_t)b * C + c0) * HW + p;
      if (inside)
        rs_fwd_pixel<T, A, KH, A>(t, planes0, win_sz, o, HW, gc);
      else
        rs_fwd_pixel<T, T, KH, A>(t, in1_0, plane_sz, o, HW, gc);
    } else if constexpr (MODE == 1) {
      const T *go = gout + ((int64_t)b * C + c0) * HW + p;
      if (inside) {
        if constexpr (FIX)
          rs_bwd1_pixel<T, PT, KH, A, LdsFixPlane>(t, go, HW, planes0, win_sz, gc, fix.up);
        else
          rs_bwd1_pixel<T, PT, KH, A, LdsPlane>(t, go, HW, planes0, win_sz, gc);
      } else
        rs_bwd1_pixel<T, T, KH, A, GlobalPlane>(t, go, HW, outp + ((int64_t)b * C + c0) * plane_sz, plane_sz, gc);
    } else {
      A rx, ry, rs;
      const T *go = gout + ((int64_t)b * C + c0) * HW + p;
      if (inside)
        rs_bwd2_pixel<T, A, KH, A>(t, planes0, win_sz, go, HW, gc, rx, ry, rs);
      else
        rs_bwd2_pixel<T, T, KH, A>(t, in1_0, plane_sz, go, HW, gc, rx, ry, rs);
      A *o = outp + (int64_t)b * 3 * HW + p;
      atomic_add(o, rx);
      atomic_add(o + HW, ry);
      atomic_add(o + 2 * HW, rs);
    }
  }
  if constexpr (MODE == 1) {
    __syncthreads();
    for (int c = 0; c < gc; ++c) {  // bit 1 of the flag word: overwrite (host-checked)
      T *dst = outp + ((int64_t)b * C + c0 + c) * plane_sz + win.lo * Wi;
      if constexpr (FIX)
        flush_planes_fix<T>(dst, planes + (size_t)c * win_sz, win_sz, split == 1 && margin < 0, (trunc & 2) != 0, fix);
      else
        flush_planes<T>(dst, planes + (size_t)c * win_sz, win_sz, split == 1 && margin < 0, (trunc & 2) != 0);
    }
  }
}


// ---- few planes, each beyond the LDS budget (BASELINE configs[1]; tile_map.h) --------------------------------------------
// forward and d/d input2 are GATHERS: thread <-> pixel, a chunk of `cpt` channels per thread with one tap setup (single-
// precision exps), taps read from global memory; blocks XCD-swizzled over (batch, pixel block) with a pixel block's
// channel chunks adjacent.  (rs_fwd_kernel / rs_bwd2_kernel above are the same bodies with one thread per (pixel, channel
// chunk) chosen for many planes; at B*C = 64 that geometry recomputed the taps for every channel.)
template <typename T, int KH>
__global__ __launch_bounds__(kBlock) void rs_fwd_big_kernel(const T *__restrict__ in1, const T *__restrict__ in2,
                                                           T *__restrict__ out, int C, int Hi, int Wi, int H, int W, int dil,
                                                           int cpt, int ncg, int nsp, int64_t nwg) {
  using A = typename Num<T>::acc;
  const int64_t v = xcd_swizzle(blockIdx.x, nwg);
  const int cg = (int)(v % ncg);
  const int64_t rest = v / ncg;
  const int sp = (int)(rest % nsp), b = (int)(rest / nsp);
  const int HW = H * W;
  const int p = sp * kBlock + threadIdx.x;
  if (p >= HW) return;
  const int y = p / W, x = p - y * W;
  const T *i2 = in2 + (int64_t)b * 3 * HW + p;
  Taps<A, KH> t;
  t.template init<2>(Num<T>::ld(i2), Num<T>::ld(i2 + HW), Num<T>::ld(i2 + 2 * HW), x, y, Hi, Wi, dil, false);
  const int c0 = cg * cpt, c1 = min(C, c0 + cpt);
  const int64_t plane_sz = (int64_t)Hi * Wi;
  rs_fwd_pixel<T, T, KH, A>(t, in1 + ((int64_t)b * C + c0) * plane_sz, plane_sz, out + ((int64_t)b * C + c0) * HW + p, HW,
                            c1 - c0);
}

template <typename T, int KH>
__global__ __launch_bounds__(kBlock) void rs_bwd2_big_kernel(const T *__restrict__ in1, const T *__restrict__ in2,
                                                            const T *__restrict__ gout, T *__restrict__ gin2, int C, int Hi,
                                                            int Wi, int H, int W, int dil, int cpt, int ncg, int nsp,
                                                            int64_t nwg) {
  using A = typename Num<T>::acc;
  const int64_t v = xcd_swizzle(blockIdx.x, nwg);
  const int cg = (int)(v % ncg);
  const int64_t rest = v / ncg;
  const int sp = (int)(rest % nsp), b = (int)(rest / nsp);
  const int HW = H * W;
  const int p = sp * kBlock + threadIdx.x;
  if (p >= HW) return;
  const int y = p / W, x = p - y * W;
  const T *i2 = in2 + (int64_t)b * 3 * HW + p;
  Taps<A, KH> t;
  t.template init<2>(Num<T>::ld(i2), Num<T>::ld(i2 + HW), Num<T>::ld(i2 + 2 * HW), x, y, Hi, Wi, dil, false);
  const int c0 = cg * cpt, c1 = min(C, c0 + cpt);
  const int64_t plane_sz = (int64_t)Hi * Wi;
  A rx, ry, rs;
  rs_bwd2_pixel<T, T, KH, A>(t, in1 + ((int64_t)b * C + c0) * plane_sz, plane_sz, gout + ((int64_t)b * C + c0) * HW + p, HW,
                             c1 - c0, rx, ry, rs);
  T *o = gin2 + (int64_t)b * 3 * HW + p;
  if (ncg == 1) {  // sole writer of this pixel
    o[0] = Num<T>::from(Num<T>::ld(o) + rx);
    o[HW] = Num<T>::from(Num<T>::ld(o + HW) + ry);
    o[2 * HW] = Num<T>::from(Num<T>::ld(o + 2 * HW) + rs);
  } else {
    atomic_add(o, (T)rx);
    atomic_add(o + HW, (T)ry);
    atomic_add(o + 2 * HW, (T)rs);
  }
}

// The gathers on LDS windows (second version; the global-gather kernels above stay as the per-tile fallback and under tuning
// key 38 = 1: at (1,64,256,176) every per-tap wave load cost ~40 CU cycles on a smooth flow -- forward 42 us whatever the
// channels per thread, profiles/r5_config2_first_kernels.txt): workgroup = (tile of th x tw pixels, G channels), one pixel per
// thread with its taps in registers, the source planes staged into the bounding-box window of the tile's taps.
// MODE 0 forward, MODE 2 d/d input2.
template <typename T, int KH, int MODE>
__global__ __launch_bounds__(512) void rs_gather_tile_kernel(const T *__restrict__ in1, const T *__restrict__ in2,
                                                            const T *__restrict__ gout, T *__restrict__ outp, int C, int Hi,
                                                            int Wi, int H, int W, int dil, int th, int tw, int ntx, int nty, int G,
                                                            int ngroups, int lds_elems, int64_t nwg, int abl) {
  // abl (tuning key 39, timing ablations -- results are garbage for bits 1-4): 1 = stop after the per-pixel setup and the
  // box, 2 = no staging, 4 = no channel loop; 8 = staging with 4-byte requests, a window row per wave (valid results)
  using A = typename Num<T>::acc;
  constexpr int N = 2 * KH;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  A *planes = reinterpret_cast<A *>(gfla_smem);
  __shared__ int s_box[4];
  const int64_t v = xcd_swizzle(blockIdx.x, nwg);
  const int g = (int)(v % ngroups);
  const int64_t rest = v / ngroups;
  const int tile = (int)(rest % ((int64_t)ntx * nty)), b = (int)(rest / ((int64_t)ntx * nty));
  const int ty = tile / ntx, tx = tile - ty * ntx;
  const int c0 = g * G, gc = min(G, C - c0);
  const int ly = threadIdx.x / tw, lx = threadIdx.x - ly * tw;
  const int y = ty * th + ly, x = tx * tw + lx;
  const bool active = ly < th && y < H && x < W;
  const int HW = H * W, plane_sz = Hi * Wi;
  const int p = active ? y * W + x : 0;
  box_init(s_box);
  __syncthreads();
  Taps<A, KH> t;   // rows as INDICES (pitch 1) until the window is known
  {
    const T *i2 = in2 + (int64_t)b * 3 * HW + p;
    t.template init<2>(Num<T>::ld(i2), Num<T>::ld(i2 + HW), Num<T>::ld(i2 + 2 * HW), x, y, Hi, Wi, dil, false, 1);
  }
  box_reduce(s_box, active ? t.row_off(0) : 0x7fffffff, active ? t.col_off(0) : 0x7fffffff, active ? t.row_off(N - 1) : -1,
             active ? t.col_off(N - 1) : -1);
  __syncthreads();
  const T *in1_0 = in1 + ((int64_t)b * C + c0) * plane_sz;
  const bool vec = window_vec_ok(in1_0, plane_sz, Wi);
  const TileWin w = tile_window_vec(s_box, Wi, vec);
  const int g_fit = window_worth_staging(w, th, tw) ? min(gc, lds_elems / max(w.size, 1)) : 0;
  const int pitch = g_fit == 0 ? Wi : w.cols;
#pragma unroll
  for (int f = 0; f < KH; ++f) {
    t.yT[f] *= pitch;
    t.yB[f] *= pitch;
  }
  const T *go0 = MODE == 2 ? gout + ((int64_t)b * C + c0) * HW + p : nullptr;
  A rx = 0, ry = 0, rs = 0;
  if (abl & 1) {
    if (active && t.sum == (A)-1.25) outp[p] = Num<T>::from(t.sum);   // (keeps the setup alive)
    return;
  }
  if (g_fit == 0) {   // the tile reaches further than one channel's window holds: global memory for this tile
    if (active) {
      if constexpr (MODE == 0) rs_fwd_pixel<T, T, KH, A>(t, in1_0, plane_sz, outp + ((int64_t)b * C + c0) * HW + p, HW, gc);
      else rs_bwd2_pixel<T, T, KH, A>(t, in1_0, plane_sz, go0, HW, gc, rx, ry, rs);
    }
  } else {
    const A *win0 = planes - (w.ymin * w.cols + w.xmin);
    for (int cb = 0; cb < gc; cb += g_fit) {
      const int n = min(g_fit, gc - cb);
      if (!(abl & 2)) stage_windows<T, A>(in1_0 + (int64_t)cb * plane_sz, plane_sz, Wi, planes, w, n, vec && !(abl & 8));
      __syncthreads();
      if (active && !(abl & 4)) {
        if constexpr (MODE == 0) {
          rs_fwd_pixel<T, A, KH, A>(t, win0, w.size, outp + ((int64_t)b * C + c0 + cb) * HW + p, HW, n);
        } else {   // linear in the channel sums: rounds combine by addition
          A ax_, ay_, as_;
          rs_bwd2_pixel<T, A, KH, A>(t, win0, w.size, go0 + (int64_t)cb * HW, HW, n, ax_, ay_, as_);
          rx += ax_, ry += ay_, rs += as_;
        }
      }
      __syncthreads();
    }
  }
  if constexpr (MODE == 2) {
    if (active) {
      T *o = outp + (int64_t)b * 3 * HW + p;
      if (ngroups == 1) {  // sole writer of this pixel
        o[0] = Num<T>::from(Num<T>::ld(o) + rx);
        o[HW] = Num<T>::from(Num<T>::ld(o + HW) + ry);
        o[2 * HW] = Num<T>::from(Num<T>::ld(o + 2 * HW) + rs);
      } else {
        atomic_add(o, (T)rx);
        atomic_add(o + HW, (T)ry);
        atomic_add(o + 2 * HW, (T)rs);
      }
    }
  }
}

// (Measured and removed, round 5: the forward as ONE workgroup per 8 x 32 tile for all channels, two halves of 256 threads,
// the windows streamed through two LDS buffers with the next stage's 16-byte requests in flight under the channel loop --
// setup 2x instead of 4-8x, staging overlapped.  22.8 us against this kernel's 19.7 at (1,64,256,176), 73 against 31 on a wild
// flow: 192 workgroups of two waves per SIMD leave a quarter of the CUs idle and hide less latency than the extra setups
// cost.  profiles/r5_config2_sweeps.txt, session s15; the kernel is in the history of this file.)

// d/d input1 is a SCATTER: workgroup = (tile of th x tw pixels, G channels), one pixel per thread with its taps kept in
// registers.  The gradient planes are accumulated in an LDS window = the bounding box of the tile's taps (from the flow, on
// the device), 64-bit fixed point for float (ds_add_u64, order-independent; lds_plane.h) / double planes for double, in as
// many channel rounds as the window allows, and leave through one atomic per touched element.  A window that does not
// hold a single channel: global atomics for that tile.
template <typename T, int KH, bool FIX>
__global__ __launch_bounds__(512) void rs_bwd1_tile_kernel(const T *__restrict__ in2, const T *__restrict__ gout,
                                                          T *__restrict__ gin1, int C, int Hi, int Wi, int H, int W, int dil,
                                                          int trunc, int th, int tw, int ntx, int nty, int G, int ngroups,
                                                          int lds_elems, int64_t nwg, int abl) {
  // abl (tuning key 39, timing ablations, results garbage): 1 = stop after setup / box / scale, 2 = no scatter, 4 = no flush
  using A = typename Num<T>::acc;
  using PT = typename std::conditional<FIX, lds_fix_t, lds_acc_t>::type;
  static_assert(!FIX || std::is_same<A, float>::value, "fixed-point planes: float scatter only");
  constexpr int N = 2 * KH;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  PT *planes = reinterpret_cast<PT *>(gfla_smem);
  __shared__ int s_box[4];
  __shared__ unsigned s_amax;
  const int64_t v = xcd_swizzle(blockIdx.x, nwg);
  const int g = (int)(v % ngroups);
  const int64_t rest = v / ngroups;
  const int tile = (int)(rest % ((int64_t)ntx * nty)), b = (int)(rest / ((int64_t)ntx * nty));
  const int ty = tile / ntx, tx = tile - ty * ntx;
  const int c0 = g * G, gc = min(G, C - c0);
  const int ly = threadIdx.x / tw, lx = threadIdx.x - ly * tw;
  const int y = ty * th + ly, x = tx * tw + lx;
  const bool active = ly < th && y < H && x < W;
  const int HW = H * W, plane_sz = Hi * Wi;
  const int p = active ? y * W + x : 0;
  box_init(s_box);
  if (threadIdx.x == 0) s_amax = 0;
  __syncthreads();
  // ---- taps of this thread's pixel (rows as INDICES: pitch 1), kept for all channel rounds
  int row[N], col[N];
  A qy[N], wx[N];
  {
    const T *i2 = in2 + (int64_t)b * 3 * HW + p;
    Taps<A, KH> t;
    t.template init<2>(Num<T>::ld(i2), Num<T>::ld(i2 + HW), Num<T>::ld(i2 + 2 * HW), x, y, Hi, Wi, dil, (trunc & 1) != 0, 1);
#pragma unroll
    for (int r = 0; r < N; ++r) {
      row[r] = t.row_off(r);
      col[r] = t.col_off(r);
      qy[r] = (A)safe_div<A>(t.row_w(r), t.sum);
      wx[r] = t.col_w(r);
    }
  }
  // clamped taps are monotone in position order: the first and the last bound the pixel's reach
  box_reduce(s_box, active ? row[0] : 0x7fffffff, active ? col[0] : 0x7fffffff, active ? row[N - 1] : -1, active ? col[N - 1] : -1);
  const T *go0 = gout + ((int64_t)b * C + c0) * HW + p;
  T *gin0 = gin1 + ((int64_t)b * C + c0) * plane_sz;
  FixScale fix{1.f, 1.0, true};
  if constexpr (FIX) {  // every contribution is one of these gradients times weights <= 1: their maximum sets the scale
    unsigned m = 0;
    if (active)
      for (int c = 0; c < gc; ++c) m = max(m, __float_as_uint(fabsf(Num<T>::ld(go0 + (int64_t)c * HW))));
    fix = fix_scale(block_umax(m, &s_amax));   // (contains the barrier that also publishes the box)
  } else {
    __syncthreads();
  }
  const int ymin = s_box[0], xmin = s_box[1];
  const int rows = s_box[2] - ymin + 1, cols = s_box[3] - xmin + 1;
  const int win = rows * cols;
  const int g_fit = min(gc, lds_elems / max(win, 1));
  if (abl & 1) {
    if (active && qy[0] == (A)-1.25) gin0[p] = Num<T>::from(qy[0] * fix.up);
    return;
  }
  if (g_fit == 0) {
    if (active) {
      int ro[N];
#pragma unroll
      for (int r = 0; r < N; ++r) ro[r] = row[r] * Wi;
      rs_bwd1_apply<T, T, N, A, GlobalPlane>(ro, col, qy, wx, go0, HW, gin0, plane_sz, gc, (A)1);
    }
    return;
  }
  int ro[N], co[N];
#pragma unroll
  for (int r = 0; r < N; ++r) {
    ro[r] = (row[r] - ymin) * cols;
    co[r] = col[r] - xmin;
  }
  for (int cb = 0; cb < gc; cb += g_fit) {
    const int n = min(g_fit, gc - cb);
    zero_planes<PT>(planes, n * win);
    __syncthreads();
    if (active && !(abl & 2)) {
      const T *go = go0 + (int64_t)cb * HW;
      if constexpr (FIX)
        rs_bwd1_apply_fix<T, N>(ro, co, qy, wx, Num<T>::ld(go), go, HW, planes, win, n, fix.up);
      else
        rs_bwd1_apply<T, PT, N, A, LdsPlane>(ro, co, qy, wx, go, HW, planes, win, n, (A)1);
    }
    __syncthreads();
    if (!(abl & 4)) {
      const TileWin w{ymin, xmin, rows, cols, win};
      const double down = fix.finite ? fix.down : __longlong_as_double(0x7ff8000000000000ll);
      const bool finite = fix.finite;
      flush_windows<T>(gin0 + (int64_t)cb * plane_sz, plane_sz, Wi, w, n, [=](int i) {
        const PT raw = planes[i];
        if constexpr (FIX) return raw == 0 ? 0.0 : (finite ? (double)raw * down : down);
        else return (double)raw;
      });
    }
    __syncthreads();
  }
}

struct Geo {
  int cpt, ncg, sp_blocks;
  int64_t blocks;
};
static Geo geometry(int64_t B, int64_t C, int64_t H, int64_t W, int max_cpt, int64_t want_waves) {
  Geo g;
  g.sp_blocks = (int)ceil_div(H * W, kBlock);
  int cpt = tuning(1) > 0 ? tuning(1) : pick_channels_per_thread((int64_t)g.sp_blocks * kBlock, C, B, max_cpt, want_waves);
  if (cpt > C) cpt = (int)C;
  g.cpt = cpt;
  g.ncg = (int)ceil_div(C, cpt);
  g.blocks = (int64_t)g.sp_blocks * g.ncg * B;
  return g;
}

// big-plane gathers: channels per thread -- all of them unless that leaves fewer than `want` waves (tuning key 33)
static int big_cpt(int64_t B, int64_t C, int64_t nsp, int64_t want) {
  if (tuning(33) > 0) return tuning(33) < C ? tuning(33) : (int)C;
  int64_t cpt = C;
  while (cpt > 1 && B * nsp * (kBlock / 64) * ceil_div(C, cpt) < want) cpt = ceil_div(cpt, 2);
  return (int)cpt;
}

template <typename T>
static int check(const void *a, const void *b, int64_t B, int64_t C, int64_t Hi, int64_t Wi, int64_t H,
                 int64_t W, int k, int dil) {
  if (!a || !b) return GFLA_ERR_NULL_POINTER;
  if (B <= 0 || C <= 0 || Hi <= 0 || Wi <= 0 || H <= 0 || W <= 0 || k < 2 || dil < 1) return GFLA_ERR_BAD_SHAPE;
  if (k / 2 > 4) return GFLA_ERR_UNSUPPORTED;  // kernel_size up to 9
  if (Hi * Wi > 0x7fffffffLL || H * W > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  return GFLA_OK;
}

#define GFLA_KH_SWITCH(KHV, ...) \
  switch (KHV) {                  \
    case 1: { constexpr int KH = 1; __VA_ARGS__; } break; \
    case 2: { constexpr int KH = 2; __VA_ARGS__; } break; \
    case 3: { constexpr int KH = 3; __VA_ARGS__; } break; \
    default: { constexpr int KH = 4; __VA_ARGS__; } break; \
  }

template <typename T>
static int resample2d_fwd(const T *in1, const T *in2, T *out, int64_t B, int64_t C, int64_t Hi,
                          int64_t Wi, int64_t H, int64_t W, int k, int dil, gfla_stream_t stream_) {
  int st = check<T>(in1, in2, B, C, Hi, Wi, H, W, k, dil);
  if (st != GFLA_OK) return st;
  if (!out) return GFLA_ERR_NULL_POINTER;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  using A = typename Num<T>::acc;
  if (tuning(6) != 1 && big_plane_regime(B, C, Hi * Wi * (int64_t)sizeof(A), lds_budget())) {
    // few planes far beyond the LDS budget (BASELINE configs[1]): one tap setup per pixel and chunk of channels, global gathers
    if (tuning(38) != 1 && sizeof(T) >= 4 && Hi * Wi <= 0x3fffffffLL) {   // planes staged into bounding-box windows
      const BigGeo bg = big_geometry(2, B, C, H, W, (k - 1) * dil + 1, (int)sizeof(A));
      const TileGeo tg = bg.tg;
      const int G = bg.G;
      const int64_t ngroups = bg.ngroups, nwg = bg.nwg;
      if (nwg <= 0x7fffffffLL) {
        const unsigned lds_bytes = bg.lds_bytes;
        GFLA_KH_SWITCH(k / 2, launch_lds(rs_gather_tile_kernel<T, KH, 0>, dim3((unsigned)nwg), dim3((unsigned)tg.threads), lds_bytes, stream, in1, in2, static_cast<const T *>(nullptr), out, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, tg.th, tg.tw, tg.ntx, tg.nty, G, (int)ngroups, (int)(lds_bytes / sizeof(A)), nwg, tile_probe_bits()));
        note_path(GFLA_PATH_RS_FWD_BIG);
        return launch_status();
      }
    }
    const int64_t nsp = ceil_div(H * W, kBlock);
    const int cpt = big_cpt(B, C, nsp, 20 * kNumCU);
    const int64_t ncg = ceil_div(C, cpt), nwg = B * nsp * ncg;
    if (nwg <= 0x7fffffffLL) {
      GFLA_KH_SWITCH(k / 2, rs_fwd_big_kernel<T, KH><<<dim3((unsigned)nwg), dim3(kBlock), 0, stream>>>(in1, in2, out, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, cpt, (int)ncg, (int)nsp, nwg));
      note_path(GFLA_PATH_RS_FWD_BIG);
      return launch_status();
    }
  }
  if (tuning(6) != 1) {
    PlaneGeo pg = lds_geometry(Hi, Wi, sizeof(A), B, C, H, W, (k - 1) * dil + 1, 1);
    if (pg.G > 0) {
      const int64_t blocks = B * pg.ngroups * pg.split;
      if (blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
      GFLA_KH_SWITCH(k / 2, if (pg.margin < 0) launch_lds(rs_lds_kernel<T, KH, 0, false>, dim3((unsigned)blocks), dim3(kLdsThreads), pg.lds_bytes, stream, in1, in2, nullptr, out, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, 0, pg.G, pg.ngroups, pg.split, pg.per, pg.margin, nullptr, 0u);
                                else launch_lds(rs_lds_kernel<T, KH, 0, true>, dim3((unsigned)blocks), dim3(kLdsThreads), pg.lds_bytes, stream, in1, in2, nullptr, out, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, 0, pg.G, pg.ngroups, pg.split, pg.per, pg.margin, nullptr, 0u));
      return launch_status();
    }
  }
  Geo g = geometry(B, C, H, W, 16, 4 * kNumCU * kWavesPerCU);
  if (g.blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  GFLA_KH_SWITCH(k / 2, rs_fwd_kernel<T, KH><<<dim3((unsigned)g.blocks), dim3(kBlock), 0, stream>>>(in1, in2, out, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, g.cpt, g.ncg, g.sp_blocks));
  return launch_status();
}

template <typename T>
static int resample2d_bwd(const T *in1, const T *in2, const T *gout, T *gin1, typename Num<T>::acc *gin2, int64_t B,
                          int64_t C, int64_t Hi, int64_t Wi, int64_t H, int64_t W, int k, int dil,
                          int trunc, gfla_stream_t stream_, const unsigned *skip_stat = nullptr,
                          unsigned skip_limit = 0, void *tap_records = nullptr) {
  int st = check<T>(in1, in2, B, C, Hi, Wi, H, W, k, dil);
  if (st != GFLA_OK) return st;
  if (!gout) return GFLA_ERR_NULL_POINTER;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  using A = typename Num<T>::acc;
  constexpr bool kBf16 = sizeof(T) == 2;
  if constexpr (!kBf16) {
    // few planes far beyond the LDS budget (BASELINE configs[1]; tile_map.h).  Not behind the matrix-core product's
    // device-side switch (skip_stat): the _ws entry point does not try the product in this regime.
    if (tuning(6) != 1 && !skip_stat && Hi * Wi <= 0x3fffffffLL &&
        big_plane_regime(B, C, Hi * Wi * (int64_t)sizeof(lds_acc_t), lds_budget())) {
      if (gin1) {
        if (trunc & 2) {  // the tiles accumulate with atomics: zero-fill here
          if (hipMemsetAsync(gin1, 0, (size_t)(B * C * Hi * Wi) * sizeof(T), stream) != hipSuccess) return GFLA_ERR_LAUNCH;
          trunc &= ~2;
        }
        const BigGeo bg = big_geometry(3, B, C, H, W, (k - 1) * dil + 1, 8);
        const TileGeo tg = bg.tg;
        const int G = bg.G;
        const int64_t ngroups = bg.ngroups, nwg = bg.nwg;
        if (nwg > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
        const unsigned lds_bytes = bg.lds_bytes;
        constexpr bool FIX = std::is_same<A, float>::value;
        GFLA_KH_SWITCH(k / 2, launch_lds(rs_bwd1_tile_kernel<T, KH, FIX>, dim3((unsigned)nwg), dim3((unsigned)tg.threads), lds_bytes, stream, in2, gout, gin1, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, trunc, tg.th, tg.tw, tg.ntx, tg.nty, G, (int)ngroups, (int)(lds_bytes / 8), nwg, tile_probe_bits()));
        note_path(GFLA_PATH_RS_BWD1_TILE);
        st = launch_status();
        if (st != GFLA_OK) return st;
      }
      if (gin2 && tuning(38) != 1) {
        const BigGeo bg = big_geometry(4, B, C, H, W, (k - 1) * dil + 1, (int)sizeof(A));
        const TileGeo tg = bg.tg;
        const int G = bg.G;
        const int64_t ngroups = bg.ngroups, nwg = bg.nwg;
        if (nwg > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
        const unsigned lds_bytes = bg.lds_bytes;
        GFLA_KH_SWITCH(k / 2, launch_lds(rs_gather_tile_kernel<T, KH, 2>, dim3((unsigned)nwg), dim3((unsigned)tg.threads), lds_bytes, stream, in1, in2, gout, gin2, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, tg.th, tg.tw, tg.ntx, tg.nty, G, (int)ngroups, (int)(lds_bytes / sizeof(A)), nwg, tile_probe_bits()));
        note_path(GFLA_PATH_RS_BWD2_BIG);
        st = launch_status();
      } else if (gin2) {
        const int64_t nsp = ceil_div(H * W, kBlock);
        const int cpt = big_cpt(B, C, nsp, 20 * kNumCU);
        const int64_t ncg = ceil_div(C, cpt), nwg = B * nsp * ncg;
        if (nwg > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
        GFLA_KH_SWITCH(k / 2, rs_bwd2_big_kernel<T, KH><<<dim3((unsigned)nwg), dim3(kBlock), 0, stream>>>(in1, in2, gout, gin2, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, cpt, (int)ncg, (int)nsp, nwg));
        note_path(GFLA_PATH_RS_BWD2_BIG);
        st = launch_status();
      }
      return st;
    }
  }
  // double scatter planes.  bf16 storage: whole planes, one owner each (the flush is a plain read-modify-write)
  PlaneGeo pg1 = kBf16 ? plane_geometry(Hi * Wi, sizeof(lds_acc_t), B, C, H * W, false)
                       : lds_geometry(Hi, Wi, sizeof(lds_acc_t), B, C, H, W, (k - 1) * dil + 1);
  PlaneGeo pg2 = lds_geometry(Hi, Wi, sizeof(A), B, C, H, W, (k - 1) * dil + 1);          // gather planes
  // flag word `trunc`: bit 0 = the reference's int() truncation quirk; bit 1 = grad_in1 arrives UNINITIALISED and is to be
  // overwritten -- honoured by the kernels where every element has exactly one writer, zero-filled here otherwise
  const bool lds_path = (tuning(6) != 1 || kBf16) && pg1.G > 0 && pg2.G > 0;
  if (gin1 && (trunc & 2) && !(lds_path && pg1.split == 1 && pg1.margin < 0) && !skip_stat) {
    if (hipMemsetAsync(gin1, 0, (size_t)(B * C * Hi * Wi) * sizeof(T), stream) != hipSuccess) return GFLA_ERR_LAUNCH;
    trunc &= ~2;
  }
  if (lds_path) {
    if (gin1) {
      const int64_t blocks = B * pg1.ngroups * pg1.split;
      if (blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
#define GFLA_RS_SCATTER(FIX_)                                                                                                                                       \
      GFLA_KH_SWITCH(k / 2, if (pg1.margin < 0) launch_lds(rs_lds_kernel<T, KH, 1, false, FIX_>, dim3((unsigned)blocks), dim3(kLdsThreads), pg1.lds_bytes, stream, in1, in2, gout, gin1, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, trunc, pg1.G, pg1.ngroups, pg1.split, pg1.per, pg1.margin, skip_stat, skip_limit); \
                                else launch_lds(rs_lds_kernel<T, KH, 1, true, FIX_>, dim3((unsigned)blocks), dim3(kLdsThreads), pg1.lds_bytes, stream, in1, in2, gout, gin1, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, trunc, pg1.G, pg1.ngroups, pg1.split, pg1.per, pg1.margin, skip_stat, skip_limit))
      // fixed-point planes (lds_plane.h); tuning key 23 = 1: round 1's double planes
      if constexpr (std::is_same<T, float>::value) {
        // scratch given, kernel_size 4 / 5: the per-pixel setup runs once (rs_tap_table_kernel); tuning key 23 = 2: never
        if (tap_records && k / 2 == 2 && Hi <= 65535 && Wi <= 65535 && tuning(23) == 0) {
          RsTapRec *tab = static_cast<RsTapRec *>(tap_records);
          rs_tap_table_kernel<<<dim3((unsigned)ceil_div(H * W, 256), (unsigned)B), 256, 0, stream>>>(
              in2, tab, (int)(H * W), (int)W, (int)Hi, (int)Wi, dil, trunc, skip_stat, skip_limit);
          const float *recs = reinterpret_cast<const float *>(tab);
          if (pg1.margin < 0)
            launch_lds(rs_lds_kernel<T, 2, 1, false, true, true>, dim3((unsigned)blocks), dim3(kLdsThreads), pg1.lds_bytes, stream, in1, recs, gout, gin1, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, trunc, pg1.G, pg1.ngroups, pg1.split, pg1.per, pg1.margin, skip_stat, skip_limit);
          else
            launch_lds(rs_lds_kernel<T, 2, 1, true, true, true>, dim3((unsigned)blocks), dim3(kLdsThreads), pg1.lds_bytes, stream, in1, recs, gout, gin1, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, trunc, pg1.G, pg1.ngroups, pg1.split, pg1.per, pg1.margin, skip_stat, skip_limit);
        } else if (tuning(23) == 1) {
          GFLA_RS_SCATTER(false);
        } else {
          GFLA_RS_SCATTER(true);
        }
      } else if constexpr (std::is_same<A, float>::value) {
        GFLA_RS_SCATTER(true);
      } else {
        GFLA_RS_SCATTER(false);
      }
#undef GFLA_RS_SCATTER
      st = launch_status();
      if (st != GFLA_OK) return st;
    }
    if (gin2) {
      if constexpr (std::is_same<A, float>::value) {
        // kernel_size 4, dilation 1: the aggregation's streaming d/d logits kernel with resample2d's epilogue
        // (rs_taps.h); tuning key 22 = 1: rs_lds_kernel<MODE 2>
        if (k == 4 && dil == 1 && tuning(22) != 1) {
          st = rs_bwd2_stream<T>(in1, in2, gout, gin2, B, C, Hi, Wi, H, W, stream);
          if (st != GFLA_ERR_UNSUPPORTED) return st;
        }
      }
      const int64_t blocks = B * pg2.ngroups * pg2.split;
      if (blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
      GFLA_KH_SWITCH(k / 2, if (pg2.margin < 0) launch_lds(rs_lds_kernel<T, KH, 2, false>, dim3((unsigned)blocks), dim3(kLdsThreads), pg2.lds_bytes, stream, in1, in2, gout, gin2, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, trunc, pg2.G, pg2.ngroups, pg2.split, pg2.per, pg2.margin, nullptr, 0u);
                                else launch_lds(rs_lds_kernel<T, KH, 2, true>, dim3((unsigned)blocks), dim3(kLdsThreads), pg2.lds_bytes, stream, in1, in2, gout, gin2, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, trunc, pg2.G, pg2.ngroups, pg2.split, pg2.per, pg2.margin, nullptr, 0u));
      st = launch_status();
    }
    return st;
  }
  if constexpr (kBf16) {
    return GFLA_ERR_UNSUPPORTED;  // bf16 storage: the planes-in-LDS kernels only
  } else {
  if (gin1) {
    Geo g = geometry(B, C, H, W, 16, 4 * kNumCU * kWavesPerCU);
    if (g.blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
    GFLA_KH_SWITCH(k / 2, rs_bwd1_kernel<T, KH><<<dim3((unsigned)g.blocks), dim3(kBlock), 0, stream>>>(in2, gout, gin1, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, trunc, g.cpt, g.ncg, g.sp_blocks));
    st = launch_status();
    if (st != GFLA_OK) return st;
  }
  if (gin2) {
    Geo g = geometry(B, C, H, W, 512, 2 * kNumCU * kWavesPerCU);
    if (g.blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
    GFLA_KH_SWITCH(k / 2, rs_bwd2_kernel<T, KH><<<dim3((unsigned)g.blocks), dim3(kBlock), 0, stream>>>(in1, in2, gout, gin2, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, g.cpt, g.ncg, g.sp_blocks));
    st = launch_status();
  }
  return st;
  }
}

}  // namespace gfla

using gfla::bf16_t;

extern "C" {
int gfla_resample2d_fwd_f32(const float *a, const float *b, float *o, int64_t B, int64_t C, int64_t Hi,
                            int64_t Wi, int64_t H, int64_t W, int k, int d, gfla_stream_t st) {
  return gfla::resample2d_fwd<float>(a, b, o, B, C, Hi, Wi, H, W, k, d, st);
}
int gfla_resample2d_fwd_f64(const double *a, const double *b, double *o, int64_t B, int64_t C,
                            int64_t Hi, int64_t Wi, int64_t H, int64_t W, int k, int d,
                            gfla_stream_t st) {
  return gfla::resample2d_fwd<double>(a, b, o, B, C, Hi, Wi, H, W, k, d, st);
}
int gfla_resample2d_fwd_bf16(const uint16_t *a, const uint16_t *b, uint16_t *o, int64_t B, int64_t C,
                             int64_t Hi, int64_t Wi, int64_t H, int64_t W, int k, int d,
                             gfla_stream_t st) {
  return gfla::resample2d_fwd<bf16_t>(reinterpret_cast<const bf16_t *>(a),
                                      reinterpret_cast<const bf16_t *>(b), reinterpret_cast<bf16_t *>(o),
                                      B, C, Hi, Wi, H, W, k, d, st);
}
int gfla_resample2d_bwd_f32(const float *a, const float *b, const float *go, float *g1, float *g2,
                            int64_t B, int64_t C, int64_t Hi, int64_t Wi, int64_t H, int64_t W, int k,
                            int d, int trunc, gfla_stream_t st) {
  return gfla::resample2d_bwd<float>(a, b, go, g1, g2, B, C, Hi, Wi, H, W, k, d, trunc, st);
}
/* As gfla_resample2d_bwd_f32 with scratch (gfla_scatter_workspace_bytes(B, H, W, k*k), may be NULL): d/d input1 runs as
 * a block-sparse product on the matrix cores (patch_mfma.hip) when dilation == 1 and the shape allows, else as before. */
int gfla_resample2d_bwd_ws_f32(const float *a, const float *b, const float *go, float *g1, float *g2, void *workspace,
                               int64_t B, int64_t C, int64_t Hi, int64_t Wi, int64_t H, int64_t W, int k, int d,
                               int trunc, gfla_stream_t st) {
  const unsigned *skip_stat = nullptr;
  unsigned skip_limit = 0;
  const bool big = gfla::tuning(6) != 1 && gfla::big_plane_regime(B, C, Hi * Wi * (int64_t)sizeof(gfla::lds_acc_t), gfla::lds_budget());
  if (g1 && workspace && d == 1 && gfla::tuning(6) != 1 && !big) {
    int rc = gfla::check<float>(a, b, B, C, Hi, Wi, H, W, k, d);
    if (rc != GFLA_OK) return rc;
    if (!go) return GFLA_ERR_NULL_POINTER;
    // adaptive only where the LDS-atomic kernel exists as the device-side fallback
    const bool lds_fallback = gfla::lds_geometry(Hi, Wi, sizeof(gfla::lds_acc_t), B, C, H, W, (k - 1) * d + 1).G > 0 &&
                              gfla::lds_geometry(Hi, Wi, sizeof(float), B, C, H, W, (k - 1) * d + 1).G > 0;
    if (trunc & 2) {  // overwrite requested: both the product and its device-side fallback must be sole writers
      const gfla::PlaneGeo pg1 = gfla::lds_geometry(Hi, Wi, sizeof(gfla::lds_acc_t), B, C, H, W, (k - 1) * d + 1);
      if (lds_fallback && !(pg1.split == 1 && pg1.margin < 0)) {
        if (hipMemsetAsync(g1, 0, (size_t)(B * C * Hi * Wi) * 4, static_cast<hipStream_t>(st)) != hipSuccess) return GFLA_ERR_LAUNCH;
        trunc &= ~2;
      }
    }
    rc = gfla::rs_input1_bwd_mfma(b, go, g1, workspace, B, C, Hi, Wi, H, W, k, trunc & 1, (trunc & 2) ? 0 : 1,
                                  lds_fallback ? 1 : 0, &skip_stat, &skip_limit, static_cast<hipStream_t>(st));
    if (rc == GFLA_OK) {
      if (!lds_fallback || skip_limit == 0xffffffffu) g1 = nullptr, skip_stat = nullptr;  // done unconditionally
    } else if (rc != GFLA_ERR_UNSUPPORTED) {
      return rc;
    } else {
      skip_stat = nullptr;
    }
    if (!g1 && !g2) return GFLA_OK;
  }
  void *taps = (workspace && g1) ? static_cast<unsigned char *>(workspace) + gfla::pm_table_bytes(B, H, W, k * k) : nullptr;
  return gfla::resample2d_bwd<float>(a, b, go, g1, g2, B, C, Hi, Wi, H, W, k, d, trunc, st, skip_stat, skip_limit, taps);
}
/* bf16 storage: grad_in1 bf16; grad_in2 (B,3,H,W) FLOAT32 (a reduction over the channels) */
int gfla_resample2d_bwd_bf16(const uint16_t *a, const uint16_t *b, const uint16_t *go, uint16_t *g1, float *g2, int64_t B,
                             int64_t C, int64_t Hi, int64_t Wi, int64_t H, int64_t W, int k, int d, int trunc,
                             gfla_stream_t st) {
  return gfla::resample2d_bwd<bf16_t>(reinterpret_cast<const bf16_t *>(a), reinterpret_cast<const bf16_t *>(b),
                                      reinterpret_cast<const bf16_t *>(go), reinterpret_cast<bf16_t *>(g1), g2, B, C, Hi,
                                      Wi, H, W, k, d, trunc, st);
}
/* Host-side launch geometry of the big-plane tile kernels (csrc/tile_map.h; no GPU needed: the CPU tests sweep it).
 * op: 0 block_extractor forward, 1 block_extractor backward (both gradients), 2 resample2d forward / d/d input2,
 * 3 resample2d d/d input1.  (H, W) = the flow / output grid, (Hs, Ws) = the source plane, span = taps per axis (kernel_size
 * + 1 for block_extractor, (kernel_size - 1) * dilation + 1 for resample2d), elem_size 4 / 8.
 * out[10] = in the regime by default (0 / 1), tile rows, tile columns, tiles along x, tiles along y, threads per workgroup,
 * channels per workgroup, channel groups, dynamic LDS bytes requested, workgroups. */
int gfla_big_plane_geometry(int op, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H, int64_t W, int span,
                            int elem_size, int64_t *out) {
  if (!out) return GFLA_ERR_NULL_POINTER;
  if (op < 0 || op > 4 || B <= 0 || C <= 0 || Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0 || span < 1 || (elem_size != 4 && elem_size != 8))
    return GFLA_ERR_BAD_SHAPE;
  const int per = op == 0 || op == 2 || op == 4 ? elem_size : op == 3 ? 8 : 8 + elem_size;
  const int64_t plane_bytes = Hs * Ws * (int64_t)(op == 1 ? 8 + elem_size : op == 3 ? 8 : elem_size);
  const gfla::BigGeo g = gfla::big_geometry(op, B, C, H, W, span, per);
  const int64_t v[10] = {gfla::big_plane_regime(B, C, plane_bytes, gfla::lds_budget()) ? 1 : 0, g.tg.th, g.tg.tw, g.tg.ntx, g.tg.nty,
                         g.tg.threads, g.G, g.ngroups, g.lds_bytes, g.nwg};
  for (int i = 0; i < 10; ++i) out[i] = v[i];
  return GFLA_OK;
}
/* The workgroup -> work-item remap of those kernels (a bijection of [0, nwg) that hands the blocks of each of the 8 XCDs a
 * contiguous range); -1 outside [0, nwg). */
int64_t gfla_xcd_swizzle(int64_t block, int64_t nwg) {
  return (block < 0 || block >= nwg) ? -1 : gfla::xcd_swizzle(block, nwg);
}
int gfla_resample2d_bwd_f64(const double *a, const double *b, const double *go, double *g1, double *g2,
                            int64_t B, int64_t C, int64_t Hi, int64_t Wi, int64_t H, int64_t W, int k,
                            int d, int trunc, gfla_stream_t st) {
  return gfla::resample2d_bwd<double>(a, b, go, g1, g2, B, C, Hi, Wi, H, W, k, d, trunc, st);
}
}
