// block_extractor backward with planes in LDS, shared by block_extractor.hip (gradient arriving as
// the (B,C,kH,kW) tensor) and local_attn_aggregate.hip (gradient arriving factored as
// attention[b,ij,p] * grad_out[b,c,p] / k^2, never materialised).
//
// workgroup <-> (b, group of G channels[, 1/split of the flow pixels]); lane <-> flow pixel.
// All k*k taps of one pixel share one fractional offset, so together they touch a dense
// (k+1)x(k+1) patch of the source plane.  The lane folds the k*k incoming gradients into that
// patch in registers (two patch rows live at a time) and issues (k+1)^2 ds_add_f64 per channel
// instead of 4*k*k; d/dflow is reduced in registers over taps and channels and leaves the lane as
// one atomic pair per group.  Source values for d/dflow come from fp32 planes staged in LDS.
#pragma once

#include "lds_plane.h"

namespace gfla {

// Gradient reaching block_source[b, c, yf*K+i, xf*K+j] for j = 0..K-1 of tap row i.
// MODE 0: gout is the reference-layout tensor (B,C,K*Hf,K*Wf)
// MODE 1: gout is (B,C,Hf,Wf) and the gradient is attn[b,ij,p] * gout[b,c,p] / K^2 (never materialised)
// MODE 2: gout is in "unfold" layout (B, C*K*K, Hf, Wf), channel index c*K*K + i*K + j
// MODE 3: the SUM of an unfold-layout gradient (gout) and a factored one (attn, gout2): both gradient
//         streams that reach block_source inside ExtractorAttn -- through the FC layer and through
//         the attention-weighted aggregation -- scattered in ONE pass over the taps.
constexpr int kGoutTensor = 0, kGoutAttn = 1, kGoutUnfold = 2, kGoutUnfoldAttn = 3;
template <typename T, int K, int MODE>
struct GoutRow {
  using A = typename Num<T>::acc;
  // tensor form: base = &gout[b, c0, yf*K, xf*K], row pitch Wo, channel pitch K*Hf*Wo
  // attention form: attn_p = &attn[b, 0, p] (channel pitch HW), go_p = &gout[b, c0, p] (pitch HW)
  const T *base;
  const T *attn_p;
  const T *base2;  // MODE 3: &gout2[b, c0, p]
  int64_t cstride;
  int pitch;
  int HW;
  A inv_kk;

  __device__ __forceinline__ void load(int c, int i, A (&g)[K]) const {
    if constexpr (MODE == kGoutUnfoldAttn) {
      const A go = Num<T>::ld(base2 + (int64_t)c * HW) * inv_kk;
      const T *r = base + ((int64_t)c * K * K + i * K) * cstride;
#pragma unroll
      for (int j = 0; j < K; ++j)
        g[j] = Num<T>::ld(r + (int64_t)j * cstride) + Num<T>::ld(attn_p + (int64_t)(i * K + j) * HW) * go;
    } else if constexpr (MODE == kGoutAttn) {
      const A go = Num<T>::ld(base + (int64_t)c * cstride) * inv_kk;
#pragma unroll
      for (int j = 0; j < K; ++j) g[j] = Num<T>::ld(attn_p + (int64_t)(i * K + j) * HW) * go;
    } else if constexpr (MODE == kGoutUnfold) {
      const T *r = base + ((int64_t)c * K * K + i * K) * cstride;
#pragma unroll
      for (int j = 0; j < K; ++j) g[j] = Num<T>::ld(r + (int64_t)j * cstride);
    } else {
      const T *r = base + (int64_t)c * cstride + (int64_t)i * pitch;
#pragma unroll
      for (int j = 0; j < K; ++j) g[j] = Num<T>::ld(r + j);
    }
  }
};

// Per-workgroup constants of the scatter kernel.
template <typename T>
struct BwdCtx {
  using A = typename Num<T>::acc;
  const T *flow, *gout, *attn, *gout2;
  const T *src0;          // &src[b, c0]
  T *gsrc0, *gflow;       // &grad_source[b, c0] (global, outlier path), grad_flow
  lds_acc_t *gplanes0;    // LDS gradient planes, shifted so plane-relative offsets index the window
  const A *splanes0;      // LDS source planes, shifted likewise
  int b, C, c0, gc, Hs, Ws, Hf, Wf, HW, plane_sz, win_sz, win_lo, win_rows;
  int64_t u_cs, u_bs;
};

// Tap geometry of one flow pixel: column taps in registers, row taps recomputed per tap row.
template <typename A, int K>
struct PixelTaps {
  int p, yf, xf, x0, y0;
  A fy0;
  int xL[K], xR[K];
  A ax[K];
  bool dense;   // floor(tap t) == floor(tap 0) + t for every t, in x and y
  bool inside;  // every source row this pixel can touch is resident in LDS

  template <typename T, bool WIN>
  __device__ __forceinline__ void init(const BwdCtx<T> &cx, int p_) {
    p = p_;
    yf = p / cx.Wf;
    xf = p - yf * cx.Wf;
    const A fx0 = Num<T>::ld(cx.flow + (int64_t)(cx.b * 2 + 0) * cx.HW + p);
    fy0 = Num<T>::ld(cx.flow + (int64_t)(cx.b * 2 + 1) * cx.HW + p);
    dense = true;
    x0 = y0 = 0;
#pragma unroll
    for (int t = 0; t < K; ++t) {
      const A dx = (fx0 + (A)(t - K / 2)) + (A)xf;  // block_extractor_kernel.cu:132-136
      const A dy = (fy0 + (A)(t - K / 2)) + (A)yf;
      const A fdx = floor_t<A>(dx), fdy = floor_t<A>(dy);
      if (t == 0) {
        x0 = (int)fdx;
        y0 = (int)fdy;
      }
      dense = dense && ((int)fdx == x0 + t) && ((int)fdy == y0 + t);
      xL[t] = clampi((int)fdx, 0, cx.Ws - 1);
      xR[t] = clampi((int)(fdx + 1), 0, cx.Ws - 1);
      ax[t] = dx - fdx;
    }
    // one row of slack covers the non-dense rounding case
    inside = !WIN || (clampi(y0 - 1, 0, cx.Hs - 1) >= cx.win_lo && clampi(y0 + K + 1, 0, cx.Hs - 1) < cx.win_lo + cx.win_rows);
  }
  // fractional y weight (bottom) of tap row i
  __device__ __forceinline__ A yfrac(int i, A &fdy_out) const {
    const A dy = (fy0 + (A)(i - K / 2)) + (A)yf;
    fdy_out = floor_t<A>(dy);
    return dy - fdy_out;
  }
};

template <typename T, int K, int MODE>
__device__ __forceinline__ GoutRow<T, K, MODE> make_gout_row(const BwdCtx<T> &cx, int p, int yf, int xf) {
  using A = typename Num<T>::acc;
  GoutRow<T, K, MODE> gr;
  gr.base2 = nullptr;
  const int Wo = K * cx.Wf;
  if constexpr (MODE == kGoutUnfoldAttn) {
    gr.base = cx.gout + (int64_t)cx.b * cx.u_bs + (int64_t)cx.c0 * K * K * cx.u_cs + p;
    gr.attn_p = cx.attn + (int64_t)cx.b * K * K * cx.HW + p;
    gr.base2 = cx.gout2 + ((int64_t)cx.b * cx.C + cx.c0) * cx.HW + p;
    gr.cstride = cx.u_cs;
    gr.pitch = 0;
  } else if constexpr (MODE == kGoutAttn) {
    gr.base = cx.gout + ((int64_t)cx.b * cx.C + cx.c0) * cx.HW + p;
    gr.attn_p = cx.attn + (int64_t)cx.b * K * K * cx.HW + p;
    gr.cstride = cx.HW;
    gr.pitch = 0;
  } else if constexpr (MODE == kGoutUnfold) {  // element (b, ch, p) at b*u_bs + ch*u_cs + p
    gr.base = cx.gout + (int64_t)cx.b * cx.u_bs + (int64_t)cx.c0 * K * K * cx.u_cs + p;
    gr.attn_p = nullptr;
    gr.cstride = cx.u_cs;
    gr.pitch = 0;
  } else {
    gr.base = cx.gout + ((int64_t)cx.b * cx.C + cx.c0) * ((int64_t)K * cx.Hf * Wo) + (int64_t)(yf * K) * Wo + xf * K;
    gr.attn_p = nullptr;
    gr.cstride = (int64_t)K * cx.Hf * Wo;
    gr.pitch = Wo;
  }
  gr.HW = cx.HW;
  gr.inv_kk = (A)1 / (A)(K * K);
  return gr;
}

// ---- one pixel --------------------------------------------------------------------------------
template <typename T, int K, bool NEED_SRC, bool NEED_FLOW, int MODE>
__device__ __forceinline__ void be_bwd_single(const BwdCtx<T> &cx, const PixelTaps<typename Num<T>::acc, K> &px) {
  using A = typename Num<T>::acc;
  const GoutRow<T, K, MODE> gr = make_gout_row<T, K, MODE>(cx, px.p, px.yf, px.xf);
  A gx_acc = 0, gy_acc = 0;
  if (!px.inside) {
    // flow beyond the window's margin: this pixel alone takes the reference's decomposition on
    // global memory (block_extractor_kernel.cu:123-168)
    for (int c = 0; c < cx.gc; ++c) {
      const T *pl = cx.src0 + (int64_t)c * cx.plane_sz;
      T *gp = NEED_SRC ? cx.gsrc0 + (int64_t)c * cx.plane_sz : nullptr;
#pragma unroll 1
      for (int i = 0; i < K; ++i) {
        A fdy;
        const A yB_P = px.yfrac(i, fdy), yT_P = 1 - yB_P;
        const int yT = clampi((int)fdy, 0, cx.Hs - 1) * cx.Ws, yB = clampi((int)(fdy + 1), 0, cx.Hs - 1) * cx.Ws;
        A gv[K];
        gr.load(c, i, gv);
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const A xL_P = 1 - px.ax[j], xR_P = px.ax[j];
          if (NEED_FLOW) {
            const A vTL = Num<T>::ld(pl + yT + px.xL[j]), vTR = Num<T>::ld(pl + yT + px.xR[j]);
            const A vBL = Num<T>::ld(pl + yB + px.xL[j]), vBR = Num<T>::ld(pl + yB + px.xR[j]);
            gy_acc += gv[j] * (-xL_P * vTL - xR_P * vTR + xL_P * vBL + xR_P * vBR);
            gx_acc += gv[j] * (-yT_P * vTL - yB_P * vBL + yT_P * vTR + yB_P * vBR);
          }
          if (NEED_SRC) {
            atomic_add(gp + yT + px.xL[j], (T)(gv[j] * xL_P * yT_P));
            atomic_add(gp + yT + px.xR[j], (T)(gv[j] * xR_P * yT_P));
            atomic_add(gp + yB + px.xL[j], (T)(gv[j] * xL_P * yB_P));
            atomic_add(gp + yB + px.xR[j], (T)(gv[j] * xR_P * yB_P));
          }
        }
      }
    }
  } else if (px.dense) {
    int col[K + 1];
#pragma unroll
    for (int q = 0; q <= K; ++q) col[q] = clampi(px.x0 + q, 0, cx.Ws - 1);
    for (int c = 0; c < cx.gc; ++c) {
      lds_acc_t *gp = cx.gplanes0 + (size_t)c * cx.win_sz;
      const A *spl = cx.splanes0 + (size_t)c * cx.win_sz;
      A rowA[K + 1], vA[K + 1];
      int offA = clampi(px.y0, 0, cx.Hs - 1) * cx.Ws;
#pragma unroll
      for (int q = 0; q <= K; ++q) {
        rowA[q] = 0;
        vA[q] = NEED_FLOW ? spl[offA + col[q]] : (A)0;
      }
#pragma unroll 1
      for (int i = 0; i < K; ++i) {
        A fdy;
        const A yB_P = px.yfrac(i, fdy), yT_P = 1 - yB_P;
        const int offB = clampi(px.y0 + i + 1, 0, cx.Hs - 1) * cx.Ws;
        A gv[K];
        gr.load(c, i, gv);
        A rowB[K + 1], vB[K + 1];
#pragma unroll
        for (int q = 0; q <= K; ++q) {
          rowB[q] = 0;
          vB[q] = NEED_FLOW ? spl[offB + col[q]] : (A)0;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const A xL_P = 1 - px.ax[j], xR_P = px.ax[j];
          if (NEED_SRC) {  // block_extractor_kernel.cu:158-161, folded into the patch
            rowA[j] += gv[j] * xL_P * yT_P;
            rowA[j + 1] += gv[j] * xR_P * yT_P;
            rowB[j] += gv[j] * xL_P * yB_P;
            rowB[j + 1] += gv[j] * xR_P * yB_P;
          }
          if (NEED_FLOW) {  // :163-164
            gy_acc += gv[j] * (-xL_P * vA[j] - xR_P * vA[j + 1] + xL_P * vB[j] + xR_P * vB[j + 1]);
            gx_acc += gv[j] * (-yT_P * vA[j] - yB_P * vB[j] + yT_P * vA[j + 1] + yB_P * vB[j + 1]);
          }
        }
        if (NEED_SRC) {
#pragma unroll
          for (int q = 0; q <= K; ++q)
            if (rowA[q] != 0) lds_add(gp + offA + col[q], (lds_acc_t)rowA[q]);
        }
#pragma unroll
        for (int q = 0; q <= K; ++q) {
          rowA[q] = rowB[q];
          vA[q] = vB[q];
        }
        offA = offB;
      }
      if (NEED_SRC) {
#pragma unroll
        for (int q = 0; q <= K; ++q)
          if (rowA[q] != 0) lds_add(gp + offA + col[q], (lds_acc_t)rowA[q]);
      }
    }
  } else {
    // a tap's floor() landed one off the dense patch (flow within rounding of an integer): the
    // reference's own tap-by-tap form, rolled
    for (int c = 0; c < cx.gc; ++c) {
      lds_acc_t *gp = cx.gplanes0 + (size_t)c * cx.win_sz;
      const A *spl = cx.splanes0 + (size_t)c * cx.win_sz;
#pragma unroll 1
      for (int i = 0; i < K; ++i) {
        A fdy;
        const A yB_P = px.yfrac(i, fdy), yT_P = 1 - yB_P;
        const int yT = clampi((int)fdy, 0, cx.Hs - 1) * cx.Ws, yB = clampi((int)(fdy + 1), 0, cx.Hs - 1) * cx.Ws;
        A gv[K];
        gr.load(c, i, gv);
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const A xL_P = 1 - px.ax[j], xR_P = px.ax[j];
          if (NEED_FLOW) {
            const A vTL = spl[yT + px.xL[j]], vTR = spl[yT + px.xR[j]], vBL = spl[yB + px.xL[j]], vBR = spl[yB + px.xR[j]];
            gy_acc += gv[j] * (-xL_P * vTL - xR_P * vTR + xL_P * vBL + xR_P * vBR);
            gx_acc += gv[j] * (-yT_P * vTL - yB_P * vBL + yT_P * vTR + yB_P * vBR);
          }
          if (NEED_SRC) {
            lds_add(gp + yT + px.xL[j], (lds_acc_t)(gv[j] * xL_P * yT_P));
            lds_add(gp + yT + px.xR[j], (lds_acc_t)(gv[j] * xR_P * yT_P));
            lds_add(gp + yB + px.xL[j], (lds_acc_t)(gv[j] * xL_P * yB_P));
            lds_add(gp + yB + px.xR[j], (lds_acc_t)(gv[j] * xR_P * yB_P));
          }
        }
      }
    }
  }
  if (NEED_FLOW) {
    atomic_add(cx.gflow + (int64_t)(cx.b * 2 + 0) * cx.HW + px.p, (T)gx_acc);
    atomic_add(cx.gflow + (int64_t)(cx.b * 2 + 1) * cx.HW + px.p, (T)gy_acc);
  }
}

// ---- two vertically adjacent pixels whose patches are one row apart -------------------------------
// Pixel B = (yf+1, xf) usually samples exactly one source row below pixel A = (yf, xf) in the same
// columns.  Their (K+1)x(K+1) patches then overlap in K of K+1 rows, so the pair is scattered as ONE
// (K+2)x(K+1) patch: (K+2)(K+1) LDS atomics and (K+3)(K+1) LDS reads for two pixels instead of
// 2(K+1)^2 and 2(K+2)(K+1) -- the LDS pipe is what bounds this kernel (SQ_LDS_IDX_ACTIVE ~75 % of
// the duration, profiles/r1_sq_counters_bench_step.txt).  Lanes stay consecutive in x.
template <typename T, int K, bool NEED_SRC, bool NEED_FLOW, int MODE>
__device__ __forceinline__ void be_bwd_pair(const BwdCtx<T> &cx, const PixelTaps<typename Num<T>::acc, K> &pa,
                                            const PixelTaps<typename Num<T>::acc, K> &pb) {
  using A = typename Num<T>::acc;
  const GoutRow<T, K, MODE> ga = make_gout_row<T, K, MODE>(cx, pa.p, pa.yf, pa.xf);
  const GoutRow<T, K, MODE> gb = make_gout_row<T, K, MODE>(cx, pb.p, pb.yf, pb.xf);
  int col[K + 1];
#pragma unroll
  for (int q = 0; q <= K; ++q) col[q] = clampi(pa.x0 + q, 0, cx.Ws - 1);
  A gxa = 0, gya = 0, gxb = 0, gyb = 0;
  for (int c = 0; c < cx.gc; ++c) {
    lds_acc_t *gp = cx.gplanes0 + (size_t)c * cx.win_sz;
    const A *spl = cx.splanes0 + (size_t)c * cx.win_sz;
    A carryA[K + 1], carryB[K + 1], vcur[K + 1];
    int offr = clampi(pa.y0, 0, cx.Hs - 1) * cx.Ws;
#pragma unroll
    for (int q = 0; q <= K; ++q) {
      carryA[q] = carryB[q] = 0;
      vcur[q] = NEED_FLOW ? spl[offr + col[q]] : (A)0;
    }
#pragma unroll 1
    for (int r = 0; r <= K + 1; ++r) {  // patch row r of the pair = source row y0_A + r
      const int offn = clampi(pa.y0 + r + 1, 0, cx.Hs - 1) * cx.Ws;
      A acc[K + 1], nA[K + 1], nB[K + 1], vnext[K + 1];
#pragma unroll
      for (int q = 0; q <= K; ++q) {
        acc[q] = carryA[q] + carryB[q];
        nA[q] = nB[q] = 0;
        vnext[q] = (NEED_FLOW && r <= K) ? spl[offn + col[q]] : (A)0;
      }
      if (r < K) {  // pixel A, tap row r: top weights land on this patch row, bottom weights on the next
        A fdy;
        const A yB_P = pa.yfrac(r, fdy), yT_P = 1 - yB_P;
        A gv[K];
        ga.load(c, r, gv);
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const A xL_P = 1 - pa.ax[j], xR_P = pa.ax[j];
          if (NEED_SRC) {
            acc[j] += gv[j] * xL_P * yT_P;
            acc[j + 1] += gv[j] * xR_P * yT_P;
            nA[j] += gv[j] * xL_P * yB_P;
            nA[j + 1] += gv[j] * xR_P * yB_P;
          }
          if (NEED_FLOW) {
            gya += gv[j] * (-xL_P * vcur[j] - xR_P * vcur[j + 1] + xL_P * vnext[j] + xR_P * vnext[j + 1]);
            gxa += gv[j] * (-yT_P * vcur[j] - yB_P * vnext[j] + yT_P * vcur[j + 1] + yB_P * vnext[j + 1]);
          }
        }
      }
      if (r >= 1 && r <= K) {  // pixel B, tap row r-1 (its patch starts one source row lower)
        A fdy;
        const A yB_P = pb.yfrac(r - 1, fdy), yT_P = 1 - yB_P;
        A gv[K];
        gb.load(c, r - 1, gv);
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const A xL_P = 1 - pb.ax[j], xR_P = pb.ax[j];
          if (NEED_SRC) {
            acc[j] += gv[j] * xL_P * yT_P;
            acc[j + 1] += gv[j] * xR_P * yT_P;
            nB[j] += gv[j] * xL_P * yB_P;
            nB[j + 1] += gv[j] * xR_P * yB_P;
          }
          if (NEED_FLOW) {
            gyb += gv[j] * (-xL_P * vcur[j] - xR_P * vcur[j + 1] + xL_P * vnext[j] + xR_P * vnext[j + 1]);
            gxb += gv[j] * (-yT_P * vcur[j] - yB_P * vnext[j] + yT_P * vcur[j + 1] + yB_P * vnext[j + 1]);
          }
        }
      }
      if (NEED_SRC) {
#pragma unroll
        for (int q = 0; q <= K; ++q)
          if (acc[q] != 0) lds_add(gp + offr + col[q], (lds_acc_t)acc[q]);
      }
#pragma unroll
      for (int q = 0; q <= K; ++q) {
        carryA[q] = nA[q];
        carryB[q] = nB[q];
        vcur[q] = vnext[q];
      }
      offr = offn;
    }
  }
  if (NEED_FLOW) {
    atomic_add(cx.gflow + (int64_t)(cx.b * 2 + 0) * cx.HW + pa.p, (T)gxa);
    atomic_add(cx.gflow + (int64_t)(cx.b * 2 + 1) * cx.HW + pa.p, (T)gya);
    atomic_add(cx.gflow + (int64_t)(cx.b * 2 + 0) * cx.HW + pb.p, (T)gxb);
    atomic_add(cx.gflow + (int64_t)(cx.b * 2 + 1) * cx.HW + pb.p, (T)gyb);
  }
}

// `per` must be a multiple of Wf (whole flow rows per workgroup) so that rows can be paired.
template <typename T, int K, bool NEED_SRC, bool NEED_FLOW, int MODE, bool WIN>
__global__ __launch_bounds__(256, 3) void be_bwd_lds_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, const T *__restrict__ gout,
    const T *__restrict__ attn, const T *__restrict__ gout2, T *__restrict__ gsrc, T *__restrict__ gflow,
    int C, int Hs, int Ws, int Hf, int Wf, int G, int ngroups, int split, int per, int margin, int64_t u_cs,
    int64_t u_bs, int pair_rows) {
  using A = typename Num<T>::acc;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  int bid = blockIdx.x;
  const int sp = bid % split;
  bid /= split;
  const int g = bid % ngroups;
  const int b = bid / ngroups;
  const int c0 = g * G;
  const int gc = min(G, C - c0);
  const int plane_sz = Hs * Ws;
  const int HW = Hf * Wf;
  const int p_begin = sp * per;
  const int p_end = min(HW, p_begin + per);
  if (p_begin >= p_end) return;
  // source rows resident in LDS: the whole plane (margin < 0) or the window this band of flow rows
  // reaches with |flow_y| <= margin
  const Window win = make_window(p_begin / Wf, (p_end - 1) / Wf, K / 2, K - K / 2, WIN ? margin : -1, Hs);
  const int win_sz = win.rows * Ws;
  lds_acc_t *gplanes = reinterpret_cast<lds_acc_t *>(gfla_smem);                                          // [G][window] double
  A *splanes = reinterpret_cast<A *>(gfla_smem + (NEED_SRC ? sizeof(lds_acc_t) * (size_t)G * win_sz : 0));  // [G][window]
  const T *src0 = src + ((int64_t)b * C + c0) * plane_sz;
  T *gsrc0 = NEED_SRC ? gsrc + ((int64_t)b * C + c0) * plane_sz : nullptr;
  if (NEED_SRC) zero_planes<lds_acc_t>(gplanes, gc * win_sz);
  if (NEED_FLOW)
    for (int c = 0; c < gc; ++c)
      stage_planes<T, A>(src0 + (int64_t)c * plane_sz + win.lo * Ws, splanes + (size_t)c * win_sz, win_sz);
  __syncthreads();

  BwdCtx<T> cx;
  cx.flow = flow; cx.gout = gout; cx.attn = attn; cx.gout2 = gout2;
  cx.src0 = src0; cx.gsrc0 = gsrc0; cx.gflow = gflow;
  cx.gplanes0 = gplanes - win.lo * Ws;   // shifted bases: plane-relative offsets index the window
  cx.splanes0 = splanes - win.lo * Ws;
  cx.b = b; cx.C = C; cx.c0 = c0; cx.gc = gc; cx.Hs = Hs; cx.Ws = Ws; cx.Hf = Hf; cx.Wf = Wf; cx.HW = HW;
  cx.plane_sz = plane_sz; cx.win_sz = win_sz; cx.win_lo = win.lo; cx.win_rows = win.rows;
  cx.u_cs = u_cs; cx.u_bs = u_bs;

  if (pair_rows) {
    // slots = (pair of flow rows, xf); lanes consecutive in xf
    const int row_a = p_begin / Wf, row_b = (p_end + Wf - 1) / Wf;
    const int nslots = ((row_b - row_a + 1) / 2) * Wf;
    for (int s = threadIdx.x; s < nslots; s += blockDim.x) {
      const int pr = s / Wf, xf = s - pr * Wf;
      const int yfa = row_a + 2 * pr;
      PixelTaps<A, K> pa, pb;
      pa.template init<T, WIN>(cx, yfa * Wf + xf);
      const bool has_b = yfa + 1 < row_b;
      if (has_b) pb.template init<T, WIN>(cx, (yfa + 1) * Wf + xf);
      if (has_b && pa.dense && pb.dense && pa.inside && pb.inside && pb.x0 == pa.x0 && pb.y0 == pa.y0 + 1) {
        be_bwd_pair<T, K, NEED_SRC, NEED_FLOW, MODE>(cx, pa, pb);
      } else {
        for (int which = 0; which < (has_b ? 2 : 1); ++which)  // one inlined copy of the single-pixel body
          be_bwd_single<T, K, NEED_SRC, NEED_FLOW, MODE>(cx, which ? pb : pa);
      }
    }
  } else {
    for (int p = p_begin + threadIdx.x; p < p_end; p += blockDim.x) {
      PixelTaps<A, K> px;
      px.template init<T, WIN>(cx, p);
      be_bwd_single<T, K, NEED_SRC, NEED_FLOW, MODE>(cx, px);
    }
  }
  if (NEED_SRC) {
    __syncthreads();
    // exclusive owner of the planes only when the whole plane is resident and not shared by bands
    for (int c = 0; c < gc; ++c)
      flush_planes<T>(gsrc0 + (int64_t)c * plane_sz + win.lo * Ws, gplanes + (size_t)c * win_sz, win_sz,
                      split == 1 && margin < 0);
  }
}

// Launch helper.  mode = kGoutTensor / kGoutAttn / kGoutUnfold (see GoutRow).  *done = false when the
// planes do not fit in LDS.
template <typename T, int K>
static int launch_be_bwd_lds(int mode, const T *src, const T *flow, const T *gout, const T *attn, T *gsrc,
                             T *gflow, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf,
                             hipStream_t stream, bool *done, int64_t u_cs = 0, int64_t u_bs = 0,
                             const T *gout2 = nullptr) {
  using A = typename Num<T>::acc;
  *done = false;
  const int bytes = (gsrc ? (int)sizeof(lds_acc_t) : 0) + (gflow ? (int)sizeof(A) : 0);
  // the factored / unfold forms are only produced for planes that fit; the tensor form may window
  PlaneGeo g = mode == kGoutTensor ? lds_geometry(Hs, Ws, bytes, B, C, Hf, Wf, K + 3)
                                   : plane_geometry(Hs * Ws, bytes, B, C, Hf * Wf, true);
  if (g.G == 0) return GFLA_OK;
  // whole flow rows per workgroup, so vertically adjacent pixels can be scattered as one patch
  const int pair_rows = tuning(2) == 2 ? 0 : 1;
  if (g.per % Wf != 0) {
    g.per = (int)(ceil_div(g.per, Wf) * Wf);
    g.split = (int)ceil_div(Hf * Wf, g.per);
  }
  const int64_t blocks = B * g.ngroups * g.split;
  if (blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)blocks), blk(256);  // 4 waves: the pair path needs ~170 VGPRs, keep >= 3 workgroups per CU
#define GFLA_BE_BWD_LAUNCH(S, F, AT)                                                                 \
  if (AT == kGoutTensor && g.margin >= 0)                                                            \
    be_bwd_lds_kernel<T, K, S, F, kGoutTensor, true><<<grid, blk, g.lds_bytes, stream>>>(            \
        src, flow, gout, attn, gout2, gsrc, gflow, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, g.G, g.ngroups,     \
        g.split, g.per, g.margin, u_cs, u_bs, pair_rows);                                            \
  else                                                                                               \
    be_bwd_lds_kernel<T, K, S, F, AT, false><<<grid, blk, g.lds_bytes, stream>>>(                    \
      src, flow, gout, attn, gout2, gsrc, gflow, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, g.G, g.ngroups, g.split,   \
      g.per, g.margin, u_cs, u_bs, pair_rows)
  if (mode == kGoutUnfoldAttn) {
    if (gsrc && gflow) GFLA_BE_BWD_LAUNCH(true, true, kGoutUnfoldAttn);
    else if (gsrc) GFLA_BE_BWD_LAUNCH(true, false, kGoutUnfoldAttn);
    else GFLA_BE_BWD_LAUNCH(false, true, kGoutUnfoldAttn);
  } else if (mode == kGoutAttn) {
    if (gsrc && gflow) GFLA_BE_BWD_LAUNCH(true, true, kGoutAttn);
    else if (gsrc) GFLA_BE_BWD_LAUNCH(true, false, kGoutAttn);
    else GFLA_BE_BWD_LAUNCH(false, true, kGoutAttn);
  } else if (mode == kGoutUnfold) {
    if (gsrc && gflow) GFLA_BE_BWD_LAUNCH(true, true, kGoutUnfold);
    else if (gsrc) GFLA_BE_BWD_LAUNCH(true, false, kGoutUnfold);
    else GFLA_BE_BWD_LAUNCH(false, true, kGoutUnfold);
  } else {
    if (gsrc && gflow) GFLA_BE_BWD_LAUNCH(true, true, kGoutTensor);
    else if (gsrc) GFLA_BE_BWD_LAUNCH(true, false, kGoutTensor);
    else GFLA_BE_BWD_LAUNCH(false, true, kGoutTensor);
  }
#undef GFLA_BE_BWD_LAUNCH
  *done = true;
  return launch_status();
}

}  // namespace gfla
