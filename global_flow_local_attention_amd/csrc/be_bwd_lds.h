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

template <typename T, int K, bool NEED_SRC, bool NEED_FLOW, int MODE, bool WIN>
__global__ __launch_bounds__(kLdsThreads) void be_bwd_lds_kernel(
    const T *__restrict__ src, const T *__restrict__ flow, const T *__restrict__ gout,
    const T *__restrict__ attn, const T *__restrict__ gout2, T *__restrict__ gsrc,
    typename Num<T>::acc *__restrict__ gflow,  // a reduction over channel groups: float even for bf16 storage
    int C, int Hs, int Ws, int Hf, int Wf, int G, int ngroups, int split, int per, int margin, int64_t u_cs,
    int64_t u_bs, const unsigned *__restrict__ skip_stat, unsigned skip_limit) {
  using A = typename Num<T>::acc;
  extern __shared__ __attribute__((aligned(16))) unsigned char gfla_smem[];
  // adaptive dispatch (patch_mfma.hip): the matrix-core path took this launch when its statistic is within the limit
  if (skip_stat) {  // the statistic is a sum over 32 counters (patch_mfma.hip: kPmStatSlots)
    unsigned tot = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) tot += skip_stat[i];
    if (tot <= skip_limit) return;
  }
  int bid = blockIdx.x;
  const int sp = bid % split;
  bid /= split;
  const int g = bid % ngroups;
  const int b = bid / ngroups;
  const int c0 = g * G;
  const int gc = min(G, C - c0);
  const int plane_sz = Hs * Ws;
  const int Wo = K * Wf;
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
  // shifted bases: plane-relative offsets index the window
  lds_acc_t *gplanes0 = gplanes - win.lo * Ws;
  const A *splanes0 = splanes - win.lo * Ws;

  for (int p = p_begin + threadIdx.x; p < p_end; p += blockDim.x) {
    const int yf = p / Wf, xf = p - yf * Wf;
    const A fx0 = Num<T>::ld(flow + (int64_t)(b * 2 + 0) * HW + p);
    const A fy0 = Num<T>::ld(flow + (int64_t)(b * 2 + 1) * HW + p);
    // column taps (registers); row taps are recomputed per row so the row loop stays rolled
    int xL[K], xR[K];
    A ax[K];
    int x0 = 0, y0 = 0;
    bool dense = true;
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
      xL[t] = clampi((int)fdx, 0, Ws - 1);
      xR[t] = clampi((int)(fdx + 1), 0, Ws - 1);
      ax[t] = dx - fdx;
    }
    GoutRow<T, K, MODE> gr;
    gr.base2 = nullptr;
    if constexpr (MODE == kGoutUnfoldAttn) {
      gr.base = gout + (int64_t)b * u_bs + (int64_t)c0 * K * K * u_cs + p;
      gr.attn_p = attn + (int64_t)b * K * K * HW + p;
      gr.base2 = gout2 + ((int64_t)b * C + c0) * HW + p;
      gr.cstride = u_cs;
      gr.pitch = 0;
    } else if constexpr (MODE == kGoutAttn) {
      gr.base = gout + ((int64_t)b * C + c0) * HW + p;
      gr.attn_p = attn + (int64_t)b * K * K * HW + p;
      gr.cstride = HW;
      gr.pitch = 0;
    } else if constexpr (MODE == kGoutUnfold) {  // element (b, ch, p) at b*u_bs + ch*u_cs + p
      gr.base = gout + (int64_t)b * u_bs + (int64_t)c0 * K * K * u_cs + p;
      gr.attn_p = nullptr;
      gr.cstride = u_cs;
      gr.pitch = 0;
    } else {
      gr.base = gout + ((int64_t)b * C + c0) * ((int64_t)K * Hf * Wo) + (int64_t)(yf * K) * Wo + xf * K;
      gr.attn_p = nullptr;
      gr.cstride = (int64_t)K * Hf * Wo;
      gr.pitch = Wo;
    }
    gr.HW = HW;
    gr.inv_kk = (A)1 / (A)(K * K);

    A gx_acc = 0, gy_acc = 0;
    // every row this pixel can touch (one row of slack covers the non-dense rounding case)
    const bool inside = !WIN || (clampi(y0 - 1, 0, Hs - 1) >= win.lo && clampi(y0 + K + 1, 0, Hs - 1) < win.lo + win.rows);
    if (!inside) {
      // flow beyond the window's margin: this pixel alone takes the reference's decomposition on
      // global memory (block_extractor_kernel.cu:123-168)
      for (int c = 0; c < gc; ++c) {
        const T *pl = src0 + (int64_t)c * plane_sz;
        T *gp = NEED_SRC ? gsrc0 + (int64_t)c * plane_sz : nullptr;
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
          const A dy = (fy0 + (A)(i - K / 2)) + (A)yf;
          const A fdy = floor_t<A>(dy);
          const int yT = clampi((int)fdy, 0, Hs - 1) * Ws, yB = clampi((int)(fdy + 1), 0, Hs - 1) * Ws;
          const A yB_P = dy - fdy, yT_P = 1 - yB_P;
          A gv[K];
          gr.load(c, i, gv);
#pragma unroll
          for (int j = 0; j < K; ++j) {
            const A xL_P = 1 - ax[j], xR_P = ax[j];
            if (NEED_FLOW) {
              const A vTL = Num<T>::ld(pl + yT + xL[j]), vTR = Num<T>::ld(pl + yT + xR[j]);
              const A vBL = Num<T>::ld(pl + yB + xL[j]), vBR = Num<T>::ld(pl + yB + xR[j]);
              gy_acc += gv[j] * (-xL_P * vTL - xR_P * vTR + xL_P * vBL + xR_P * vBR);
              gx_acc += gv[j] * (-yT_P * vTL - yB_P * vBL + yT_P * vTR + yB_P * vBR);
            }
            if (NEED_SRC) {
              atomic_add(gp + yT + xL[j], (T)(gv[j] * xL_P * yT_P));
              atomic_add(gp + yT + xR[j], (T)(gv[j] * xR_P * yT_P));
              atomic_add(gp + yB + xL[j], (T)(gv[j] * xL_P * yB_P));
              atomic_add(gp + yB + xR[j], (T)(gv[j] * xR_P * yB_P));
            }
          }
        }
      }
    } else if (dense) {
      int col[K + 1];
#pragma unroll
      for (int q = 0; q <= K; ++q) col[q] = clampi(x0 + q, 0, Ws - 1);
      for (int c = 0; c < gc; ++c) {
        lds_acc_t *gp = gplanes0 + (size_t)c * win_sz;
        const A *spl = splanes0 + (size_t)c * win_sz;
        A rowA[K + 1], vA[K + 1];
        int offA = clampi(y0, 0, Hs - 1) * Ws;
#pragma unroll
        for (int q = 0; q <= K; ++q) {
          rowA[q] = 0;
          vA[q] = NEED_FLOW ? spl[offA + col[q]] : (A)0;
        }
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
          const A dy = (fy0 + (A)(i - K / 2)) + (A)yf;
          const A yB_P = dy - floor_t<A>(dy), yT_P = 1 - yB_P;
          const int offB = clampi(y0 + i + 1, 0, Hs - 1) * Ws;
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
            const A xL_P = 1 - ax[j], xR_P = ax[j];
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
      for (int c = 0; c < gc; ++c) {
        lds_acc_t *gp = gplanes0 + (size_t)c * win_sz;
        const A *spl = splanes0 + (size_t)c * win_sz;
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
          const A dy = (fy0 + (A)(i - K / 2)) + (A)yf;
          const A fdy = floor_t<A>(dy);
          const int yT = clampi((int)fdy, 0, Hs - 1) * Ws, yB = clampi((int)(fdy + 1), 0, Hs - 1) * Ws;
          const A yB_P = dy - fdy, yT_P = 1 - yB_P;
          A gv[K];
          gr.load(c, i, gv);
#pragma unroll
          for (int j = 0; j < K; ++j) {
            const A xL_P = 1 - ax[j], xR_P = ax[j];
            if (NEED_FLOW) {
              const A vTL = spl[yT + xL[j]], vTR = spl[yT + xR[j]], vBL = spl[yB + xL[j]], vBR = spl[yB + xR[j]];
              gy_acc += gv[j] * (-xL_P * vTL - xR_P * vTR + xL_P * vBL + xR_P * vBR);
              gx_acc += gv[j] * (-yT_P * vTL - yB_P * vBL + yT_P * vTR + yB_P * vBR);
            }
            if (NEED_SRC) {
              lds_add(gp + yT + xL[j], (lds_acc_t)(gv[j] * xL_P * yT_P));
              lds_add(gp + yT + xR[j], (lds_acc_t)(gv[j] * xR_P * yT_P));
              lds_add(gp + yB + xL[j], (lds_acc_t)(gv[j] * xL_P * yB_P));
              lds_add(gp + yB + xR[j], (lds_acc_t)(gv[j] * xR_P * yB_P));
            }
          }
        }
      }
    }
    if (NEED_FLOW) {
      atomic_add(gflow + (int64_t)(b * 2 + 0) * HW + p, gx_acc);
      atomic_add(gflow + (int64_t)(b * 2 + 1) * HW + p, gy_acc);
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
                             typename Num<T>::acc *gflow, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf,
                             hipStream_t stream, bool *done, int64_t u_cs = 0, int64_t u_bs = 0,
                             const T *gout2 = nullptr, const unsigned *skip_stat = nullptr, unsigned skip_limit = 0) {
  using A = typename Num<T>::acc;
  *done = false;
  const int bytes = (gsrc ? (int)sizeof(lds_acc_t) : 0) + (gflow ? (int)sizeof(A) : 0);
  // the factored / unfold forms are only produced for planes that fit; the tensor form may window
  constexpr bool kBf16 = sizeof(T) == 2;  // bf16 storage: whole planes, one owner per plane (no atomics on the planes)
  PlaneGeo g = (mode == kGoutTensor && !kBf16) ? lds_geometry(Hs, Ws, bytes, B, C, Hf, Wf, K + 3, 1)
                                               : plane_geometry(Hs * Ws, bytes, B, C, Hf * Wf, !kBf16, 1);
  if (g.G == 0) return GFLA_OK;
  const int64_t blocks = B * g.ngroups * g.split;
  if (blocks > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)blocks), blk(kLdsThreads);
#define GFLA_BE_BWD_LAUNCH(S, F, AT)                                                                 \
  if (AT == kGoutTensor && g.margin >= 0)                                                            \
    launch_lds(be_bwd_lds_kernel<T, K, S, F, kGoutTensor, true>, grid, blk, g.lds_bytes, stream,             \
        src, flow, gout, attn, gout2, gsrc, gflow, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, g.G, g.ngroups,     \
        g.split, g.per, g.margin, u_cs, u_bs, skip_stat, skip_limit);                                \
  else                                                                                               \
    launch_lds(be_bwd_lds_kernel<T, K, S, F, AT, false>, grid, blk, g.lds_bytes, stream,                     \
      src, flow, gout, attn, gout2, gsrc, gflow, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, g.G, g.ngroups, g.split,   \
      g.per, g.margin, u_cs, u_bs, skip_stat, skip_limit)
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
