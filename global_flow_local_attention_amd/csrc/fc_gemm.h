// Shared declarations of the MFMA kernels behind ExtractorAttn's first FC layer (fc_gemm.hip, fc_sample.hip,
// fc_block.hip).  See fc_gemm.hip for the formulation.
#pragma once

#include "gfla_common.h"

namespace gfla {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int kFcChunk = 16;   // channels per K chunk (one 16-channel record per pixel)
constexpr int kFcTN = 128;     // output channels per workgroup
constexpr int kFcHidden = 128; // hidden_nc of ExtractorAttn (base_function.py:794)

// Arithmetic of the contraction ("mode").  All three accumulate in f32 inside the MFMA.
//   0: v_mfma_f32_32x32x2_f32 on f32 operands -- a k-ordered fmaf chain, bit-for-bit f32 (157 TF/s peak).
//   3: operands split into THREE f16 terms (x*s = x1 + x2 + x3 exactly, s a power of two chosen from the
//      tensor's max |x| so that nothing under/overflows), 6 cross products on v_mfma_f32_32x32x16_f16:
//      every product term above 2^-32 of the product is kept (an f32 multiply rounds at 2^-24), so the
//      result is f32-grade or better at 2.67x the f32 MFMA rate.
//   2: two f16 terms, 3 cross products: per-product error 2^-21 -- far below the f32 accumulation
//      error of a K = 2304..3200 dot product -- at 5.3x the f32 MFMA rate.
//   1: ONE f16 term per operand (round to 11 significant bits after the power-of-two scaling): exact for operands
//      that are bf16 values (8 significant bits) -- the arithmetic of the bf16-feature path -- at the full f16 MFMA
//      rate, 16x the f32 one.
template <int MODE>
struct Fc;
template <>
struct Fc<0> {
  static constexpr int NS = 1, ESZ = 4, REC = 64, PIECES = 4, PITCH = 80, KB = 2;
};
template <>
struct Fc<1> {
  static constexpr int NS = 1, ESZ = 2, REC = 32, PIECES = 2, PITCH = 48, KB = 1;
};
template <>
struct Fc<2> {
  static constexpr int NS = 2, ESZ = 2, REC = 32, PIECES = 2, PITCH = 48, KB = 1;
};
template <>
struct Fc<3> {
  static constexpr int NS = 3, ESZ = 2, REC = 32, PIECES = 2, PITCH = 48, KB = 1;
};

//   4: float32 throughout, Winograd domain (fc_wino.hip): F(2x2,5x5) / F(4x4,3x3) on the same 6 points, the 36 point-wise
//      products as f32 MFMA GEMMs over the channels -- 2.78x / 4x fewer multiplies; operands, packing and every other
//      kernel are mode 0's.
//   5: float32 tensors and transforms, Winograd domain as in 4, but the 36 point-wise GEMMs run on the f16 matrix cores: every
//      transformed input / weight value is split into two f16 terms after a power-of-two scaling (hi + lo = the value to
//      2^-24, a float32's own rounding) and all four cross products are accumulated in f32 (fc_wino16.hip): 4x less
//      matrix-core time than mode 4, the same measured error.  The weight gradient is mode 4's.
inline bool fc_is_wino(int mode) { return mode == 4 || mode == 5; }
inline int fc_base_mode(int mode) { return fc_is_wino(mode) ? 0 : mode; }  // operand format / non-convolution kernels
inline int fc_nsplit(int mode) { return mode == 0 ? 1 : mode; }
inline int fc_esz(int mode) { return mode == 0 ? 4 : 2; }
inline bool fc_mode_ok(int mode) { return mode >= 0 && mode <= 5; }

// An activation operand: 16-channel records, pixel-linear inside a sample (row pitch = the padded width, so a
// k x k tap is a constant pixel offset), chunk-major.  Strides in bytes.
struct PackedDesc {
  const unsigned char *base;
  int64_t split_stride;  // between the f16 terms (mode 2/3)
  int64_t batch_stride;  // between samples
  int64_t chunk_stride;  // between 16-channel chunks of one sample
  int pix_stride;        // between consecutive pixels of one chunk
};

// scale = 2^(14 - e) for a tensor whose max |x| has binary exponent e: scaled values stay below 2^15 (f16 max
// 65504) and the smallest f16 subnormal is 2^-39 of the largest element.  amax_bits = float bits of max |x|.
__host__ __device__ __forceinline__ int fc_scale_exp(uint32_t amax_bits) {
  const int eb = (int)((amax_bits >> 23) & 0xffu);
  if (eb == 0 || eb == 255) return 127;  // all-zero (or non-finite) tensor: scale 1
  int se = 268 - eb;
  return se < 2 ? 2 : (se > 252 ? 252 : se);
}
__device__ __forceinline__ float fc_scale(const uint32_t *amax) {
  return amax ? __uint_as_float((uint32_t)fc_scale_exp(*amax) << 23) : 1.f;
}
__device__ __forceinline__ float fc_inv_scale(const uint32_t *amax) {
  return amax ? __uint_as_float((uint32_t)(254 - fc_scale_exp(*amax)) << 23) : 1.f;
}

// (hi0, hi1) and (lo0, lo1) f16 words of two float values, hi = RN16(v), lo = RN16(v - hi), in four instructions:
// v_cvt_pk_f16_f32 (both hi), two v_fma_mix_f32 (v * 1 - hi with hi read as f16 from either half: the exact remainders),
// v_cvt_pk_f16_f32 (both lo).  The values are made opaque first: with the arithmetic that produced them in sight hipcc fuses
// it into the convert (v_fma_mixlo_f16 of the unrounded result) and hi is no longer the half the remainder was taken from.
typedef _Float16 fc_f16x2 __attribute__((ext_vector_type(2)));
typedef float fc_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fc_split_pair(float v0, float v1, uint32_t &hi, uint32_t &lo) {
  asm("" : "+v"(v0));
  asm("" : "+v"(v1));
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(fc_f32x2{v0, v1}, fc_f16x2));
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(v0), "v"(hi));
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(v1), "v"(hi));
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(fc_f32x2{r0, r1}, fc_f16x2));
}

// ---- geometry of one half (source or target) of the layer, shared by host code ------------------
struct FcHalf {
  int Hp, Wp;      // replicate-padded input (Wp = the row pitch of the linearised input and gradient maps of this half)
  int Ho, Wo;      // convolution output domain; the convolved map is stored compactly, row pitch Wo
  int pad_t, pad_l, pad_b, pad_r;
  int M;           // Ho * Wp: the output positions in the input's linearisation (weight-gradient reduction range)
  int Mv;          // Ho * Wo valid outputs per sample = rows of the convolved map
  int Md;          // Hp * Wp = outputs per sample of the data-gradient convolution
  int lead;        // (k-1)*(Wp+1): zero pixels ahead of the gradient map ("Z layout")
  int64_t Sx;      // pixels per sample of the packed input (with read slack)
  int64_t Sz;      // pixels per sample of the Z-layout gradient map
  int64_t Mg;      // rows per sample allocated for the f32 convolved map
  int64_t Mdg;     // rows per sample allocated for the f32 data-gradient output
};

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

inline FcHalf fc_half(int H, int W, int k, bool source) {
  FcHalf g;
  const int lo = k / 2, hi = k - 1 - lo;
  if (source) {  // sampled at p + flow: the convolved map is needed on [-hi, H-1+lo] x [-hi, W-1+lo]
    g.pad_t = g.pad_l = g.pad_b = g.pad_r = k - 1;
  } else {       // zero flow: plain replicate-padded unfold
    g.pad_t = g.pad_l = lo;
    g.pad_b = g.pad_r = hi;
  }
  g.Hp = H + g.pad_t + g.pad_b;
  g.Wp = W + g.pad_l + g.pad_r;
  g.Ho = g.Hp - k + 1;
  g.Wo = g.Wp - k + 1;
  g.M = g.Ho * g.Wp;
  g.Mv = g.Ho * g.Wo;
  g.Md = g.Hp * g.Wp;
  g.lead = (k - 1) * (g.Wp + 1);
  g.Mg = round_up(g.Mv, 16);
  g.Mdg = round_up(g.Md, 16);
  // a row tile of the convolution reads [pix(m0), pix(m0) + span + halo) with pix(Mv-1) + halo + 1 = Hp*Wp and
  // span <= 256 + (255 / Wo + 1) * (k-1) for the largest tile (fc_gemm.hip: fc_conv_tile_pixels)
  g.Sx = round_up((int64_t)g.Md + 256 + (255 / g.Wo + 1) * (k - 1) + 16, 16);
  const int64_t need_w = g.lead + round_up(g.M, 64) + 64;  // weight-gradient kernel: K range in 64-pixel steps
  const int64_t need_d = (int64_t)g.Md + 256 + g.lead;     // data-gradient convolution reads Z[m + tap], 256-row tiles
  g.Sz = round_up((need_w > need_d ? need_w : need_d) + 16, 16);
  return g;
}

// kernels / launchers defined in fc_gemm.hip
int fc_maxabs(const float *x, int64_t n, uint32_t *slot, hipStream_t stream);
int fc_maxabs_multi(const float *x0, int64_t n0, uint32_t *slot0, const float *x1, int64_t n1, uint32_t *slot1,
                    const float *x2, int64_t n2, uint32_t *slot2, hipStream_t stream);
int fc_pack_act(const float *src, const uint32_t *amax, void *out, int64_t B, int C, int H, int W, const FcHalf &g,
                int mode, hipStream_t stream);
int fc_pack_act2(const float *src_s, const uint32_t *amax_s, void *out_s, const FcHalf &gs, const float *src_t,
                 const uint32_t *amax_t, void *out_t, const FcHalf &gt, int64_t B, int C, int H, int W, int mode,
                 hipStream_t stream);
int fc_pack_z(const float *z, const uint32_t *amax, void *out, int64_t B, int64_t S, int Cz, int mode,
              hipStream_t stream);
int fc_pack_z2(const float *z_s, const uint32_t *amax_s, void *out_s, int64_t S_s, const float *z_t, const uint32_t *amax_t,
               void *out_t, int64_t S_t, int64_t B, int Cz, int mode, hipStream_t stream);
int fc_unpack_act(const void *x16, const uint32_t *amax, float *x32, int64_t B, int nch, int64_t S, hipStream_t stream);
int fc_pack_weights(const float *w0, const uint32_t *amax, void *wf_t, void *wf_s, void *wd_t, void *wd_s, int C,
                    int k, int mode, hipStream_t stream);
int fc_conv(const PackedDesc &X, const void *wk, int64_t w_split_stride, float *out, int64_t out_bs, int ldo,
            int n_valid, int64_t B, int nch, int M, int Wv, int Wp, int k, int mode, const uint32_t *amax_x,
            const uint32_t *amax_w, hipStream_t stream);
// mode 2's arithmetic (two f16 terms per operand, three cross products) on a FLOAT32 input map (packed records or the
// (B, S, C) gradient map in place): the split happens when a chunk's pixels are written to LDS; weights = mode 2's pack
int fc_conv_f32src(const PackedDesc &X32, const void *wk, int64_t w_split_stride, float *out, int64_t out_bs, int ldo,
                   int n_valid, int64_t B, int nch, int M, int Wv, int Wp, int k, const uint32_t *amax_x,
                   const uint32_t *amax_w, hipStream_t stream);
bool fc_conv_fits(int Wv, int Wp, int k, int mode);
int fc_wgrad(const PackedDesc &X, const PackedDesc &Y, int64_t y_lead, float *dwacc, int cpad, int64_t B, int Mk,
             int Wp, int k, int mode, hipStream_t stream);
int fc_wgrad_splits(int64_t B, int Mk, int cpad);
int fc_wgrad_f32(const PackedDesc &X, const PackedDesc &Y, int64_t y_lead, float *part, int cpad, int64_t B, int Mk,
                 int Wp, int k, hipStream_t stream);
int fc_wgrad_reduce(const float *part, int nsplit, float *grad_w0, int C, int c_off, int cpad, int k,
                    hipStream_t stream);
int fc_wgrad_reduce2(const float *part_s, int nsplit_s, const float *part_t, int nsplit_t, float *grad_w0, int C, int cpad,
                     int k, hipStream_t stream);
int fc_unpack_wgrad(const float *dw_t, const float *dw_s, const uint32_t *amax_xt, const uint32_t *amax_xs,
                    const uint32_t *amax_zt, const uint32_t *amax_zs, float *grad_w0, int C, int cpad, int k,
                    hipStream_t stream);
PackedDesc fc_desc_packed(const void *base, int64_t B, int nch, int64_t S, int mode);
PackedDesc fc_desc_nhwc(const float *base, int64_t S, int Cz);
int64_t fc_packed_bytes(int64_t B, int nch, int64_t S, int mode);
int64_t fc_wpack_bytes(int ntiles, int nch, int k, int mode);
int fc_tr_probe(const short *image, int n_halves, const int *offsets, short *out, hipStream_t stream);

// fc_wino.hip (arithmetic mode 4)
int64_t fc_wino_wpack_bytes(int n_in, int n_out);
int fc_wino_pack_weights(const float *w0, float *u_ft, float *u_fs, float *u_dt, float *u_ds, int C, int k,
                         hipStream_t stream);
bool fc_wino_fits(int M, int Wv, int Wp, int k);
struct WnConvJob {   // one convolution of fc_wino_conv_jobs: the arguments of fc_wino_conv that may differ between the jobs
  PackedDesc X;
  const float *U;
  float *out;
  int64_t out_bs;
  int ldo, n_valid, M, Wv, Wp;
  int64_t S;
};
int fc_wino_conv_jobs(const WnConvJob *jobs, int njobs, int64_t B, int nch, int k, hipStream_t stream);
int fc_wino_conv(const PackedDesc &X, const float *U, float *out, int64_t out_bs, int ldo, int n_valid, int64_t B, int nch,
                 int M, int Wv, int Wp, int64_t S, int k, hipStream_t stream);
int fc_wino_wgrad_splits(int64_t B, int Ho, int Wo, int cpad, int k);
struct WwJob {   // one Winograd-domain weight gradient (fc_wino_wgrad's arguments)
  PackedDesc X;
  const float *Z;
  float *part;
  int64_t z_bs, z_lead, SX;
  int Ho, Wo, Wp;
};
int fc_wino_wgrad_jobs(const WwJob *jobs, int njobs, int cpad, int64_t B, int k, hipStream_t stream);
int fc_wino16_wgrad_jobs(const WwJob *jobs, int njobs, int cpad, int64_t B, int k, const uint32_t *const *amax_x,
                         const uint32_t *const *amax_z, hipStream_t stream);
int fc_wino_wgrad(const PackedDesc &X, const float *Z, int64_t z_bs, int64_t z_lead, float *part, int cpad, int64_t B, int Ho,
                  int Wo, int Wp, int64_t SX, int k, hipStream_t stream);
int fc_wino_wgrad_reduce(float *part, int nsplit, float *grad_w0, int C, int c_off, int cpad, int k, hipStream_t stream);
int fc_wino_wgrad_reduce2(float *part_s, int nsplit_s, float *part_t, int nsplit_t, float *grad_w0, int C, int cpad, int k,
                          hipStream_t stream);

// fc_wino16.hip (arithmetic mode 5)
struct Wn16ConvJob {   // WnConvJob with the two-term f16 weights of fc_wino16_pack_weights and the max |x| slot of the input
  PackedDesc X;
  const uint32_t *U;
  const uint32_t *amax_x;
  float *out;
  int64_t out_bs;
  int ldo, n_valid, M, Wv, Wp;
  int64_t S;
};
int fc_wino16_pack_weights(const float *w0, const uint32_t *amax_w, float *u_ft, float *u_fs, float *u_dt, float *u_ds, int C,
                           int k, hipStream_t stream);
bool fc_wino16_fits(int M, int Wv, int Wp, int k);
int fc_wino16_conv_jobs(const Wn16ConvJob *jobs, int njobs, int64_t B, int nch, int k, const uint32_t *amax_w,
                        hipStream_t stream);

// fc_sample.hip
int fc_sample_tail_fwd(const float *gs, const float *gt, const float *flow, const float *b0, const float *w1,
                       const float *b1, float *hid, float *logits, int64_t B, int H, int W, int k, int64_t gs_bs,
                       int64_t gt_bs, int wps, int wpt, float slope, hipStream_t stream);
int fc_sample_tail_bwd(const float *gs, const float *flow, const float *hid, const float *w1, const float *g_logits,
                       float *dzs, float *dzt, float *gflow, float *b0_partials, int64_t B, int H, int W, int k,
                       int64_t gs_bs, int wps, int wpz, int wpt, int64_t zs_bs, int64_t zt_bs, int lead_s, int lead_t,
                       float slope, int acc_flow, hipStream_t stream, uint32_t *amax_d = nullptr);
// d Gs by owner-computes (no global atomics, no memset, max |d Gs| as a by-product): fc_sample.hip
int fc_scatter_own_rows(int64_t B, int Ho, int Wo);
int fc_sample_scatter_own(const float *flow, const float *dzt, float *dzs, const uint32_t *amax_d, uint32_t *amax_out,
                          int64_t B, int H, int W, int k, int Ho, int Wo, int wpz, int wpt, int64_t zs_bs, int64_t zt_bs,
                          int lead_s, int lead_t, int64_t Sz, hipStream_t stream);
int fc_dw1(const float *hid, const float *g_logits, float *partials, int64_t B, int HW, int KK, int tiles_per_sample,
           float slope, hipStream_t stream);
int fc_fold(const float *dxpad, float *grad, int64_t B, int C, int H, int W, const FcHalf &g, int64_t dx_bs,
            int accumulate, hipStream_t stream);
int fc_fold2(const float *dx_s, float *grad_s, const FcHalf &gs, int64_t dxs_bs, int acc_s, const float *dx_t, float *grad_t,
             const FcHalf &gt, int64_t dxt_bs, int acc_t, int64_t B, int C, int H, int W, hipStream_t stream);
constexpr int kFcRedTmpFloats = 32 * (32 * kFcHidden + 32 + kFcHidden);  // first-pass scratch: 32 splits x (d W1 | d b1 row + d b0 row)
int fc_reduce_rows(const float *partials, float *out, int64_t rows, int cols, float scale, float *tmp,
                   hipStream_t stream);
int fc_reduce_bias_w1(const float *b0_partials, int64_t rows_b0, float *g_b0, const float *dw1_partials, int64_t rows_w1,
                      float *g_w1, float *g_b1, int KK, float *tmp, hipStream_t stream);

}  // namespace gfla
