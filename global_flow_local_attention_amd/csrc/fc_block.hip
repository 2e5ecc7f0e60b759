// C ABI of the MFMA path of ExtractorAttn's fully_connect_layer (fc_gemm.hip, fc_sample.hip): one call per
// direction, all kernels enqueued on the caller's stream, scratch memory supplied by the caller.
//
// Reference: model/networks/base_function.py:799-807 -- logits = Conv2d(128,k*k,1)(nonlinearity(
// Conv2d(2C,128,k,stride k)(cat(BlockExtractor(target, 0), BlockExtractor(source, flow))))).
#include "fc_gemm.h"

namespace gfla {

static int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

// amax slots (uint32 each): max |x| of the tensors that get split into f16 terms
enum { kAmaxSrc = 0, kAmaxTgt = 1, kAmaxW = 2, kAmaxZs = 3, kAmaxZt = 4, kAmaxSlots = 8 };

struct FcLayout {
  FcHalf hs, ht;
  int nch_c, cpad, nt_d, KK, dw1_tiles;
  // forward workspace, kept for backward
  int64_t amax, xs, xt, gs, hid, wd_t, wd_s, gt, wf_t, wf_s, wu_ft, wu_fs, wu_dt, wu_ds, fwd_total;
  // backward scratch: [dzs, dzt, dw_s, dw_t] are zeroed by one memset
  int64_t dzs, dzt, dw_s, dw_t, zero_bytes, zs_pk, zt_pk, dxs, dxt, b0p, dw1p, red, red_tmp, dwp, dwp2, x32, x32b, bwd_total;
  bool wgrad_f32_wino;   // mode 1, k = 5: the weight gradient runs in the float32 Winograd domain on unpacked activations
};

static FcLayout fc_layout(int64_t B, int C, int H, int W, int k, int mode_) {
  const bool wino = fc_is_wino(mode_);
  const int mode = fc_base_mode(mode_);
  FcLayout L;
  L.hs = fc_half(H, W, k, true);
  L.ht = fc_half(H, W, k, false);
  L.nch_c = (int)ceil_div(C, kFcChunk);
  L.cpad = L.nch_c * kFcChunk;
  L.nt_d = (int)ceil_div(C, kFcTN);
  L.KK = k * k;
  const int nch_h = kFcHidden / kFcChunk;
  int64_t o = 0;
  auto take = [&](int64_t bytes) {
    const int64_t at = o;
    o += align256(bytes);
    return at;
  };
  L.amax = take(kAmaxSlots * 4);
  L.xs = take(fc_packed_bytes(B, L.nch_c, L.hs.Sx, mode));
  L.xt = take(fc_packed_bytes(B, L.nch_c, L.ht.Sx, mode));
  L.gs = take(B * L.hs.Mg * kFcHidden * 4);
  L.hid = take(B * (int64_t)H * W * kFcHidden * 4);
  // mode 5 runs its data-gradient convolutions (and the k = 5 forward ones) on the direct f16x2 kernels: mode 2's weight packs
  const bool hyb = mode_ == 5;
  L.wd_t = take(wino ? (hyb ? fc_wpack_bytes(L.nt_d, nch_h, k, 2) : 0) : fc_wpack_bytes(L.nt_d, nch_h, k, mode));
  L.wd_s = take(wino ? (hyb ? fc_wpack_bytes(L.nt_d, nch_h, k, 2) : 0) : fc_wpack_bytes(L.nt_d, nch_h, k, mode));
  L.gt = take(B * L.ht.Mg * kFcHidden * 4);
  L.wf_t = take(wino ? (hyb ? fc_wpack_bytes(1, L.nch_c, k, 2) : 0) : fc_wpack_bytes(1, L.nch_c, k, mode));
  L.wf_s = take(wino ? (hyb ? fc_wpack_bytes(1, L.nch_c, k, 2) : 0) : fc_wpack_bytes(1, L.nch_c, k, mode));
  // Winograd mode: U = G w G^T of the forward (C -> 128) and data-gradient (128 -> C) convolutions of both halves
  L.wu_ft = take(wino ? fc_wino_wpack_bytes(C, kFcHidden) : 0);
  L.wu_fs = take(wino ? fc_wino_wpack_bytes(C, kFcHidden) : 0);
  L.wu_dt = take(wino ? fc_wino_wpack_bytes(kFcHidden, C) : 0);
  L.wu_ds = take(wino ? fc_wino_wpack_bytes(kFcHidden, C) : 0);
  L.fwd_total = o;

  o = 0;
  L.dzs = take(B * L.hs.Sz * kFcHidden * 4);
  L.dzt = take(B * L.ht.Sz * kFcHidden * 4);
  L.dw_s = take((int64_t)L.KK * L.cpad * kFcHidden * 4);
  L.dw_t = take((int64_t)L.KK * L.cpad * kFcHidden * 4);
  L.zero_bytes = o;
  L.zs_pk = take(mode ? fc_packed_bytes(B, nch_h, L.hs.Sz, mode) : 0);
  L.zt_pk = take(mode ? fc_packed_bytes(B, nch_h, L.ht.Sz, mode) : 0);
  L.dxs = take(B * L.hs.Mdg * (int64_t)C * 4);
  L.dxt = take(B * L.ht.Mdg * (int64_t)C * 4);
  const int64_t tiles = ceil_div((int64_t)H * W, 64);
  L.b0p = take(B * tiles * kFcHidden * 4);
  // d W1: enough workgroups to fill the chip, few enough rows to reduce
  int t = 1;
  while (B * t < 2 * kNumCU && t * 128 < H * W) t *= 2;
  L.dw1_tiles = t;
  L.dw1p = take(B * t * (32 * kFcHidden + 32) * 4);
  L.red = take((32 * kFcHidden + 32 + kFcHidden) * 4);
  L.red_tmp = take((int64_t)kFcRedTmpFloats * 4);
  // exact-f32 weight gradient: per-split partial sums (the two halves run one after the other and share it)
  const int64_t sp_s = fc_wgrad_splits(B, L.hs.M, L.cpad), sp_t = fc_wgrad_splits(B, L.ht.M, L.cpad);
  int64_t dwp = mode == 0 ? (sp_s > sp_t ? sp_s : sp_t) * L.KK * L.cpad * kFcHidden * 4 : 0;
  // bf16 features (mode 1) at k = 5: the one-f16-term weight-gradient kernel is this path's slowest (226 us per half at
  // B = 8, 64x64, against ~125 for the float32 Winograd-domain kernel, profiles/r2_face_bf16_kernel_stats.txt); its operands
  // are exact in float32, so the f32 kernel takes over (tuning key 19 = 1: keep the f16 kernel)
  L.wgrad_f32_wino = mode_ == 1 && k == 5 && tuning(19) != 1;
  if (wino || L.wgrad_f32_wino) {  // Winograd-domain partials: 36 points per (c, n)
    const int64_t ws_s = fc_wino_wgrad_splits(B, L.hs.Ho, L.hs.Wo, L.cpad, k), ws_t = fc_wino_wgrad_splits(B, L.ht.Ho, L.ht.Wo, L.cpad, k);
    const int64_t w = (ws_s > ws_t ? ws_s : ws_t) * 36 * L.cpad * kFcHidden * 4;
    if (w > dwp) dwp = w;
  }
  L.dwp = take(dwp);
  L.dwp2 = take(dwp);   // the target half's partials: both halves are reduced by one launch pair
  const int64_t x32_s = fc_packed_bytes(B, L.nch_c, L.hs.Sx, 0), x32_t = fc_packed_bytes(B, L.nch_c, L.ht.Sx, 0);
  L.x32 = take(L.wgrad_f32_wino ? (x32_s > x32_t ? x32_s : x32_t) : 0);
  L.x32b = take(L.wgrad_f32_wino ? x32_t : 0);   // the target half's copy: both weight gradients run as one grid
  L.bwd_total = o;
  return L;
}

static int fc_args_ok(int64_t B, int64_t C, int64_t H, int64_t W, int k, int mode) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0 || k < 1) return GFLA_ERR_BAD_SHAPE;
  if (!fc_mode_ok(mode) || (k != 3 && k != 5)) return GFLA_ERR_UNSUPPORTED;
  if (B > 65535 || C > 4096 || H > 2048 || W > 2048) return GFLA_ERR_UNSUPPORTED;
  // the smallest input tile of each convolution (64 outputs + the tap halo) has to fit the LDS of a CU
  const FcHalf hs = fc_half((int)H, (int)W, k, true), ht = fc_half((int)H, (int)W, k, false);
  if (mode == 4) {
    if (!fc_wino_fits(hs.Mv, hs.Wo, hs.Wp, k) || !fc_wino_fits(hs.Md, hs.Wp, hs.Wp, k) ||
        !fc_wino_fits(ht.Mv, ht.Wo, ht.Wp, k) || !fc_wino_fits(ht.Md, ht.Wp, ht.Wp, k))
      return GFLA_ERR_UNSUPPORTED;
  } else if (mode == 5) {
    if (!fc_wino16_fits(hs.Mv, hs.Wo, hs.Wp, k) || !fc_wino16_fits(hs.Md, hs.Wp, hs.Wp, k) ||
        !fc_wino16_fits(ht.Mv, ht.Wo, ht.Wp, k) || !fc_wino16_fits(ht.Md, ht.Wp, ht.Wp, k))
      return GFLA_ERR_UNSUPPORTED;
  } else if (!fc_conv_fits(hs.Wo, hs.Wp, k, mode) || !fc_conv_fits(hs.Wp, hs.Wp, k, mode) ||
             !fc_conv_fits(ht.Wo, ht.Wp, k, mode) || !fc_conv_fits(ht.Wp, ht.Wp, k, mode))
    return GFLA_ERR_UNSUPPORTED;
  if ((int64_t)64 * (W + 1) * 4 > 64 * 1024) return GFLA_ERR_UNSUPPORTED;
  return GFLA_OK;
}

#define GFLA_TRY(expr)           \
  do {                           \
    const int rc_ = (expr);      \
    if (rc_ != GFLA_OK) return rc_; \
  } while (0)

// Mode 5 runs a convolution on the two-term f16 kernel where that kernel measured faster than the float32 one: everywhere at
// k = 5 and the forward at k = 3; the k = 3 data gradient (128 -> C channels on a 32x22 map: 4 x 4 outputs per tile make the
// f16 kernel's epilogue its largest part) stays on fc_wino.hip: 141 against 166 us at (32,256,32,22).  Tuning key 43 = 1: f16
// kernels everywhere (tests).
static bool fc_w16_dgrad(int k) { return k == 5 || tuning(43) == 1; }
// ... and the weight gradient of the k = 5 layer (both halves in one launch; needs max |dz| of both gradient maps, which the
// data-gradient convolutions of that layer need anyway)
static bool fc_w16_wgrad(int mode_, int k) { return mode_ == 5 && k == 5 && tuning(49) != 1; }

// Mode 5, later in round 6: WHICH convolution runs on which kernel.  Measured per launch at B = 32 (tools/probe_modes.py): the
// DIRECT kernels with two f16 terms per operand and three cross products (mode 2's arithmetic, fc_conv_impl.h) beat the
// Winograd-domain f16 kernel on the k = 5 layer -- forward 216 + 189 against 437 us, data gradient 236 + 211 against 527 -- and
// the float32 Winograd kernel on the k = 3 data gradient (58 + 56 against 142); the k = 3 forward stays in the Winograd domain
// (114 against 65 + 65).  Round 6's first half had found the same and gained nothing, because mode 2 wanted its own packed f16
// copies of the activations and of the gradient maps (a pack pass each, and the weight gradients want float32 anyway); the
// SRC32 form of the direct kernel reads the float32 maps every other kernel of the mode uses and splits while it stages.
// The weight gradients stay in the Winograd domain (k = 5: two-term f16, k = 3: float32): direct 418 + 359 / 113 + 95 us.
// Tuning key 52 = 1: Winograd-domain convolutions everywhere (the first half of round 6).
static bool fc_hyb(int mode_) { return mode_ == 5 && tuning(52) != 1; }
static bool fc_hyb_fwd(int mode_, int k) { return fc_hyb(mode_) && k == 5; }
static bool fc_hyb_fits(const FcLayout &L, int k) {
  return fc_conv_fits(L.hs.Wo, L.hs.Wp, k, 2) && fc_conv_fits(L.hs.Wp, L.hs.Wp, k, 2) && fc_conv_fits(L.ht.Wo, L.ht.Wp, k, 2) &&
         fc_conv_fits(L.ht.Wp, L.ht.Wp, k, 2);
}

// the four Winograd weight sets of one layer (mode 4; mode 5: two-term f16 words, scaled by the slot kAmaxW; fwd_only: the
// data-gradient sets are not wanted)
static int fc_wino_pack_all(const FcLayout &L, const float *w0, unsigned char *ws, int C, int k, hipStream_t stream,
                            bool w16 = false, bool fwd_only = false) {
  float *u_ft = reinterpret_cast<float *>(ws + L.wu_ft), *u_fs = reinterpret_cast<float *>(ws + L.wu_fs);
  float *u_dt = reinterpret_cast<float *>(ws + L.wu_dt), *u_ds = reinterpret_cast<float *>(ws + L.wu_ds);
  if (w16) {
    const bool d16 = fc_w16_dgrad(k) && !fwd_only;
    GFLA_TRY(fc_wino16_pack_weights(w0, reinterpret_cast<const uint32_t *>(ws + L.amax) + kAmaxW, u_ft, u_fs, d16 ? u_dt : nullptr,
                                    d16 ? u_ds : nullptr, C, k, stream));
    return (d16 || fwd_only) ? GFLA_OK : fc_wino_pack_weights(w0, nullptr, nullptr, u_dt, u_ds, C, k, stream);
  }
  return fc_wino_pack_weights(w0, reinterpret_cast<float *>(ws + L.wu_ft), reinterpret_cast<float *>(ws + L.wu_fs),
                              reinterpret_cast<float *>(ws + L.wu_dt), reinterpret_cast<float *>(ws + L.wu_ds), C, k, stream);
}

// one or two Winograd-domain convolutions in mode 4 (float32 operands) or 5 (two-term f16 operands: amax[j] = the max |x| slot
// of job j's input, amax_w the weights')
static int fc_wino_jobs(const WnConvJob *jobs, int njobs, const uint32_t *const *amax, const uint32_t *amax_w, bool w16, int64_t B,
                        int nch, int k, hipStream_t stream) {
  if (!w16) return fc_wino_conv_jobs(jobs, njobs, B, nch, k, stream);
  Wn16ConvJob j16[2];
  for (int j = 0; j < njobs && j < 2; ++j)
    j16[j] = Wn16ConvJob{jobs[j].X, reinterpret_cast<const uint32_t *>(jobs[j].U), amax[j], jobs[j].out, jobs[j].out_bs, jobs[j].ldo,
                         jobs[j].n_valid, jobs[j].M, jobs[j].Wv, jobs[j].Wp, jobs[j].S};
  return fc_wino16_conv_jobs(j16, njobs, B, nch, k, amax_w, stream);
}

static int fc_forward(const float *source, const float *target, const float *flow, const float *w0, const float *b0,
                      const float *w1, const float *b1, void *ws_, float *logits, int64_t B, int C, int H, int W,
                      int k, float slope, int mode_, hipStream_t stream) {
  if (!source || !target || !flow || !w0 || !w1 || !ws_ || !logits) return GFLA_ERR_NULL_POINTER;
  GFLA_TRY(fc_args_ok(B, C, H, W, k, mode_));
  if (B == 0) return GFLA_OK;
  note_path(mode_ == 5 ? GFLA_PATH_FC_FWD_MODE5 : GFLA_PATH_FC_FWD_MODE0 + mode_);
  const FcLayout L = fc_layout(B, C, H, W, k, mode_);
  const bool wino = fc_is_wino(mode_), w16 = mode_ == 5;
  const int mode = fc_base_mode(mode_);
  unsigned char *ws = static_cast<unsigned char *>(ws_);
  uint32_t *amax = reinterpret_cast<uint32_t *>(ws + L.amax);
  const uint32_t *a_src = mode ? amax + kAmaxSrc : nullptr, *a_tgt = mode ? amax + kAmaxTgt : nullptr;
  const uint32_t *a_w = mode ? amax + kAmaxW : nullptr;
  if (mode || w16) {   // the f16-split modes scale by max |x|; the float32 modes never read the slots
    if (hipMemsetAsync(amax, 0, kAmaxSlots * 4, stream) != hipSuccess) return GFLA_ERR_LAUNCH;
    GFLA_TRY(fc_maxabs_multi(source, B * (int64_t)C * H * W, amax + kAmaxSrc, target, B * (int64_t)C * H * W, amax + kAmaxTgt,
                             w0, (int64_t)kFcHidden * 2 * C * k * k, amax + kAmaxW, stream));
  }
  const bool hyb = fc_hyb(mode_) && fc_hyb_fits(L, k), hyb_f = hyb && fc_hyb_fwd(mode_, k);
  if (hyb)   // mode 2's packs of the sets the direct kernels take (scaled by the slot kAmaxW)
    GFLA_TRY(fc_pack_weights(w0, amax + kAmaxW, hyb_f ? ws + L.wf_t : nullptr, hyb_f ? ws + L.wf_s : nullptr, ws + L.wd_t, ws + L.wd_s,
                             C, k, 2, stream));
  if (wino && !hyb_f) GFLA_TRY(fc_wino_pack_all(L, w0, ws, C, k, stream, w16, hyb));
  GFLA_TRY(fc_pack_act2(source, a_src, ws + L.xs, L.hs, target, a_tgt, ws + L.xt, L.ht, B, C, H, W, mode, stream));
  float *gs = reinterpret_cast<float *>(ws + L.gs), *gt = reinterpret_cast<float *>(ws + L.gt);
  const PackedDesc xs = fc_desc_packed(ws + L.xs, B, L.nch_c, L.hs.Sx, mode);
  const PackedDesc xt = fc_desc_packed(ws + L.xt, B, L.nch_c, L.ht.Sx, mode);
  if (hyb_f) {
    const int64_t wsplit_f = fc_wpack_bytes(1, L.nch_c, k, 2) / 2;
    GFLA_TRY(fc_conv_f32src(xs, ws + L.wf_s, wsplit_f, gs, L.hs.Mg * kFcHidden, kFcHidden, kFcHidden, B, L.nch_c, L.hs.Mv,
                            L.hs.Wo, L.hs.Wp, k, amax + kAmaxSrc, amax + kAmaxW, stream));
    GFLA_TRY(fc_conv_f32src(xt, ws + L.wf_t, wsplit_f, gt, L.ht.Mg * kFcHidden, kFcHidden, kFcHidden, B, L.nch_c, L.ht.Mv,
                            L.ht.Wo, L.ht.Wp, k, amax + kAmaxTgt, amax + kAmaxW, stream));
  } else if (wino) {
    const WnConvJob jobs[2] = {   // both halves in one launch (fc_wino.hip: they share the half-empty last round)
        {xs, reinterpret_cast<const float *>(ws + L.wu_fs), gs, L.hs.Mg * kFcHidden, kFcHidden, kFcHidden, L.hs.Mv, L.hs.Wo,
         L.hs.Wp, L.hs.Sx},
        {xt, reinterpret_cast<const float *>(ws + L.wu_ft), gt, L.ht.Mg * kFcHidden, kFcHidden, kFcHidden, L.ht.Mv, L.ht.Wo,
         L.ht.Wp, L.ht.Sx}};
    const uint32_t *const am[2] = {amax + kAmaxSrc, amax + kAmaxTgt};
    GFLA_TRY(fc_wino_jobs(jobs, 2, am, amax + kAmaxW, w16, B, L.nch_c, k, stream));
  } else {
    GFLA_TRY(fc_pack_weights(w0, a_w, ws + L.wf_t, ws + L.wf_s, ws + L.wd_t, ws + L.wd_s, C, k, mode, stream));
    const int64_t wsplit_f = fc_wpack_bytes(1, L.nch_c, k, mode) / fc_nsplit(mode);
    GFLA_TRY(fc_conv(xs, ws + L.wf_s, wsplit_f, gs, L.hs.Mg * kFcHidden, kFcHidden, kFcHidden, B, L.nch_c, L.hs.Mv,
                     L.hs.Wo, L.hs.Wp, k, mode, a_src, a_w, stream));
    GFLA_TRY(fc_conv(xt, ws + L.wf_t, wsplit_f, gt, L.ht.Mg * kFcHidden, kFcHidden, kFcHidden, B, L.nch_c, L.ht.Mv,
                     L.ht.Wo, L.ht.Wp, k, mode, a_tgt, a_w, stream));
  }
  return fc_sample_tail_fwd(gs, gt, flow, b0, w1, b1, reinterpret_cast<float *>(ws + L.hid), logits, B, H, W, k,
                            L.hs.Mg * kFcHidden, L.ht.Mg * kFcHidden, L.hs.Wo, L.ht.Wo, slope, stream);
}

// mode 4: the weight gradient in the Winograd domain (tuning key 19: 1 = the direct kernel, 3 = round 3's choice: Winograd for
// k = 5 only).  Round 3 kept the direct kernel for k = 3 (135 vs 172 us at C256 32x22: units of one tile row, 6 of 16 tiles);
// with units of whole tile rows and both halves in one grid it is 102 / 88 us against 151 / 129
// (profiles/r4_wino_wgrad_k3_multirow.txt).
static bool fc_wgrad_in_wino_domain(int mode_, int k) {
  return fc_is_wino(mode_) && tuning(19) != 1 && (k == 5 || tuning(19) != 3);
}

// data gradient (transposed convolution + replicate-pad fold) and weight gradient of one half, from its f32
// Z-layout gradient map
static int fc_half_backward(const FcLayout &L, const FcHalf &g, bool source, unsigned char *ws, unsigned char *sc,
                            float *g_x, float *g_w0, int64_t B, int C, int H, int W, int k, int mode_,
                            hipStream_t stream, int acc_x = 0, bool dgrad_done = false, bool reduce_now = true,
                            bool wgrad_done = false, bool z_ready = false, bool defer_fold = false) {
  // z_ready: max |dz| and the packed gradient map already exist; defer_fold: fc_backward folds both halves in one launch
  // dgrad_done: convolution AND fold already enqueued; reduce_now = false: the weight-gradient partials stay in this half's
  // buffer (source: dwp, target: dwp2) and fc_backward reduces both halves together
  const bool wino = fc_is_wino(mode_), w16 = mode_ == 5;
  const int mode = fc_base_mode(mode_);
  const bool want_w = g_w0 != nullptr;
  uint32_t *amax = reinterpret_cast<uint32_t *>(ws + L.amax);
  const int nch_h = kFcHidden / kFcChunk;
  float *dz = reinterpret_cast<float *>(sc + (source ? L.dzs : L.dzt));
  unsigned char *zpk = sc + (source ? L.zs_pk : L.zt_pk);
  uint32_t *a_z = mode ? amax + (source ? kAmaxZs : kAmaxZt) : nullptr;
  const uint32_t *a_x = mode ? amax + (source ? kAmaxSrc : kAmaxTgt) : nullptr;
  const uint32_t *a_w = mode ? amax + kAmaxW : nullptr;
  PackedDesc Z;
  if (mode) {
    if (!z_ready) {
      GFLA_TRY(fc_maxabs(dz, B * g.Sz * kFcHidden, a_z, stream));
      GFLA_TRY(fc_pack_z(dz, a_z, zpk, B, g.Sz, kFcHidden, mode, stream));
    }
    Z = fc_desc_packed(zpk, B, nch_h, g.Sz, mode);
  } else {
    Z = fc_desc_nhwc(dz, g.Sz, kFcHidden);
  }
  if (g_x) {
    float *dx = reinterpret_cast<float *>(sc + (source ? L.dxs : L.dxt));
    if (wino && fc_hyb(mode_) && fc_hyb_fits(L, k)) {
      if (!dgrad_done) {   // the direct f16x2 kernel on the float32 gradient map (see fc_hyb)
        uint32_t *a_z16 = amax + (source ? kAmaxZs : kAmaxZt);
        GFLA_TRY(fc_maxabs(dz, B * g.Sz * kFcHidden, a_z16, stream));   // (raises a slot that is zero or already holds the maximum)
        GFLA_TRY(fc_conv_f32src(Z, ws + (source ? L.wd_s : L.wd_t), fc_wpack_bytes(L.nt_d, nch_h, k, 2) / 2, dx, g.Mdg * (int64_t)C, C,
                                C, B, nch_h, g.Md, g.Wp, g.Wp, k, a_z16, amax + kAmaxW, stream));
      }
    } else if (wino) {
      if (!dgrad_done) {
        const WnConvJob job{Z, reinterpret_cast<const float *>(ws + (source ? L.wu_ds : L.wu_dt)), dx, g.Mdg * (int64_t)C, C, C, g.Md,
                            g.Wp, g.Wp, g.Sz};
        uint32_t *a_z16 = amax + (source ? kAmaxZs : kAmaxZt);
        const bool d16 = w16 && fc_w16_dgrad(k);
        if (d16) GFLA_TRY(fc_maxabs(dz, B * g.Sz * kFcHidden, a_z16, stream));   // (the slot was zeroed by the caller)
        const uint32_t *const am[1] = {a_z16};
        GFLA_TRY(fc_wino_jobs(&job, 1, am, amax + kAmaxW, d16, B, nch_h, k, stream));
      }
    } else {
      const int64_t wsplit_d = fc_wpack_bytes(L.nt_d, nch_h, k, mode) / fc_nsplit(mode);
      GFLA_TRY(fc_conv(Z, ws + (source ? L.wd_s : L.wd_t), wsplit_d, dx, g.Mdg * (int64_t)C, C, C, B, nch_h, g.Md, g.Wp,
                       g.Wp, k, mode, a_z, a_w, stream));
    }
    if (!dgrad_done && !defer_fold) GFLA_TRY(fc_fold(dx, g_x, B, C, H, W, g, g.Mdg * (int64_t)C, acc_x, stream));
  }
  if (want_w) {
    const PackedDesc X = fc_desc_packed(ws + (source ? L.xs : L.xt), B, L.nch_c, g.Sx, mode);
    float *part = reinterpret_cast<float *>(sc + (source ? L.dwp : L.dwp2));
    if (L.wgrad_f32_wino) {
      float *x32 = reinterpret_cast<float *>(sc + (source ? L.x32 : L.x32b));
      GFLA_TRY(fc_unpack_act(ws + (source ? L.xs : L.xt), a_x, x32, B, L.nch_c, g.Sx, stream));
      const PackedDesc X32 = fc_desc_packed(x32, B, L.nch_c, g.Sx, 0);
      if (!wgrad_done)
        GFLA_TRY(fc_wino_wgrad(X32, dz, g.Sz * kFcHidden, g.lead, part, L.cpad, B, g.Ho, g.Wo, g.Wp, g.Sx, k, stream));
      if (reduce_now)
        GFLA_TRY(fc_wino_wgrad_reduce(part, fc_wino_wgrad_splits(B, g.Ho, g.Wo, L.cpad, k), g_w0, C, source ? C : 0, L.cpad, k,
                                      stream));
    } else if (fc_wgrad_in_wino_domain(mode_, k)) {
      if (!wgrad_done) {   // (fc_backward launches both halves' kernels as one grid)
        if (fc_w16_wgrad(mode_, k)) {   // two-term f16 operands (one job: the per-half entry points and one-sided backward calls)
          uint32_t *a_z16 = amax + (source ? kAmaxZs : kAmaxZt);
          GFLA_TRY(fc_maxabs(dz, B * g.Sz * kFcHidden, a_z16, stream));   // (raises a slot that is zero or already holds the maximum)
          const WwJob job{X, dz, part, g.Sz * kFcHidden, g.lead, g.Sx, g.Ho, g.Wo, g.Wp};
          const uint32_t *const ax[1] = {amax + (source ? kAmaxSrc : kAmaxTgt)}, *const az[1] = {a_z16};
          GFLA_TRY(fc_wino16_wgrad_jobs(&job, 1, L.cpad, B, k, ax, az, stream));
        } else {
          GFLA_TRY(fc_wino_wgrad(X, dz, g.Sz * kFcHidden, g.lead, part, L.cpad, B, g.Ho, g.Wo, g.Wp, g.Sx, k, stream));
        }
      }
      if (reduce_now)
        GFLA_TRY(fc_wino_wgrad_reduce(part, fc_wino_wgrad_splits(B, g.Ho, g.Wo, L.cpad, k), g_w0, C, source ? C : 0, L.cpad, k,
                                      stream));
    } else if (mode == 0) {
      GFLA_TRY(fc_wgrad_f32(X, Z, g.lead, part, L.cpad, B, g.M, g.Wp, k, stream));
      if (reduce_now) GFLA_TRY(fc_wgrad_reduce(part, fc_wgrad_splits(B, g.M, L.cpad), g_w0, C, source ? C : 0, L.cpad, k, stream));
    } else {
      GFLA_TRY(fc_wgrad(X, Z, g.lead, reinterpret_cast<float *>(sc + (source ? L.dw_s : L.dw_t)), L.cpad, B, g.M, g.Wp,
                        k, mode, stream));
    }
  }
  return GFLA_OK;
}

// (Round 4, measured and dropped: the weight gradients of both halves on a library-owned side stream, forked from and joined
// to the caller's stream with events, concurrently with the data-gradient convolutions and their folds -- 4.75 ms per
// step against 4.66 on one stream, profiles/r4_fc_backward_side_stream.txt.  Both chains are full-chip kernels that own a
// CU's whole register file; side by side they only contend.)
static int fc_backward(void *ws_, const float *flow, const float *w1, const float *g_logits, void *scratch_,
                       float *g_source, float *g_target, float *g_flow, float *g_w0, float *g_b0, float *g_w1,
                       float *g_b1, int64_t B, int C, int H, int W, int k, float slope, int mode_, int flags,
                       hipStream_t stream) {
  if (!ws_ || !flow || !w1 || !g_logits || !scratch_) return GFLA_ERR_NULL_POINTER;
  GFLA_TRY(fc_args_ok(B, C, H, W, k, mode_));
  if (B == 0) return GFLA_OK;
  note_path(mode_ == 5 ? GFLA_PATH_FC_BWD_MODE5 : GFLA_PATH_FC_BWD_MODE0 + mode_);
  const FcLayout L = fc_layout(B, C, H, W, k, mode_);
  const int mode = fc_base_mode(mode_);
  unsigned char *ws = static_cast<unsigned char *>(ws_), *sc = static_cast<unsigned char *>(scratch_);
  uint32_t *amax = reinterpret_cast<uint32_t *>(ws + L.amax);
  const float *hid = reinterpret_cast<const float *>(ws + L.hid);
  float *red_tmp = reinterpret_cast<float *>(sc + L.red_tmp);
  const bool w16 = mode_ == 5;
  const float *gs = reinterpret_cast<const float *>(ws + L.gs);
  const bool need_s = g_source || g_w0, need_t = g_target || g_w0;
  float *dzs = need_s ? reinterpret_cast<float *>(sc + L.dzs) : nullptr;
  float *dzt = need_t ? reinterpret_cast<float *>(sc + L.dzt) : nullptr;
  float *b0p = g_b0 ? reinterpret_cast<float *>(sc + L.b0p) : nullptr;
  const int64_t tiles = ceil_div((int64_t)H * W, 64);
  // d Gs by owner-computes (fc_sample.hip: fc_scatter_own_kernel) whenever both gradient maps are wanted and a row of the
  // map fits the LDS: no global atomics, no memset of the source half's map, and max |dz| of both maps as by-products
  // (tuning key 46 = 1: round 2's atomics)
  const bool own = dzs && dzt && tuning(46) != 1 && fc_scatter_own_rows(B, L.hs.Ho, L.hs.Wo) >= 1;
  if (hipMemsetAsync(sc + (own ? L.dzt : 0), 0, L.zero_bytes - (own ? L.dzt : 0), stream) != hipSuccess) return GFLA_ERR_LAUNCH;
  if ((mode || w16 || own) && hipMemsetAsync(amax + kAmaxZs, 0, 8, stream) != hipSuccess) return GFLA_ERR_LAUNCH;
  GFLA_TRY(fc_sample_tail_bwd(gs, flow, hid, w1, g_logits, own ? nullptr : dzs, dzt, g_flow, b0p, B, H, W, k, L.hs.Mg * kFcHidden,
                              L.hs.Wo, L.hs.Wp, L.ht.Wp, L.hs.Sz * kFcHidden, L.ht.Sz * kFcHidden, L.hs.lead, L.ht.lead, slope,
                              flags & GFLA_FC_ACCUMULATE_FLOW, stream, own ? amax + kAmaxZt : nullptr));
  if (own)
    GFLA_TRY(fc_sample_scatter_own(flow, dzt, dzs, amax + kAmaxZt, amax + kAmaxZs, B, H, W, k, L.hs.Ho, L.hs.Wo, L.hs.Wp, L.ht.Wp,
                                   L.hs.Sz * kFcHidden, L.ht.Sz * kFcHidden, L.hs.lead, L.ht.lead, L.hs.Sz, stream));
  const float *dw1p = nullptr;
  if (g_w1 || g_b1) {
    float *part = reinterpret_cast<float *>(sc + L.dw1p);
    GFLA_TRY(fc_dw1(hid, g_logits, part, B, H * W, L.KK, L.dw1_tiles, slope, stream));
    dw1p = part;
  }
  // d b0, d W1, d b1: one two-pass reduction that writes all three in place (8 reduce launches + 4 device copies before).
  // (Round 4, measured and dropped: this chain and the forward's weight transform forked onto a library-owned second
  // stream -- 4.449 ms per step against 4.460 on the caller's stream alone, profiles/r4_fc_small_kernels_side_stream.txt.)
  GFLA_TRY(fc_reduce_bias_w1(b0p, B * tiles, g_b0, dw1p, B * L.dw1_tiles, g_w1, g_b1, L.KK, red_tmp, stream));
  // mode 4: the data-gradient convolutions of both halves in one launch (fc_wino.hip)
  const bool both_dgrads = fc_is_wino(mode_) && g_source && g_target;
  if (both_dgrads) {
    const int nch_h = kFcHidden / kFcChunk;
    const bool hyb_d = fc_hyb(mode_) && fc_hyb_fits(L, k);
    const bool d16 = w16 && fc_w16_dgrad(k);
    if ((d16 || hyb_d) && !own)   // max |dz| of both gradient maps: the scale of their two-term f16 split
      GFLA_TRY(fc_maxabs_multi(dzs, B * L.hs.Sz * kFcHidden, amax + kAmaxZs, dzt, B * L.ht.Sz * kFcHidden, amax + kAmaxZt, nullptr,
                               0, nullptr, stream));
    if (hyb_d) {   // the direct f16x2 kernels on the float32 gradient maps (see fc_hyb)
      const int64_t wsplit_d = fc_wpack_bytes(L.nt_d, nch_h, k, 2) / 2;
      GFLA_TRY(fc_conv_f32src(fc_desc_nhwc(dzs, L.hs.Sz, kFcHidden), ws + L.wd_s, wsplit_d, reinterpret_cast<float *>(sc + L.dxs),
                              L.hs.Mdg * (int64_t)C, C, C, B, nch_h, L.hs.Md, L.hs.Wp, L.hs.Wp, k, amax + kAmaxZs, amax + kAmaxW, stream));
      GFLA_TRY(fc_conv_f32src(fc_desc_nhwc(dzt, L.ht.Sz, kFcHidden), ws + L.wd_t, wsplit_d, reinterpret_cast<float *>(sc + L.dxt),
                              L.ht.Mdg * (int64_t)C, C, C, B, nch_h, L.ht.Md, L.ht.Wp, L.ht.Wp, k, amax + kAmaxZt, amax + kAmaxW, stream));
    } else {
    const WnConvJob jobs[2] = {
        {fc_desc_nhwc(dzs, L.hs.Sz, kFcHidden), reinterpret_cast<const float *>(ws + L.wu_ds),
         reinterpret_cast<float *>(sc + L.dxs), L.hs.Mdg * (int64_t)C, C, C, L.hs.Md, L.hs.Wp, L.hs.Wp, L.hs.Sz},
        {fc_desc_nhwc(dzt, L.ht.Sz, kFcHidden), reinterpret_cast<const float *>(ws + L.wu_dt),
         reinterpret_cast<float *>(sc + L.dxt), L.ht.Mdg * (int64_t)C, C, C, L.ht.Md, L.ht.Wp, L.ht.Wp, L.ht.Sz}};
    const uint32_t *const am[2] = {amax + kAmaxZs, amax + kAmaxZt};
    GFLA_TRY(fc_wino_jobs(jobs, 2, am, amax + kAmaxW, d16, B, nch_h, k, stream));
    }
    // ... and their replicate-pad folds in one launch (fc_sample.hip)
    GFLA_TRY(fc_fold2(reinterpret_cast<const float *>(sc + L.dxs), g_source, L.hs, L.hs.Mdg * (int64_t)C,
                      (flags & GFLA_FC_ACCUMULATE_SOURCE) ? 1 : 0, reinterpret_cast<const float *>(sc + L.dxt), g_target, L.ht,
                      L.ht.Mdg * (int64_t)C, 0, B, C, H, W, stream));
  }
  // weight-gradient partials of both halves: summed (and, in the Winograd domain, transformed back) by ONE launch (pair)
  const bool wino_w = L.wgrad_f32_wino || fc_wgrad_in_wino_domain(mode_, k);
  const bool defer = g_w0 && need_s && need_t && (wino_w || mode == 0);
  // mode 4, k = 5: the two Winograd-domain weight-gradient kernels as ONE grid (each is one round of workgroups)
  // (bf16 features, k = 5: the same kernel on the unpacked activations of both halves)
  const bool both_wgrads = defer && wino_w && g_source && g_target && tuning(21) != 2;
  // f16-split modes: max |dz| and the packed gradient maps of both halves by one launch each, and (any mode without the
  // joint Winograd launch above) both replicate-pad folds by one launch behind the two data-gradient convolutions
  const bool z_both = mode != 0 && need_s && need_t;
  if (z_both) {
    if (!own)
      GFLA_TRY(fc_maxabs_multi(dzs, B * L.hs.Sz * kFcHidden, amax + kAmaxZs, dzt, B * L.ht.Sz * kFcHidden, amax + kAmaxZt, nullptr,
                             0, nullptr, stream));
    GFLA_TRY(fc_pack_z2(dzs, amax + kAmaxZs, sc + L.zs_pk, L.hs.Sz, dzt, amax + kAmaxZt, sc + L.zt_pk, L.ht.Sz, B, kFcHidden,
                        mode, stream));
  }
  const bool fold_both = !both_dgrads && g_source && g_target;
  if (need_s)
    GFLA_TRY(fc_half_backward(L, L.hs, true, ws, sc, g_source, g_w0, B, C, H, W, k, mode_, stream,
                              (flags & GFLA_FC_ACCUMULATE_SOURCE) ? 1 : 0, both_dgrads, !defer, both_wgrads, z_both, fold_both));
  if (need_t)
    GFLA_TRY(fc_half_backward(L, L.ht, false, ws, sc, g_target, g_w0, B, C, H, W, k, mode_, stream, 0, both_dgrads, !defer,
                              both_wgrads, z_both, fold_both));
  if (fold_both)
    GFLA_TRY(fc_fold2(reinterpret_cast<const float *>(sc + L.dxs), g_source, L.hs, L.hs.Mdg * (int64_t)C,
                      (flags & GFLA_FC_ACCUMULATE_SOURCE) ? 1 : 0, reinterpret_cast<const float *>(sc + L.dxt), g_target, L.ht,
                      L.ht.Mdg * (int64_t)C, 0, B, C, H, W, stream));
  if (both_wgrads) {
    const PackedDesc Xs = L.wgrad_f32_wino ? fc_desc_packed(sc + L.x32, B, L.nch_c, L.hs.Sx, 0)
                                           : fc_desc_packed(ws + L.xs, B, L.nch_c, L.hs.Sx, mode);
    const PackedDesc Xt = L.wgrad_f32_wino ? fc_desc_packed(sc + L.x32b, B, L.nch_c, L.ht.Sx, 0)
                                           : fc_desc_packed(ws + L.xt, B, L.nch_c, L.ht.Sx, mode);
    const WwJob jobs[2] = {
        {Xs, dzs, reinterpret_cast<float *>(sc + L.dwp), L.hs.Sz * kFcHidden, L.hs.lead, L.hs.Sx, L.hs.Ho, L.hs.Wo, L.hs.Wp},
        {Xt, dzt, reinterpret_cast<float *>(sc + L.dwp2), L.ht.Sz * kFcHidden, L.ht.lead, L.ht.Sx, L.ht.Ho, L.ht.Wo, L.ht.Wp}};
    // mode 5, k = 5: both operands as two-term f16 values on the f16 matrix cores (fc_wino.hip: fc_wino16_wgrad_kernel); the
    // max |x| slots of the activations are the forward's, those of the gradient maps this call's (tuning key 49 = 1: float32)
    // (bf16 features, k = 5: the same kernel on the unpacked activations -- their max |x| slots and those of the gradient maps
    // were filled for the one-f16-term convolutions)
    if (fc_w16_wgrad(mode_, k) || (L.wgrad_f32_wino && k == 5 && tuning(49) != 1)) {
      const uint32_t *const ax[2] = {amax + kAmaxSrc, amax + kAmaxTgt}, *const az[2] = {amax + kAmaxZs, amax + kAmaxZt};
      GFLA_TRY(fc_wino16_wgrad_jobs(jobs, 2, L.cpad, B, k, ax, az, stream));
    } else {
      GFLA_TRY(fc_wino_wgrad_jobs(jobs, 2, L.cpad, B, k, stream));
    }
  }
  if (defer) {
    float *part_s = reinterpret_cast<float *>(sc + L.dwp), *part_t = reinterpret_cast<float *>(sc + L.dwp2);
    if (wino_w)
      GFLA_TRY(fc_wino_wgrad_reduce2(part_s, fc_wino_wgrad_splits(B, L.hs.Ho, L.hs.Wo, L.cpad, k), part_t,
                                     fc_wino_wgrad_splits(B, L.ht.Ho, L.ht.Wo, L.cpad, k), g_w0, C, L.cpad, k, stream));
    else
      GFLA_TRY(fc_wgrad_reduce2(part_s, fc_wgrad_splits(B, L.hs.M, L.cpad), part_t, fc_wgrad_splits(B, L.ht.M, L.cpad), g_w0, C,
                                L.cpad, k, stream));
  }
  if (g_w0 && mode != 0 && !L.wgrad_f32_wino) {
    const uint32_t *a = mode ? amax : nullptr;
    GFLA_TRY(fc_unpack_wgrad(reinterpret_cast<float *>(sc + L.dw_t), reinterpret_cast<float *>(sc + L.dw_s),
                             a ? a + kAmaxTgt : nullptr, a ? a + kAmaxSrc : nullptr, a ? a + kAmaxZt : nullptr,
                             a ? a + kAmaxZs : nullptr, g_w0, C, L.cpad, k, stream));
  }
  return GFLA_OK;
}

}  // namespace gfla

extern "C" {
using namespace gfla;

int gfla_fc_supported(int64_t C, int64_t H, int64_t W, int kernel_size, int mode) {
  return fc_args_ok(1, C, H, W, kernel_size, mode) == GFLA_OK ? 1 : 0;
}

int64_t gfla_fc_workspace_bytes(int64_t B, int64_t C, int64_t H, int64_t W, int kernel_size, int mode, int which) {
  if (fc_args_ok(B, C, H, W, kernel_size, mode) != GFLA_OK) return -1;
  const FcLayout L = fc_layout(B, (int)C, (int)H, (int)W, kernel_size, mode);
  return which == 0 ? L.fwd_total : L.bwd_total;
}

int gfla_fc_forward_f32(const float *source, const float *target, const float *flow, const float *w0,
                        const float *b0, const float *w1, const float *b1, void *workspace, float *logits, int64_t B,
                        int64_t C, int64_t H, int64_t W, int kernel_size, double slope, int mode,
                        gfla_stream_t stream) {
  return fc_forward(source, target, flow, w0, b0, w1, b1, workspace, logits, B, (int)C, (int)H, (int)W, kernel_size,
                    (float)slope, mode, static_cast<hipStream_t>(stream));
}

int gfla_fc_backward_f32(void *workspace, const float *flow, const float *w1, const float *grad_logits,
                         void *scratch, float *grad_source, float *grad_target, float *grad_flow, float *grad_w0,
                         float *grad_b0, float *grad_w1, float *grad_b1, int64_t B, int64_t C, int64_t H, int64_t W,
                         int kernel_size, double slope, int mode, int flags, gfla_stream_t stream) {
  return fc_backward(workspace, flow, w1, grad_logits, scratch, grad_source, grad_target, grad_flow, grad_w0, grad_b0,
                     grad_w1, grad_b1, B, (int)C, (int)H, (int)W, kernel_size, (float)slope, mode, flags,
                     static_cast<hipStream_t>(stream));
}

/* ---- pieces of the above, exposed for the parity tests ---- */

/* out[0..11] = Hp, Wp, Ho, Wo, pad_t, pad_l, M, Md, lead, Sx, Sz, Mg;  out[12] = Mdg */
int gfla_fc_geometry(int64_t H, int64_t W, int kernel_size, int is_source, int64_t *out) {
  if (!out) return GFLA_ERR_NULL_POINTER;
  if (H <= 0 || W <= 0 || kernel_size < 1) return GFLA_ERR_BAD_SHAPE;
  const FcHalf g = fc_half((int)H, (int)W, kernel_size, is_source != 0);
  const int64_t v[13] = {g.Hp, g.Wp, g.Ho, g.Wo, g.pad_t, g.pad_l, g.M, g.Md, g.lead, g.Sx, g.Sz, g.Mg, g.Mdg};
  for (int i = 0; i < 13; ++i) out[i] = v[i];
  return GFLA_OK;
}

/* convolved map of one half: out (B, Mg, 128) f32, row m = yo*Wo + xo (gfla_fc_geometry) */
int gfla_fc_conv_fwd_f32(const float *x, const float *w0, int is_source, void *workspace, float *out, int64_t B,
                         int64_t C_, int64_t H_, int64_t W_, int kernel_size, int mode, gfla_stream_t stream_) {
  if (!x || !w0 || !workspace || !out) return GFLA_ERR_NULL_POINTER;
  const int C = (int)C_, H = (int)H_, W = (int)W_, k = kernel_size;
  GFLA_TRY(fc_args_ok(B, C, H, W, k, mode));
  if (B == 0) return GFLA_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const FcLayout L = fc_layout(B, C, H, W, k, mode);
  const FcHalf &g = is_source ? L.hs : L.ht;
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  uint32_t *amax = reinterpret_cast<uint32_t *>(ws + L.amax);
  if (hipMemsetAsync(amax, 0, kAmaxSlots * 4, stream) != hipSuccess) return GFLA_ERR_LAUNCH;
  if (fc_is_wino(mode)) {
    const bool w16 = mode == 5;
    unsigned char *xq = ws + (is_source ? L.xs : L.xt);
    uint32_t *a_x16 = amax + (is_source ? kAmaxSrc : kAmaxTgt);
    if (w16) {
      GFLA_TRY(fc_maxabs(x, B * (int64_t)C * H * W, a_x16, stream));
      GFLA_TRY(fc_maxabs(w0, (int64_t)kFcHidden * 2 * C * k * k, amax + kAmaxW, stream));
    }
    GFLA_TRY(fc_pack_act(x, nullptr, xq, B, C, H, W, g, 0, stream));
    const bool hyb = fc_hyb(mode) && fc_hyb_fits(L, k), hyb_f = hyb && fc_hyb_fwd(mode, k);
    if (hyb)
      GFLA_TRY(fc_pack_weights(w0, amax + kAmaxW, hyb_f ? ws + L.wf_t : nullptr, hyb_f ? ws + L.wf_s : nullptr, ws + L.wd_t, ws + L.wd_s,
                               C, k, 2, stream));
    if (hyb_f)
      return fc_conv_f32src(fc_desc_packed(xq, B, L.nch_c, g.Sx, 0), ws + (is_source ? L.wf_s : L.wf_t),
                            fc_wpack_bytes(1, L.nch_c, k, 2) / 2, out, g.Mg * kFcHidden, kFcHidden, kFcHidden, B, L.nch_c, g.Mv, g.Wo,
                            g.Wp, k, a_x16, amax + kAmaxW, stream);
    GFLA_TRY(fc_wino_pack_all(L, w0, ws, C, k, stream, w16, hyb));
    const WnConvJob job{fc_desc_packed(xq, B, L.nch_c, g.Sx, 0), reinterpret_cast<const float *>(ws + (is_source ? L.wu_fs : L.wu_ft)),
                        out, g.Mg * kFcHidden, kFcHidden, kFcHidden, g.Mv, g.Wo, g.Wp, g.Sx};
    const uint32_t *const am[1] = {a_x16};
    return fc_wino_jobs(&job, 1, am, amax + kAmaxW, w16, B, L.nch_c, k, stream);
  }
  uint32_t *a_x = mode ? amax + (is_source ? kAmaxSrc : kAmaxTgt) : nullptr, *a_w = mode ? amax + kAmaxW : nullptr;
  if (mode) {
    GFLA_TRY(fc_maxabs(x, B * (int64_t)C * H * W, a_x, stream));
    GFLA_TRY(fc_maxabs(w0, (int64_t)kFcHidden * 2 * C * k * k, a_w, stream));
  }
  unsigned char *xp = ws + (is_source ? L.xs : L.xt);
  GFLA_TRY(fc_pack_act(x, a_x, xp, B, C, H, W, g, mode, stream));
  GFLA_TRY(fc_pack_weights(w0, a_w, ws + L.wf_t, ws + L.wf_s, ws + L.wd_t, ws + L.wd_s, C, k, mode, stream));
  const int64_t wsplit_f = fc_wpack_bytes(1, L.nch_c, k, mode) / fc_nsplit(mode);
  const PackedDesc X = fc_desc_packed(xp, B, L.nch_c, g.Sx, mode);
  return fc_conv(X, ws + (is_source ? L.wf_s : L.wf_t), wsplit_f, out, g.Mg * kFcHidden, kFcHidden, kFcHidden, B,
                 L.nch_c, g.Mv, g.Wo, g.Wp, k, mode, a_x, a_w, stream);
}

/* gradients of one half from its Z-layout gradient map z (B, Sz, 128) f32 (zero outside the data, see
 * gfla_fc_geometry): grad_x (B,C,H,W), grad_w0 (128, 2C, k, k) with the other half zero.  `workspace` must hold the
 * result of gfla_fc_conv_fwd_f32 for the same x / w0 / half. */
int gfla_fc_conv_bwd_f32(const float *z, int is_source, void *workspace, void *scratch, float *grad_x, float *grad_w0,
                         int64_t B, int64_t C_, int64_t H_, int64_t W_, int kernel_size, int mode,
                         gfla_stream_t stream_) {
  if (!z || !workspace || !scratch) return GFLA_ERR_NULL_POINTER;
  const int C = (int)C_, H = (int)H_, W = (int)W_, k = kernel_size;
  GFLA_TRY(fc_args_ok(B, C, H, W, k, mode));
  if (B == 0) return GFLA_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const FcLayout L = fc_layout(B, C, H, W, k, mode);
  const FcHalf &g = is_source ? L.hs : L.ht;
  unsigned char *ws = static_cast<unsigned char *>(workspace), *sc = static_cast<unsigned char *>(scratch);
  uint32_t *amax = reinterpret_cast<uint32_t *>(ws + L.amax);
  if (hipMemsetAsync(sc, 0, L.zero_bytes, stream) != hipSuccess) return GFLA_ERR_LAUNCH;
  if (hipMemsetAsync(amax + kAmaxZs, 0, 8, stream) != hipSuccess) return GFLA_ERR_LAUNCH;
  if (hipMemcpyAsync(sc + (is_source ? L.dzs : L.dzt), z, (size_t)(B * g.Sz * kFcHidden * 4), hipMemcpyDeviceToDevice,
                     stream) != hipSuccess)
    return GFLA_ERR_LAUNCH;
  if (grad_w0 && fc_base_mode(mode) == 0 &&  // the other half of conv0.weight.grad is zero by contract
      hipMemsetAsync(grad_w0, 0, (size_t)kFcHidden * 2 * C * k * k * 4, stream) != hipSuccess)
    return GFLA_ERR_LAUNCH;
  GFLA_TRY(fc_half_backward(L, g, is_source != 0, ws, sc, grad_x, grad_w0, B, C, H, W, k, mode, stream));
  if (grad_w0 && fc_base_mode(mode) != 0) {
    const uint32_t *a = mode ? amax : nullptr;
    GFLA_TRY(fc_unpack_wgrad(reinterpret_cast<float *>(sc + L.dw_t), reinterpret_cast<float *>(sc + L.dw_s),
                             a ? a + kAmaxTgt : nullptr, a ? a + kAmaxSrc : nullptr, a ? a + kAmaxZt : nullptr,
                             a ? a + kAmaxZs : nullptr, grad_w0, C, L.cpad, k, stream));
  }
  return GFLA_OK;
}

/* ONE internal kernel of the path, on the state a forward + backward of the same shape left in workspace / scratch:
 * per-kernel timing for bench.py and the profiles (results land in scratch areas the next real call overwrites).
 * which: 0 / 1 convolution forward source / target half, 2 / 3 data-gradient convolution, 4 / 5 weight gradient;
 * mode 4 only: 6 / 7 = the forward / data-gradient convolutions of BOTH halves in one launch, as the step issues them. */
int gfla_fc_kernel_f32(int which, void *workspace, void *scratch, int64_t B, int64_t C_, int64_t H_, int64_t W_,
                       int kernel_size, int mode, gfla_stream_t stream_) {
  if (!workspace || !scratch) return GFLA_ERR_NULL_POINTER;
  if (which < 0 || which > 8 || (which > 5 && !fc_is_wino(mode))) return GFLA_ERR_BAD_SHAPE;
  const int C = (int)C_, H = (int)H_, W = (int)W_, k = kernel_size;
  GFLA_TRY(fc_args_ok(B, C, H, W, k, mode));
  if (B == 0) return GFLA_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const FcLayout L = fc_layout(B, C, H, W, k, mode);
  const bool source = (which & 1) == 0;
  const FcHalf &g = source ? L.hs : L.ht;
  unsigned char *ws = static_cast<unsigned char *>(workspace), *sc = static_cast<unsigned char *>(scratch);
  const bool w16 = mode == 5;
  const uint32_t *amx = reinterpret_cast<const uint32_t *>(ws + L.amax);
  const uint32_t *const am_f[2] = {amx + kAmaxSrc, amx + kAmaxTgt}, *const am_d[2] = {amx + kAmaxZs, amx + kAmaxZt};
  const bool hyb = fc_hyb(mode) && fc_hyb_fits(L, k), hyb_f = hyb && fc_hyb_fwd(mode, k);
  const int64_t wsplit_hf = fc_wpack_bytes(1, L.nch_c, k, 2) / 2, wsplit_hd = fc_wpack_bytes(L.nt_d, kFcHidden / kFcChunk, k, 2) / 2;
  auto hyb_fwd = [&](bool src) {
    const FcHalf &h = src ? L.hs : L.ht;
    return fc_conv_f32src(fc_desc_packed(ws + (src ? L.xs : L.xt), B, L.nch_c, h.Sx, 0), ws + (src ? L.wf_s : L.wf_t), wsplit_hf,
                          reinterpret_cast<float *>(ws + (src ? L.gs : L.gt)), h.Mg * kFcHidden, kFcHidden, kFcHidden, B, L.nch_c, h.Mv,
                          h.Wo, h.Wp, k, amx + (src ? kAmaxSrc : kAmaxTgt), amx + kAmaxW, stream);
  };
  auto hyb_dgrad = [&](bool src) {
    const FcHalf &h = src ? L.hs : L.ht;
    return fc_conv_f32src(fc_desc_nhwc(reinterpret_cast<float *>(sc + (src ? L.dzs : L.dzt)), h.Sz, kFcHidden),
                          ws + (src ? L.wd_s : L.wd_t), wsplit_hd, reinterpret_cast<float *>(sc + (src ? L.dxs : L.dxt)),
                          h.Mdg * (int64_t)C, C, C, B, kFcHidden / kFcChunk, h.Md, h.Wp, h.Wp, k, amx + (src ? kAmaxZs : kAmaxZt),
                          amx + kAmaxW, stream);
  };
  if (hyb_f && which == 6) {   // (two launches: the direct kernels take one half each)
    GFLA_TRY(hyb_fwd(true));
    return hyb_fwd(false);
  }
  if (hyb && which == 7) {
    GFLA_TRY(hyb_dgrad(true));
    return hyb_dgrad(false);
  }
  if (hyb_f && which < 2) return hyb_fwd(source);
  if (hyb && (which == 2 || which == 3)) return hyb_dgrad(source);
  if (which == 6) {
    const WnConvJob jobs[2] = {
        {fc_desc_packed(ws + L.xs, B, L.nch_c, L.hs.Sx, 0), reinterpret_cast<const float *>(ws + L.wu_fs),
         reinterpret_cast<float *>(ws + L.gs), L.hs.Mg * kFcHidden, kFcHidden, kFcHidden, L.hs.Mv, L.hs.Wo, L.hs.Wp, L.hs.Sx},
        {fc_desc_packed(ws + L.xt, B, L.nch_c, L.ht.Sx, 0), reinterpret_cast<const float *>(ws + L.wu_ft),
         reinterpret_cast<float *>(ws + L.gt), L.ht.Mg * kFcHidden, kFcHidden, kFcHidden, L.ht.Mv, L.ht.Wo, L.ht.Wp, L.ht.Sx}};
    return fc_wino_jobs(jobs, 2, am_f, amx + kAmaxW, w16, B, L.nch_c, k, stream);
  }
  if (which == 7) {
    const WnConvJob jobs[2] = {
        {fc_desc_nhwc(reinterpret_cast<float *>(sc + L.dzs), L.hs.Sz, kFcHidden), reinterpret_cast<const float *>(ws + L.wu_ds),
         reinterpret_cast<float *>(sc + L.dxs), L.hs.Mdg * (int64_t)C, C, C, L.hs.Md, L.hs.Wp, L.hs.Wp, L.hs.Sz},
        {fc_desc_nhwc(reinterpret_cast<float *>(sc + L.dzt), L.ht.Sz, kFcHidden), reinterpret_cast<const float *>(ws + L.wu_dt),
         reinterpret_cast<float *>(sc + L.dxt), L.ht.Mdg * (int64_t)C, C, C, L.ht.Md, L.ht.Wp, L.ht.Wp, L.ht.Sz}};
    return fc_wino_jobs(jobs, 2, am_d, amx + kAmaxW, w16 && fc_w16_dgrad(k), B, kFcHidden / kFcChunk, k, stream);
  }
  if (which == 8) {
    if (!fc_wgrad_in_wino_domain(mode, k)) return GFLA_ERR_UNSUPPORTED;
    const WwJob jobs[2] = {
        {fc_desc_packed(ws + L.xs, B, L.nch_c, L.hs.Sx, 0), reinterpret_cast<float *>(sc + L.dzs), reinterpret_cast<float *>(sc + L.dwp),
         L.hs.Sz * kFcHidden, L.hs.lead, L.hs.Sx, L.hs.Ho, L.hs.Wo, L.hs.Wp},
        {fc_desc_packed(ws + L.xt, B, L.nch_c, L.ht.Sx, 0), reinterpret_cast<float *>(sc + L.dzt), reinterpret_cast<float *>(sc + L.dwp2),
         L.ht.Sz * kFcHidden, L.ht.lead, L.ht.Sx, L.ht.Ho, L.ht.Wo, L.ht.Wp}};
    if (fc_w16_wgrad(mode, k)) {
      const uint32_t *const ax[2] = {amx + kAmaxSrc, amx + kAmaxTgt}, *const az[2] = {amx + kAmaxZs, amx + kAmaxZt};
      return fc_wino16_wgrad_jobs(jobs, 2, L.cpad, B, k, ax, az, stream);
    }
    return fc_wino_wgrad_jobs(jobs, 2, L.cpad, B, k, stream);
  }
  if (fc_is_wino(mode)) {
    const PackedDesc X4 = fc_desc_packed(ws + (source ? L.xs : L.xt), B, L.nch_c, g.Sx, 0);
    const PackedDesc Z4 = fc_desc_nhwc(reinterpret_cast<float *>(sc + (source ? L.dzs : L.dzt)), g.Sz, kFcHidden);
    if (which < 2) {
      const WnConvJob job{X4, reinterpret_cast<const float *>(ws + (source ? L.wu_fs : L.wu_ft)),
                          reinterpret_cast<float *>(ws + (source ? L.gs : L.gt)), g.Mg * kFcHidden, kFcHidden, kFcHidden, g.Mv, g.Wo,
                          g.Wp, g.Sx};
      return fc_wino_jobs(&job, 1, am_f + (source ? 0 : 1), amx + kAmaxW, w16, B, L.nch_c, k, stream);
    }
    if (which < 4) {
      const WnConvJob job{Z4, reinterpret_cast<const float *>(ws + (source ? L.wu_ds : L.wu_dt)),
                          reinterpret_cast<float *>(sc + (source ? L.dxs : L.dxt)), g.Mdg * (int64_t)C, C, C, g.Md, g.Wp, g.Wp, g.Sz};
      return fc_wino_jobs(&job, 1, am_d + (source ? 0 : 1), amx + kAmaxW, w16 && fc_w16_dgrad(k), B, kFcHidden / kFcChunk, k, stream);
    }
    if (!fc_wgrad_in_wino_domain(mode, k))
      return fc_wgrad_f32(X4, Z4, g.lead, reinterpret_cast<float *>(sc + L.dwp), L.cpad, B, g.M, g.Wp, k, stream);
    return fc_wino_wgrad(X4, reinterpret_cast<float *>(sc + (source ? L.dzs : L.dzt)), g.Sz * kFcHidden, g.lead,
                         reinterpret_cast<float *>(sc + L.dwp), L.cpad, B, g.Ho, g.Wo, g.Wp, g.Sx, k, stream);
  }
  uint32_t *amax = reinterpret_cast<uint32_t *>(ws + L.amax);
  const uint32_t *a_x = mode ? amax + (source ? kAmaxSrc : kAmaxTgt) : nullptr, *a_w = mode ? amax + kAmaxW : nullptr;
  const uint32_t *a_z = mode ? amax + (source ? kAmaxZs : kAmaxZt) : nullptr;
  const int nch_h = kFcHidden / kFcChunk;
  const PackedDesc X = fc_desc_packed(ws + (source ? L.xs : L.xt), B, L.nch_c, g.Sx, mode);
  const PackedDesc Z = mode ? fc_desc_packed(sc + (source ? L.zs_pk : L.zt_pk), B, nch_h, g.Sz, mode)
                            : fc_desc_nhwc(reinterpret_cast<float *>(sc + (source ? L.dzs : L.dzt)), g.Sz, kFcHidden);
  if (which < 2) {
    const int64_t wsplit_f = fc_wpack_bytes(1, L.nch_c, k, mode) / fc_nsplit(mode);
    return fc_conv(X, ws + (source ? L.wf_s : L.wf_t), wsplit_f, reinterpret_cast<float *>(ws + (source ? L.gs : L.gt)),
                   g.Mg * kFcHidden, kFcHidden, kFcHidden, B, L.nch_c, g.Mv, g.Wo, g.Wp, k, mode, a_x, a_w, stream);
  }
  if (which < 4) {
    const int64_t wsplit_d = fc_wpack_bytes(L.nt_d, nch_h, k, mode) / fc_nsplit(mode);
    return fc_conv(Z, ws + (source ? L.wd_s : L.wd_t), wsplit_d, reinterpret_cast<float *>(sc + (source ? L.dxs : L.dxt)),
                   g.Mdg * (int64_t)C, C, C, B, nch_h, g.Md, g.Wp, g.Wp, k, mode, a_z, a_w, stream);
  }
  if (mode == 0)
    return fc_wgrad_f32(X, Z, g.lead, reinterpret_cast<float *>(sc + L.dwp), L.cpad, B, g.M, g.Wp, k, stream);
  return fc_wgrad(X, Z, g.lead, reinterpret_cast<float *>(sc + (source ? L.dw_s : L.dw_t)), L.cpad, B, g.M, g.Wp, k, mode,
                  stream);
}

int gfla_fc_tr_probe(const int16_t *image, int n_halves, const int32_t *offsets, int16_t *out, gfla_stream_t stream) {
  if (!image || !offsets || !out) return GFLA_ERR_NULL_POINTER;
  return fc_tr_probe(image, n_halves, offsets, out, static_cast<hipStream_t>(stream));
}
}
