/*
 * gfla_hip.h -- C ABI of libgfla_hip.so, the MI355X (gfx950) implementation of the
 * Global-Flow-Local-Attention feature-warping hot path.
 *
 * This is the drop-in boundary.  Each entry point replaces one `forward`/`backward` of the
 * reference's three pybind modules (paths relative to the reference checkout):
 *
 *   block_extractor_cuda      model/networks/block_extractor/block_extractor_cuda.cc:5-33
 *   local_attn_reshape_cuda   model/networks/local_attn_reshape/local_attn_reshape_cuda.cc:5-29
 *   resample2d_cuda           model/networks/resample2d_package/resample2d_cuda.cc:6-33
 *
 * plus one fused entry point pair for the softmax -> reshape -> multiply -> avg_pool tail of
 * ExtractorAttn.forward (model/networks/base_function.py:803,808-809), which the reference runs
 * as four separate torch/custom ops.
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes; no torch types; tensors are contiguous NCHW, exactly what
 *     the reference wrappers assert (block_extractor.py:9-10, local_attn_reshape.py:9,
 *     resample2d.py:10-11);
 *   - the caller owns every buffer (the reference allocates outputs in Function.forward,
 *     block_extractor.py:21, local_attn_reshape.py:18, resample2d.py:19); the library never
 *     allocates, never synchronises, never touches the host copy of the data;
 *   - `stream` is a hipStream_t (NULL = the null stream); kernels are enqueued asynchronously
 *     on it, the way the reference enqueues on the current torch stream
 *     (block_extractor_kernel.cu:197);
 *   - return 0 on success, a negative gfla_status otherwise (the reference returns 1 always and
 *     swallows launch errors, block_extractor_cuda.cc:11, block_extractor_kernel.cu:215);
 *   - suffix = storage type: _f32, _f64 (the two types the reference dispatches,
 *     AT_DISPATCH_FLOATING_TYPES) and _bf16 (new; forward AND backward, fp32 arithmetic inside, reductions over
 *     channels returned in float32).  bf16 buffers are raw uint16_t bit patterns.
 */
#ifndef GFLA_HIP_H_
#define GFLA_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gfla_status {
  GFLA_OK = 0,
  GFLA_ERR_NULL_POINTER = -1,   /* a required buffer is NULL */
  GFLA_ERR_BAD_SHAPE = -2,      /* non-positive dimension, kernel_size < 1, C != k*k ... */
  GFLA_ERR_UNSUPPORTED = -3,    /* shape is valid but outside what the kernels index */
  GFLA_ERR_LAUNCH = -4          /* hipGetLastError() reported a failure after the launch */
} gfla_status;

typedef void *gfla_stream_t; /* hipStream_t */

/* Bumped whenever an entry point is added or a signature changes.
 *   1: round 1 (the three ops + aggregate)   2: round 2 (fc_*, *_ws, bf16 backward, max_cosine, correctness_map)
 *   3: round 3 (gfla_path_count, process-global tuning, arithmetic mode 4)
 *   4: round 3 (gfla_fc_kernel_f32 which = 6 / 7; the scatter workspace also carries resample2d's tap records)
 *   5: round 4 (gfla_aggregate_bwd_supported, gfla_mask_blend_*; tuning keys 24-27; path id GFLA_PATH_BE_FWD_PIX)
 *   6: round 4 (gfla_convert_multi)
 *   7: round 5 (path ids 13-17, tuning keys 30-41: the big-plane kernels of csrc/tile_map.h; gfla_big_plane_geometry,
 *      gfla_xcd_swizzle)
 *   8: round 6 (arithmetic mode 5 of gfla_fc_*: Winograd domain with two-term f16 operands on the f16 matrix cores,
 *      csrc/fc_wino16.hip; path ids 18 / 19; tuning keys 43, 46, 49, 52) */
#define GFLA_ABI_VERSION 8
int gfla_abi_version(void);
const char *gfla_status_string(int status);

/* Tuning knobs (benchmarks/tests only; defaults are chosen per shape).  Returns the old value.
 *   key 0: block_extractor forward   0 auto (planes in LDS when they fit: flow rows of up to 64 pixels -> the
 *          wave-per-flow-row kernel, csrc/be_fwd_wrow.h; wider -> the lane-per-pixel kernel, csrc/be_fwd_pix.h),
 *          1 force the global-gather kernel, 2 force round 1's planes-in-LDS kernel (lane = four consecutive outputs),
 *          3 force the lane-per-pixel kernel, 4 force the wave-per-flow-row kernel
 *   key 1: channels per thread of the global kernels (0 auto)
 *   key 2: block_extractor backward  0 auto, 1 force global-atomics kernel
 *   key 3: aggregate fwd/bwd         0 auto, 1 force global kernels
 *   key 4: cap on G, the channel planes one workgroup keeps in LDS (0 auto)
 *   key 5: split, workgroups sharing one (b, channel group) (0 auto)
 *   key 6: resample2d fwd/bwd        0 auto, 1 force global kernels
 *   key 7: row windows for planes larger than the LDS budget   0 on, 1 off (use global kernels)
 *   key 10: LDS budget per workgroup in KB (0 = 64; up to 160)
 *   key 19: FC weight gradient in arithmetic mode 4   0 auto (Winograd domain), 1 direct, 3 Winograd for k = 5 only (round 3)
 *   key 20: timing ablations of the Winograd kernels -- only in `make PROBES=1` builds (results are garbage;
 *           tools/probe_wino.py); a default build ignores the key
 *   key 21: Winograd convolutions   1 single raw buffer, 2 one launch per half instead of both halves in one (also
 *           separates the two weight-gradient kernels of a layer again)
 *   key 23: resample2d d/d input1 LDS planes   0 fixed point + tap records (with scratch), 1 double planes (round 1)
 *   key 24: be_fwd_pix_kernel / be_fwd_wrow_kernel: threads per workgroup (0 auto; multiples of 64 up to 1024)
 *   key 25: be_fwd_pix_kernel: 1 = non-temporal output stores (A/B only: half the rate)
 *   key 27: (make PROBES=1 builds) timing ablations of the two round-4 block_extractor forward kernels
 *   key 29: Winograd-domain weight gradient: 1 = units of one tile row everywhere (round 3); 0 = whole tile rows per unit
 *           on maps whose tile rows fill at most half a unit (csrc/fc_wino.hip: MR)
 *   key 30: big-plane kernels (few planes, each beyond the LDS budget: csrc/tile_map.h)   0 auto, 1 never (round 1's
 *           row-window kernels), 2 always (tests drive them at small shapes)
 *   key 31 / 32: rows / columns of a tile (0 auto: 16 x 32)   key 35 / 36: the same for block_extractor's forward (0 auto:
 *           8 x 32)   key 34 / 37: channels per workgroup of the scatter / gather tiles (0 auto: 8 / 8-16)
 *   key 39: timing ablations of the tile kernels -- only in `make PROBES=1` builds (results are garbage); a default build
 *           ignores the key
 *   key 40: channels per pixel chunk of block_extractor's forward tiles (0 auto)
 *   key 41: 1 = block_extractor's backward tiles without the cross-lane fold of the patch rows (csrc/be_tile.h: BeLinks)
 *   key 43: arithmetic mode 5: 1 = the two-term f16 kernel also for the k = 3 data gradient (default: float32 Winograd kernel there)
 *   key 46: 1 = gfla_fc_backward_f32 scatters the gradient of the convolved source map with global float atomics (round 2)
 *           instead of the owner-computes kernel of round 6 (csrc/fc_sample.hip: fc_scatter_own_kernel)
 *   key 49: arithmetic mode 5: 1 = the k = 5 weight gradient on the float32 Winograd kernel (rounds 3-5) instead of the
 *           two-term f16 kernel (csrc/fc_wino.hip: fc_wino16_wgrad_kernel)
 *   key 52: arithmetic mode 5: 1 = Winograd-domain kernels for every convolution (the first half of round 6) instead of the
 *           hybrid dispatch (direct f16x2 kernels fed from the float32 maps for the k = 5 convolutions and all data gradients:
 *           csrc/fc_block.hip: fc_hyb, csrc/fc_conv_impl.h: SRC32)
 *   key 38: 1 = the first version of the gathers (taps read from global memory, no LDS window); key 33 = its channels per wave
 * (the other keys select experiments of individual kernels; see the tuning(...) calls in csrc/)                  */
int gfla_set_tuning(int key, int value);

/* Dispatch trace (tests): number of times a kernel path has been enqueued by this process, from any host thread
 * (autograd runs backward on its own worker threads).  -1 for an unknown id. */
enum gfla_path {
  GFLA_PATH_BE_BWD_LDS = 0,     /* block_extractor backward: planes-in-LDS kernel */
  GFLA_PATH_BE_BWD_GLOBAL = 1,  /* block_extractor backward: global-atomics kernel (tuning key 2) */
  GFLA_PATH_FC_FWD_MODE0 = 2,   /* gfla_fc_forward_f32 in arithmetic mode 0 .. 4 = ids 2 .. 6 */
  GFLA_PATH_FC_FWD_MODE1 = 3,
  GFLA_PATH_FC_FWD_MODE2 = 4,
  GFLA_PATH_FC_FWD_MODE3 = 5,
  GFLA_PATH_FC_FWD_MODE4 = 6,
  GFLA_PATH_FC_BWD_MODE0 = 7,   /* gfla_fc_backward_f32, ids 7 .. 11 */
  GFLA_PATH_FC_BWD_MODE1 = 8,
  GFLA_PATH_FC_BWD_MODE2 = 9,
  GFLA_PATH_FC_BWD_MODE3 = 10,
  GFLA_PATH_FC_BWD_MODE4 = 11,
  GFLA_PATH_BE_FWD_PIX = 12,   /* block_extractor forward: lane = flow pixel, padded planes in LDS (round 4) */
  GFLA_PATH_BE_FWD_GPIX = 13,  /* round 5, few planes beyond the LDS budget (csrc/tile_map.h): block_extractor forward */
  GFLA_PATH_BE_BWD_TILE = 14,  /*   block_extractor backward, flow-pixel tiles with bounding-box LDS windows */
  GFLA_PATH_RS_FWD_BIG = 15,   /*   resample2d forward */
  GFLA_PATH_RS_BWD1_TILE = 16, /*   resample2d d/d input1, tiles with bounding-box LDS windows */
  GFLA_PATH_RS_BWD2_BIG = 17,  /*   resample2d d/d input2 */
  GFLA_PATH_FC_FWD_MODE5 = 18, /* round 6: gfla_fc_forward_f32 / gfla_fc_backward_f32 in arithmetic mode 5 */
  GFLA_PATH_FC_BWD_MODE5 = 19,
  GFLA_PATH_COUNT = 20
};
int64_t gfla_path_count(int path);

/* Round 5, host logic of the big-plane tile kernels (csrc/tile_map.h), for tests -- no GPU needed.
 * gfla_big_plane_geometry: op 0 block_extractor forward, 1 block_extractor backward, 2 resample2d forward, 4 resample2d d/d input2,
 *   3 resample2d d/d input1; (H, W) the flow / output grid, (Hs, Ws) the source plane, span = taps per axis; out[10] = in the
 *   regime by default, tile rows, tile columns, tiles along x, tiles along y, threads per workgroup, channels per workgroup,
 *   channel groups, dynamic LDS bytes requested, workgroups.
 * gfla_xcd_swizzle: the block -> work item remap of those kernels (a bijection of [0, nwg)); -1 outside the range. */
int gfla_big_plane_geometry(int op, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H, int64_t W, int span,
                            int elem_size, int64_t *out);
int64_t gfla_xcd_swizzle(int64_t block, int64_t nwg);

/* ---- block_extractor ---------------------------------------------------------------------
 * forward : replaces block_extractor_cuda.forward(source, flow_field, output, kernel_size)
 *           (block_extractor_cuda.cc:5-12 -> block_extractor_kernel.cu:20-85)
 *   source (B,C,Hs,Ws)   flow (B,2,Hf,Wf): channel 0 = x, 1 = y displacement in source pixels
 *   out    (B,C,k*Hf,k*Wf), fully overwritten (need not be zeroed)
 * backward: replaces block_extractor_cuda.backward(source, flow_field, grad_output, grad_source,
 *           grad_flow_field, kernel_size) (block_extractor_cuda.cc:14-25 -> .cu:89-170)
 *   grad_source (B,C,Hs,Ws) is ACCUMULATED into and must arrive zeroed, as in the reference
 *   (block_extractor.py:35); grad_flow (B,2,Hf,Wf) likewise.  Either may be NULL to skip that
 *   gradient (the reference always computes both).                                             */
#define GFLA_DECL_BLOCK_EXTRACTOR(SFX, T)                                                          \
  int gfla_block_extractor_fwd_##SFX(const T *source, const T *flow, T *out, int64_t B, int64_t C, \
                                     int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf,               \
                                     int kernel_size, gfla_stream_t stream);
GFLA_DECL_BLOCK_EXTRACTOR(f32, float)
GFLA_DECL_BLOCK_EXTRACTOR(f64, double)
GFLA_DECL_BLOCK_EXTRACTOR(bf16, uint16_t)
#undef GFLA_DECL_BLOCK_EXTRACTOR

#define GFLA_DECL_BLOCK_EXTRACTOR_BWD(SFX, T)                                                      \
  int gfla_block_extractor_bwd_##SFX(const T *source, const T *flow, const T *grad_out,            \
                                     T *grad_source, T *grad_flow, int64_t B, int64_t C,           \
                                     int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf,               \
                                     int kernel_size, gfla_stream_t stream);
GFLA_DECL_BLOCK_EXTRACTOR_BWD(f32, float)
GFLA_DECL_BLOCK_EXTRACTOR_BWD(f64, double)
#undef GFLA_DECL_BLOCK_EXTRACTOR_BWD

/* ---- block_extractor, "unfold" layout (new; what the fused ExtractorAttn feeds its first FC layer) ----
 * Same samples as gfla_block_extractor_fwd, arranged as out (B, C*k*k, Hf, Wf) with channel index
 * c*k*k + i*k + j = tap (i,j) of source channel c, i.e. out_unfold[b, c*k*k+i*k+j, yf, xf] ==
 * out[b, c, yf*k+i, xf*k+j].  In this layout the stride-k convolution of ExtractorAttn
 * (base_function.py:800) is the batched GEMM  W(128, C*k*k) @ out_unfold[b](C*k*k, Hf*Wf)  and all
 * kernel traffic is coalesced along the pixel index.  layout = 1 stores (C*k*k, B, Hf, Wf) instead,
 * so the whole batch is ONE GEMM  W(128, C*k*k) @ out_unfold(C*k*k, B*Hf*Wf)  (and one each for the
 * data and weight gradients).  backward takes the gradient in the same layout;
 * grad_source / grad_flow accumulate (pass zeroed buffers), either may be NULL.
 * Only for kernel_size <= 5 and source planes that fit the LDS budget: gfla_unfold_supported() says
 * so (1/0); otherwise the entry points return GFLA_ERR_UNSUPPORTED and the caller uses the reference
 * layout.                                                                                          */
int gfla_unfold_supported(int64_t Hs, int64_t Ws, int kernel_size, int elem_size);
#define GFLA_DECL_UNFOLD_FWD(SFX, T)                                                               \
  int gfla_block_extractor_unfold_fwd_##SFX(const T *source, const T *flow, T *out_unfold,         \
                                            int64_t B, int64_t C, int64_t Hs, int64_t Ws,          \
                                            int64_t Hf, int64_t Wf, int kernel_size, int layout,   \
                                            gfla_stream_t stream);
GFLA_DECL_UNFOLD_FWD(f32, float)
GFLA_DECL_UNFOLD_FWD(f64, double)
GFLA_DECL_UNFOLD_FWD(bf16, uint16_t)
#undef GFLA_DECL_UNFOLD_FWD

#define GFLA_DECL_UNFOLD_BWD(SFX, T)                                                               \
  int gfla_block_extractor_unfold_bwd_##SFX(const T *source, const T *flow, const T *grad_unfold,  \
                                            T *grad_source, T *grad_flow, int64_t B, int64_t C,    \
                                            int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf,        \
                                            int kernel_size, int layout, gfla_stream_t stream);
GFLA_DECL_UNFOLD_BWD(f32, float)
GFLA_DECL_UNFOLD_BWD(f64, double)
#undef GFLA_DECL_UNFOLD_BWD

/* ---- local_attn_reshape ------------------------------------------------------------------
 * forward : replaces local_attn_reshape_cuda.forward(inputs, output, kernel_size)
 *           (local_attn_reshape_cuda.cc:5-11 -> local_attn_reshape_kernel.cu:20-61)
 *   in (B,k*k,H,W) -> out (B,1,k*H,k*W): out[b,0,ys*k+i,xs*k+j] = in[b,i*k+j,ys,xs]; bit-exact
 * backward: replaces local_attn_reshape_cuda.backward(inputs, grad_output, grad_inputs, k)
 *           (local_attn_reshape_cuda.cc:13-21 -> .cu:65-108); the inverse permutation.
 *   grad_in is fully OVERWRITTEN (the reference atomically adds into a zeroed buffer; the map
 *   is a bijection so the result is identical).                                                */
#define GFLA_DECL_RESHAPE(SFX, T)                                                                  \
  int gfla_local_attn_reshape_fwd_##SFX(const T *in, T *out, int64_t B, int64_t H, int64_t W,      \
                                        int kernel_size, gfla_stream_t stream);                    \
  int gfla_local_attn_reshape_bwd_##SFX(const T *grad_out, T *grad_in, int64_t B, int64_t H,       \
                                        int64_t W, int kernel_size, gfla_stream_t stream);
GFLA_DECL_RESHAPE(f32, float)
GFLA_DECL_RESHAPE(f64, double)
GFLA_DECL_RESHAPE(bf16, uint16_t)
#undef GFLA_DECL_RESHAPE

/* ---- resample2d --------------------------------------------------------------------------
 * forward : replaces resample2d_cuda.forward(input1, input2, output, kernel_size, dilation)
 *           (resample2d_cuda.cc:6-14 -> resample2d_kernel.cu:20-95)
 *   in1 (B,C,Hi,Wi)   in2 (B,3,H,W) = (dx, dy, sigma)   out (B,C,H,W), fully overwritten
 * backward: replaces resample2d_cuda.backward(input1, input2, gradOutput, gradInput1,
 *           gradInput2, kernel_size, dilation) (resample2d_cuda.cc:16-26 -> .cu:98-202, :204-330)
 *   grad_in1 (B,C,Hi,Wi) is ACCUMULATED into, must arrive zeroed (resample2d.py:32);
 *   grad_in2 (B,3,H,W) must arrive zeroed as well.  Either may be NULL.
 *   trunc_compat is a flag word.  Bit 0 set reproduces the reference's `xf - int(xf)` in the input1 gradient
 *   (resample2d_kernel.cu:137-138); clear uses floor, which is the true gradient of the forward.  Bit 1 set
 *   (GFLA_RESAMPLE_OVERWRITE_IN1): grad_in1 may arrive UNINITIALISED and is overwritten -- by plain stores where every
 *   element has exactly one writer, after an internal zero fill where the kernels have to accumulate with atomics.       */
#define GFLA_RESAMPLE_TRUNC_COMPAT 1
#define GFLA_RESAMPLE_OVERWRITE_IN1 2
#define GFLA_DECL_RESAMPLE_FWD(SFX, T)                                                             \
  int gfla_resample2d_fwd_##SFX(const T *in1, const T *in2, T *out, int64_t B, int64_t C,          \
                                int64_t Hi, int64_t Wi, int64_t H, int64_t W, int kernel_size,     \
                                int dilation, gfla_stream_t stream);
GFLA_DECL_RESAMPLE_FWD(f32, float)
GFLA_DECL_RESAMPLE_FWD(f64, double)
GFLA_DECL_RESAMPLE_FWD(bf16, uint16_t)
#undef GFLA_DECL_RESAMPLE_FWD

#define GFLA_DECL_RESAMPLE_BWD(SFX, T)                                                             \
  int gfla_resample2d_bwd_##SFX(const T *in1, const T *in2, const T *grad_out, T *grad_in1,        \
                                T *grad_in2, int64_t B, int64_t C, int64_t Hi, int64_t Wi,         \
                                int64_t H, int64_t W, int kernel_size, int dilation,               \
                                int trunc_compat, gfla_stream_t stream);
GFLA_DECL_RESAMPLE_BWD(f32, float)
GFLA_DECL_RESAMPLE_BWD(f64, double)
#undef GFLA_DECL_RESAMPLE_BWD

/* ---- local-attention softmax + aggregate (fused) -------------------------------------------
 * Replaces, in ExtractorAttn.forward (base_function.py:804-810), the chain
 *   Softmax(dim=1) (:803) -> LocalAttnReshape (:808) -> attn * block_source -> avg_pool2d (:809)
 * together with the block_source extraction that feeds the product (:805), without ever
 * materialising the (B,C,kH,kW) product:
 *   out[b,c,y,x] = (1/k^2) * sum_{i,j} a[b,i*k+j,y,x] * bilinear(source[b,c], tap_ij(flow[b,:,y,x]))
 *   a = softmax over the k*k channel of `logits` if apply_softmax != 0, else `logits` as given
 *   (ExtractorAttn swaps the softmax for the plain nonlinearity when built with softmax=None,
 *   base_function.py:794).
 *   source (B,C,Hs,Ws)  flow (B,2,H,W)  logits (B,k*k,H,W)  out (B,C,H,W)
 *   attn_out (B,k*k,H,W) optional (NULL to skip): the post-softmax weights, i.e. what
 *   hook_attn_param returns as attn_param_ (base_function.py:815,818) and what backward needs.
 * backward: given grad_out (B,C,H,W) and the saved `attn` (post-softmax), produces
 *   grad_source (B,C,Hs,Ws; accumulated, must arrive zeroed), grad_flow (B,2,H,W; zeroed) and
 *   grad_logits (B,k*k,H,W; zeroed) -- the gradient w.r.t. the pre-softmax logits when
 *   apply_softmax != 0, else w.r.t. the weights.  Any of the three may be NULL.                 */
#define GFLA_DECL_AGGREGATE_FWD(SFX, T)                                                            \
  int gfla_local_attn_aggregate_fwd_##SFX(const T *source, const T *flow, const T *logits, T *out, \
                                          T *attn_out, int64_t B, int64_t C, int64_t Hs,           \
                                          int64_t Ws, int64_t H, int64_t W, int kernel_size,       \
                                          int apply_softmax, gfla_stream_t stream);
GFLA_DECL_AGGREGATE_FWD(f32, float)
GFLA_DECL_AGGREGATE_FWD(f64, double)
GFLA_DECL_AGGREGATE_FWD(bf16, uint16_t)
#undef GFLA_DECL_AGGREGATE_FWD

/* Forward with scratch (f32 / bf16 storage): softmax, tap geometry and the (k+1)x(k+1) patch coefficients of every
 * flow pixel are computed ONCE (a table of (k+1)(k+2) floats + one word per pixel in `workspace`,
 * gfla_aggregate_fwd_workspace_bytes(B, H, W, k) bytes, 16-byte aligned) instead of once per channel group, and the
 * aggregation reads patch rows from LDS with paired 64-bit reads.  workspace == NULL, even k or Ws < k + 1: the
 * plain entry point's kernels.  Same results up to f32 summation order (the coefficient of a patch word is summed
 * over its taps before it meets the source value).                                                            */
int64_t gfla_aggregate_fwd_workspace_bytes(int64_t B, int64_t H, int64_t W, int kernel_size);
/* 1 when gfla_local_attn_aggregate_bwd_<storage> takes Hs x Ws source planes (elem_size 2 / 4 / 8 bytes); bf16 storage
 * is limited to planes that fit the LDS accumulators (round 4; replaces a constant duplicated on the host side) */
int gfla_aggregate_bwd_supported(int64_t Hs, int64_t Ws, int elem_size);
/* host-side launch geometry of that path (no GPU needed; tests): out[9] = channels per chunk, channels per range,
 * ranges, tile groups, threads, LDS row-pair pitch in words, tile width, tiles per sample, dynamic LDS bytes */
int gfla_aggregate_fwd_geometry(int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H, int64_t W, int kernel_size,
                                int64_t *out);
int gfla_local_attn_aggregate_fwd_ws_f32(const float *source, const float *flow, const float *logits, float *out,
                                         float *attn_out, void *workspace, int64_t B, int64_t C, int64_t Hs,
                                         int64_t Ws, int64_t H, int64_t W, int kernel_size, int apply_softmax,
                                         gfla_stream_t stream);
int gfla_local_attn_aggregate_fwd_ws_bf16(const uint16_t *source, const uint16_t *flow, const uint16_t *logits,
                                          uint16_t *out, uint16_t *attn_out, void *workspace, int64_t B, int64_t C,
                                          int64_t Hs, int64_t Ws, int64_t H, int64_t W, int kernel_size,
                                          int apply_softmax, gfla_stream_t stream);

#define GFLA_DECL_AGGREGATE_BWD(SFX, T)                                                            \
  int gfla_local_attn_aggregate_bwd_##SFX(const T *source, const T *flow, const T *attn,           \
                                          const T *grad_out, T *grad_source, T *grad_flow,         \
                                          T *grad_logits, int64_t B, int64_t C, int64_t Hs,        \
                                          int64_t Ws, int64_t H, int64_t W, int kernel_size,       \
                                          int apply_softmax, gfla_stream_t stream);
GFLA_DECL_AGGREGATE_BWD(f32, float)
GFLA_DECL_AGGREGATE_BWD(f64, double)
#undef GFLA_DECL_AGGREGATE_BWD


/* bf16 storage for the backward entry points (mixed-precision features): feature-map gradients (grad_source,
 * grad_in1) are bf16 and accumulated into like their f32 counterparts; every gradient that is a REDUCTION over channels
 * -- grad_flow (B,2,H,W), grad_logits (B,k*k,H,W), grad_in2 (B,3,H,W) -- is float32 (accumulated across workgroups with
 * float atomics; 8 mantissa bits would not survive C*k*k terms).  Planes must fit LDS (the attention-layer shapes do);
 * anything else returns GFLA_ERR_UNSUPPORTED.  Accumulation is f32 / f64-in-LDS exactly as in the f32 entry points. */
int gfla_block_extractor_bwd_bf16(const uint16_t *source, const uint16_t *flow, const uint16_t *grad_out,
                                  uint16_t *grad_source, float *grad_flow, int64_t B, int64_t C, int64_t Hs, int64_t Ws,
                                  int64_t Hf, int64_t Wf, int kernel_size, gfla_stream_t stream);
int gfla_block_extractor_unfold_bwd_bf16(const uint16_t *source, const uint16_t *flow, const uint16_t *grad_unfold,
                                         uint16_t *grad_source, float *grad_flow, int64_t B, int64_t C, int64_t Hs,
                                         int64_t Ws, int64_t Hf, int64_t Wf, int kernel_size, int layout,
                                         gfla_stream_t stream);
int gfla_local_attn_aggregate_bwd_bf16(const uint16_t *source, const uint16_t *flow, const uint16_t *attn,
                                       const uint16_t *grad_out, uint16_t *grad_source, float *grad_flow,
                                       float *grad_logits, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H,
                                       int64_t W, int kernel_size, int apply_softmax, gfla_stream_t stream);
int gfla_local_attn_source_bwd_bf16(const uint16_t *source, const uint16_t *flow, const uint16_t *grad_unfold,
                                    const uint16_t *attn, const uint16_t *grad_out, uint16_t *grad_source,
                                    float *grad_flow, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H,
                                    int64_t W, int kernel_size, int layout, gfla_stream_t stream);
int gfla_resample2d_bwd_bf16(const uint16_t *in1, const uint16_t *in2, const uint16_t *grad_out, uint16_t *grad_in1,
                             float *grad_in2, int64_t B, int64_t C, int64_t Hi, int64_t Wi, int64_t H, int64_t W,
                             int kernel_size, int dilation, int trunc_compat, gfla_stream_t stream);

/* ---- scatters as block-sparse products on the matrix cores (csrc/patch_mfma.hip) -----------------------------
 * The two backward passes that scatter into a feature plane -- the aggregation's d/d source and resample2d's
 * d/d input1 (resample2d_kernel.cu:98-202; replaces its atomicAdd scatter) -- spread each flow pixel's gradient over
 * a dense patch with channel-independent weights: a sparse x dense product.  The *_ws entry points take scratch for
 * the per-pixel patch table (gfla_scatter_workspace_bytes(B, H, W, entries): entries = (k+1)^2 for the aggregation,
 * k*k for resample2d; 256-byte aligned; may be NULL = the LDS-atomic kernels of the plain entry points) and multiply
 * the tiled sparse matrix on v_mfma_f32_32x32x2_f32 with output-stationary accumulators: no atomics, fixed summation
 * order.  With a workspace the aggregation's d/d flow comes out of the d/d logits pass.  Same accumulate-into
 * contract as the plain entry points; unsupported shapes (plane rows wider than 192, dilation != 1) fall back.   */
int64_t gfla_scatter_workspace_bytes(int64_t B, int64_t H, int64_t W, int patch_entries);
int gfla_local_attn_aggregate_bwd_ws_f32(const float *source, const float *flow, const float *attn,
                                         const float *grad_out, float *grad_source, float *grad_flow,
                                         float *grad_logits, void *workspace, int64_t B, int64_t C, int64_t Hs,
                                         int64_t Ws, int64_t H, int64_t W, int kernel_size, int apply_softmax,
                                         gfla_stream_t stream);
int gfla_resample2d_bwd_ws_f32(const float *in1, const float *in2, const float *grad_out, float *grad_in1,
                               float *grad_in2, void *workspace, int64_t B, int64_t C, int64_t Hi, int64_t Wi,
                               int64_t H, int64_t W, int kernel_size, int dilation, int trunc_compat,
                               gfla_stream_t stream);

/* ---- everything in ExtractorAttn that flows back into block_source(source, flow), in ONE pass ----------
 * block_source receives two gradient streams (base_function.py:805-809): through the first FC layer
 * (grad_unfold, in the unfold layout above; may be NULL) and through the attention-weighted aggregation
 * (attn (B,k*k,H,W) post-softmax and grad_out (B,C,H,W): attn[b,ij,p]*grad_out[b,c,p]/k^2, never
 * materialised; both may be NULL together).  Their sum is scattered into grad_source / grad_flow
 * (accumulated; pass zeroed buffers; either may be NULL).  The reference runs the block_extractor
 * backward kernel twice for this (once per stream, plus the reshape/multiply/avg-pool backward).      */
#define GFLA_DECL_SOURCE_BWD(SFX, T)                                                               \
  int gfla_local_attn_source_bwd_##SFX(const T *source, const T *flow, const T *grad_unfold,       \
                                       const T *attn, const T *grad_out, T *grad_source,           \
                                       T *grad_flow, int64_t B, int64_t C, int64_t Hs, int64_t Ws, \
                                       int64_t H, int64_t W, int kernel_size, int layout,          \
                                       gfla_stream_t stream);
GFLA_DECL_SOURCE_BWD(f32, float)
GFLA_DECL_SOURCE_BWD(f64, double)
#undef GFLA_DECL_SOURCE_BWD

/* ---- tail of ExtractorAttn's fully_connect_layer: nonlinearity + 1x1 convolution (base_function.py:799-803)
 *   logits[b,q,p] = b1[q] + sum_o w1[q,o] * lrelu(hs[b,o,p] + ht[b,o,p] + b0[o], slope)
 * hs: the source half of the first FC convolution, addressed as hs[b*hs_sb + o*hs_so + p] (so both the
 * GEMM's (Hc,B,HW) and the plain (B,Hc,HW) layout are accepted); ht: the target half, (B,Hc,HW) contiguous;
 * b0 (Hc) and b1 (KK) may be NULL; w1 (KK,Hc); logits (B,KK,HW) is overwritten.  KK in {1,4,9,16,25};
 * slope = 0 gives ReLU.  backward: given g_logits (B,KK,HW) overwrites g_hs (hs's addressing), g_ht
 * (B,Hc,HW; may be NULL) with the gradient w.r.t. hs/ht (identical values, two layouts) and act (B,Hc,HW;
 * may be NULL) with the activations (for dW1 = sum_b g_logits_b act_b^T, left to the caller's GEMM) and
 * bias_partials (B * ceil(HW / 64), Hc + KK; may be NULL) with one row per workgroup: its sums of the
 * hidden gradient (-> d b0) followed by its sums of g_logits (-> d b1); the caller adds the rows up.    */
#define GFLA_DECL_FC_TAIL(SFX, T)                                                                    \
  int gfla_fc_tail_fwd_##SFX(const T *hs, int64_t hs_sb, int64_t hs_so, const T *ht, const T *b0,   \
                             const T *w1, const T *b1, T *logits, int64_t B, int64_t Hc, int64_t HW, \
                             int KK, double slope, gfla_stream_t stream);                           \
  int gfla_fc_tail_bwd_##SFX(const T *hs, int64_t hs_sb, int64_t hs_so, const T *ht, const T *b0,   \
                             const T *w1, const T *g_logits, T *g_hs, T *g_ht, T *act,              \
                             T *bias_partials, int64_t B, int64_t Hc, int64_t HW, int KK,           \
                             double slope, gfla_stream_t stream);
GFLA_DECL_FC_TAIL(f32, float)
GFLA_DECL_FC_TAIL(f64, double)
#undef GFLA_DECL_FC_TAIL

/* ---- first FC layer of ExtractorAttn on the matrix cores (csrc/fc_gemm.hip, fc_sample.hip, fc_block.hip) -----
 * Replaces, in ExtractorAttn.forward (model/networks/base_function.py:799-807),
 *   block_source = extractor(source, flow); block_target = extractor(target, 0)
 *   logits = Conv2d(128, k*k, 1)(nonlinearity(Conv2d(2C, 128, k, stride k)(cat(block_target, block_source))))
 * i.e. the two BlockExtractor launches, the cat and both convolutions of fully_connect_layer (its softmax stays
 * with gfla_local_attn_aggregate_*).  Neither block tensor is built:  FC0(source half) = bilinear sample, at
 * p + flow(p), of conv_kxk(replicate-extended source, W[:, C:]) -- exact, because all k*k taps of a position share
 * one fractional offset -- and FC0(target half) = conv_kxk(replicate-padded target, W[:, :C]); the convolutions
 * are implicit GEMMs on MFMA, hand-written for gfx950.
 *   source, target (B,C,H,W)  flow (B,2,H,W)  w0 = conv0.weight (128, 2C, k, k)  b0 (128) or NULL
 *   w1 = conv1.weight (k*k, 128)  b1 (k*k) or NULL  logits (B,k*k,H,W), overwritten.  fp32; k in {3, 5}.
 *   slope: LeakyReLU negative slope (0 = ReLU).
 *   mode: arithmetic of the contraction -- 4: float32 throughout in the Winograd domain (F(2x2,5x5) / F(4x4,3x3) on the
 *   points {0, 1, -1, 2, -1/2, inf}: 36 multiplies per 6x6 tile instead of 100 / 144, v_mfma_f32_16x16x4_f32; the
 *   host-side default, csrc/fc_wino.hip); 0: v_mfma_f32_32x32x2_f32, direct convolution (a k-ordered f32 fma chain per
 *   output); 3: operands as three f16 terms, six cross products, f32 accumulate (every product term above 2^-32 kept:
 *   f32-grade); 2: two f16 terms, three products (2^-21 per product); 1: one f16 term (exact for bf16 values).
 *   workspace: gfla_fc_workspace_bytes(..., which = 0) bytes, 256-byte aligned; forward fills it and backward
 *   reads it (packed inputs, convolved source map, hidden activations).  scratch: (..., which = 1) bytes.
 * backward: given grad_logits, overwrites whichever of grad_source, grad_target (B,C,H,W), grad_flow (B,2,H,W),
 *   grad_w0, grad_b0, grad_w1, grad_b1 is not NULL (no buffer needs zeroing).  flags: GFLA_FC_ACCUMULATE_SOURCE /
 *   GFLA_FC_ACCUMULATE_FLOW add into grad_source / grad_flow instead (they then hold the gradient another path of the
 *   same block already produced, e.g. the aggregation's).
 * gfla_fc_supported(C, H, W, k, mode): 1 if the shape is handled (the convolution's input tile must fit LDS).  */
#define GFLA_FC_ACCUMULATE_SOURCE 1
#define GFLA_FC_ACCUMULATE_FLOW 2
int gfla_fc_supported(int64_t C, int64_t H, int64_t W, int kernel_size, int mode);
int64_t gfla_fc_workspace_bytes(int64_t B, int64_t C, int64_t H, int64_t W, int kernel_size, int mode,
                                int which);
int gfla_fc_forward_f32(const float *source, const float *target, const float *flow, const float *w0,
                        const float *b0, const float *w1, const float *b1, void *workspace, float *logits,
                        int64_t B, int64_t C, int64_t H, int64_t W, int kernel_size, double slope, int mode,
                        gfla_stream_t stream);
int gfla_fc_backward_f32(void *workspace, const float *flow, const float *w1, const float *grad_logits,
                         void *scratch, float *grad_source, float *grad_target, float *grad_flow, float *grad_w0,
                         float *grad_b0, float *grad_w1, float *grad_b1, int64_t B, int64_t C, int64_t H,
                         int64_t W, int kernel_size, double slope, int mode, int flags, gfla_stream_t stream);
/* Pieces of the above for the parity tests.  gfla_fc_geometry: out[0..12] = Hp, Wp, Ho, Wo, pad_top, pad_left, M,
 * Md, lead, Sx, Sz, Mg, Mdg of one half (is_source: the source half is extended by k-1, the target half padded by
 * k/2).  gfla_fc_conv_fwd: out (B, Mg, 128) = the convolved map of one half, row yo*Wo + xo.  gfla_fc_conv_bwd:
 * from z (B, Sz, 128), the gradient of that map in "Z layout" (row lead + yo*Wp + xo, zero elsewhere), grad_x
 * (B,C,H,W) and grad_w0 (128,2C,k,k; the other half zero); needs the workspace of gfla_fc_conv_fwd.
 * gfla_fc_tr_probe: raw result of ds_read_b64_tr_b16 for an LDS image and 64 per-lane byte offsets.            */
int gfla_fc_geometry(int64_t H, int64_t W, int kernel_size, int is_source, int64_t *out);
int gfla_fc_conv_fwd_f32(const float *x, const float *w0, int is_source, void *workspace, float *out, int64_t B,
                         int64_t C, int64_t H, int64_t W, int kernel_size, int mode, gfla_stream_t stream);
int gfla_fc_conv_bwd_f32(const float *z, int is_source, void *workspace, void *scratch, float *grad_x,
                         float *grad_w0, int64_t B, int64_t C, int64_t H, int64_t W, int kernel_size, int mode,
                         gfla_stream_t stream);
/* gfla_fc_kernel: ONE internal kernel of the path on the state a forward + backward of the same shape left in
 * workspace / scratch, for per-kernel timing (bench.py, profiles).  which: 0 / 1 convolution forward of the source /
 * target half, 2 / 3 data-gradient convolution, 4 / 5 weight gradient; modes 4 / 5 only: 6 / 7 = the forward / data-gradient
 * convolutions of both halves in ONE launch, 8 = both Winograd-domain weight gradients in one launch, the way
 * gfla_fc_forward / gfla_fc_backward issue them.                                                                  */
int gfla_fc_kernel_f32(int which, void *workspace, void *scratch, int64_t B, int64_t C, int64_t H, int64_t W,
                       int kernel_size, int mode, gfla_stream_t stream);
int gfla_fc_tr_probe(const int16_t *image, int n_halves, const int32_t *offsets, int16_t *out,
                     gfla_stream_t stream);
/* tools only: a device buffer (workgroups x 8 waves x 6 uint64) that the timing instantiation of the Winograd
 * convolution kernel (arithmetic mode 4, tuning key 20 = 16) fills with per-wave phase times in shader cycles;
 * NULL switches it off.                                                                                          */
int gfla_fc_wino_debug_buffer(void *buffer);

/* ---- gradient of the replicate padding in front of the target half of ExtractorAttn's first FC layer ---
 * block_target = extractor(target, zero flow) (base_function.py:806) is the replicate-padded unfold of
 * target; its half of the FC layer runs as a stride-1 convolution of the padded target.  Given the gradient
 * w.r.t. the padded tensor (planes, H+top+bottom, W+left+right), overwrites grad_in (planes, H, W): border
 * strips are folded onto the edge pixels (a gather, no atomics).                                          */
int gfla_replicate_pad_bwd_f32(const float *grad_padded, float *grad_in, int64_t planes, int64_t H,
                               int64_t W, int pad_left, int pad_right, int pad_top, int pad_bottom,
                               gfla_stream_t stream);
int gfla_replicate_pad_bwd_f64(const double *grad_padded, double *grad_in, int64_t planes, int64_t H,
                               int64_t W, int pad_left, int pad_right, int pad_top, int pad_bottom,
                               gfla_stream_t stream);

/* ---- best-match cosine similarity of the sampling-correctness loss ----------------
 * Replaces, in PerceptualCorrectness.calculate_loss (external_function.py:255-268),
 *   source_norm = source / (||source||_c + eps); target_norm = target / (||target||_c + eps)
 *   correction = bmm(source_norm^T, target_norm)          [B, Ns, Nt], materialised by the reference
 *   correction_max, max_indices = max(correction, dim=1)
 * source (B,C,Ns), target (B,C,Nt): contiguous views of the NCHW feature maps.  out_max (B,Nt) and
 * out_idx (B,Nt; int32 row of the maximum, may be NULL) are fully overwritten.  `workspace` is caller-
 * provided scratch of gfla_max_cosine_workspace_bytes(B, Ns, Nt) bytes, 16-byte aligned (the inverse
 * norms and the packed running maxima live there).  fp32 only (exact-f32 MFMA); the [Ns,Nt] matrix is
 * never written.  NaN similarities are ignored by the max (the reference's torch.max propagates them). */
int64_t gfla_max_cosine_workspace_bytes(int64_t B, int64_t Ns, int64_t Nt);
int gfla_max_cosine_fwd_f32(const float *source, const float *target, void *workspace, float *out_max,
                            int32_t *out_idx, int64_t B, int64_t C, int64_t Ns, int64_t Nt, double eps,
                            gfla_stream_t stream);

/* ---- per-position map of the sampling-correctness loss (external_function.py:275-276) -----------------
 *   loss_map[b,n] = exp(-cosine_similarity(warped[b,:,n], target[b,:,n]) / (best[b,n] + eps))
 * warped, target (B,C,N) contiguous; best (B,N) = gfla_max_cosine_fwd's out_max; eps_cos = the eps of
 * F.cosine_similarity (1e-8), eps = the loss's own (1e-8).  forward overwrites loss_map (B,N) and
 * stats (B,N,3) = (cos, |warped|, |target|), which backward reads back.  backward overwrites any of
 * grad_warped (B,C,N), grad_target (B,C,N), grad_best (B,N) that is not NULL, given grad_map (B,N).    */
int gfla_correctness_map_fwd_f32(const float *warped, const float *target, const float *best,
                                 float *loss_map, float *stats, int64_t B, int64_t C, int64_t N,
                                 double eps_cos, double eps, gfla_stream_t stream);
int gfla_correctness_map_bwd_f32(const float *warped, const float *target, const float *best,
                                 const float *stats, const float *loss_map, const float *grad_map,
                                 float *grad_warped, float *grad_target, float *grad_best, int64_t B,
                                 int64_t C, int64_t N, double eps_cos, double eps, gfla_stream_t stream);

/* Storage-type conversion of up to four contiguous tensors in one launch (unused jobs: n = 0).  to_bf16 = 0: bfloat16 ->
 * float32 (exact); 1: float32 -> bfloat16, round to nearest even.  The bf16 feature path of ExtractorAttn widens three
 * operands and narrows three gradients per call; at the face model's batch every one of them was a launch of its own. */
int gfla_convert_multi(const void *src0, void *dst0, int64_t n0, const void *src1, void *dst1, int64_t n1, const void *src2,
                       void *dst2, int64_t n2, const void *src3, void *dst3, int64_t n3, int to_bf16, gfla_stream_t stream);

/* ---- mask blend of the face model's attention pair (generator.py:496-499; csrc/mask_blend.hip, round 4) ----
 *   y = (out*(1-mask_p) + attn_p*mask_p) + (out*(1-mask_r) + attn_r*mask_r)
 * out, attn_p, attn_r, y: (B,C,HW) contiguous; mask_p, mask_r: (B,1,HW).  Forward: the op-by-op result bit for bit (every
 * intermediate rounded to the storage type).  Backward: any gradient pointer may be NULL; g_mask_p / g_mask_r are FLOAT32
 * (B,1,HW), ACCUMULATED into (pass zeroed buffers) whatever the storage type. */
int gfla_mask_blend_fwd_f32(const float *out, const float *attn_p, const float *attn_r, const float *mask_p,
                            const float *mask_r, float *y, int64_t B, int64_t C, int64_t HW, gfla_stream_t stream);
int gfla_mask_blend_fwd_bf16(const uint16_t *out, const uint16_t *attn_p, const uint16_t *attn_r, const uint16_t *mask_p,
                             const uint16_t *mask_r, uint16_t *y, int64_t B, int64_t C, int64_t HW, gfla_stream_t stream);
int gfla_mask_blend_bwd_f32(const float *out, const float *attn_p, const float *attn_r, const float *mask_p,
                            const float *mask_r, const float *grad_y, float *g_out, float *g_attn_p, float *g_attn_r,
                            float *g_mask_p, float *g_mask_r, int64_t B, int64_t C, int64_t HW, gfla_stream_t stream);
int gfla_mask_blend_bwd_bf16(const uint16_t *out, const uint16_t *attn_p, const uint16_t *attn_r, const uint16_t *mask_p,
                             const uint16_t *mask_r, const uint16_t *grad_y, uint16_t *g_out, uint16_t *g_attn_p,
                             uint16_t *g_attn_r, float *g_mask_p, float *g_mask_r, int64_t B, int64_t C, int64_t HW,
                             gfla_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GFLA_HIP_H_ */
