// Per-position map of the sampling-correctness loss, gfx950.
//
// Reference: PerceptualCorrectness.calculate_loss, external_function.py:275-276 --
//   correction_sample = F.cosine_similarity(input_sample, target_all)        [b, N]   (over channels)
//   loss_map = exp(-correction_sample / (correction_max + eps))
// torch evaluates this as ~10 elementwise/reduction kernels forward and as many backward, each a pass
// over the (B,C,N) features.  Here: one pass forward (three channel sums per position, exp) and one
// pass backward (the per-position factors once, then one fused multiply-add per feature element).
// All of it is HBM-bound: 2 reads of (B,C,N) forward; 2 reads + 1..2 writes backward.
//
// cosine_similarity semantics are those of the torch this package runs on (2.x): each norm is clamped from
// below by eps_cos (1e-8), and the clamp is not differentiated:
//   cos = <x,t> / (max(|x|,e) max(|t|,e));   dcos/dx_c = t_c / (nx' nt') - cos x_c / (|x| nx'),  0 at |x| = 0.
#include "gfla_common.h"

namespace gfla {

constexpr int kSlices = 4;  // channel slices per position (256 threads = 64 positions x 4 slices)

__global__ __launch_bounds__(256) void correctness_map_fwd_kernel(const float *__restrict__ x,
                                                                 const float *__restrict__ t,
                                                                 const float *__restrict__ best,
                                                                 float *__restrict__ loss_map,
                                                                 float *__restrict__ stats, int C, int N,
                                                                 float eps_cos, float eps) {
  __shared__ float part[3][kSlices][64];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const int64_t b = blockIdx.y;
  float sxt = 0.f, sxx = 0.f, stt = 0.f;
  if (n < N) {
    const int64_t base = b * C * (int64_t)N + n;
#pragma unroll 4
    for (int c = slice; c < C; c += kSlices) {
      const float xv = x[base + (int64_t)c * N], tv = t[base + (int64_t)c * N];
      sxt = fmaf(xv, tv, sxt);
      sxx = fmaf(xv, xv, sxx);
      stt = fmaf(tv, tv, stt);
    }
  }
  part[0][slice][lane] = sxt;
  part[1][slice][lane] = sxx;
  part[2][slice][lane] = stt;
  __syncthreads();
  if (slice == 0 && n < N) {
    sxt = (part[0][0][lane] + part[0][1][lane]) + (part[0][2][lane] + part[0][3][lane]);
    sxx = (part[1][0][lane] + part[1][1][lane]) + (part[1][2][lane] + part[1][3][lane]);
    stt = (part[2][0][lane] + part[2][1][lane]) + (part[2][2][lane] + part[2][3][lane]);
    const float nx = sqrtf(sxx), nt = sqrtf(stt);
    const float cosv = sxt / (fmaxf(nx, eps_cos) * fmaxf(nt, eps_cos));
    const int64_t i = b * N + n;
    loss_map[i] = expf(-cosv / (best[i] + eps));
    stats[3 * i + 0] = cosv;
    stats[3 * i + 1] = nx;
    stats[3 * i + 2] = nt;
  }
}

template <bool GX, bool GT>
__global__ __launch_bounds__(256) void correctness_map_bwd_kernel(const float *__restrict__ x,
                                                                 const float *__restrict__ t,
                                                                 const float *__restrict__ best,
                                                                 const float *__restrict__ stats,
                                                                 const float *__restrict__ loss_map,
                                                                 const float *__restrict__ grad_map,
                                                                 float *__restrict__ gx, float *__restrict__ gt,
                                                                 float *__restrict__ gbest, int C, int N,
                                                                 float eps_cos, float eps) {
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const int64_t b = blockIdx.y;
  if (n >= N) return;
  const int64_t i = b * N + n;
  const float cosv = stats[3 * i], nx = stats[3 * i + 1], nt = stats[3 * i + 2];
  const float den = best[i] + eps;
  const float g = grad_map[i] * loss_map[i];
  const float gcos = -g / den;                                  // d loss_map / d cos, times upstream
  if (gbest && slice == 0) gbest[i] = g * cosv / (den * den);
  if (!GX && !GT) return;
  const float nxc = fmaxf(nx, eps_cos), ntc = fmaxf(nt, eps_cos);
  const float cross = gcos / (nxc * ntc);
  // d/dx of max(|x|, eps) is x/|x| above eps and 0 below it (torch's clamp_min): no self term for a sub-eps norm
  const float selfx = nx > eps_cos ? gcos * cosv / (nx * nxc) : 0.f;
  const float selft = nt > eps_cos ? gcos * cosv / (nt * ntc) : 0.f;
  const int64_t base = b * C * (int64_t)N + n;
#pragma unroll 4
  for (int c = slice; c < C; c += kSlices) {
    const int64_t o = base + (int64_t)c * N;
    const float xv = x[o], tv = t[o];
    if (GX) gx[o] = fmaf(tv, cross, -xv * selfx);
    if (GT) gt[o] = fmaf(xv, cross, -tv * selft);
  }
}

static int map_check(int64_t B, int64_t C, int64_t N) {
  if (B < 0 || C <= 0 || N < 0) return GFLA_ERR_BAD_SHAPE;
  if (N > 0x7fffff00LL || C > 0x7fffff00LL || B > 65535) return GFLA_ERR_UNSUPPORTED;
  return GFLA_OK;
}

}  // namespace gfla

extern "C" {
int gfla_correctness_map_fwd_f32(const float *warped, const float *target, const float *best, float *loss_map,
                                 float *stats, int64_t B, int64_t C, int64_t N, double eps_cos, double eps,
                                 gfla_stream_t stream) {
  using namespace gfla;
  if (!warped || !target || !best || !loss_map || !stats) return GFLA_ERR_NULL_POINTER;
  if (int rc = map_check(B, C, N)) return rc;
  if (B == 0 || N == 0) return GFLA_OK;
  correctness_map_fwd_kernel<<<dim3((unsigned)ceil_div(N, 64), (unsigned)B), 256, 0,
                               static_cast<hipStream_t>(stream)>>>(warped, target, best, loss_map, stats, (int)C,
                                                                   (int)N, (float)eps_cos, (float)eps);
  return launch_status();
}

int gfla_correctness_map_bwd_f32(const float *warped, const float *target, const float *best, const float *stats,
                                 const float *loss_map, const float *grad_map, float *grad_warped,
                                 float *grad_target, float *grad_best, int64_t B, int64_t C, int64_t N,
                                 double eps_cos, double eps, gfla_stream_t stream) {
  using namespace gfla;
  if (!warped || !target || !best || !stats || !loss_map || !grad_map) return GFLA_ERR_NULL_POINTER;
  if (int rc = map_check(B, C, N)) return rc;
  if (B == 0 || N == 0 || (!grad_warped && !grad_target && !grad_best)) return GFLA_OK;
  const dim3 grid((unsigned)ceil_div(N, 64), (unsigned)B);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define GFLA_MAP_BWD(GX_, GT_)                                                                                   \
  correctness_map_bwd_kernel<GX_, GT_><<<grid, 256, 0, st>>>(warped, target, best, stats, loss_map, grad_map,    \
                                                             grad_warped, grad_target, grad_best, (int)C, (int)N, \
                                                             (float)eps_cos, (float)eps)
  if (grad_warped && grad_target) GFLA_MAP_BWD(true, true);
  else if (grad_warped) GFLA_MAP_BWD(true, false);
  else if (grad_target) GFLA_MAP_BWD(false, true);
  else GFLA_MAP_BWD(false, false);
#undef GFLA_MAP_BWD
  return launch_status();
}
}
