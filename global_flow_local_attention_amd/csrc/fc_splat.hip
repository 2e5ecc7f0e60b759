// Gradient of the sampled source map (the "Z layout" map the data-gradient convolution reads) as a GATHER, gfx950.
//
// Backward of hidden[p] += sum_{4 corners} w_corner(p) * Gs[q(p) + corner] (fc_sample.hip; the reference's bilinear rule,
// block_extractor_kernel.cu:58-76, applied to the convolved map) is a bilinear splat of d hidden[p] (128 channels) into
// d Gs.  fc_tail_bwd_kernel used to issue it as coalesced float atomics while walking the positions: 4-8 wave-wide
// atomics per position, and the L2's float-atomic rate (~250 G lane-ops/s, profiles/r1_ubench_lds_atomics.txt) WAS the
// kernel: 52 M lane-atomics = the 195 us it took at C128 64x44 B=32.
//
// The flow is shared by all 128 channels, so the scatter is inverted once per step instead:
//   * fc_splat_cells_kernel: every position is filed under the CELL of its anchor q(p) = floor(p + flow(p)) -- already
//     clamped by the reference rule to the finite range [-(hi+1), W+lo] x [-(hi+1), H+lo], so a far or non-finite flow is
//     just a border cell -- with its two fractional weights and the address of its d hidden row (slot allocation by an
//     integer atomic; a cell holds kSplatCap positions, the rare position beyond that goes to an overflow list);
//   * fc_splat_gather_kernel: one wave per pixel of the gradient map, channels on the lanes: the pixel collects from the
//     2 x 2 cells whose corners can land on it (border pixels: the extra border cells that clamp onto them), weight =
//     (sum of matching x weights) * (sum of matching y weights), and writes its 512 bytes ONCE -- plain stores, no float
//     atomic, and it also writes the zeros of the layout (lead, wrap columns, tail), so the map needs no memset;
//   * fc_splat_overflow_kernel: the overflow list, with the old atomics, after the stores.
// d hidden itself is read from the target half's gradient map, which fc_tail_bwd_kernel writes anyway.
#include "fc_gemm.h"

namespace gfla {

constexpr int kSplatCap = 4;      // positions per cell before the overflow list takes them
constexpr int kSplatPixPerWave = 8;

struct alignas(16) SplatSlot {
  int zt;        // pixel index of the position's d hidden row in the target map's Z layout
  float xr, yb;  // the reference's xR_P, yB_P (xL_P = 1 - xR_P, yT_P = 1 - yB_P)
  int p;         // the position (overflow / debugging)
};

struct SplatGeo {
  int H, W, lo, hi, cw, ncells;   // cells: (H + k + 1) x (W + k + 1), anchor (qy, qx) -> (qy + hi + 1) * cw + qx + hi + 1
  int hm, wm, wpz, lead_s;        // the source map: hm x wm pixels, row pitch wpz, lead_s zero pixels ahead
  int wpt, lead_t;                // the target map's Z layout
};

__device__ __forceinline__ void splat_anchor(float fx, float fy, int x, int y, const SplatGeo &g, int &qx, int &qy, float &xr,
                                             float &yb) {
  const float dx = fx + (float)x, dy = fy + (float)y;       // fc_sample.hip: corners()
  const float fdx = floorf(dx), fdy = floorf(dy);
  xr = dx - fdx;
  yb = dy - fdy;
  qx = (int)fminf(fmaxf(fdx, -(float)(g.hi + 1)), (float)(g.W + g.lo));
  qy = (int)fminf(fmaxf(fdy, -(float)(g.hi + 1)), (float)(g.H + g.lo));
}

__global__ __launch_bounds__(256) void fc_splat_cells_kernel(const float *__restrict__ flow, int *__restrict__ cnt,
                                                            SplatSlot *__restrict__ cells, int *__restrict__ ovf,
                                                            SplatGeo g) {
  const int HW = g.H * g.W;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int64_t b = blockIdx.y;
  if (p >= HW) return;
  const int y = p / g.W, x = p - y * g.W;
  int qx, qy;
  SplatSlot s;
  splat_anchor(flow[(b * 2 + 0) * HW + p], flow[(b * 2 + 1) * HW + p], x, y, g, qx, qy, s.xr, s.yb);
  s.zt = g.lead_t + y * g.wpt + x;
  s.p = p;
  const int64_t cell = b * g.ncells + (qy + g.hi + 1) * g.cw + (qx + g.hi + 1);
  const int slot = atomicAdd(cnt + cell, 1);
  if (slot < kSplatCap) {
    cells[cell * kSplatCap + slot] = s;
  } else {
    const int o = atomicAdd(ovf, 1);
    ovf[1 + o] = (int)(b * HW + p);
  }
}

// weight of the anchor coordinate q for the map coordinate u (both in map coordinates [-hi, n-1+lo]): the reference clamps
// the two corner indices q, q + 1 separately and keeps both weights
__device__ __forceinline__ float splat_axis_weight(int q, int u, int lo_c, int hi_c, float frac) {
  const int c0 = min(max(q, lo_c), hi_c), c1 = min(max(q + 1, lo_c), hi_c);
  return (c0 == u ? 1.f - frac : 0.f) + (c1 == u ? frac : 0.f);
}

__global__ __launch_bounds__(256) void fc_splat_gather_kernel(const float *__restrict__ dzt, const int *__restrict__ cnt,
                                                             const SplatSlot *__restrict__ cells, float *__restrict__ dzs,
                                                             int64_t zt_bs, int64_t zs_bs, int sz, SplatGeo g) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t b = blockIdx.y;
  const int i0 = (blockIdx.x * 4 + wave) * kSplatPixPerWave;
  const float *src = dzt + b * zt_bs + lane * 2;
  float *dst = dzs + b * zs_bs + lane * 2;
  const int *cb = cnt + b * g.ncells;
  const SplatSlot *cellb = cells + b * (int64_t)g.ncells * kSplatCap;
  const int y_lo = -g.hi, y_hi = g.H - 1 + g.lo, x_lo = -g.hi, x_hi = g.W - 1 + g.lo;
#pragma unroll 2
  for (int j = 0; j < kSplatPixPerWave; ++j) {
    const int i = i0 + j;
    if (i >= sz) break;
    float a0 = 0.f, a1 = 0.f;
    const int r = i - g.lead_s;
    const int gy = r >= 0 ? r / g.wpz : -1, gx = r - gy * g.wpz;
    if (r >= 0 && gy < g.hm && gx < g.wm) {
      const int u = gy - g.hi, v = gx - g.hi;  // map coordinates
      // anchors whose corners can be clamped onto (u, v): u - 1 .. u, plus the border cells behind a border pixel
      const int qy0 = u == y_lo ? y_lo - 1 : u - 1, qy1 = u == y_hi ? y_hi + 1 : u;
      const int qx0 = v == x_lo ? x_lo - 1 : v - 1, qx1 = v == x_hi ? x_hi + 1 : v;
      for (int qy = qy0; qy <= qy1; ++qy)
        for (int qx = qx0; qx <= qx1; ++qx) {
          const int cell = (qy + g.hi + 1) * g.cw + (qx + g.hi + 1);
          const int n = min(cb[cell], kSplatCap);
          for (int s = 0; s < n; ++s) {
            const SplatSlot sl = cellb[(int64_t)cell * kSplatCap + s];
            const float w = splat_axis_weight(qx, v, x_lo, x_hi, sl.xr) * splat_axis_weight(qy, u, y_lo, y_hi, sl.yb);
            const float2 d = *reinterpret_cast<const float2 *>(src + (int64_t)sl.zt * kFcHidden);
            a0 = fmaf(w, d.x, a0);
            a1 = fmaf(w, d.y, a1);
          }
        }
    }
    *reinterpret_cast<float2 *>(dst + (int64_t)i * kFcHidden) = make_float2(a0, a1);
  }
}

__global__ __launch_bounds__(256) void fc_splat_overflow_kernel(const float *__restrict__ flow, const float *__restrict__ dzt,
                                                               const int *__restrict__ ovf, float *__restrict__ dzs,
                                                               int64_t zt_bs, int64_t zs_bs, SplatGeo g) {
  const int n = ovf[0];
  const int lane = threadIdx.x & 63;
  const int HW = g.H * g.W;
  const int y_lo = -g.hi, y_hi = g.H - 1 + g.lo, x_lo = -g.hi, x_hi = g.W - 1 + g.lo;
  for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < n; e += gridDim.x * 4) {
    const int bp = ovf[1 + e];
    const int64_t b = bp / HW;
    const int p = bp - (int)b * HW, y = p / g.W, x = p - y * g.W;
    int qx, qy;
    float xr, yb;
    splat_anchor(flow[(b * 2 + 0) * HW + p], flow[(b * 2 + 1) * HW + p], x, y, g, qx, qy, xr, yb);
    const float2 d = *reinterpret_cast<const float2 *>(dzt + b * zt_bs + (int64_t)(g.lead_t + y * g.wpt + x) * kFcHidden + lane * 2);
#pragma unroll
    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
      for (int cx = 0; cx < 2; ++cx) {
        const int u = min(max(qy + cy, y_lo), y_hi), v = min(max(qx + cx, x_lo), x_hi);
        const float w = (cx ? xr : 1.f - xr) * (cy ? yb : 1.f - yb);
        float *o = dzs + b * zs_bs + (int64_t)(g.lead_s + (u + g.hi) * g.wpz + (v + g.hi)) * kFcHidden + lane * 2;
        atomic_add(o, w * d.x);
        atomic_add(o + 1, w * d.y);
      }
  }
}

int64_t fc_splat_scratch_bytes(int64_t B, int H, int W, int k) {
  const int64_t ncells = (int64_t)(H + k + 1) * (W + k + 1);
  const int64_t cnt = ((B * ncells + 1 + B * H * W) * 4 + 255) & ~(int64_t)255;   // counts | overflow count + list
  return cnt + B * ncells * kSplatCap * (int64_t)sizeof(SplatSlot);
}
// bytes at the head of the scratch that must be zero when fc_splat_gather runs (cell counts + overflow count)
int64_t fc_splat_zero_bytes(int64_t B, int H, int W, int k) {
  return (B * (int64_t)(H + k + 1) * (W + k + 1) + 1) * 4;
}

// dzs (B, sz, 128) <- splat of the d hidden rows in dzt; `scratch`: fc_splat_scratch_bytes, its first fc_splat_zero_bytes zeroed
int fc_splat_gather(const float *flow, const float *dzt, float *dzs, void *scratch, int64_t B, int H, int W, int k, int wpz,
                    int wpt, int64_t zs_bs, int64_t zt_bs, int lead_s, int lead_t, int64_t sz, hipStream_t stream) {
  if (!flow || !dzt || !dzs || !scratch) return GFLA_ERR_NULL_POINTER;
  if (B <= 0) return GFLA_OK;
  if (B > 65535 || sz > 0x7fffffffLL || B * (int64_t)H * W > 0x7fffffffLL) return GFLA_ERR_UNSUPPORTED;
  SplatGeo g;
  g.H = H, g.W = W, g.lo = k / 2, g.hi = k - 1 - k / 2;
  g.cw = W + k + 1, g.ncells = (H + k + 1) * (W + k + 1);
  g.hm = H + k - 1, g.wm = W + k - 1, g.wpz = wpz, g.lead_s = lead_s, g.wpt = wpt, g.lead_t = lead_t;
  unsigned char *sc = static_cast<unsigned char *>(scratch);
  int *cnt = reinterpret_cast<int *>(sc);
  int *ovf = cnt + B * g.ncells;
  SplatSlot *cells = reinterpret_cast<SplatSlot *>(sc + (((B * g.ncells + 1 + B * H * W) * 4 + 255) & ~(int64_t)255));
  fc_splat_cells_kernel<<<dim3((unsigned)ceil_div((int64_t)H * W, 256), (unsigned)B), 256, 0, stream>>>(flow, cnt, cells, ovf, g);
  fc_splat_gather_kernel<<<dim3((unsigned)ceil_div(sz, 4 * kSplatPixPerWave), (unsigned)B), 256, 0, stream>>>(
      dzt, cnt, cells, dzs, zt_bs, zs_bs, (int)sz, g);
  fc_splat_overflow_kernel<<<dim3(256), 256, 0, stream>>>(flow, dzt, ovf, dzs, zt_bs, zs_bs, g);
  return launch_status();
}

}  // namespace gfla
