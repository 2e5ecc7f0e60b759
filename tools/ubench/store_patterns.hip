// Micro-benchmark (round 4): how fast does MI355X absorb the output of block_extractor in the reference layout
// (B, C, K*Hf, K*Wf), K = 5, as a function of HOW the store instructions cover it?  No arithmetic, constant data; the
// address streams are the ones the candidate kernels produce.  A workgroup of 8 waves owns 4 consecutive planes and walks
// the flow field in blocks of 64 pixels per wave (pattern of be_fwd_pix_kernel) or the plane flat (round 1's kernel).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/store_patterns.hip -o tools/ubench/store_patterns.bin
//   tools/ubench/store_patterns.bin [Wf ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4a __attribute__((ext_vector_type(4)));
constexpr int K = 5;

// mode 0: flat, 16 B per lane, lanes consecutive (a fill)
// mode 1: flat, 4 B per lane, lanes consecutive
// mode 2: lane = pixel, K floats per lane per row: one 16-byte + one 4-byte store at a 20-byte lane stride
// mode 3: lane = pixel, K dword stores at a 20-byte lane stride
// mode 4: the wave's row (64*K floats) stored as K dword stores with lanes consecutive (256 B per instruction)
// mode 5: the wave's row stored as 16-byte chunks aligned in global memory, lanes consecutive (+ scalar head / tail)
template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(float *__restrict__ out, int Hf, int Wf, int planes_per_wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int Wo = K * Wf, Ho = K * Hf;
  const long oplane = (long)Ho * Wo;
  float *base = out + (long)blockIdx.x * planes_per_wg * oplane;
  const float v = (float)lane;
  if (MODE == 0 || MODE == 1) {
    const long n = planes_per_wg * oplane;
    if (MODE == 0) {
      f4a val = {v, v, v, v};
      for (long i = threadIdx.x; i < n / 4; i += blockDim.x) reinterpret_cast<f4a *>(base)[i] = val;
    } else {
      for (long i = threadIdx.x; i < n; i += blockDim.x) base[i] = v;
    }
    return;
  }
  const int HW = Hf * Wf, nblk = (HW + 63) / 64;
  for (int blk = wave; blk < nblk; blk += nwaves) {
    const int p = blk * 64 + lane;
    const bool active = p < HW;
    const int pc = active ? p : HW - 1;
    const int yf = pc / Wf, xf = pc - yf * Wf;
    const int ooff = (K * yf) * Wo + K * xf;
    int goff[K];
    for (int r = 0; r < K; ++r) {
      const int t = 64 * r + lane, q = t / K, j = t - q * K, p2 = blk * 64 + q;
      const int pp = p2 < HW ? p2 : HW - 1;
      const int y2 = pp / Wf, x2 = pp - y2 * Wf;
      goff[r] = p2 < HW ? (K * y2) * Wo + K * x2 + j : -1;
    }
    for (int cc = 0; cc < planes_per_wg; ++cc) {
      float *oc = base + cc * oplane;
#pragma unroll
      for (int i = 0; i < K; ++i) {
        float *row = oc + (long)i * Wo;
        if (MODE == 2) {
          if (active) {
            f4u val = {v, v, v, v};
            *reinterpret_cast<f4u *>(row + ooff) = val;
            row[ooff + 4] = v;
          }
        } else if (MODE == 3) {
          if (active) {
#pragma unroll
            for (int j = 0; j < K; ++j) row[ooff + j] = v;
          }
        } else if (MODE == 4) {
#pragma unroll
          for (int r = 0; r < K; ++r) if (goff[r] >= 0) row[goff[r]] = v;
        } else if (MODE == 5) {
          // the wave's elements of this row: up to two contiguous runs (the block may straddle a flow-row boundary);
          // emulate with aligned 16-byte chunks over [first, last] of each run
          const int first_p = blk * 64, last_p = min(HW, first_p + 64) - 1;
          const int ya = first_p / Wf, yb = last_p / Wf;
          for (int y = ya; y <= yb; ++y) {
            const int xs = y == ya ? first_p - ya * Wf : 0, xe = y == yb ? last_p - yb * Wf : Wf - 1;
            float *run = row + (long)(K * y) * Wo + K * xs;
            const int len = K * (xe - xs + 1);
            const int head = (int)((4 - ((reinterpret_cast<unsigned long>(run) / 4) & 3)) & 3);
            const int h = head < len ? head : len;
            if (lane < h) run[lane] = v;
            const int body = (len - h) / 4;
            f4a val = {v, v, v, v};
            for (int m = lane; m < body; m += 64) reinterpret_cast<f4a *>(run + h)[m] = val;
            const int t0 = h + body * 4;
            if (lane < len - t0) run[t0 + lane] = v;
          }
        }
      }
    }
  }
}

template <int MODE>
static double run(float *out, int B, int C, int Hf, int Wf, int iters) {
  const int ppw = 4;
  const int wgs = B * C / ppw;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) store_kernel<MODE><<<wgs, 512>>>(out, Hf, Wf, ppw);
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) store_kernel<MODE><<<wgs, 512>>>(out, Hf, Wf, ppw);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters * 1e3;
}

int main(int argc, char **argv) {
  std::vector<int> widths;
  for (int i = 1; i < argc; ++i) widths.push_back(atoi(argv[i]));
  if (widths.empty()) widths = {44, 64, 32, 48, 22};
  const int B = 32, C = 128, Hf = 64;
  float *out;
  CHECK(hipMalloc(&out, (size_t)B * C * K * K * Hf * 96 * 4 + 4096));
  const char *names[6] = {"flat 16 B/lane", "flat 4 B/lane", "pixel lanes: 16+4 B at 20 B stride", "pixel lanes: 5 x 4 B at 20 B stride",
                          "row transposed: 5 x 4 B, lanes consecutive", "row transposed: aligned 16 B chunks + head/tail"};
  for (int Wf : widths) {
    const double bytes = (double)B * C * K * K * Hf * Wf * 4;
    double us[6];
    us[0] = run<0>(out, B, C, Hf, Wf, 10);
    us[1] = run<1>(out, B, C, Hf, Wf, 10);
    us[2] = run<2>(out, B, C, Hf, Wf, 10);
    us[3] = run<3>(out, B, C, Hf, Wf, 10);
    us[4] = run<4>(out, B, C, Hf, Wf, 10);
    us[5] = run<5>(out, B, C, Hf, Wf, 10);
    for (int m = 0; m < 6; ++m)
      printf("{\"Wf\": %d, \"row_bytes\": %d, \"MB\": %.1f, \"pattern\": \"%s\", \"us\": %.1f, \"TBps\": %.2f}\n", Wf, K * Wf * 4,
             bytes / 1e6, names[m], us[m], bytes / us[m] / 1e6);
  }
  return 0;
}
