// Micro-benchmark 3: ds_add_f64 throughput for lane->address PATTERNS (which lanes conflict with which?).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int TPB = 512;
constexpr int N64 = 4096;  // doubles (32 KB)
__device__ __forceinline__ int pattern(int p, int l) {
  const int q = l >> 4, i = (l >> 2) & 3, j = l & 3;
  switch (p) {
    case 0: return l;                                   // unit stride
    case 1: return q * 700 + i * 44 + j;                // 4x4 window per 16 lanes, pitch 44, pixels far apart
    case 2: return q * 701 + i * 44 + j;                // same, odd pixel distance
    case 3: return (l & 15) + q * 272;                  // every 16-lane group covers banks 0..15 (other rows)
    case 4: return (l & 7) + (l >> 3) * 272;            // every 8-lane group covers banks 0..7
    case 5: return q * 700 + i * 48 + j;                // pitch 48: rows of a window share banks (4-way)
    case 6: return q + i * 44 + j;                      // adjacent pixels: overlapping windows (same addresses)
    case 7: return (l & 31) + (l >> 5) * 528;           // 32 consecutive per half wave
    case 8: return ((l & 15) * 17) & 4095;              // 16 lanes: stride 17 (distinct banks), 4 groups same addresses
    case 9: return q * 700 + j * 44 + i;                // transposed window
    case 10: return (l & 3) * 16 + (l >> 2);            // lanes l, l+4, ... consecutive: 4-lane groups hit banks 0,16->0?  
    default: return l;
  }
}
__global__ __launch_bounds__(TPB) void k(double *gout, int iters, int p) {
  __shared__ double lds64[N64];
  for (int i = threadIdx.x; i < N64; i += TPB) lds64[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int base = pattern(p, lane) + wave * 5;
  double v = 1.0 + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int a = (base + u * 256 + it * 16) & (N64 - 1);
      __hip_atomic_fetch_add(&lds64[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  if (lds64[threadIdx.x] == 12345.0) gout[0] = v;
}
int main() {
  double *d; (void)hipMalloc(&d, 1 << 20);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 512, iters = 1000;
  const char *names[] = {"unit stride", "4x4 window/16 lanes pitch 44, far pixels", "same, odd distance", "16-lane groups on banks 0..15",
                         "8-lane groups on banks 0..7", "window pitch 48 (rows share banks)", "adjacent pixels (same addresses)",
                         "32 consecutive per half wave", "16 lanes stride 17, groups share addresses", "transposed window", "pattern 10"};
  for (int p = 0; p <= 10; ++p) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(TPB), 0, 0, d, iters, p);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double ops = (double)blocks * TPB * iters * 16;
    const double rate = ops / (ms * 1e-3) / 256 / 2.4e9;
    printf("pattern %2d %-46s ds_add_f64 %6.2f lanes/clk/CU  (%5.1f clk per wave atomic)\n", p, names[p], rate, 64 / rate);
  }
  return 0;
}
