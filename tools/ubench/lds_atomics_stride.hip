// Micro-benchmark 2: ds_add_f64 / ds_add_u64 throughput vs the lane->address stride (in 8-byte units).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int TPB = 512;
constexpr int N64 = 4096;  // doubles (32 KB)
template <int MODE>
__global__ __launch_bounds__(TPB) void k(double *gout, int iters, int stride) {
  __shared__ unsigned long long lds64[N64];
  for (int i = threadIdx.x; i < N64; i += TPB) lds64[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x;
  const int base = (lane * stride) & (N64 - 1);
  double v = 1.0 + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int a = (base + u * 263 + it * 17) & (N64 - 1);
      if (MODE == 0) __hip_atomic_fetch_add(reinterpret_cast<double *>(&lds64[a]), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(&lds64[a], (unsigned long long)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  if (lds64[threadIdx.x] == 12345ull) gout[0] = v;
}
int main() {
  double *d; (void)hipMalloc(&d, 1 << 20);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 512, iters = 1000;
  for (int stride : {1, 2, 3, 4, 5, 8, 9, 16, 17, 32, 33, 64, 65}) {
    float ms[2];
    for (int m = 0; m < 2; ++m) {
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(TPB), 0, 0, d, iters, stride);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(TPB), 0, 0, d, iters, stride);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms[m], e0, e1);
      }
    }
    const double ops = (double)blocks * TPB * iters * 16;
    printf("stride %2d (x8 B): ds_add_f64 %6.2f lanes/clk/CU   ds_add_u64 %6.2f lanes/clk/CU\n", stride,
           ops / (ms[0] * 1e-3) / 256 / 2.4e9, ops / (ms[1] * 1e-3) / 256 / 2.4e9);
  }
  return 0;
}
