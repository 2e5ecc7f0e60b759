// Micro-benchmark 4: ds_add_f64 throughput vs waves per CU and vs how many atomics a wave issues between two
// waits on LDS reads (is the LDS atomic pipe throughput- or latency-limited per wave?).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int N64 = 4096;
template <int BURST>
__global__ void k(double *gout, int iters) {
  __shared__ double lds64[N64];
  __shared__ float tab[1024];
  for (int i = threadIdx.x; i < N64; i += blockDim.x) lds64[i] = 0;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) tab[i] = 1.f + i;
  __syncthreads();
  const int l = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = l >> 4, i = (l >> 2) & 3, j = l & 3;
  const int base = q * 700 + i * 44 + j + wave * 5;
  float acc = 0;
  for (int it = 0; it < iters; ++it) {
    const float w = tab[(it * 67 + l) & 1023];  // an LDS read the atomics depend on
#pragma unroll
    for (int u = 0; u < BURST; ++u) {
      const int a = (base + u * 256 + it * 16) & (N64 - 1);
      __hip_atomic_fetch_add(&lds64[a], (double)(w + u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    acc += w;
  }
  __syncthreads();
  if (lds64[threadIdx.x] == 12345.0 || acc == 1.5f) gout[0] = acc;
}
template <int BURST>
void run(double *d, int tpb, int blocks_per_cu) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * blocks_per_cu, iters = 4000 / BURST * 4;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<BURST>, dim3(blocks), dim3(tpb), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double wave_atomics_per_cu = (double)blocks_per_cu * (tpb / 64) * iters * BURST;
  printf("waves/CU %2d  burst %2d : %6.1f clk per wave atomic at the CU  (%.0f clk per atomic seen by one wave)\n",
         blocks_per_cu * tpb / 64, BURST, ms * 1e-3 * 2.4e9 / wave_atomics_per_cu,
         ms * 1e-3 * 2.4e9 / (iters * (double)BURST));
}
int main() {
  double *d; (void)hipMalloc(&d, 1 << 20);
  for (int tpb : {64, 256, 512, 1024}) { run<1>(d, tpb, 1); run<4>(d, tpb, 1); run<16>(d, tpb, 1); }
  run<1>(d, 1024, 2); run<4>(d, 1024, 2); run<16>(d, 1024, 2);
  return 0;
}
