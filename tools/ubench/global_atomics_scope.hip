// Global float atomics: agent scope (what atomicAdd emits: the RMW is forwarded past the XCD's L2) against workgroup scope
// (performed in the issuing XCD's L2), and where blocks run (HW_REG_XCC_ID against blockIdx % 8).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int SCOPE>
__global__ __launch_bounds__(256) void g(float *buf, int iters, int n, int pattern) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  unsigned r = tid * 2654435761u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int a;
      if (pattern == 0) a = (tid + (it * 8 + u) * 4099) % n;          // coalesced, distinct within a wave
      else if (pattern == 1) { r = r * 1664525u + 1013904223u; a = (r >> 8) % n; }   // random
      else a = ((blockIdx.x % 8) * (n / 8) + (threadIdx.x + (blockIdx.x / 8) * 256 + (it * 8 + u) * 4099) % (n / 8));   // coalesced, XCD-private eighth
      if (SCOPE == 0) __hip_atomic_fetch_add(&buf[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_fetch_add(&buf[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}
__global__ void where(int *hist) {   // hist[xcc * 8 + blockIdx % 8]
  if (threadIdx.x == 0) {
    const unsigned x = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 15;   // HW_REG_XCC_ID, bits 3:0
    atomicAdd(&hist[(x & 7) * 8 + (blockIdx.x & 7)], 1);
  }
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize(); hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float *d; hipMalloc(&d, 256 << 20); hipMemset(d, 0, 256 << 20);
  int *h; hipMalloc(&h, 64 * 4); hipMemset(h, 0, 64 * 4);
  hipLaunchKernelGGL(where, dim3(4096), dim3(64), 0, 0, h); hipDeviceSynchronize();
  int hh[64]; hipMemcpy(hh, h, 256, hipMemcpyDeviceToHost);
  printf("blocks by (HW_REG_XCC_ID row, blockIdx %% 8 column):\n");
  for (int x = 0; x < 8; ++x) { for (int b = 0; b < 8; ++b) printf("%6d", hh[x * 8 + b]); printf("\n"); }
  const int gb = 256 * 16, git = 200;
  for (int pattern = 0; pattern <= 2; ++pattern)
    for (int nMB : {1, 64}) {
      const int n = nMB * 262144;
      float a = timeit([&] { hipLaunchKernelGGL(g<0>, dim3(gb), dim3(256), 0, 0, d, git, n, pattern); });
      float b = timeit([&] { hipLaunchKernelGGL(g<1>, dim3(gb), dim3(256), 0, 0, d, git, n, pattern); });
      const double ops = (double)gb * 256 * git * 8;
      printf("pattern=%s footprint=%3d MB  agent scope %8.3f ms %7.1f Gops/s | workgroup scope %8.3f ms %7.1f Gops/s\n",
             pattern == 0 ? "coalesced" : pattern == 1 ? "random   " : "xcd-eighth", nMB, a, ops / a / 1e6, b, ops / b / 1e6);
    }
  // correctness of workgroup-scope sums when every address is touched from one XCD only (pattern 2): each element of an eighth
  hipMemset(d, 0, 64 << 20);
  hipLaunchKernelGGL(g<1>, dim3(gb), dim3(256), 0, 0, d, 10, 16 * 262144, 2); hipDeviceSynchronize();
  float *host = (float *)malloc(64 << 20); hipMemcpy(host, d, 64 << 20, hipMemcpyDeviceToHost);
  double tot = 0; for (int i = 0; i < 16 * 262144; ++i) tot += host[i];
  printf("xcd-private workgroup-scope atomics: sum %.0f expected %.0f\n", tot, (double)gb * 256 * 10 * 8);
  hipMemset(d, 0, 64 << 20);
  hipLaunchKernelGGL(g<1>, dim3(gb), dim3(256), 0, 0, d, 10, 262144, 0); hipDeviceSynchronize();
  hipMemcpy(host, d, 1 << 20, hipMemcpyDeviceToHost);
  tot = 0; for (int i = 0; i < 262144; ++i) tot += host[i];
  printf("SHARED addresses, workgroup-scope atomics (expected to LOSE updates across XCDs): sum %.0f expected %.0f\n", tot, (double)gb * 256 * 10 * 8);
  return 0;
}
