// Micro-benchmark: throughput of LDS / global atomic adds on gfx950 (design input for the scatter kernels).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_atomics.hip -o /tmp/lds_atomics && /tmp/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int TPB = 512;
constexpr int N_LDS = 8192;  // floats of LDS used (32 KB)

template <int MODE>
__global__ __launch_bounds__(TPB) void k(float *gout, int iters, int stride, int same) {
  __shared__ unsigned long long lds64[N_LDS / 2];
  float *lds = reinterpret_cast<float *>(lds64);
  for (int i = threadIdx.x; i < N_LDS; i += TPB) lds[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x;
  int base = same ? (threadIdx.x >> 6) * 64 : (lane * stride) % N_LDS;
  float v = 1.0f + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int a = (base + u * 523 + it * 17) & (N_LDS - 1);
      if (MODE == 0) {
        __hip_atomic_fetch_add(&lds[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (MODE == 1) {
        __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(&lds[a]), (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (MODE == 2) {
        lds[a] = lds[a] + v;  // racy read-modify-write: rate of ds_read + ds_write only
      } else if (MODE == 3) {
        __hip_atomic_fetch_add(&lds64[a >> 1], (unsigned long long)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else if (MODE == 4) {
        float old = __hip_atomic_fetch_add(&lds[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        v += old * 1e-30f;
      } else if (MODE == 5) {  // packed bf16 add? use f64 atomic instead
        __hip_atomic_fetch_add(reinterpret_cast<double *>(&lds64[a >> 1]), (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  if (lds[threadIdx.x] == 12345.678f) gout[0] = v;
}

template <int MODE>
__global__ __launch_bounds__(256) void g(float *buf, int iters, int n, int pattern) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  float v = 1.0f;
  unsigned r = tid * 2654435761u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int a;
      if (pattern == 0) a = (tid + (it * 8 + u) * 4099) % n;                       // coalesced, distinct
      else { r = r * 1664525u + 1013904223u; a = (r >> 8) % n; }                   // random
      if (MODE == 0) __hip_atomic_fetch_add(&buf[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else buf[a] = buf[a] + v;
    }
  }
}

template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float *d; hipMalloc(&d, 256 << 20);
  hipMemset(d, 0, 256 << 20);
  const int blocks = 256 * 2, iters = 2000;
  const char *names[] = {"ds_add_f32 (no rtn)", "ds_add_u32", "ds_read+ds_write (racy RMW)", "ds_add_u64", "ds_add_rtn_f32", "ds_add_f64"};
  for (int stride = 1; stride <= 33; stride += 32) {
    for (int same = 0; same <= 1; ++same) {
      if (same && stride > 1) continue;
      float ms[6];
      ms[0] = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(TPB), 0, 0, d, iters, stride, same); });
      ms[1] = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(TPB), 0, 0, d, iters, stride, same); });
      ms[2] = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(TPB), 0, 0, d, iters, stride, same); });
      ms[3] = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(TPB), 0, 0, d, iters, stride, same); });
      ms[4] = timeit([&] { hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(TPB), 0, 0, d, iters, stride, same); });
      ms[5] = timeit([&] { hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(TPB), 0, 0, d, iters, stride, same); });
      const double ops = (double)blocks * TPB * iters * 16;
      for (int m = 0; m < 6; ++m)
        printf("LDS  stride=%2d same_addr=%d  %-28s %8.3f ms  %7.1f Glane-ops/s  %6.2f lane-ops/clk/CU (2.4GHz, 256 CU)\n", stride, same,
               names[m], ms[m], ops / ms[m] / 1e6, ops / (ms[m] * 1e-3) / 256 / 2.4e9);
    }
  }
  for (int pattern = 0; pattern <= 1; ++pattern) {
    for (int nMB : {1, 64}) {
      const int n = nMB * 262144;
      const int gb = 256 * 16, git = 200;
      float a = timeit([&] { hipLaunchKernelGGL(g<0>, dim3(gb), dim3(256), 0, 0, d, git, n, pattern); });
      float b = timeit([&] { hipLaunchKernelGGL(g<1>, dim3(gb), dim3(256), 0, 0, d, git, n, pattern); });
      const double ops = (double)gb * 256 * git * 8;
      printf("GLOBAL pattern=%s footprint=%3d MB  atomic_add_f32 %8.3f ms %7.1f Gops/s | plain RMW %8.3f ms %7.1f Gops/s\n",
             pattern ? "random" : "coalesced", nMB, a, ops / a / 1e6, b, ops / b / 1e6);
    }
  }
  return 0;
}
