// Micro-benchmark: what does an LDS atomic cost when only SOME lanes of the wave are active?  (Design input, round 5: folding
// the overlapping patch columns of neighbouring lanes with DPP shifts leaves one full-wave atomic plus K sparse ones per
// patch row -- worth it only if the sparse ones are cheap.)
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_atomics_sparse.hip -o /tmp/lds_sparse && /tmp/lds_sparse
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int TPB = 512;
constexpr int N64 = 4096;  // 64-bit words of LDS (32 KB)

// MODE 0 ds_add_f64, 1 ds_add_u64.  Active lanes: (lane % every) == 0.
template <int MODE>
__global__ __launch_bounds__(TPB) void k(float *gout, int iters, int every) {
  __shared__ unsigned long long lds64[N64];
  for (int i = threadIdx.x; i < N64; i += TPB) lds64[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x;
  const bool on = (lane % every) == 0;
  const double v = 1.0 + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int a = (lane + u * 261 + it * 17) & (N64 - 1);
      if (on) {
        if (MODE == 0) __hip_atomic_fetch_add(reinterpret_cast<double *>(&lds64[a]), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(&lds64[a], (unsigned long long)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  if (lds64[threadIdx.x] == 12345ull) gout[0] = (float)v;
}

// The folding itself: K wave_shr:1 shift-adds (v_add_f32_dpp) in front of one atomic, against K+1 atomics.
template <int K, bool FOLD>
__global__ __launch_bounds__(TPB) void fold(float *gout, int iters) {
  __shared__ unsigned long long lds64[N64];
  for (int i = threadIdx.x; i < N64; i += TPB) lds64[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x;
  const bool last = (lane & 31) == 31;   // run ends: one lane in 32
  float a[K + 1];
#pragma unroll
  for (int q = 0; q <= K; ++q) a[q] = 1.0f + lane + q;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int base = (lane + u * 261 + it * 17) & (N64 - 1 - 8);
      double *p = reinterpret_cast<double *>(lds64) + base;
      if (FOLD) {
        float t = a[K];
#pragma unroll
        for (int q = K - 1; q >= 0; --q) {
          if (last) __hip_atomic_fetch_add(p + q + 1, (double)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          t = a[q] + __builtin_amdgcn_update_dpp(0.f, t, 0x138, 0xf, 0xf, false);
        }
        __hip_atomic_fetch_add(p, (double)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
#pragma unroll
        for (int q = 0; q <= K; ++q) __hip_atomic_fetch_add(p + q, (double)a[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
#pragma unroll
      for (int q = 0; q <= K; ++q) a[q] += 0.25f;
    }
  }
  __syncthreads();
  if (lds64[threadIdx.x] == 12345ull) gout[0] = a[0];
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
  float *d; hipMalloc(&d, 1 << 20);
  const int blocks = 256 * 2, iters = 1000;
  for (int every : {1, 2, 4, 8, 16, 32, 64}) {
    float m0 = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(TPB), 0, 0, d, iters, every); });
    float m1 = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(TPB), 0, 0, d, iters, every); });
    const double insts = (double)blocks * (TPB / 64) * iters * 16;   // wave-level atomic instructions
    printf("active lanes %2d of 64:  ds_add_f64 %7.3f ms = %5.1f CU-clk per wave instruction | ds_add_u64 %7.3f ms = %5.1f CU-clk  (2.4 GHz, 256 CU)\n",
           64 / every, m0, m0 * 1e-3 * 2.4e9 * 256 / insts, m1, m1 * 1e-3 * 2.4e9 * 256 / insts);
  }
  {
    float a = timeit([&] { hipLaunchKernelGGL((fold<3, false>), dim3(blocks), dim3(TPB), 0, 0, d, iters); });
    float b = timeit([&] { hipLaunchKernelGGL((fold<3, true>), dim3(blocks), dim3(TPB), 0, 0, d, iters); });
    float c = timeit([&] { hipLaunchKernelGGL((fold<5, false>), dim3(blocks), dim3(TPB), 0, 0, d, iters); });
    float e = timeit([&] { hipLaunchKernelGGL((fold<5, true>), dim3(blocks), dim3(TPB), 0, 0, d, iters); });
    printf("patch row of K+1 columns, ds_add_f64:  K=3 plain %7.3f ms, folded %7.3f ms | K=5 plain %7.3f ms, folded %7.3f ms\n", a, b, c, e);
  }
  return 0;
}
