// checks the three-instruction (hi, lo) split of fc_wino16.hip against the plain C form, bit for bit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef _Float16 f16x2w __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ uint32_t split_c(float v) {
  const _Float16 h = (_Float16)v;
  const _Float16 l = (_Float16)(v - (float)h);
  return (uint32_t)__builtin_bit_cast(unsigned short, h) | ((uint32_t)__builtin_bit_cast(unsigned short, l) << 16);
}
__device__ uint32_t split_3(float v) {
  const _Float16 h = (_Float16)v;
  float rem;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(rem) : "v"(v), "v"(h));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2v{v, rem}, f16x2w));
}
__global__ void k(const float *x, uint32_t *a, uint32_t *b, float *r, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  a[i] = split_c(x[i]);
  b[i] = split_3(x[i]);
  const _Float16 h = (_Float16)x[i];
  float rem;
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(rem) : "v"(x[i]), "v"(h));
  r[i] = rem;
}
int main() {
  const int n = 1 << 16;
  std::vector<float> x(n);
  uint32_t s = 12345;
  for (int i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    const float m = (float)(s >> 8) / (1 << 24) * 2.f - 1.f;
    s = s * 1664525u + 1013904223u;
    x[i] = ldexpf(m, (int)(s >> 27) - 20);
  }
  float *dx, *dr; uint32_t *da, *db;
  hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dr, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, da, db, dr, n);
  std::vector<uint32_t> a(n), b(n); std::vector<float> r(n);
  hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) if (a[i] != b[i]) { if (bad < 8) printf("x=%a c=%08x new=%08x rem=%a\n", x[i], a[i], b[i], r[i]); ++bad; }
  printf("mismatches: %d of %d\n", bad, n);
  return bad != 0;
}
