// Shared device/host helpers for the gfx950 kernels of libgfla_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gfla_hip.h"

namespace gfla {

constexpr int kBlock = 256;       // 4 wave64 per workgroup
constexpr int kNumCU = 256;       // MI355X: 8 XCD x 32 CU
constexpr int kNumXCD = 8;
constexpr int kWavesPerCU = 32;

// ---- storage <-> arithmetic types --------------------------------------------------------
// f32 and bf16 storage compute in float (the reference's float instantiation), f64 in double.
struct bf16_t {
  uint16_t bits;
  bf16_t() = default;
  // (T)x in the shared templates: round-to-nearest-even conversion (defined below, after Num<bf16_t>)
  __host__ __device__ explicit bf16_t(float v);
  static __host__ __device__ __forceinline__ bf16_t from_bits(uint16_t b) {
    bf16_t r;
    r.bits = b;
    return r;
  }
};

template <typename T>
struct Num;
template <>
struct Num<float> {
  using acc = float;
  static __device__ __forceinline__ float ld(const float *p) { return *p; }
  static __device__ __forceinline__ float from(float v) { return v; }
};
template <>
struct Num<double> {
  using acc = double;
  static __device__ __forceinline__ double ld(const double *p) { return *p; }
  static __device__ __forceinline__ double from(double v) { return v; }
};
template <>
struct Num<bf16_t> {
  using acc = float;
  static __device__ __forceinline__ float ld(const bf16_t *p) {
    return __uint_as_float(static_cast<uint32_t>(p->bits) << 16);
  }
  static __host__ __device__ __forceinline__ uint16_t pack(float v) {  // round-to-nearest-even
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return static_cast<uint16_t>(u >> 16);
  }
  static __device__ __forceinline__ bf16_t from(float v) { return bf16_t::from_bits(pack(v)); }
};
__host__ __device__ inline bf16_t::bf16_t(float v) : bits(Num<bf16_t>::pack(v)) {}

// Vector of V storage elements written with ONE store instruction (V*sizeof(T) in {4,8,16}).
template <typename T, int V>
struct alignas(sizeof(T) * V) Pack {
  T v[V];
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return max(min(v, hi), lo); }

template <typename A>
__device__ __forceinline__ A floor_t(A v);
template <>
__device__ __forceinline__ float floor_t<float>(float v) { return floorf(v); }
template <>
__device__ __forceinline__ double floor_t<double>(double v) { return floor(v); }

// atomic add in the storage type (f32 / f64 only): relaxed, device scope, no return value.
__device__ __forceinline__ void atomic_add(float *p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_add(double *p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// bf16 storage has no atomic add.  The bf16 backward entry points only exist for the "planes in LDS" kernels, where
// feature-map gradients leave through an exclusive read-modify-write and every cross-workgroup reduction (flow,
// logits, (dx,dy,sigma)) targets a float32 buffer; the host side returns GFLA_ERR_UNSUPPORTED for anything else.
// This overload only lets the shared templates compile.
__device__ __forceinline__ void atomic_add(bf16_t *, bf16_t) { __builtin_trap(); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? GFLA_OK : GFLA_ERR_LAUNCH;
}

// Pick how many channels one thread walks: as many as possible (amortises the per-pixel index
// and weight setup) while the launch still has >= `want_waves` wavefronts to fill 256 CUs.
inline int pick_channels_per_thread(int64_t threads_per_channel_group_unit, int64_t C,
                                    int64_t B, int max_cpt, int64_t want_waves = 4 * kNumCU * kWavesPerCU) {
  int cpt = max_cpt;
  while (cpt > 1) {
    int64_t groups = ceil_div(C, cpt);
    int64_t waves = threads_per_channel_group_unit * groups * B / 64;
    if (waves >= want_waves) break;
    cpt >>= 1;
  }
  return cpt < 1 ? 1 : cpt;
}

int tuning(int key);  // defined in abi.hip
void note_path(int id);  // dispatch trace (gfla_path_count)

}  // namespace gfla
