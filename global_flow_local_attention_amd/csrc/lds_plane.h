// "Plane in LDS" building blocks shared by the gather/scatter kernels.
//
// Every op on the hot path gathers from (or scatters into) ONE (b,c) feature plane at positions
// chosen by the flow field.  Planes at the attention layers are small (32x22 .. 64x64 floats =
// 2.8 .. 16 KB), so a workgroup can hold a GROUP of G whole planes in the CU's 160 KB LDS:
//   * gathers become ds_read_b32 at 128 B/clk/CU instead of L1/TA-limited global loads, after
//     ONE coalesced 16-byte-per-lane read of the planes from HBM;
//   * scatters become ds_add_f64 into a double-precision LDS plane followed by ONE coalesced flush,
//     instead of k^2*4 global atomics per pixel and channel (the reference's decomposition and the reason its
//     backward kernels run at ~0.5 % of HBM bandwidth).
// Arbitrary flow needs no special case: the whole plane is resident, clamping is index math.
// Planes that do not fit (e.g. 256x176) use the global-memory kernels.
#pragma once

#include "gfla_common.h"

namespace gfla {

constexpr int kLdsBudgetDefault = 64 * 1024;  // per workgroup: two workgroups of 512 threads per CU
// tuning key 10 (KB) overrides it; more than 64 KB needs the kernel attribute raised (launch_lds below)
inline int64_t lds_budget() {
  const int kb = tuning(10);
  return kb >= 16 && kb <= 160 ? (int64_t)kb * 1024 : kLdsBudgetDefault;
}
#define kLdsBudget (::gfla::lds_budget())
constexpr int kLdsThreads = 512;

// LDS atomic add without return value.  Measured on MI355X (profiles/r1_ubench_lds_atomics.txt):
// ds_add_f64 runs at 3.2-3.9 lanes/clk/CU, ds_add_f32 at 0.33 -- a 10x gap -- so every scatter
// accumulator plane in LDS is DOUBLE, whatever the tensor type.  As a bonus the sum is (almost)
// order-independent: float gradients come out bit-reproducible run to run in practice, unlike the
// reference's float atomics.
using lds_acc_t = double;
__device__ __forceinline__ void lds_add(double *p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// (Tried and rejected, round 1: merging neighbouring lanes' overlapping patch columns with a chain of
// ds_bpermute before the atomics.  It removed ~60 % of the ds_add_f64 but the K dependent shuffle rounds
// per patch row are latency-bound: be_bwd 461 -> 517 us, resample2d d/d input1 528 -> 587 us.
// Also rejected: scattering two vertically adjacent pixels as one (K+2)x(K+1) patch (-42 % atomics when
// the pair lines up) -- with smooth random flows most waves hold both paired and unpaired lanes, so the
// wave executes both code paths: be_bwd 457 -> 909 us (profiles/r1_s16_vertical_pairing_rejected_*).)
struct PlaneGeo {
  int G;        // channels (planes) per workgroup; 0 = does not fit, use the global kernels
  int ngroups;  // ceil(C / G)
  int split;    // workgroups sharing one (b, group): each takes `per` of the work items
  unsigned lds_bytes;
  int per;      // work items (pixels / output positions) per workgroup
  int margin;   // < 0: the whole plane is resident.  >= 0: only a ROW WINDOW of the plane is -- the
                // rows a band of flow rows can reach when |flow_y| <= margin; lanes whose taps fall
                // outside the window take the global-memory path for that pixel
};

// Source rows [lo, lo+rows) resident in LDS for the flow rows [yf_first, yf_last] of a workgroup whose
// taps span yf - k_lo .. yf + k_hi before the flow is added.
struct Window {
  int lo, rows;
};
__device__ __forceinline__ Window make_window(int yf_first, int yf_last, int k_lo, int k_hi, int margin, int H) {
  if (margin < 0) return Window{0, H};
  const int lo = max(0, yf_first - k_lo - margin);
  const int hi = min(H, yf_last + k_hi + 1 + margin);
  return Window{lo, max(hi - lo, 0)};
}

// plane_elems: elements of one plane; bytes_per_elem: LDS bytes one channel needs per plane element
// (sizeof(arithmetic type) for a gather plane, 8 for a scatter plane, their sum for both);
// work_items: positions one (b, group) iterates over.
// chunk = 0: as many planes as fit (halved while the launch has < 2 workgroups per CU).
// chunk >= 1 (gather kernels; the kernel processes channels `chunk` at a time): G is chosen to minimise
//   rounds(G) * (chunk * ceil(G / chunk) + 3),   rounds = ceil(B * ceil(C / G) / (2 workgroups * 256 CUs)),
// i.e. launches whose last round of workgroups is nearly empty and channel slots that are computed but
// not used are avoided; the 3 stands for the per-pixel setup a workgroup pays once for its G channels.
// Measured on MI355X at the bench shapes (profiles/r1_g_sweep.txt): aggregate fwd L2 177 -> 116 us (G 5 -> 4),
// resample fwd relu4_1 58 -> 42 us (G 23 -> 16), relu3_1 90 -> 80 (5 -> 4).
inline PlaneGeo plane_geometry(int64_t plane_elems, int bytes_per_elem, int64_t B, int64_t C,
                               int64_t work_items, bool allow_split, int chunk = 0) {
  PlaneGeo g{0, 0, 1, 0, 0, -1};
  const int64_t per_channel = plane_elems * bytes_per_elem;
  if (per_channel > kLdsBudget) return g;
  int64_t G = kLdsBudget / per_channel;
  if (G > C) G = C;
  if (tuning(4) > 0) {
    if (tuning(4) < G) G = tuning(4);
  } else if (chunk > 0) {
    int64_t best = G, best_cost = -1;
    for (int64_t cand = G; cand >= 1; --cand) {
      const int64_t rounds = ceil_div(B * ceil_div(C, cand), 2 * kNumCU);
      const int64_t cost = rounds * (ceil_div(cand, chunk) * chunk + 3);
      if (best_cost < 0 || cost < best_cost) {
        best = cand;
        best_cost = cost;
      }
    }
    G = best;
  } else {
    // keep >= 2 workgroups per CU in flight when the batch is small
    while (G > 1 && B * ceil_div(C, G) < 2 * kNumCU) G = (G + 1) / 2;
  }
  g.G = (int)G;
  g.ngroups = (int)ceil_div(C, G);
  int split = 1;
  if (allow_split) {
    const int64_t wgs = B * g.ngroups;
    while (wgs * split < 4 * kNumCU && work_items / (split * 2) >= 2 * kLdsThreads && split < 64) split *= 2;
  }
  if (tuning(5) > 0) split = tuning(5);
  g.split = split;
  g.per = (int)ceil_div(work_items, split);
  g.lds_bytes = (unsigned)(G * per_channel);
  return g;
}

// Planes too large for LDS (e.g. 256x176): keep a row WINDOW per workgroup instead.  A workgroup
// takes a band of `hb` flow rows (items_per_row work items each); the window holds the band plus
// k+1 tap rows plus `margin` = hb rows of slack on either side for the flow, so flows up to ~hb
// pixels stay in LDS and larger ones degrade gracefully (per pixel) to the global-memory path.
// Returns G = 0 when even one row band of one channel does not fit.
inline PlaneGeo band_geometry(int64_t H, int64_t W, int bytes_per_elem, int64_t B, int64_t C, int64_t Hf,
                              int64_t items_per_row, int k_span) {
  PlaneGeo g{0, 0, 1, 0, 0, -1};
  const int64_t row_bytes = W * bytes_per_elem;
  int64_t G = C < 4 ? C : 4;
  int64_t cap = 0, hb = 0;
  for (; G >= 1; G /= 2) {
    cap = kLdsBudget / (row_bytes * G);  // window rows that fit
    hb = (cap - k_span) / 3;             // band + 2 * margin(= band) + tap span <= cap
    // fewer channels per workgroup rather than a margin real flows (a few pixels) overflow
    if ((hb >= 2 && (cap - k_span - hb) / 2 >= 6) || (G == 1 && hb >= 1)) break;
  }
  if (G < 1 || hb < 1) return g;
  if (hb > Hf) hb = Hf;
  g.G = (int)G;
  g.ngroups = (int)ceil_div(C, G);
  g.split = (int)ceil_div(Hf, hb);
  g.per = (int)(hb * items_per_row);
  g.margin = (int)((cap - k_span - hb) / 2);
  g.lds_bytes = (unsigned)(cap * row_bytes * G);
  return g;
}

// Either the whole plane (plane_geometry) or a row window (band_geometry).
inline PlaneGeo lds_geometry(int64_t H, int64_t W, int bytes_per_elem, int64_t B, int64_t C, int64_t Hf,
                             int64_t items_per_row, int k_span, int chunk = 0) {
  PlaneGeo g = plane_geometry(H * W, bytes_per_elem, B, C, Hf * items_per_row, true, chunk);
  if (g.G > 0) return g;
  if (tuning(7) == 1) return g;  // windows disabled
  return band_geometry(H, W, bytes_per_elem, B, C, Hf, items_per_row, k_span);
}

// Launch a kernel with `lds` bytes of dynamic LDS; above 64 KB the per-kernel limit has to be raised first.
template <typename... KArgs, typename... Args>
inline void launch_lds(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned lds, hipStream_t stream,
                       Args... args) {
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  kernel<<<grid, block, lds, stream>>>(args...);
}

// global (n contiguous storage elements) -> LDS (arithmetic type), all threads of the block
template <typename T, typename A>
__device__ __forceinline__ void stage_planes(const T *__restrict__ g, A *lds, int n) {
  if constexpr (sizeof(T) == 4 && sizeof(A) == 4) {
    // both sides 16-byte aligned (the LDS side is not when a plane has an odd number of elements and is not the first)
    if (((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(lds)) & 15) == 0) {
      const int n4 = n >> 2;
      const float4 *g4 = reinterpret_cast<const float4 *>(g);
      float4 *l4 = reinterpret_cast<float4 *>(lds);
      for (int i = threadIdx.x; i < n4; i += blockDim.x) l4[i] = g4[i];
      for (int i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) lds[i] = Num<T>::ld(g + i);
      return;
    }
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = Num<T>::ld(g + i);
}

template <typename A>
__device__ __forceinline__ void zero_planes(A *lds, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = 0;
}

// g[i] += val(i).  exclusive = this workgroup is the only writer of these planes (plain
// read-modify-write, coalesced); otherwise device-scope atomics.
template <typename T, typename F>
__device__ __forceinline__ void flush_planes_with(T *__restrict__ g, int n, bool exclusive, bool overwrite, F val) {
  using A = typename Num<T>::acc;
  if (exclusive && overwrite) {  // sole writer of planes the caller did not initialise: plain stores
    for (int i = threadIdx.x; i < n; i += blockDim.x) g[i] = Num<T>::from((A)val(i));
    return;
  }
  if (exclusive) {
    if constexpr (sizeof(T) == 4) {
      if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        const int n4 = n >> 2;
        float4 *g4 = reinterpret_cast<float4 *>(g);
        for (int i = threadIdx.x; i < n4; i += blockDim.x) {
          float4 a = g4[i];
          a.x += (float)val(4 * i); a.y += (float)val(4 * i + 1); a.z += (float)val(4 * i + 2); a.w += (float)val(4 * i + 3);
          g4[i] = a;
        }
        for (int i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) g[i] = Num<T>::from(Num<T>::ld(g + i) + (A)val(i));
        return;
      }
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) g[i] = Num<T>::from(Num<T>::ld(g + i) + (A)val(i));
  } else {
    for (int i = threadIdx.x; i < n; i += blockDim.x) atomic_add(g + i, (T)val(i));
  }
}
template <typename T>
__device__ __forceinline__ void flush_planes(T *__restrict__ g, const lds_acc_t *lds, int n, bool exclusive,
                                             bool overwrite = false) {
  flush_planes_with<T>(g, n, exclusive, overwrite, [lds](int i) { return lds[i]; });
}

// ---- fixed-point scatter planes (round 3) ------------------------------------------------------------------------------
// The same microbenchmark (profiles/r1_ubench_lds_atomics.txt) has ds_add_u64 at 5.2-7.7 lanes/clk/CU against ds_add_f64's
// 3.2-3.9, and a scatter kernel is bound by exactly that rate.  So a kernel that can bound its contributions -- |v| <= max
// |g| over the gradients the workgroup reads, known after one pass over them -- accumulates 64-bit FIXED POINT instead: the
// scale is the power of two that puts max |g| at 2^40 (every float contribution down to 2^-16 of the maximum is represented
// exactly; 22 bits of headroom = 4 M contributions per element before the sum could overflow; smaller ones round at
// 2^-40 of the maximum), the float -> integer conversion is the 1.5 * 2^52 magic-number add, and the integer sum is
// exactly associative: the result does not depend on the order the lanes arrive in (bit-reproducible run to run, which
// neither the reference's float atomics nor the double planes guarantee).  A non-finite gradient cannot be represented:
// the workgroup then writes NaN to the planes it owns (the reference would poison only the taps the value reaches).
using lds_fix_t = long long;
struct FixScale {
  float up;      // contributions are multiplied by this power of two before the conversion
  double down;   // its inverse
  bool finite;
};
// amax_bits: the largest |x| bit pattern the workgroup will scatter (non-finite patterns compare above every finite one)
__device__ __forceinline__ FixScale fix_scale(unsigned amax_bits) {
  const int e = (int)(amax_bits >> 23);                      // biased exponent of the maximum
  const int se = min(253, 127 + 40 - (e - 127));             // biased exponent of the scale
  FixScale f;
  f.up = __uint_as_float((unsigned)se << 23);
  f.down = __longlong_as_double((long long)(1023 - (se - 127)) << 52);
  f.finite = amax_bits < 0x7f800000u;
  return f;
}
constexpr double kFixMagic = 6755399441055744.0;   // 1.5 * 2^52: (x + magic) carries rint(x) in its low mantissa bits
// d = x + kFixMagic, |x| < 2^51
__device__ __forceinline__ void lds_add_fix_biased(lds_fix_t *p, double d) {
  const unsigned long long bits = (unsigned long long)__double_as_longlong(d) - 0x4338000000000000ull;
  __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add_fix(lds_fix_t *p, float v_scaled) { lds_add_fix_biased(p, (double)v_scaled + kFixMagic); }
// workgroup-wide maximum of `v` through the LDS word `slot` (zeroed by the caller before its last barrier)
__device__ __forceinline__ unsigned block_umax(unsigned v, unsigned *slot) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
  if ((threadIdx.x & 63) == 0) atomicMax(slot, v);
  __syncthreads();
  return *slot;
}
template <typename T>
__device__ __forceinline__ void flush_planes_fix(T *__restrict__ g, const lds_fix_t *lds, int n, bool exclusive,
                                                 bool overwrite, FixScale f) {
  const double down = f.finite ? f.down : __longlong_as_double(0x7ff8000000000000ll);
  const bool finite = f.finite;
  flush_planes_with<T>(g, n, exclusive, overwrite, [=](int i) { return finite ? (double)lds[i] * down : down; });
}

}  // namespace gfla
