// MFMA operand fragments shared by the convolution (fc_conv_impl.h) and weight-gradient (fc_gemm.hip) kernels.
#pragma once

#include "fc_gemm.h"

namespace gfla {

// ------------------------------------------------------------------------------------------ MFMA fragments
// One fragment = the K-slice of a 32-row operand block a lane feeds to the matrix core:
//   mode 0: 4 consecutive channels (one 16-byte LDS slot); lanes 0-31 take slot 2*kb, lanes 32-63 slot 2*kb+1, and
//           the four v_mfma_f32_32x32x2_f32 of a fragment pair element e of both halves (a permutation of the 8
//           channels of the K block, the same for A and B);
//   mode 2/3: 8 consecutive channels of each f16 term: lanes 0-31 channels 0-7, lanes 32-63 channels 8-15 -- the
//           A/B layout of v_mfma_f32_32x32x16_f16.
template <int MODE>
struct Frag;
template <>
struct Frag<0> {
  float4 v;
};
template <>
struct Frag<1> {
  f16x8 s[1];
};
template <>
struct Frag<2> {
  f16x8 s[2];
};
template <>
struct Frag<3> {
  f16x8 s[3];
};

template <int MODE>
__device__ __forceinline__ Frag<MODE> load_frag(const unsigned char *rec, int plane_stride, int kb, int kh) {
  Frag<MODE> f;
  if constexpr (MODE == 0) {
    f.v = *reinterpret_cast<const float4 *>(rec + (2 * kb + kh) * 16);
  } else {
#pragma unroll
    for (int sp = 0; sp < Fc<MODE>::NS; ++sp)
      f.s[sp] = *reinterpret_cast<const f16x8 *>(rec + sp * plane_stride + kh * 16);
  }
  return f;
}

template <int MODE>
__device__ __forceinline__ f32x16 mma(const Frag<MODE> &a, const Frag<MODE> &b, f32x16 acc) {
  if constexpr (MODE == 0) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v.x, b.v.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v.y, b.v.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v.z, b.v.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v.w, b.v.w, acc, 0, 0, 0);
  } else if constexpr (MODE == 1) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.s[0], b.s[0], acc, 0, 0, 0);
  } else if constexpr (MODE == 2) {  // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.s[1], b.s[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.s[0], b.s[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.s[0], b.s[0], acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.s[2], b.s[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.s[0], b.s[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.s[1], b.s[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.s[1], b.s[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.s[0], b.s[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.s[0], b.s[0], acc, 0, 0, 0);
  }
  return acc;
}

}  // namespace gfla
