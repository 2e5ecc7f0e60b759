// TEST INFRASTRUCTURE (oracle/_ref build only); force-included in front of the reference *.cu.
// torch 2.10 no longer converts DeprecatedTypeProperties (tensor.type()) to ScalarType, which
// is the only thing that stops the reference launchers from compiling
// (block_extractor_kernel.cu:196,253; local_attn_reshape_kernel.cu:130,176;
// resample2d_kernel.cu:354,400,427).  Re-point the dispatch macro; the sources stay untouched.
#pragma once
#include <hip/hip_runtime.h>
#include <ATen/ATen.h>
#include <ATen/Dispatch.h>
#undef AT_DISPATCH_FLOATING_TYPES
#define AT_DISPATCH_FLOATING_TYPES(TYPE, NAME, ...) \
  AT_DISPATCH_SWITCH((TYPE).scalarType(), NAME, AT_DISPATCH_CASE_FLOATING_TYPES(__VA_ARGS__))
