// TEST INFRASTRUCTURE (oracle/_ref build only).
// The reference kernels include <ATen/cuda/CUDAContext.h> only to obtain the current stream
// (e.g. block_extractor_kernel.cu:197).  On a ROCm build of torch that header drags in
// cuda_runtime_api.h, so this directory is put in front of torch's include path and forwards
// the one symbol the reference uses to torch's HIP stream.
#pragma once
#include <ATen/hip/HIPContext.h>
namespace at { namespace cuda {
inline hipStream_t getCurrentCUDAStream() {
  return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
}
}}  // namespace at::cuda
