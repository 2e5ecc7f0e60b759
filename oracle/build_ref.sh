#!/usr/bin/env bash
# TEST INFRASTRUCTURE.  Compiles the REAL reference extensions (block_extractor_cuda,
# local_attn_reshape_cuda, resample2d_cuda) for gfx950 from the sources WHERE THEY LIE under
# $GFLA_REFERENCE (default /root/reference) -- no copy, no hipify, no edit: hipcc reads the
# .cu as HIP, ref_shim/ supplies the one stream accessor and the one dispatch macro that
# torch 2.10 changed.  Outputs go to oracle/_ref/ only (git-ignored, shipped to the GPU box).
# The reference's own build (setup.py + nvcc, sm_60/61/70) is not used.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${GFLA_REFERENCE:-/root/reference}"
NET="$REF/model/networks"
OUT="$HERE/_ref"
[ -d "$NET/block_extractor" ] || { echo "reference not present at $REF -- skipping _ref build"; exit 0; }
mkdir -p "$OUT"
PY="${PYTHON:-python}"
TI="$($PY -c 'import torch.utils.cpp_extension as c; print(c.include_paths()[0])')"
TL="$($PY -c 'import torch.utils.cpp_extension as c; print(c.library_paths()[0])')"
PYI="$($PY -c 'import sysconfig; print(sysconfig.get_paths()["include"])')"
ABI="$($PY -c 'import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))')"
COMMON=(-O2 -std=c++17 -fPIC -I"$HERE/ref_shim" -I"$TI" -I"$TI/torch/csrc/api/include" -I"$PYI"
        -I/opt/rocm/include -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -D_GLIBCXX_USE_CXX11_ABI="$ABI" -w)
build_one() {  # dir  stem  module
  local dir="$1" stem="$2" mod="$3"
  if [ "$OUT/$mod.so" -nt "$NET/$dir/${stem}_kernel.cu" ] && [ "$OUT/$mod.so" -nt "$HERE/build_ref.sh" ]; then
    echo "[_ref] $mod.so up to date"; return
  fi
  echo "[_ref] building $mod from $NET/$dir"
  hipcc -x hip --offload-arch=gfx950 "${COMMON[@]}" -include "$HERE/ref_shim/ref_preinclude.h" \
        -c "$NET/$dir/${stem}_kernel.cu" -o "$OUT/${stem}_kernel.o"
  g++ "${COMMON[@]}" -DTORCH_EXTENSION_NAME="$mod" -DTORCH_API_INCLUDE_EXTENSION_H \
        -c "$NET/$dir/${stem}_cuda.cc" -o "$OUT/${stem}_cuda.o"
  hipcc --offload-arch=gfx950 -shared -fPIC "$OUT/${stem}_kernel.o" "$OUT/${stem}_cuda.o" \
        -L"$TL" -ltorch -ltorch_cpu -ltorch_hip -lc10 -lc10_hip -ltorch_python -Wl,-rpath,"$TL" \
        -o "$OUT/$mod.so"
  rm -f "$OUT/${stem}_kernel.o" "$OUT/${stem}_cuda.o"
}
build_one block_extractor     block_extractor     block_extractor_cuda &
build_one local_attn_reshape  local_attn_reshape  local_attn_reshape_cuda &
build_one resample2d_package  resample2d          resample2d_cuda &
wait
ls -la "$OUT"
