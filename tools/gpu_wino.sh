#!/usr/bin/env bash
# Winograd (mode 4) bring-up: isolated convolution tests, then whole-layer tests, then per-kernel probes.
set -uo pipefail
TAG="${1:-wino}"; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fc_mfma_gpu.py -m gpu -q -x -s -k "conv_fwd_bwd_one_half and 4-" --timeout=300 > $OUT/pytest_conv.log 2>&1; echo "conv pytest rc=$?"
grep -E "^mode 4|^(FAILED|ERROR)|passed|failed|Error|rel err" $OUT/pytest_conv.log | cut -c1-200 | tail -30
timeout 600 python -m pytest tests/test_fc_mfma_gpu.py -m gpu -q -s -k "4-" --timeout=300 > $OUT/pytest_layer.log 2>&1; echo "layer pytest rc=$?"
grep -E "^mode 4|^(FAILED|ERROR)|passed|failed|Error|rel err" $OUT/pytest_layer.log | cut -c1-200 | tail -30
timeout 600 python -m pytest tests/test_bench_shapes_gpu.py -m gpu -q -s -k "mfma-4" --timeout=500 > $OUT/pytest_bench_shapes.log 2>&1; echo "bench-shape pytest rc=$?"
grep -E "mfma/4|^(FAILED|ERROR)|passed|failed|Error|rel err" $OUT/pytest_bench_shapes.log | cut -c1-300 | tail -30
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --fc-mode 4 > $OUT/bench_mode4.json 2> $OUT/bench_mode4.err; echo "bench rc=$?"; cut -c1-260 $OUT/bench_mode4.json; tail -2 $OUT/bench_mode4.err | cut -c1-300
python - <<'PY'
import json
d=json.load(open("gpurun_out/%s/bench_mode4.json" % "$TAG"))
for r in d.get("fc_kernels", []): print(r["kernel"], r["avg_us"], r["TFLOPs"], r["frac_mfma_f32_peak"])
PY
