#!/usr/bin/env bash
# Winograd tuning loop: parity of the isolated convolutions + layer + bench shapes (mode 4 only), then kernel probes.
set -uo pipefail
TAG="${1:-wino}"; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fc_mfma_gpu.py -m gpu -q -x --timeout=300 > $OUT/pytest_layer.log 2>&1; echo "layer pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|Error|rel err" $OUT/pytest_layer.log | cut -c1-200 | tail -8
timeout 600 python -m pytest tests/test_bench_shapes_gpu.py -m gpu -q -s -k "mfma-4" --timeout=500 > $OUT/pytest_bench_shapes.log 2>&1; echo "bench-shape pytest rc=$?"
grep -E "mfma/4|^(FAILED|ERROR)|passed|failed|Error|rel err" $OUT/pytest_bench_shapes.log | cut -c1-300 | tail -8
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --no-legs --fc-mode 4 > $OUT/bench_mode4.json 2> $OUT/bench_mode4.err; echo "bench rc=$?"; cut -c1-200 $OUT/bench_mode4.json; tail -2 $OUT/bench_mode4.err | cut -c1-300
python - "$OUT/bench_mode4.json" <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
for r in d.get("fc_kernels", []): print(r["kernel"], r["avg_us"], r["TFLOPs"], r["frac_mfma_f32_peak"])
PY
