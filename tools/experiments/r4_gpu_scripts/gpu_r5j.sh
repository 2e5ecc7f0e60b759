#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r5j}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_default_path_gpu.py tests/test_bench_shapes_gpu.py -x -q -m gpu -k "aggregate or attn or resample or scatter" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout 600 python tools/bench_scatter.py --flows smooth,wild --rows 0 > $OUT/scatter.jsonl 2> $OUT/scatter.err; cut -c1-300 $OUT/scatter.jsonl
timeout 600 python bench.py --no-cpu-baseline --no-legs --no-variants > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
l=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("default", l["ms_per_step"], l["value"])
for k in l["kernels"]:
    if "aggregate_bwd" in k["entry"] or "resample2d_bwd_ws" in k["entry"]: print("  %-46s %-34s %8.1f us" % (k["entry"], k["dims"], k["avg_us"]))
PY
