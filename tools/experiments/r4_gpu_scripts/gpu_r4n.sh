#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r4n}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fc_mfma_gpu.py tests/test_fc_wino_gpu.py tests/test_bench_shapes_gpu.py tests/test_bench_tools_gpu.py -q --timeout=600 > $OUT/pytest_fc.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_fc.log | tail
for T in "30=0" "29=1"; do
  timeout 600 python bench.py --no-legs --no-cpu-baseline --tuning "$T" > $OUT/bench_$T.json 2> $OUT/bench.err; echo "bench [$T] rc=$?"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$T.json"))
print("tuning [$T] value", d["value"], "ms", d["ms_per_step"], "variants", json.dumps({k:v["ms_per_step"] for k,v in d.get("variants",{}).items()}))
PY
done
