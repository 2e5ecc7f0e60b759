#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r5q}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for T in "" "13=2" "13=3"; do
timeout 600 python tools/opbench.py --only agg_bwd --no-ref --tuning "$T" > $OUT/opbench_$T.jsonl 2> $OUT/opbench.err
python - <<PY
import json
for l in open("$OUT/opbench_$T.jsonl"):
    d=json.loads(l)
    if "cfg3" in d["case"]: print("tuning '$T'", d["case"][:64], d["us"])
PY
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_default_path_gpu.py -x -q -m gpu -k "aggregate or scatter" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
