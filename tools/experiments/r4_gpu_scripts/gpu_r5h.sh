#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r5h}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_default_path_gpu.py -x -q -m gpu -k "aggregate or attn or extractor" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout 600 python tools/opbench.py --only agg_bwd,rs_bwd --no-ref > $OUT/opbench.jsonl 2> $OUT/opbench.err
python - <<PY
import json
for l in open("$OUT/opbench.jsonl"):
    d=json.loads(l)
    print(d["case"][:64], d["us"])
PY
timeout 600 python bench.py --no-cpu-baseline --no-legs --no-variants > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
l=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("default", l["ms_per_step"], l["value"])
for k in l["kernels"]:
    if "aggregate_bwd" in k["entry"]: print("  %-46s %-34s %8.1f us" % (k["entry"], k["dims"], k["avg_us"]))
PY
