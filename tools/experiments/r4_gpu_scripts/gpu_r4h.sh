#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r4h}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_default_path_gpu.py tests/test_bench_tools_gpu.py -q --timeout=600 -s > $OUT/pytest_new.log 2>&1; echo "pytest new rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert|sigma" $OUT/pytest_new.log | tail -30
timeout 300 python tools/bench_north_star.py --iters 20 --sweep none > $OUT/ns.jsonl 2> $OUT/ns.err
timeout 300 python tools/bench_north_star.py --iters 20 --sweep none --face > $OUT/ns_face.jsonl 2>> $OUT/ns.err
python tools/fmt_north_star.py $OUT/ns*.jsonl
timeout 600 python bench.py --steps 10 --warmup 3 --no-variants > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
for k in ("value","ms_per_step","hipgraph_step","north_star","oracle_check","cpu_baseline","vendor_fallback_calls"):
    print(k, json.dumps(d.get(k))[:1500])
print("legs", json.dumps({k:{kk:vv for kk,vv in v.items() if kk!='what'} for k,v in d.get("legs",{}).items()}))
PY
