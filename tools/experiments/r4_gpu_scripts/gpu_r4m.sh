#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r4m}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
( time timeout 900 python bench.py --no-legs --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; tail -2 $OUT/bench.err | cut -c1-300
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"])
print("variants", json.dumps({k:v["ms_per_step"] for k,v in d.get("variants",{}).items()}))
PY
timeout 900 python tools/opbench.py --iters 20 --out $OUT/opbench.jsonl > $OUT/opbench.log 2>&1; echo "opbench rc=$?"; grep -E "agg_fwd|agg_bwd" $OUT/opbench.log | cut -c1-260 | head -20
