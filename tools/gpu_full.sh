#!/usr/bin/env bash
# The round-end sequence the driver runs (full GPU test suite, smoke, default bench) + the north-star sweep.
# usage: gpurun --timeout 2400 -- 'bash tools/gpu_full.sh <tag>'
set -uo pipefail
TAG="${1:-full}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $OUT/pytest_gpu.log 2>&1 ) 2> $OUT/pytest.time; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | tail -30; tail -3 $OUT/pytest.time
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; tail -3 $OUT/bench.err; tail -3 $OUT/bench.time
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roofline", json.dumps(d.get("roofline"))[:300])
ns=d.get("north_star",{}).get("layers",{})
for k,v in ns.items(): print(k, {kk:(vv["us"],vv["frac"]) for kk,vv in v.items() if isinstance(vv,dict)})
print("legs", json.dumps({k:{kk:vv for kk,vv in v.items() if kk!='what'} for k,v in d.get("legs",{}).items()}))
print("oracle max_abs", d.get("oracle_check",{}).get("max_abs"))
PY
