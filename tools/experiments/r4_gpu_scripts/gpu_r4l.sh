#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r4l}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_face_step_gpu.py tests/test_trainer_gpu.py tests/test_bench_tools_gpu.py tests/test_fc_mfma_gpu.py -q --timeout=600 > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_sel.log | tail; grep -E "^E " $OUT/pytest_sel.log | head -20
( time timeout 900 python bench.py --no-variants > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"])
print("legs", json.dumps({k:{kk:(vv if not isinstance(vv,(dict,list)) else '...') for kk,vv in v.items() if kk!='what'} for k,v in d.get("legs",{}).items()}))
print("oracle max_rel", d.get("oracle_check",{}).get("max_rel"))
PY
