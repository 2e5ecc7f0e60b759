#!/usr/bin/env bash
# launch-count reductions + side stream in fc_forward/backward: parity subset, bench A/B (key 28), aggregate forward sweep
set -uo pipefail
TAG="${1:-r4s}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fc_mfma_gpu.py tests/test_fc_wino_gpu.py tests/test_bench_shapes_gpu.py tests/test_trainer_gpu.py tests/test_face_step_gpu.py tests/test_bench_tools_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-legs > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
l=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("default", l["ms_per_step"], l["value"], {k:v["ms_per_step"] for k,v in l.get("variants",{}).items()})
PY
timeout 600 python bench.py --tuning 28=1 --no-cpu-baseline --no-legs --no-variants > $OUT/bench_noside.json 2> $OUT/bench_noside.err; echo "bench(no side) rc=$?"
python - <<PY
import json
l=json.loads(open("$OUT/bench_noside.json").read().strip().splitlines()[-1])
print("key 28=1", l["ms_per_step"], l["value"])
PY
timeout 600 python tools/bench_north_star.py --sweep agg --iters 20 > $OUT/agg_sweep.jsonl 2> $OUT/agg_sweep.err; echo "sweep rc=$?"
python tools/fmt_north_star.py $OUT/agg_sweep.jsonl 2>/dev/null | head -90
