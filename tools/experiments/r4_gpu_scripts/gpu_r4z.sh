#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r4z}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_fc_mfma_gpu.py tests/test_fc_wino_gpu.py tests/test_face_step_gpu.py tests/test_gpu_parity.py tests/test_trainer_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 python tools/probe_face_host.py > $OUT/face_host.jsonl 2> $OUT/face_host.err; cat $OUT/face_host.jsonl
timeout 600 python bench.py --workload face_bf16 --no-cpu-baseline --no-legs --no-variants > $OUT/bench_face.json 2> $OUT/bench_face.err; echo "bench rc=$?"
python - <<PY
import json
l=json.loads(open("$OUT/bench_face.json").read().strip().splitlines()[-1])
print("face", l["ms_per_step"], l["value"], l.get("unit"))
PY
timeout 600 python bench.py --no-cpu-baseline --no-legs --no-variants > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
l=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("default", l["ms_per_step"], l["value"])
PY
