#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r4p}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fc_mfma_gpu.py tests/test_face_step_gpu.py tests/test_gpu_parity.py -q --timeout=600 -k "bf16 or dual or face or blend" > $OUT/pytest_bf16.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_bf16.log | tail -5; grep -E "^E " $OUT/pytest_bf16.log | head
for T in "30=0" "19=1"; do
  timeout 600 python bench.py --workload face_bf16 --batch 8 --steps 5 --warmup 2 --tuning "$T" > $OUT/face_$T.json 2> $OUT/face.err; echo "face [$T] rc=$?"
  python - <<PY
import json
d=json.load(open("$OUT/face_$T.json"))
print("tuning [$T] frames/s", d["value"], "ms", d["ms_per_step"])
PY
done
timeout 600 python bench.py --workload face_bf16 --batch 8 --steps 5 --warmup 2 --face-one-stream > $OUT/face_one.json 2>> $OUT/face.err; python -c "
import json; d=json.load(open('$OUT/face_one.json')); print('one stream: ms', d['ms_per_step'])"
