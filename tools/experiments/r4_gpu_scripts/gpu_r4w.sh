#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r4w}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_fc_mfma_gpu.py tests/test_fc_wino_gpu.py tests/test_bench_shapes_gpu.py tests/test_default_path_gpu.py tests/test_trainer_gpu.py tests/test_face_step_gpu.py tests/test_bench_tools_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-legs --no-variants > $OUT/bench$i.json 2> $OUT/bench$i.err; echo "bench rc=$?"
python - <<PY
import json
l=json.loads(open("$OUT/bench$i.json").read().strip().splitlines()[-1])
print("default", l["ms_per_step"], l["value"])
for k in l["kernels"][:8]: print("  %-46s %-34s %8.1f us" % (k["entry"], k["dims"], k["avg_us"]))
PY
done
