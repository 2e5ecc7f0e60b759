#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r5a}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_fc_wino_gpu.py tests/test_face_step_gpu.py -x -q -m gpu -k "weight_gradient_forms or convert_many or ragged or two_job" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for T in "" "29=2" "29=1" "19=1"; do
timeout 600 python bench.py --tuning "$T" --no-cpu-baseline --no-legs --no-variants > $OUT/bench_$T.json 2> $OUT/bench_$T.err; echo "bench rc=$?"
python - <<PY
import json
l=json.loads(open("$OUT/bench_$T.json").read().strip().splitlines()[-1])
print("tuning '$T'", l["ms_per_step"], l["value"])
for k in l["kernels"]:
    if k["entry"]=="gfla_fc_backward_f32": print("  %-30s %-34s %8.1f us" % (k["entry"], k["dims"], k["avg_us"]))
for k in l.get("fc_kernels",[]):
    if "weight" in k["kernel"]: print("   ", k["kernel"][:80], k["avg_us"])
PY
done
