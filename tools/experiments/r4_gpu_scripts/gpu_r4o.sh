#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r4o}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fc_mfma_gpu.py tests/test_face_step_gpu.py -q --timeout=600 -k "bf16 or dual or face" > $OUT/pytest_bf16.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_bf16.log | tail -5; grep -E "^E " $OUT/pytest_bf16.log | head
for T in "30=0" "19=1"; do
  timeout 600 python bench.py --workload face_bf16 --batch 8 --steps 5 --warmup 2 --tuning "$T" > $OUT/face_$T.json 2> $OUT/face.err; echo "face [$T] rc=$?"
  python - <<PY
import json
d=json.load(open("$OUT/face_$T.json"))
print("tuning [$T] frames/s", d["value"], "ms", d["ms_per_step"])
PY
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_face -o face -- python $OLDPWD/bench.py --workload face_bf16 --batch 8 --steps 5 --warmup 2 > $OUT/rocprof_face.log 2>&1; echo "rocprof rc=$?"
python - <<PY
import csv, glob
f = glob.glob("/tmp/prof_face/**/face_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
out = open("$OUT/face_kernel_stats.txt", "w")
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    line = "%9.2f ms %6s calls %8.1f us %5.1f%%  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:110])
    print(line); out.write(line + "\n")
PY
