#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r4y}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python tools/probe_face_host.py > $OUT/face_host.jsonl 2> $OUT/face_host.err; cat $OUT/face_host.jsonl; tail -2 $OUT/face_host.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/face_trace -o face -- python $OLDPWD/bench.py --workload face_bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-legs > $OUT/rocprof_face.log 2>&1); echo "trace rc=$?"
python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/face_trace/face_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
out=open("$OUT/face_kernel_stats.txt","w")
for r in rows[:45]:
    line="%9.2f ms %6s calls %8.1f us %5.1f%%  %s" % (float(r["TotalDurationNs"])/1e6, r["Calls"], float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot, r["Name"][:110])
    print(line); out.write(line+"\n")
print("total kernel ms", tot/1e6, "dispatches", sum(int(r["Calls"]) for r in rows))
PY
tail -3 $OUT/rocprof_face.log | cut -c1-400
