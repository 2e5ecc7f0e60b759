#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s5; mkdir -p $O
run() { python tools/bench_config2.py --tag "$1" --tuning "$2" --no-ref --flows smooth,zero --out $O/config2.jsonl > /dev/null 2>&1; }
for t in "39=0" "39=1" "39=2" "39=4" "39=6" "39=0,37=8" "39=1,37=8" "39=2,37=8" "39=4,37=8" "39=6,37=8" "39=0,37=64" "39=1,37=64" ; do run abl $t; done
python - <<'PY'
import json
for l in open("gpurun_out/r5_s5/config2.jsonl"):
    r = json.loads(l)
    if "resample2d" in r["op"]:
        print("%-14s %-26s %-38s %-8s %7.1f us" % (r["tag"], r["tuning"], r["op"], r["flow"], r["us"]))
PY
