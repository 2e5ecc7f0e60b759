#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s7; mkdir -p $O
timeout 900 python -m pytest tests/test_big_plane_gpu.py -q --maxfail=8 --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; tail -12 $O/pytest.log
run() { python tools/bench_config2.py --tag "$1" --tuning "$2" --no-ref --split --flows smooth,zero --out $O/config2.jsonl > /dev/null 2>&1; }
for t in "39=0" "39=1" "39=2" "39=4" "37=8" "37=32" "31=8,32=32,37=8" "31=8,32=64" "40=1" "40=2" "40=4" "35=4,36=32" "35=8,36=32,37=16" "35=8,36=32,10=32" "35=4,36=64,37=8" "35=8,36=16"; do run v $t; done
python - <<'PY'
import json
for l in open("gpurun_out/r5_s7/config2.jsonl"):
    r = json.loads(l)
    if r["flow"] == "zero" and r["tuning"] not in ("39=0",): continue
    print("%-26s %-38s %-8s %7.1f us  frac %.3f" % (r["tuning"], r["op"], r["flow"], r["us"], r["frac"]))
PY
