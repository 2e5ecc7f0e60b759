#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s6; mkdir -p $O
timeout 900 python -m pytest tests/test_big_plane_gpu.py -q --maxfail=8 --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; tail -5 $O/pytest.log
run() { python tools/bench_config2.py --tag "$1" --tuning "$2" --no-ref --flows smooth,zero --out $O/config2.jsonl > /dev/null 2>&1; }
for t in "39=0" "39=8" "39=4" "39=12" "39=2" "37=8" "37=8,39=4" "31=8,32=64" "31=8,32=64,39=4" "31=16,32=32,37=32" "35=8,36=32" "35=16,36=32" "35=8,36=64" "35=8,36=32,37=4"; do run vec $t; done
python - <<'PY'
import json
for l in open("gpurun_out/r5_s6/config2.jsonl"):
    r = json.loads(l)
    if "fwd" in r["op"]:
        print("%-14s %-26s %-38s %-8s %7.1f us  frac %.3f" % (r["tag"], r["tuning"], r["op"], r["flow"], r["us"], r["frac"]))
PY
