#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s8; mkdir -p $O
timeout 900 python -m pytest tests/test_big_plane_gpu.py -q --maxfail=8 --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; tail -5 $O/pytest.log
run() { python tools/bench_config2.py --tag "$1" --tuning "$2" --no-ref --split --flows smooth --out $O/config2.jsonl > /dev/null 2>&1; }
for t in "39=0" "39=1" "39=2" "39=4" "39=6" "39=8" "39=14" "34=4" "34=16" "34=4,39=4" "34=16,39=4" "31=16,32=64" "31=8,32=64" "36=64"; do run v $t; done
python - <<'PY'
import json
for l in open("gpurun_out/r5_s8/config2.jsonl"):
    r = json.loads(l)
    print("%-26s %-38s %-8s %7.1f us  frac %.3f" % (r["tuning"], r["op"], r["flow"], r["us"], r["frac"]))
PY
