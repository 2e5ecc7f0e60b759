#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s17; mkdir -p $O
timeout 900 python -m pytest tests/test_big_plane_gpu.py -q --maxfail=8 --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; tail -5 $O/pytest.log
run() { python tools/bench_config2.py --tag "$1" --tuning "$2" --no-ref --flows smooth,zero,wild --out $O/config2.jsonl > /dev/null 2>&1; }
for t in "40=0" "40=1" "40=4" "40=4,37=16" "40=2,37=4" "40=4,10=48" "38=1"; do run v $t; done
python - <<'PY'
import json
for l in open("gpurun_out/r5_s17/config2.jsonl"):
    r = json.loads(l)
    if "fwd k5" in r["op"] or ("fwd" in r["op"] and r["tuning"] == "40=0"):
        print("%-26s %-38s %-8s %7.1f us  frac %.3f" % (r["tuning"], r["op"], r["flow"], r["us"], r["frac"]))
PY
