#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s13; mkdir -p $O
timeout 900 python -m pytest tests/test_big_plane_gpu.py -q --maxfail=8 --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; tail -8 $O/pytest.log
run() { python tools/bench_config2.py --tag "$1" --tuning "$2" --no-ref --flows smooth,zero,wild --out $O/config2.jsonl > /dev/null 2>&1; }
for t in "39=0" "41=1" "37=4" "37=16" "35=16,36=32" "35=4,36=32"; do run v $t; done
python - <<'PY'
import json
for l in open("gpurun_out/r5_s13/config2.jsonl"):
    r = json.loads(l)
    if "fwd k5" in r["op"] or (r["tuning"] == "39=0"):
        print("%-26s %-38s %-8s %7.1f us  frac %.3f" % (r["tuning"], r["op"], r["flow"], r["us"], r["frac"]))
PY
