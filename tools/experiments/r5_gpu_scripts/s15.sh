#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s15; mkdir -p $O
timeout 900 python -m pytest tests/test_big_plane_gpu.py -q --maxfail=8 --tb=short -p no:cacheprovider -k "resample" > $O/pytest.log 2>&1; tail -8 $O/pytest.log
run() { python tools/bench_config2.py --tag "$1" --tuning "$2" --no-ref --flows smooth,zero,wild --out $O/config2.jsonl > /dev/null 2>&1; }
for t in "42=0" "42=1" "43=64" "43=96" "43=128" "31=4" "31=6" "31=7"; do run v $t; done
python - <<'PY'
import json
for l in open("gpurun_out/r5_s15/config2.jsonl"):
    r = json.loads(l)
    if "resample2d_fwd" in r["op"]:
        print("%-26s %-38s %-8s %7.1f us  frac %.3f" % (r["tuning"], r["op"], r["flow"], r["us"], r["frac"]))
PY
