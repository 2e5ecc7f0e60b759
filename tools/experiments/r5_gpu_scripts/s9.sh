#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s9; mkdir -p $O
timeout 900 python -m pytest tests/test_big_plane_gpu.py -q --maxfail=8 --tb=short -p no:cacheprovider > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python tools/bench_config2.py --tag default --split --out $O/config2.jsonl > $O/config2_default.log 2>&1
run() { python tools/bench_config2.py --tag "$1" --tuning "$2" --no-ref --split --flows smooth --out $O/config2.jsonl > /dev/null 2>&1; }
for t in "39=2" "39=4" "37=8" "37=32" "40=4" "34=4"; do run v $t; done
python - <<'PY'
import json
for l in open("gpurun_out/r5_s9/config2.jsonl"):
    r = json.loads(l)
    print("%-10s %-12s %-38s %-12s %7.1f us  frac %.3f ref %s err %s" % (r["tag"], r["tuning"], r["op"], r["flow"], r["us"], r["frac"], r.get("ref_us"), r.get("max_abs_vs_ref")))
PY
tail -2 $O/config2_default.log
