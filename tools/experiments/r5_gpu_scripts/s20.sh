#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s20; mkdir -p $O
timeout 900 python -m pytest tests/test_big_plane_gpu.py -x -q -k "block_extractor or reproducible" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for t in "" "41=1" "34=4" "34=16"; do
  python tools/bench_config2.py --tag "fix_$t" ${t:+--tuning $t} --no-ref --flows smooth,zero,wild,compress --out $O/config2.jsonl > /dev/null 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r5_s20/config2.jsonl"):
    r = json.loads(l)
    if "be_bwd" in r["op"]:
        print("%-12s %-30s %-8s %7.1f us  frac %.3f" % (r.get("tag"), r["op"], r["flow"], r["us"], r["frac"]))
PY
