#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s19; mkdir -p $O
python tools/bench_config2.py --tag probe --no-ref --split --flows smooth,zero,expand,compress --out $O/config2.jsonl > /dev/null 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r5_s19/config2.jsonl"):
    r = json.loads(l)
    if "bwd" in r["op"]:
        print("%-38s %-8s %7.1f us  frac %.3f" % (r["op"], r["flow"], r["us"], r["frac"]))
PY
