#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s18; mkdir -p $O
run() { python tools/bench_config2.py --tag "$1" --tuning "$2" --no-ref --flows smooth,zero --out $O/config2.jsonl > /dev/null 2>&1; }
run default ""
GFLA_HIP_LIBRARY=$PWD/global_flow_local_attention_amd/libgfla_hip_roll3.so run roll3 ""
GFLA_HIP_LIBRARY=$PWD/global_flow_local_attention_amd/libgfla_hip_roll3.so run roll3 "40=4"
python - <<'PY'
import json
for l in open("gpurun_out/r5_s18/config2.jsonl"):
    r = json.loads(l)
    if "block_extractor_fwd" in r["op"]:
        print("%-10s %-10s %-38s %-8s %7.1f us  frac %.3f" % (r["tag"], r["tuning"], r["op"], r["flow"], r["us"], r["frac"]))
PY
