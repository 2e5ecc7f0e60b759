#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s42; mkdir -p $O
V=$GRAFT_REPO_ROOT/global_flow_local_attention_amd/variants
timeout 900 python -m pytest tests/test_big_plane_gpu.py -x -q -k "block_extractor or reproducible" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { GFLA_HIP_LIBRARY=$2 python tools/bench_config2.py --tag "$1" ${3:+--tuning $3} --no-ref --split --flows smooth,zero,wild,integer --out $O/config2.jsonl > /dev/null 2>$O/err_$1.log; }
for i in 1 2; do
run diet "" ""
run base $V/libgfla_hip_base.so ""
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r5_s42/config2.jsonl")]
tags=["diet","base"]
for op in sorted({r["op"] for r in rows if "block_extractor_bwd" in r["op"]}):
    for fl in ("smooth","zero","wild","integer"):
        print("%-40s %-8s"%(op,fl)+"  ".join("%s %s"%(t,[r["us"] for r in rows if r["op"]==op and r["flow"]==fl and r["tag"]==t]) for t in tags))
PY
