#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s27; mkdir -p $O
V=$GRAFT_REPO_ROOT/global_flow_local_attention_amd/variants
timeout 900 python -m pytest tests/test_big_plane_gpu.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { GFLA_HIP_LIBRARY=$2 python tools/bench_config2.py --tag "$1" ${3:+--tuning $3} --no-ref --flows smooth,zero,wild,expand,compress --out $O/config2.jsonl > /dev/null 2>$O/err_$1.log; }
run new "" ""
run base $V/libgfla_hip_base.so ""
run new "" ""
run base $V/libgfla_hip_base.so ""
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r5_s27/config2.jsonl")]
tags=[]
for r in rows:
    if r["tag"] not in tags: tags.append(r["tag"])
for op in sorted({r["op"] for r in rows if "block_extractor_bwd" in r["op"]}):
    for fl in ("smooth","zero","wild","expand","compress"):
        print("%-40s %-8s"%(op,fl)+"  ".join("%s %s"%(t,[r["us"] for r in rows if r["op"]==op and r["flow"]==fl and r["tag"]==t]) for t in tags))
PY
