#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s41; mkdir -p $O
export GFLA_HIP_LIBRARY=$GRAFT_REPO_ROOT/global_flow_local_attention_amd/variants/libgfla_hip_probes.so
for t in "" "39=1" "39=2" "39=4" "39=6" "39=8" ""; do
  python tools/bench_config2.py --tag "$t" ${t:+--tuning $t} --no-ref --split --flows smooth,zero --out $O/config2.jsonl > /dev/null 2>&1
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r5_s41/config2.jsonl")]
tags=[]
for r in rows:
    if r["tag"] not in tags: tags.append(r["tag"])
ops=sorted({r["op"] for r in rows})
print("%-8s"%"abl"+"".join("%16s"%o.replace("block_extractor","be").replace("resample2d","rs").replace(" only)",")")[:15] for o in ops))
for t in tags:
    print("%-8s"%t+"".join("%16s"%("/".join("%.0f"%([r["us"] for r in rows if r["op"]==o and r["flow"]==fl and r["tag"]==t]+[0])[0] for fl in ("smooth","zero"))) for o in ops))
PY
