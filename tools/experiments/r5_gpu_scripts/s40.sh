#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s40; mkdir -p $O
for t in "" "31=16,32=16" "31=16,32=16,34=4" "31=32,32=16" "31=32,32=16,34=4" "31=8,32=16" "35=16,36=16" "35=8,36=16" "35=32,36=16" "35=16,36=16,37=4" "35=16,36=16,37=16" ""; do
  python tools/bench_config2.py --tag "$t" ${t:+--tuning $t} --no-ref --split --flows smooth,zero,wild --out $O/config2.jsonl > /dev/null 2>&1
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r5_s40/config2.jsonl")]
tags=[]
for r in rows:
    if r["tag"] not in tags: tags.append(r["tag"])
ops=["block_extractor_fwd k3","block_extractor_fwd k5","block_extractor_bwd k3","block_extractor_bwd k5","resample2d_bwd k4 (input1 only)","resample2d_bwd k4 (input2 only)","resample2d_fwd k4"]
print("%-22s"%"tuning"+"".join("%24s"%o.replace("block_extractor","be").replace("resample2d","rs")[:22] for o in ops))
for t in tags:
    print("%-22s"%t+"".join("%24s"%("/".join("%.0f"%([r["us"] for r in rows if r["op"]==o and r["flow"]==fl and r["tag"]==t]+[0])[0] for fl in ("smooth","zero","wild"))) for o in ops))
PY
