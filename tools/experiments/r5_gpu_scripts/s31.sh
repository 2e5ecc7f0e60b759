#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s31; mkdir -p $O
for t in "" "37=4" "37=16" "35=16,36=32" "35=4,36=64" "35=8,36=64" "40=1" "40=4" "10=32" "10=48" "35=16,36=32,37=4" "35=4,36=32" "35=4,36=32,37=16" ""; do
  python tools/bench_config2.py --tag "$t" ${t:+--tuning $t} --no-ref --flows smooth,zero --out $O/config2.jsonl > /dev/null 2>&1
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r5_s31/config2.jsonl")]
tags=[]
for r in rows:
    if r["tag"] not in tags: tags.append(r["tag"])
ops=sorted({r["op"] for r in rows if "fwd" in r["op"]})
print("%-22s"%"tuning"+"".join("%30s"%(o[:20]+" sm/zero") for o in ops))
for t in tags:
    print("%-22s"%t+"".join("%30s"%(" ".join("%.1f/%.1f"%(a,b) for a,b in zip([r["us"] for r in rows if r["op"]==o and r["flow"]=="smooth" and r["tag"]==t],[r["us"] for r in rows if r["op"]==o and r["flow"]=="zero" and r["tag"]==t]))) for o in ops))
PY
