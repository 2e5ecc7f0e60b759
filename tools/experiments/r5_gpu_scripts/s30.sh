#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s30; mkdir -p $O
for t in "" "34=4" "34=16" "31=8,32=32" "31=8,32=64" "31=32,32=16" "31=8,32=32,34=4" "31=8,32=32,34=16" "10=96" "10=48"; do
  python tools/bench_config2.py --tag "$t" ${t:+--tuning $t} --no-ref --flows smooth,zero --out $O/config2.jsonl > /dev/null 2>&1
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r5_s30/config2.jsonl")]
tags=[]
for r in rows:
    if r["tag"] not in tags: tags.append(r["tag"])
ops=sorted({r["op"] for r in rows if "bwd" in r["op"]})
print("%-22s"%"tuning"+"".join("%30s"%(o[:16]+" sm/zero") for o in ops))
for t in tags:
    print("%-22s"%t+"".join("%30s"%("%.1f / %.1f"%tuple([r["us"] for r in rows if r["op"]==o and r["flow"]==fl and r["tag"]==t][0] for fl in ("smooth","zero"))) for o in ops))
PY
