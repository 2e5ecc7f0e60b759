#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s33; mkdir -p $O
for t in "" "31=8,32=32" "31=8,32=32,34=4" "31=8,32=32,37=8" "34=4" "37=8" "31=8,32=32,34=4,37=8" "31=8,32=64,34=4" "31=4,32=64,34=4" ""; do
  python tools/bench_config2.py --tag "$t" ${t:+--tuning $t} --no-ref --split --flows smooth,zero,wild --out $O/config2.jsonl > /dev/null 2>&1
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r5_s33/config2.jsonl")]
tags=[]
for r in rows:
    if r["tag"] not in tags: tags.append(r["tag"])
ops=sorted({r["op"] for r in rows if "resample" in r["op"]})
print("%-26s"%"tuning"+"".join("%28s"%(o.replace("resample2d_","")[:24]) for o in ops))
for t in tags:
    print("%-26s"%t+"".join("%28s"%(" ".join("/".join("%.1f"%x for x in [r["us"] for r in rows if r["op"]==o and r["flow"]==fl and r["tag"]==t]) for fl in ("smooth","zero","wild"))) for o in ops))
PY
