#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s39; mkdir -p $O
for t in "" "31=16,32=16,37=8" "31=16,32=16,37=16" "31=32,32=16,37=8" "31=32,32=16,37=16" "31=16,32=16,37=4" "31=8,32=16,37=16" "31=16,32=8,37=8" "31=32,32=8,37=8" "31=16,32=16,37=8" ""; do
  python tools/bench_config2.py --tag "$t" ${t:+--tuning $t} --no-ref --flows smooth,zero,wild --out $O/config2.jsonl > /dev/null 2>&1
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r5_s39/config2.jsonl")]
tags=[]
for r in rows:
    if r["tag"] not in tags: tags.append(r["tag"])
for t in tags:
    print("%-22s"%t+"  fwd sm/zero/wild: "+" | ".join("/".join("%.1f"%x for x in [r["us"] for r in rows if r["op"]=="resample2d_fwd k4" and r["flow"]==fl and r["tag"]==t]) for fl in ("smooth","zero","wild")))
PY
