#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s32; mkdir -p $O
for t in "" "40=1" "" "40=1" "" "40=1"; do
  python tools/bench_config2.py --tag "$t" ${t:+--tuning $t} --no-ref --flows smooth,zero,wild --out $O/config2.jsonl > /dev/null 2>&1
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r5_s32/config2.jsonl")]
for op in ("block_extractor_fwd k3","block_extractor_fwd k5"):
    for fl in ("smooth","zero","wild"):
        print(op, fl, " ".join("%s:%s"%(t or "dflt",[r["us"] for r in rows if r["op"]==op and r["flow"]==fl and r["tag"]==t]) for t in ("","40=1")))
PY
