#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s23; mkdir -p $O
timeout 900 python -m pytest tests/test_big_plane_gpu.py -x -q -k "block_extractor or reproducible" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for t in "" "41=1" "" "41=1"; do
  python tools/bench_config2.py --tag "fold_$t" ${t:+--tuning $t} --no-ref --flows smooth,zero,wild,compress,expand --out $O/config2.jsonl > /dev/null 2>&1
done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r5_s23/config2.jsonl")]
for op in ("block_extractor_bwd k3","block_extractor_bwd k5"):
    for fl in ("smooth","zero","wild","compress","expand"):
        v=lambda t:[r["us"] for r in rows if r["op"]==op and r["flow"]==fl and r["tag"]==t]
        print("%-26s %-9s fold %s   plain %s"%(op,fl,v("fold_"),v("fold_41=1")))
PY
