#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s21; mkdir -p $O
V=$GRAFT_REPO_ROOT/global_flow_local_attention_amd/variants
timeout 900 python -m pytest tests/test_big_plane_gpu.py -x -q -k "resample2d or reproducible" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { # tag lib tuning
  GFLA_HIP_LIBRARY=$2 python tools/bench_config2.py --tag "$1" ${3:+--tuning $3} --no-ref --flows smooth,wild,zero --out $O/config2.jsonl > /dev/null 2>$O/err_$1.log
}
run merged_eu4 "" ""
run split_eu4 "" "42=1"
run merged_eu2 $V/libgfla_hip_eu2.so ""
run split_eu2 $V/libgfla_hip_eu2.so "42=1"
run be4 $V/libgfla_hip_be4.so ""
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r5_s21/config2.jsonl")]
tags=[]
for r in rows:
    if r["tag"] not in tags: tags.append(r["tag"])
print(" "*36+"  ".join("%10s"%t for t in tags))
for op in sorted({r["op"] for r in rows}):
    for fl in ("smooth","wild","zero"):
        print("%-26s %-8s"%(op,fl)+"  ".join("%10.1f"%([r["us"] for r in rows if r["op"]==op and r["flow"]==fl and r["tag"]==t]+[float('nan')])[0] for t in tags))
PY
