#!/bin/bash
# round 5, GPU session 2: the LDS-window gathers; d/dflow mismatch localisation; sweeps
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s2; mkdir -p $O
python tools/experiments/r5_gpu_scripts/debug_gflow.py > $O/debug_gflow.log 2>&1; cat $O/debug_gflow.log
timeout 1200 python -m pytest tests/test_big_plane_gpu.py -q --maxfail=12 --tb=short -p no:cacheprovider > $O/pytest_big_plane.log 2>&1
tail -25 $O/pytest_big_plane.log
python tools/bench_config2.py --tag default --out $O/config2.jsonl > $O/config2_default.log 2>&1
run() { python tools/bench_config2.py --tag "$1" --tuning "$2" --no-ref --flows smooth,wild --out $O/config2.jsonl > /dev/null 2>&1; }
for t in "35=1,36=176" "35=2,36=176" "35=16,36=32" "35=8,36=64" "35=4,36=88" "35=2,36=176,37=4" "35=2,36=176,37=16" "35=2,36=176,10=96" "35=2,36=176,10=32"; do run be_fwd_tiles $t; done
for t in "37=8" "37=32" "37=64" "31=8,32=64" "31=8,32=32" "10=32" "10=96" "31=8,32=32,37=32" "31=16,32=32,37=16,10=48"; do run rs_gather_tiles $t; done
for t in "34=4" "34=16" "34=16,10=128" "31=8,32=64,34=8" "31=11,32=44,34=8" "31=11,32=44,34=16,10=128" "31=8,32=32,34=8"; do run scatter_tiles $t; done
python - <<'PY'
import json
for l in open("gpurun_out/r5_s2/config2.jsonl"):
    r = json.loads(l)
    sel = {"be_fwd_tiles": "block_extractor_fwd", "rs_gather_tiles": "resample2d", "scatter_tiles": "bwd"}.get(r["tag"], "")
    if sel in r["op"]:
        print("%-16s %-28s %-26s %-12s %7.1f us  frac %.3f  ref %s err %s" % (r["tag"], r["tuning"], r["op"], r["flow"], r["us"], r["frac"], r.get("ref_us"), r.get("max_abs_vs_ref")))
PY
