#!/bin/bash
# round 5, GPU session 3: pipelined staging, tile-sized LDS requests; sweeps with split backward timings
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s3; mkdir -p $O
timeout 1200 python -m pytest tests/test_big_plane_gpu.py -q --maxfail=12 --tb=short -p no:cacheprovider > $O/pytest_big_plane.log 2>&1
tail -15 $O/pytest_big_plane.log
python tools/bench_config2.py --tag default --split --out $O/config2.jsonl > $O/config2_default.log 2>&1
run() { python tools/bench_config2.py --tag "$1" --tuning "$2" --no-ref --split --flows smooth,wild --out $O/config2.jsonl > /dev/null 2>&1; }
for t in "35=16,36=32" "35=8,36=64" "35=8,36=32" "35=4,36=64" "35=4,36=88" "35=2,36=176" "35=8,36=64,37=4" "35=8,36=64,37=16" "35=8,36=32,37=16" "38=1"; do run be_fwd_tiles $t; done
for t in "31=16,32=32" "31=8,32=64" "31=8,32=32" "31=4,32=64" "31=8,32=32,37=8,34=4" "31=8,32=32,37=32,34=16" "31=16,32=32,37=8" "31=8,32=64,37=8,34=4" "31=16,32=16" "10=128"; do run tiles $t; done
python - <<'PY'
import json
for l in open("gpurun_out/r5_s3/config2.jsonl"):
    r = json.loads(l)
    if r["tag"] == "be_fwd_tiles" and "fwd" not in r["op"]: continue
    if r["tag"] == "tiles" and "block_extractor_fwd" in r["op"]: continue
    print("%-14s %-26s %-38s %-12s %7.1f us  frac %.3f  ref %s err %s" % (r["tag"], r["tuning"], r["op"], r["flow"], r["us"], r["frac"], r.get("ref_us"), r.get("max_abs_vs_ref")))
PY
