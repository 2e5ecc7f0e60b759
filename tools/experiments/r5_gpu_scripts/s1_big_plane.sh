#!/bin/bash
# round 5, GPU session 1: parity of the big-plane kernels + first timings of BASELINE configs[1] with geometry sweeps
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s1; mkdir -p $O
timeout 1200 python -m pytest tests/test_big_plane_gpu.py -q --maxfail=12 --tb=short -p no:cacheprovider > $O/pytest_big_plane.log 2>&1
tail -40 $O/pytest_big_plane.log
python tools/bench_config2.py --tag default --out $O/config2.jsonl > $O/config2_default.log 2>&1
python tools/bench_config2.py --tag round1_windowed --tuning 30=1 --no-ref --flows smooth,wild --out $O/config2.jsonl > /dev/null 2>&1
for t in 33=2 33=4 33=8 33=16 33=32 33=64; do python tools/bench_config2.py --tag gather_cpw --tuning $t --no-ref --flows smooth,wild --out $O/config2.jsonl > /dev/null 2>&1; done
for t in "31=8,32=64" "31=11,32=44" "31=32,32=16" "31=5,32=88" "31=16,32=32,34=2" "31=16,32=32,34=8" "31=11,32=44,34=8" "31=16,32=32,10=128" "31=11,32=44,34=16,10=128" "31=2,32=176,34=8" ; do
  python tools/bench_config2.py --tag scatter_tiles --tuning $t --no-ref --flows smooth,wild --out $O/config2.jsonl > /dev/null 2>&1; done
python - <<'PY'
import json
for l in open("gpurun_out/r5_s1/config2.jsonl"):
    r = json.loads(l)
    print("%-16s %-28s %-26s %-12s %7.1f us  frac %.3f  ref %s err %s" % (r["tag"], r["tuning"], r["op"], r["flow"], r["us"], r["frac"], r.get("ref_us"), r.get("max_abs_vs_ref")))
PY
