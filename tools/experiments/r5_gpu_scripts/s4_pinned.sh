#!/bin/bash
# round 5, GPU session 4: taps pinned (requests issued together), tighter LDS requests
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s4; mkdir -p $O
timeout 1200 python -m pytest tests/test_big_plane_gpu.py tests/test_default_path_gpu.py tests/test_gpu_parity.py -k "big_plane or config2 or resample" -q --maxfail=12 --tb=short -p no:cacheprovider > $O/pytest.log 2>&1
tail -12 $O/pytest.log
python tools/bench_config2.py --tag default --split --out $O/config2.jsonl > $O/config2_default.log 2>&1
run() { python tools/bench_config2.py --tag "$1" --tuning "$2" --no-ref --split --flows smooth,wild --out $O/config2.jsonl > /dev/null 2>&1; }
for t in "35=16,36=32" "35=8,36=64" "35=8,36=32" "35=4,36=64" "35=8,36=32,37=4" "35=8,36=32,37=16" "35=4,36=32,37=8" "35=8,36=44" "38=1"; do run be_fwd_tiles $t; done
for t in "31=16,32=32" "31=8,32=64" "31=8,32=32" "31=8,32=32,37=8,34=4" "31=8,32=32,37=32,34=16" "31=16,32=32,37=8,34=4" "31=8,32=44,37=16,34=8" "31=4,32=64,37=16"; do run tiles $t; done
python tools/opbench.py --only rs_fwd,rs_bwd --iters 20 --out $O/opbench_rs.jsonl > $O/opbench_rs.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r5_s4/config2.jsonl"):
    r = json.loads(l)
    if r["tag"] == "be_fwd_tiles" and "block_extractor_fwd" not in r["op"]: continue
    if r["tag"] == "tiles" and "block_extractor_fwd" in r["op"]: continue
    if r["tag"] == "default" and r["flow"] not in ("smooth", "zero", "wild"): continue
    print("%-14s %-26s %-38s %-8s %7.1f us  frac %.3f  ref %s err %s" % (r["tag"], r["tuning"], r["op"], r["flow"], r["us"], r["frac"], r.get("ref_us"), r.get("max_abs_vs_ref")))
for l in open("gpurun_out/r5_s4/opbench_rs.jsonl"):
    r = json.loads(l)
    print("opbench %-60s %7.1f us frac %.3f ref %s" % (r["case"], r["us"], r["frac_peak"], r["ref_us"]))
PY
