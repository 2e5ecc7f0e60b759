#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r4v}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_default_path_gpu.py tests/test_bench_shapes_gpu.py tests/test_fc_mfma_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 python tools/opbench.py --only agg_fwd,agg_bwd --no-ref > $OUT/opbench.jsonl 2> $OUT/opbench.err
python - <<PY
import json
for l in open("$OUT/opbench.jsonl"):
    d=json.loads(l)
    if "x176" in d["case"] or "cfg3" in d["case"]: print(d["case"][:64], d["us"], d["frac_peak"])
PY
timeout 300 python tools/bench_north_star.py --sweep none --iters 30 > $OUT/ns.jsonl 2> $OUT/ns.err; python tools/fmt_north_star.py $OUT/ns.jsonl
timeout 600 python bench.py --no-cpu-baseline --no-legs --no-variants > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
l=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("default", l["ms_per_step"], l["value"])
for k in l["kernels"]: print("  %-46s %-34s %8.1f us" % (k["entry"], k["dims"], k["avg_us"]))
PY
