#!/usr/bin/env bash
# aggregate forward (results of a chunk in registers, no waits in the channel loop) + separable bilinear in block_extractor
set -uo pipefail
TAG="${1:-r4t}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_default_path_gpu.py tests/test_bench_shapes_gpu.py -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 python tools/bench_north_star.py --sweep none --iters 30 > $OUT/ns.jsonl 2> $OUT/ns.err; python tools/fmt_north_star.py $OUT/ns.jsonl
timeout 600 python tools/bench_north_star.py --sweep none --iters 30 --face > $OUT/ns_face.jsonl 2>> $OUT/ns.err; python tools/fmt_north_star.py $OUT/ns_face.jsonl
timeout 600 python bench.py --no-cpu-baseline --no-legs --no-variants > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
l=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("default", l["ms_per_step"], l["value"])
print({k:(v.get("us") if isinstance(v,dict) else v) for k,v in l.get("kernels",{}).items()})
PY
