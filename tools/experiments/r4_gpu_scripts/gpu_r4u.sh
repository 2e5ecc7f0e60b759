#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r4u}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python tools/bench_agg_abl.py > $OUT/agg_abl.jsonl 2> $OUT/agg_abl.err; cat $OUT/agg_abl.jsonl; tail -3 $OUT/agg_abl.err
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_default_path_gpu.py tests/test_bench_shapes_gpu.py -x -q -m gpu -k "aggregate or attn or resample or golden" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 600 python tools/opbench.py --only agg_fwd,agg_bwd,rs_bwd --no-ref > $OUT/opbench.jsonl 2> $OUT/opbench.err; cut -c1-400 $OUT/opbench.jsonl | head -30
timeout 600 python bench.py --no-cpu-baseline --no-legs --no-variants > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
l=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("default", l["ms_per_step"], l["value"]); print(json.dumps(l.get("kernels"))[:1500])
PY
