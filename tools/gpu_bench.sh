#!/usr/bin/env bash
# The driver's bench invocation (default flags) + a 2-rank run of the same script on the one GPU (gloo ranks sharing device 0).
set -uo pipefail
TAG="${1:-bench}"; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
( time timeout 600 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json; tail -3 $OUT/bench.err | cut -c1-300; cat $OUT/bench.time
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "roofline", "variants", "oracle_check", "cpu_baseline") if k in d}, indent=1)[:3000])
for r in d.get("fc_kernels", []): print(r)
PY
GFLA_DIST_BACKEND=gloo GFLA_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-variants > $OUT/bench_2ranks.json 2> $OUT/bench_2ranks.err; echo "2-rank rc=$?"; cut -c1-300 $OUT/bench_2ranks.json; tail -3 $OUT/bench_2ranks.err | cut -c1-300
