#!/usr/bin/env bash
# Whole GPU suite + default bench + kernel trace.  usage: gpurun --timeout 1800 -- 'bash tools/gpu_all.sh <tag> [bench args]'
set -uo pipefail
TAG="${1:-all}"; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --maxfail=30 --timeout=900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest_gpu.log | cut -c1-300 | tail -30
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-260 $OUT/bench.json; tail -2 $OUT/bench.err | cut -c1-300
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants "$@" > $OLDPWD/$OUT/rocprof_bench.log 2>&1); echo "rocprof rc=$?"
cp /tmp/prof_$TAG/bench_kernel_stats.csv $OUT/ 2>/dev/null
python tools/trace_steps.py /tmp/prof_$TAG/bench_kernel_trace.csv "fc_tail_fwd_kernel<3>" > $OUT/steady_state_steps.txt 2>&1; head -50 $OUT/steady_state_steps.txt
