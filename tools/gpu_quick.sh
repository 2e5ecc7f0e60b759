#!/usr/bin/env bash
# Short session: bench + steady-state kernel trace (no tests).  usage: gpurun -- 'bash tools/gpu_quick.sh <tag> [bench args]'
set -uo pipefail
TAG="${1:-q}"; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json; tail -3 $OUT/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OLDPWD/$OUT/rocprof_bench.log 2>&1); echo "rocprof rc=$?"
cp /tmp/prof_$TAG/bench_kernel_stats.csv $OUT/ 2>/dev/null
python tools/trace_steps.py /tmp/prof_$TAG/bench_kernel_trace.csv "fc_tail_fwd_kernel<3>" > $OUT/steady_state_steps.txt 2>&1; head -60 $OUT/steady_state_steps.txt
