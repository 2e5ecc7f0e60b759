#!/usr/bin/env bash
# Kernel trace of the step with everything on ONE stream: per-kernel durations that no second stream inflates (the anatomy
# of profiles/r4_final_steady_state_steps.txt is of the two-stream headline).  usage: gpurun -- 'bash tools/gpu_trace_one_stream.sh <tag>'
set -uo pipefail
TAG="${1:-trace1}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/t1 -o bench -- python $OLDPWD/bench.py --one-stream --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-legs > $OUT/rocprof_bench.log 2>&1); echo "trace rc=$?"
python tools/trace_steps.py /tmp/t1/bench_kernel_trace.csv "fc_tail_fwd_kernel<3>" > $OUT/steady_state_steps_one_stream.txt 2>&1; head -60 $OUT/steady_state_steps_one_stream.txt
