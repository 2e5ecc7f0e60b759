#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/r5_s36; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-legs > $O/rocprof.log 2>&1)
cp /tmp/tr/bench_kernel_stats.csv $O/ 2>/dev/null
python tools/trace_steps.py /tmp/tr/bench_kernel_trace.csv "fc_tail_fwd_kernel<3>" > $O/steps.txt 2>&1; head -30 $O/steps.txt
