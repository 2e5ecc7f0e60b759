#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s35; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fc_wino_gpu.py tests/test_fc_mfma_gpu.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --no-legs > $O/bench.json 2> $O/bench.err; cut -c1-260 $O/bench.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-legs > $O/rocprof.log 2>&1)
cp /tmp/tr/bench_kernel_stats.csv $O/ 2>/dev/null
python tools/trace_steps.py /tmp/tr/bench_kernel_trace.csv "fc_tail_fwd_kernel<3>" > $O/steps.txt 2>&1; head -24 $O/steps.txt
