#!/usr/bin/env bash
# FC-on-MFMA session: its parity tests, then the bench in each arithmetic mode, then a kernel trace of the default.
# usage: gpurun --timeout 1200 -- 'bash tools/gpu_fc.sh <tag> [modes...]'
set -uo pipefail
TAG="${1:-fc}"; shift || true
MODES="${*:-0 3 2}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fc_mfma_gpu.py tests/test_bench_shapes_gpu.py -q -s --maxfail=60 --timeout=600 > $OUT/pytest_fc.log 2>&1; echo "pytest fc rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_fc.log | cut -c1-250 | tail -30
for M in $MODES; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --fc-mode $M > $OUT/bench_mfma_$M.json 2> $OUT/bench_mfma_$M.err
  echo "bench mode $M rc=$?"; cut -c1-260 $OUT/bench_mfma_$M.json; tail -2 $OUT/bench_mfma_$M.err | cut -c1-300
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants > $OLDPWD/$OUT/rocprof_bench.log 2>&1); echo "rocprof rc=$?"
cp /tmp/prof_$TAG/bench_kernel_stats.csv $OUT/ 2>/dev/null
python tools/trace_steps.py /tmp/prof_$TAG/bench_kernel_trace.csv "fc_tail_fwd_kernel<3>" > $OUT/steady_state_steps.txt 2>&1; head -50 $OUT/steady_state_steps.txt
