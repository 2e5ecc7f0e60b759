#!/usr/bin/env bash
# FC-on-MFMA session: its parity tests, then the bench in each arithmetic mode + the vendor-library path, then a kernel trace.
# usage: gpurun --timeout 1200 -- 'bash tools/gpu_fc.sh <tag>'
set -uo pipefail
TAG="${1:-fc}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fc_mfma_gpu.py -q -s --maxfail=60 --timeout=300 > $OUT/pytest_fc.log 2>&1; echo "pytest fc rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_fc.log | tail -30
for V in "mfma 0" "mfma 3" "mfma 2" "library 0"; do
  set -- $V
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --fc-impl $1 --fc-mode $2 > $OUT/bench_$1_$2.json 2> $OUT/bench_$1_$2.err
  echo "bench $1 mode $2 rc=$?"; cut -c1-260 $OUT/bench_$1_$2.json; tail -2 $OUT/bench_$1_$2.err
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OLDPWD/$OUT/rocprof_bench.log 2>&1); echo "rocprof rc=$?"
cp /tmp/prof_$TAG/bench_kernel_stats.csv $OUT/ 2>/dev/null
python tools/trace_steps.py /tmp/prof_$TAG/bench_kernel_trace.csv "fc_tail_fwd_kernel<3>" > $OUT/steady_state_steps.txt 2>&1; head -70 $OUT/steady_state_steps.txt
