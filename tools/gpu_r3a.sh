#!/usr/bin/env bash
# Round-3 session A: whole GPU suite, default bench + trace, config-3 inference in modes 0 and 3 (labelled).
set -uo pipefail
TAG="${1:-r3a}"; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --maxfail=30 --timeout=600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest_gpu.log | cut -c1-300 | tail -30
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-260 $OUT/bench.json; tail -2 $OUT/bench.err | cut -c1-300
for M in 0 3; do timeout 200 python tools/bench_inference.py --fc-mode $M > $OUT/inference_mode$M.jsonl 2> $OUT/inference_mode$M.err; echo "inference mode $M rc=$?"; tail -1 $OUT/inference_mode$M.jsonl | cut -c1-400; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants "$@" > $OLDPWD/$OUT/rocprof_bench.log 2>&1); echo "rocprof rc=$?"
cp /tmp/prof_$TAG/bench_kernel_stats.csv $OUT/ 2>/dev/null
python tools/trace_steps.py /tmp/prof_$TAG/bench_kernel_trace.csv "fc_tail_fwd_kernel<3>" > $OUT/steady_state_steps.txt 2>&1; head -50 $OUT/steady_state_steps.txt
