#!/usr/bin/env bash
# One gpurun session: goldens from the real reference kernels, GPU parity tests, op bench, bench, rocprof.
# usage: gpurun --timeout 1500 -- 'bash tools/gpu_session.sh <tag>'
set -uo pipefail
TAG="${1:-s}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== env" | tee $OUT/env.txt
(nproc; rocm-smi --showproductname 2>/dev/null | head -8; rocminfo | grep -E "Marketing Name|Compute Unit|gfx" | head -6) >> $OUT/env.txt 2>&1
echo "== golden"
timeout 300 python tests/golden/make_ref_golden.py $OUT/ref_golden.npz > $OUT/golden.log 2>&1; echo "golden rc=$?"
echo "== pytest gpu"
timeout 900 python -m pytest tests -m gpu -q --maxfail=40 --timeout=600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_gpu.log | tail -45
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
echo "== diag conv (skipped)"
true
echo "== opbench"
timeout 600 python tools/opbench.py --iters 20 ${OPBENCH_ARGS:-} --out $OUT/opbench.jsonl > $OUT/opbench.log 2>&1; echo "opbench rc=$?"; tail -5 $OUT/opbench.log
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== rocprof"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OLDPWD/$OUT/rocprof_bench.log 2>&1); echo "rocprof rc=$?"
find /tmp/prof_$TAG -name "*stats*.csv" -exec cp {} $OUT/ \; 2>/dev/null; find /tmp/prof_$TAG -type f | head -20; cat $OUT/*kernel_stats.csv 2>/dev/null | head -30 | cut -c1-220
python tools/trace_steps.py /tmp/prof_$TAG/bench_kernel_trace.csv "fc_tail_fwd_kernel<3>" > $OUT/steady_state_steps.txt 2>&1; head -45 $OUT/steady_state_steps.txt
echo "== pmc (HBM traffic of the gfla kernels; separate passes, kernel-trace only)"
for CNT in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d /tmp/pmc_${TAG}_$CNT -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/pmc_$CNT.log 2>&1); echo "pmc $CNT rc=$?"
  find /tmp/pmc_${TAG}_$CNT -name "*counter_collection.csv" -exec cp {} $OUT/pmc_${CNT}_counter_collection.csv \; 2>/dev/null
done
python tools/pmc_summary.py $OUT/pmc_FETCH_SIZE_counter_collection.csv $OUT/pmc_WRITE_SIZE_counter_collection.csv $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1; cat $OUT/pmc_traffic.txt
ls -la $OUT
