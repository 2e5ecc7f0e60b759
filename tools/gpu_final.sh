#!/usr/bin/env bash
# Round-end evidence in one GPU call: the default bench; HBM traffic counters of the bench step (-> profiles/pmc_traffic.json,
# which bench.py's roofline.traffic reads) and of the north star's two forward ops; MFMA-pipe utilisation of the FC kernels;
# LDS counters of the block_extractor forward and aggregation forward kernels; kernel trace of the default bench
# (steady-state per-step table).  Counter passes are separate rocprofv3 runs with --kernel-trace only.
# usage: gpurun --timeout 2400 -- 'bash tools/gpu_final.sh <tag>'
set -uo pipefail
TAG="${1:-final}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# the default bench FIRST: after rocprofv3 --pmc passes in the same job the f32-MFMA kernels run ~10 % slower for a while
# (the last stdout line is the short one; the full record is written to the detail file)
( time BENCH_DETAIL=$OUT/bench_detail.json timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json; cat $OUT/bench.time | tail -3
BENCH="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-legs"
NS="python $PWD/tools/bench_north_star.py --sweep none --iters 3"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/fin_$C -o pmc -- $BENCH > $OUT/pmc_$C.log 2>&1); echo "$C rc=$?"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/fin_ns_$C -o pmc -- $NS > $OUT/pmc_ns_$C.log 2>&1); echo "north star $C rc=$?"
done
F=$(find /tmp/fin_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find /tmp/fin_WRITE_SIZE -name "*counter_collection.csv" | head -1)
# (only gpurun_out/ travels back: copy $OUT/pmc_traffic.json to profiles/pmc_traffic.json in the build container afterwards)
python tools/pmc_summary.py "$F" "$W" $OUT/pmc_traffic.json "$TAG" > $OUT/pmc_traffic.txt 2>&1
grep -E "fc_conv|fc_wgrad|fc_wino|agg_|be_bwd|rs_lds|patch_" $OUT/pmc_traffic.txt | cut -c1-160 | head -40
F=$(find /tmp/fin_ns_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find /tmp/fin_ns_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py "$F" "$W" $OUT/north_star_pmc_traffic.json "$TAG north star" > $OUT/north_star_pmc_traffic.txt 2>&1; cat $OUT/north_star_pmc_traffic.txt | cut -c1-170
# BASELINE configs[1]: traffic of the big-plane kernels (smooth flow, no reference kernels in the traced process)
C2="python $PWD/tools/bench_config2.py --no-ref --flows smooth --iters 3"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/fin_c2_$C -o pmc -- $C2 > $OUT/pmc_c2_$C.log 2>&1); echo "config2 $C rc=$?"
done
F=$(find /tmp/fin_c2_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find /tmp/fin_c2_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py "$F" "$W" $OUT/config2_pmc_traffic.json "$TAG config2" > $OUT/config2_pmc_traffic.txt 2>&1; cat $OUT/config2_pmc_traffic.txt | cut -c1-170
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fin_c2_trace -o c2 -- python $OLDPWD/tools/bench_config2.py --no-ref --flows smooth --iters 10 > $OUT/rocprof_config2.log 2>&1); cp /tmp/fin_c2_trace/c2_kernel_stats.csv $OUT/config2_kernel_stats.csv 2>/dev/null
bash tools/gpu_pmc_fc.sh $TAG/mfma --no-variants --no-legs > $OUT/mfma.log 2>&1; grep -E "fc_conv|fc_wgrad|fc_wino" $OUT/mfma/pmc_set1.txt 2>/dev/null | cut -c1-250 | head -12
bash tools/gpu_pmc_lds.sh $TAG/ns_lds "_kernel" -- $NS > /dev/null 2>&1
grep -E "be_fwd|agg_" $OUT/ns_lds/pmc_summary.txt | cut -c1-330
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fin_trace -o bench -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-legs > $OUT/rocprof_bench.log 2>&1); echo "trace rc=$?"
cp /tmp/fin_trace/bench_kernel_stats.csv $OUT/ 2>/dev/null
python tools/trace_steps.py /tmp/fin_trace/bench_kernel_trace.csv "fc_tail_fwd_kernel<3>" > $OUT/steady_state_steps.txt 2>&1; head -50 $OUT/steady_state_steps.txt
