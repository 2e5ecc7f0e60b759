#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out/r5_s44; mkdir -p $O; export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_fc_mfma_gpu.py tests/test_fc_wino_gpu.py tests/test_bench_shapes_gpu.py tests/test_default_path_gpu.py -x -q -k "not block_extractor and not resample2d" ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for t in "" "43=1" "" "43=1"; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants --no-legs ${t:+--tuning $t} > $O/bench_$t.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_$t.json')); print('tuning [$t] step ms', d['ms_per_step'], [ (r['name'][:40], r.get('avg_us')) for r in d.get('kernels', []) if 'fc_backward' in r.get('name','')][:2])"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-legs > $O/rocprof.log 2>&1)
grep -E "fc_tail_bwd" /tmp/tr/bench_kernel_stats.csv | cut -c1-60,200-330
