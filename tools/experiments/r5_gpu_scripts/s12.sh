#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s12; mkdir -p $O
# the self-spawn launch with the real workload: two gloo ranks sharing the one GPU (test hooks GFLA_DIST_BACKEND / GFLA_DEVICE)
( time GFLA_DIST_BACKEND=gloo GFLA_DEVICE=0 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-legs ) > $O/bench_2rank_selfspawn.json 2> $O/bench_2rank_selfspawn.err; echo "rc=$?"; cut -c1-400 $O/bench_2rank_selfspawn.json; tail -5 $O/bench_2rank_selfspawn.err
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fc_mfma_gpu.py tests/test_bench_shapes_gpu.py -q --maxfail=8 --tb=short -p no:cacheprovider -k "beyond_the_lds or exactly_zero or market or reshape or resample" > $O/pytest.log 2>&1; tail -6 $O/pytest.log
( time python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('step ms', d['ms_per_step'], 'value', d['value']); print(d['cpu_baseline']); print(d['roofline']['traffic'], d['roofline']['traffic_source'][-60:])"; tail -3 $O/bench.err
python tools/opbench.py --only reshape --iters 50 > $O/opbench_reshape.log 2>&1; grep reshape $O/opbench_reshape.log | cut -c1-200
