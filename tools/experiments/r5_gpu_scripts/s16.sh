#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s16; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 --tb=short -p no:cacheprovider -k "aggregate or attn or extractor_attn or softmax or bench_shape" > $O/pytest.log 2>&1; tail -6 $O/pytest.log
python tools/bench_north_star.py --sweep none --iters 20 > $O/north_star.log 2>&1; tail -12 $O/north_star.log | cut -c1-400
python bench.py --no-cpu-baseline --no-variants --no-legs --steps 20 --warmup 5 > $O/bench_quick.json 2> $O/bench_quick.err; python -c "
import json; d=json.load(open('$O/bench_quick.json')); print('step ms', d['ms_per_step'], 'value', d['value'])"
