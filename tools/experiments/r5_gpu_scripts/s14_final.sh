#!/bin/bash
# round 5, last pass: the whole GPU suite + smoke + default bench on the final tree
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_last; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --tb=short -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time python bench.py ) > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('step ms', d['ms_per_step'], 'value', d['value'], 'roofline', d['roofline']['frac'], d['roofline']['traffic_source'][-50:])
print({k: (v['pair']['frac'], v['block_extractor_fwd']['frac'], v['local_attn_fwd']['frac']) for k, v in d['north_star']['layers'].items()})
for r in d['legs']['config2_ops']['rows']:
    if r['flow'] == 'smooth': print(r['op'], r['us'], r['frac'], r.get('ref_us'))
print(d['legs']['config2_ops']['slower_than_reference_on'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print({k: v.get('ms', v.get('ms_per_step')) for k, v in d['legs'].items() if isinstance(v, dict) and k != 'config2_ops'})"; tail -3 $O/bench.err
