#!/bin/bash
# round 5, last pass on the final tree: evidence (tools/gpu_final.sh: default bench first, PMC tables, traces), then the whole GPU suite and smoke
cd "$GRAFT_REPO_ROOT"
bash tools/gpu_final.sh r5_final3 > gpurun_out/r5_final3.log 2>&1; tail -5 gpurun_out/r5_final3.log
O=gpurun_out/r5_last3; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --tb=short -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python -c "
import json; d=json.load(open('gpurun_out/r5_final3/bench.json')); print('step ms', d['ms_per_step'], 'value', d['value'], 'roofline', d['roofline']['frac'], d['roofline']['traffic_source'][-50:])
print({k: (v['pair']['frac'], v['block_extractor_fwd']['frac'], v['local_attn_fwd']['frac']) for k, v in d['north_star']['layers'].items()})
for r in d['legs']['config2_ops']['rows']:
    if r['flow'] == 'smooth': print(r['op'], r['us'], r['frac'], r.get('ref_us'))
print(d['legs']['config2_ops']['slower_than_reference_on'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print({k: v.get('ms', v.get('ms_per_step')) for k, v in d['legs'].items() if isinstance(v, dict) and k != 'config2_ops'})"
