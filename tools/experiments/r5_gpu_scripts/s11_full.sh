#!/bin/bash
# round 5: the whole GPU suite, smoke, then the round-end evidence (tools/gpu_final.sh)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_full; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --tb=short -p no:cacheprovider ) > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash tools/gpu_final.sh r5_final > $O/gpu_final.log 2>&1; tail -60 $O/gpu_final.log
