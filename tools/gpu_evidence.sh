#!/usr/bin/env bash
# Everything the round's claims rest on that is NOT a hardware counter, in one GPU call (run BEFORE tools/gpu_final.sh's
# counter passes: after rocprofv3 --pmc the f32-MFMA kernels run slower for a while):
#   full GPU test suite + smoke(), the bench-shape parity log, config-3 inference per arithmetic mode, the trainer-step
#   workload, the two-rank (gloo, one shared GPU) bench, the Winograd kernel's ablations and per-wave phase times.
# usage: gpurun --timeout 2400 -- 'bash tools/gpu_evidence.sh <tag>'
set -uo pipefail
TAG="${1:-evidence}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x --timeout=900 > $OUT/pytest_gpu.log 2>&1 ) 2> $OUT/pytest_gpu.time; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 600 python -m pytest tests/test_bench_shapes_gpu.py tests/test_fc_wino_gpu.py -m gpu -v --timeout=600 > $OUT/bench_shape_parity_pytest.log 2>&1; echo "bench-shape rc=$?"; tail -2 $OUT/bench_shape_parity_pytest.log
for M in 4 0; do
  timeout 300 python tools/bench_inference.py --fc-mode $M > $OUT/inference_config3_mode$M.jsonl 2> $OUT/inference_mode$M.err; echo "inference mode $M rc=$?"; tail -1 $OUT/inference_config3_mode$M.jsonl | cut -c1-300
done
timeout 600 python bench.py --workload trainer_step --no-variants --no-legs > $OUT/bench_trainer_step.json 2> $OUT/bench_trainer_step.err; echo "trainer_step rc=$?"; cut -c1-400 $OUT/bench_trainer_step.json
GFLA_DIST_BACKEND=gloo GFLA_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-legs > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "2-rank rc=$?"; cut -c1-300 $OUT/bench_2rank_gloo.json
GFLA_DIST_BACKEND=gloo GFLA_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --workload trainer_step --no-cpu-baseline --no-variants --no-legs > $OUT/bench_trainer_2rank_gloo.json 2> $OUT/bench_trainer_2rank_gloo.err; echo "2-rank trainer rc=$?"; cut -c1-300 $OUT/bench_trainer_2rank_gloo.json
# Winograd convolution (k = 5, forward of the source half = which 0): whole kernel, then with parts compiled out
{
  echo "fc_wino_conv_kernel<5> at C128 64x44 B=32 (tools/probe_wino.py, PROBE_K=5, 10 launches each; which 0..3 = forward t/s, data gradient t/s)"
  echo "tuning key 20 bits: 1 = no input transform, 2 = no MFMA multiply, 4 = no weight reloads, 8 = no staging"
  for D in 0 1 2 4 8 3 5 13; do
    echo "--- tuning 20 = $D"
    PROBE_K=5 timeout 120 python tools/probe_wino.py 4 10 20=$D 2>&1 | grep "which [0-3]"
  done
  echo "--- mode 0 (direct f32 MFMA convolution), same launches"
  PROBE_K=5 timeout 120 python tools/probe_wino.py 0 10 2>&1 | grep "which [0-3]"
  echo
  echo "fc_wino_wgrad_kernel<5>, same shapes (which 4 / 5 = weight gradient of the source / target half)"
  echo "tuning key 20 = 32 + bits: 1 no input transform, 2 no MFMAs / A reads, 4 no dY loads, 8 no lift of dY to the 36 points, 16 no raw staging"
  for D in 0 1 2 4 8 16 12 29; do
    echo "--- tuning 20 = 32 + $D"
    PROBE_K=5 timeout 120 python tools/probe_wino.py 4 10 20=$((32+D)) 2>&1 | grep "which [45]"
  done
} > $OUT/wino_ablations.txt 2>&1; tail -5 $OUT/wino_ablations.txt
timeout 120 python tools/probe_wino_phases.py > $OUT/wino_phases.txt 2>&1; cat $OUT/wino_phases.txt
