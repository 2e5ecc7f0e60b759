#!/usr/bin/env bash
# Round-3 session B: trainer-step tests (f4 on hardware), the re-run of round A's failures, bench --workload trainer_step.
set -uo pipefail
TAG="${1:-r3b}"; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_trainer_gpu.py -m gpu -q -x --timeout=600 -s > $OUT/pytest_trainer.log 2>&1; echo "trainer pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|Error|worst" $OUT/pytest_trainer.log | cut -c1-300 | tail -20
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "bf16 or dispatch or tuning or default_fc or empty_flow" --timeout=600 > $OUT/pytest_bf16.log 2>&1; echo "bf16 pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|Error" $OUT/pytest_bf16.log | cut -c1-300 | tail -20
timeout 300 python bench.py --workload trainer_step --steps 5 --warmup 2 > $OUT/bench_trainer.json 2> $OUT/bench_trainer.err; echo "bench rc=$?"; cut -c1-1500 $OUT/bench_trainer.json; tail -3 $OUT/bench_trainer.err | cut -c1-300
