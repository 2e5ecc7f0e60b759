#!/usr/bin/env bash
# Round 4, call A: the new block_extractor forward kernel -- sweep over its launch geometry, parity tests of the default
# path on rough flows, the existing block_extractor tests.
# usage: gpurun --timeout 900 -- 'bash tools/gpu_r4a.sh <tag>'
set -uo pipefail
TAG="${1:-r4a}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python tools/bench_north_star.py --iters 20 --sweep be > $OUT/north_star_sweep.jsonl 2> $OUT/north_star_sweep.err; echo "sweep rc=$?"; tail -3 $OUT/north_star_sweep.err
head -4 $OUT/north_star_sweep.jsonl | cut -c1-400
timeout 200 python tools/bench_north_star.py --iters 20 --sweep none --face > $OUT/north_star_face.jsonl 2>> $OUT/north_star_sweep.err; cat $OUT/north_star_face.jsonl | cut -c1-400
timeout 200 python tools/bench_north_star.py --iters 20 --sweep none --flow zero > $OUT/north_star_zero.jsonl 2>> $OUT/north_star_sweep.err; cat $OUT/north_star_zero.jsonl | cut -c1-400
timeout 900 python -m pytest tests/test_default_path_gpu.py -q -x --timeout=600 -s > $OUT/pytest_default_path.log 2>&1; echo "pytest default path rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $OUT/pytest_default_path.log | tail -30
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "block_extractor or config2 or config3 or golden or real_reference or empty or 2_to_31" --timeout=600 > $OUT/pytest_be.log 2>&1; echo "pytest be rc=$?"; tail -5 $OUT/pytest_be.log
