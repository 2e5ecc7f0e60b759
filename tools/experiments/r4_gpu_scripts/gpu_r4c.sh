#!/usr/bin/env bash
# Round 4: ablations / sweeps of the block_extractor forward kernel + the default-path parity tests.
set -uo pipefail
TAG="${1:-r4c}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python tools/bench_north_star.py --iters 20 --sweep abl > $OUT/abl.jsonl 2> $OUT/abl.err; echo "abl rc=$?"; tail -2 $OUT/abl.err
timeout 300 python tools/bench_north_star.py --iters 20 --sweep be > $OUT/sweep.jsonl 2>> $OUT/abl.err; echo "sweep rc=$?"
timeout 300 python tools/bench_north_star.py --iters 20 --sweep abl --face > $OUT/abl_face.jsonl 2>> $OUT/abl.err; echo "abl face rc=$?"
python tools/fmt_north_star.py $OUT/*.jsonl
timeout 900 python -m pytest tests/test_default_path_gpu.py -q --timeout=600 -s > $OUT/pytest_default_path.log 2>&1; echo "pytest default path rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $OUT/pytest_default_path.log | tail -30
