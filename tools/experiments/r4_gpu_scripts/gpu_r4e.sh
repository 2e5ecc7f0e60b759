#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r4e}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_default_path_gpu.py -q --timeout=600 -s -k "block_extractor or tiny_sigma" > $OUT/pytest_be.log 2>&1; echo "pytest be rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $OUT/pytest_be.log | tail -20
timeout 300 python tools/bench_north_star.py --iters 20 --sweep be > $OUT/sweep.jsonl 2> $OUT/sweep.err; echo "sweep rc=$?"; tail -2 $OUT/sweep.err
timeout 300 python tools/bench_north_star.py --iters 20 --sweep none --face > $OUT/face.jsonl 2>> $OUT/sweep.err
python tools/fmt_north_star.py $OUT/*.jsonl
timeout 200 python tools/bench_be_widths.py > $OUT/widths.jsonl 2>> $OUT/sweep.err; cut -c1-330 $OUT/widths.jsonl
