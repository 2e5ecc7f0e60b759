#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r4q}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_default_path_gpu.py tests/test_face_step_gpu.py -q --timeout=600 -k "block_extractor or blend" > $OUT/pytest_be.log 2>&1; echo "pytest rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest_be.log | tail -5
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "block_extractor or config2 or config3 or golden or real_reference" --timeout=600 > $OUT/pytest_be2.log 2>&1; echo "pytest be2 rc=$?"; tail -2 $OUT/pytest_be2.log
timeout 300 python tools/bench_north_star.py --iters 20 --sweep none > $OUT/ns.jsonl 2> $OUT/ns.err
timeout 300 python tools/bench_north_star.py --iters 20 --sweep none --face > $OUT/ns_face.jsonl 2>> $OUT/ns.err
python tools/fmt_north_star.py $OUT/ns*.jsonl
