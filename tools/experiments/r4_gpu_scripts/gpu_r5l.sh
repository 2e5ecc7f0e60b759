#!/usr/bin/env bash
set -uo pipefail
TAG="${1:-r5l}"; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python tools/bench_scatter.py --tuning 14=2 --flows smooth,coherent,wild,zero --rows 0 > $OUT/scatter.jsonl 2> $OUT/scatter.err; cut -c1-400 $OUT/scatter.jsonl; tail -2 $OUT/scatter.err
