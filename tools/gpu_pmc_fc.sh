#!/usr/bin/env bash
# MFMA-pipe utilisation and effective clock of the FC kernels in one bench step (rocprofv3 --pmc, kernel trace only).
# usage: gpurun -- 'bash tools/gpu_pmc_fc.sh <tag> [bench args]'
set -uo pipefail
TAG="${1:-pmcfc}"; shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for SET in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVES"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmcfc_${TAG}_$i -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OLDPWD/$OUT/pmc_$i.log 2>&1); echo "set $i rc=$?"
  c=$(find /tmp/pmcfc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  k=$(find /tmp/pmcfc_${TAG}_$i -name "*kernel_trace.csv" | head -1)
  [ -n "$c" ] && python tools/pmc_fc_summary.py "$c" "$k" > $OUT/pmc_set$i.txt 2>&1
  cat $OUT/pmc_set$i.txt | cut -c1-330
done
