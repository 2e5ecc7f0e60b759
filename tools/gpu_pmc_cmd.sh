#!/usr/bin/env bash
# Counter passes over an arbitrary command, summarised per kernel.
# usage: gpurun -- 'bash tools/gpu_pmc_cmd.sh <tag> <kernel-name-substring> -- <command...>'
set -uo pipefail
TAG="$1"; MATCH="$2"; shift 3
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for SET in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmc_${TAG}_$i -o pmc -- "$@" > $OUT/pmc_$i.log 2>&1); echo "set $i rc=$?"
  f=$(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$MATCH" >> $OUT/pmc_summary.txt <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if sys.argv[2] not in n: continue
    short = n.split("(")[0].replace("void gfla::", "")
    d = agg.setdefault((short, r["Grid_Size"]), collections.OrderedDict())
    d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for (k, g), d in agg.items():
    print("%-40s grid=%-8s " % (k, g) + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in d.items()))
PY
done
cat $OUT/pmc_summary.txt
