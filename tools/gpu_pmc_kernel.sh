#!/usr/bin/env bash
# SQ counters of kernels matching <substring> over a command, two counter passes (issue/wait mix, LDS).
# usage: bash tools/gpu_pmc_kernel.sh <tag> <substring> -- <command...>
set -uo pipefail
TAG="$1"; MATCH="$2"; shift 3
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"; rm -f $OUT/pmc_summary.txt
export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pk_${TAG}_$i -o pmc -- "$@" > $OUT/pmc_$i.log 2>&1); echo "set $i rc=$?"
  f=$(find /tmp/pk_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$MATCH" >> $OUT/pmc_summary.txt <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if sys.argv[2] not in n: continue
    short = n.split("(")[0].replace("void gfla::", "")
    d = agg.setdefault((short, r["Grid_Size"], r.get("LDS_Block_Size", "")), collections.OrderedDict())
    d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for (k, g, l), d in agg.items():
    print("%-50s grid=%-8s lds=%-7s n=%d\n     " % (k, g, l, len(list(d.values())[0])) + "  ".join("%s=%.4g" % (c, max(v)) for c, v in d.items()))
PY
done
cat $OUT/pmc_summary.txt
