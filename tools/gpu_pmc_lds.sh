#!/usr/bin/env bash
# LDS counters of kernels matching <substring> over a command.  usage: bash tools/gpu_pmc_lds.sh <tag> <substring> -- <command...>
set -uo pipefail
TAG="$1"; MATCH="$2"; shift 3
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"; rm -f $OUT/pmc_summary.txt
export TMPDIR=/tmp
i=0
for SET in "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
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
    d = agg.setdefault((short, r["Grid_Size"], r["Workgroup_Size"], r.get("LDS_Block_Size", "")), collections.OrderedDict())
    d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for (k, g, w, l), d in agg.items():
    print("%-36s grid=%-8s wg=%-5s lds=%-7s n=%d " % (k, g, w, l, len(list(d.values())[0])) + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in d.items()))
PY
done
cat $OUT/pmc_summary.txt
