#!/usr/bin/env bash
# SQ-level counters for the gfla kernels of one bench step (diagnosis of what bounds them).
set -uo pipefail
TAG="${1:-sq}"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TA_[A-Z_0-9]+" | sort -u > $OUT/counters_available.txt; wc -l $OUT/counters_available.txt
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/sq_${TAG}_$i -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/sq_$i.log 2>&1); echo "set $i rc=$?"
  f=$(find /tmp/sq_${TAG}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" > $OUT/sq_set$i.txt <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "gfla::" not in n: continue
    short = n.split("(")[0].replace("void gfla::", "")
    key = (short, r["Grid_Size"])
    d = agg.setdefault(key, collections.OrderedDict())
    d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for (k, g), d in agg.items():
    print("%-56s grid=%-8s " % (k, g) + "  ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in d.items()))
PY
  tail -2 $OUT/sq_$i.log | cut -c1-200
done
cat $OUT/sq_set*.txt | grep -E "be_bwd_lds|rs_lds_kernel<float, 2, 1|agg_ga|agg_fwd_lds_kernel<float, 5|be_unfold_fwd_lds_kernel<float, 5" 
