#!/bin/bash
# usage: bash tools/gpu_prof_cmd.sh <tag> <command...> -- rocprofv3 kernel trace of a command, per-kernel averages into gpurun_out/<tag>_kernels.txt
TAG=$1; shift
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o t -- "$@" > /tmp/prof_$TAG.log 2>&1)
python - "$TAG" > gpurun_out/${TAG}_kernels.txt 2>&1 <<'PY'
import csv, sys, glob
f = glob.glob("/tmp/prof_%s/**/t_kernel_stats.csv" % sys.argv[1], recursive=True)
rows = list(csv.DictReader(open(f[0])))
print("%10s %8s %10s %10s %10s  %s" % ("total_ms", "calls", "avg_us", "min_us", "max_us", "kernel"))
for r in rows[:40]:
    print("%10.2f %8s %10.1f %10.1f %10.1f  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3,
          float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Name"][:120]))
PY
head -30 gpurun_out/${TAG}_kernels.txt
