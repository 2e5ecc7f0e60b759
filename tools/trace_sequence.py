#!/usr/bin/env python3
"""The dispatch sequence of ONE steady-state step from a rocprofv3 --kernel-trace CSV of bench.py: start offset, duration and the
idle gap in front of every dispatch (us).    python tools/trace_sequence.py bench_kernel_trace.csv [marker_substring]"""
import csv
import sys


def main(path, marker="fc_maxabs_kernel"):
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(path)))
    tail = [i for i, r in enumerate(rows) if "fc_tail_fwd_kernel<3>" in r[2]]
    lo, hi = tail[-2], tail[-1]
    # a step starts a few dispatches before the first tail kernel: back up to the previous step's last kernel
    t0 = rows[lo][0]
    prev_end = rows[lo - 1][1]
    for s, e, n, q in rows[lo:hi]:
        short = n.split("(")[0].replace("void ", "").replace("gfla::", "")[:70]
        print("%9.1f  dur %8.1f  gap %6.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, short))
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main(*sys.argv[1:])
