#!/usr/bin/env python3
"""Steady-state per-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py.

    python tools/trace_steps.py bench_kernel_trace.csv [marker_substring] [n_last_steps]

A step starts at each dispatch whose name contains the marker (default: the first gfx950 kernel
of a step).  Only the last n steps are aggregated, so MIOpen's find-mode candidates (run during
warm-up) do not pollute the picture the way they do in --stats.
"""
import collections
import csv
import sys


def main(path, marker="be_unfold_fwd_lds_kernel<float, 3>", n_last=3):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(starts) < n_last + 1:
        print("not enough steps found", len(starts))
        return
    lo, hi = starts[-(n_last + 1)], starts[-1]
    sel = rows[lo:hi]
    wall = (rows[hi][0] - rows[lo][0]) / n_last / 1e3
    agg = collections.OrderedDict()
    busy = 0
    for s, e, n in sel:
        short = n.split("(")[0].replace("void ", "")[:100]
        a = agg.setdefault(short, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
        busy += (e - s) / 1e3
    print("steady state over last %d steps: wall %.1f us/step, kernel-busy %.1f us/step, %d dispatches/step" %
          (n_last, wall, busy / n_last, len(sel) // n_last))
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%9.1f us/step  x%-5.1f  %s" % (t / n_last, c / n_last, k))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2] if len(a) > 2 else "be_unfold_fwd_lds_kernel<float, 3>", int(a[3]) if len(a) > 3 else 3)
