#!/usr/bin/env python3
"""Per-kernel counter averages of a rocprofv3 --pmc run + kernel durations from the kernel trace of the same run.

For the MFMA kernels: effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration, and MFMA-pipe utilisation =
SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8)."""
import collections
import csv
import sys


def short(n):
    return n.split("(")[0].replace("void gfla::", "").replace("void ", "")[:64]


def main(counters, trace):
    dur = collections.defaultdict(list)
    if trace:
        for r in csv.DictReader(open(trace)):
            dur[(short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "")))].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(counters)):
        n = r["Kernel_Name"]
        if "gfla::" not in n:
            continue
        key = (short(n), r["Grid_Size"])
        agg.setdefault(key, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for key, d in agg.items():
        vals = {c: sum(v) / len(v) for c, v in d.items()}
        us = dur.get(key) or [x for (kn, _), v in dur.items() if kn == key[0] for x in v]
        t = sum(us) / len(us) if us else float("nan")
        extra = ""
        gui = vals.get("GRBM_GUI_ACTIVE")
        if gui:
            extra += "  clk=%.2fGHz" % (gui / 8 / (t * 1e3))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
                extra += "  mfma_util=%.3f" % (vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * gui / 8))
            if "SQ_LDS_IDX_ACTIVE" in vals:
                extra += "  lds_busy=%.3f" % (vals["SQ_LDS_IDX_ACTIVE"] / (256 * gui / 8))
        print("%-52s grid=%-9s us=%8.1f%s  " % (key[0], key[1], t, extra) + "  ".join("%s=%.4g" % kv for kv in vals.items()))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
