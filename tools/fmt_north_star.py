#!/usr/bin/env python3
"""One line per configuration of tools/bench_north_star.py's JSON lines."""
import json
import sys

for f in sys.argv[1:]:
    print(f)
    for line in open(f):
        d = json.loads(line)
        if "tag" not in d:
            print("  " + line.strip()[:200])
            continue
        print("  %-64s L3 be %6.1f us %.3f agg %5.1f pair %.3f | L2 be %6.1f us %.3f agg %5.1f pair %.3f" % (
            d["tag"][:64], d["attn3"]["be_us"], d["attn3"]["be_frac"], d["attn3"]["agg_us"], d["attn3"]["pair_frac"],
            d["attn2"]["be_us"], d["attn2"]["be_frac"], d["attn2"]["agg_us"], d["attn2"]["pair_frac"]))
