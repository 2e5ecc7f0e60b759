#!/usr/bin/env python3
"""Print VGPRs / scratch / occupancy / LDS of every kernel in the library (compile-only, no GPU)."""
import os, re, subprocess, sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "global_flow_local_attention_amd", "csrc")
files = sys.argv[1:] or ["block_extractor", "local_attn_reshape", "resample2d", "local_attn_aggregate"]
for f in files:
    out = subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", f + ".hip", "-o", "/dev/null",
                          "-Rpass-analysis=kernel-resource-usage"], cwd=CSRC, capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|TotalSGPRs): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1).split(" ")[0], m.group(2)
        cur[k] = v
        if k == "LDS":
            name = subprocess.run(["c++filt", cur["Function"]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"^void gfla::", "", name).split("(")[0]
            print("%-52s vgpr=%-4s sgpr=%-4s scratch=%-5s occ=%-2s lds=%s" % (name, cur.get("VGPRs"), cur.get("TotalSGPRs"), cur.get("ScratchSize"), cur.get("Occupancy"), v))
            cur = {}
