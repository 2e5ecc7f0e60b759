#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s34; mkdir -p $O
timeout 900 python -m pytest tests/test_big_plane_gpu.py -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python tools/bench_config2.py --tag final --out $O/config2.jsonl > /dev/null 2>$O/err.log
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r5_s34/config2.jsonl")]
for r in rows:
    print("%-26s %-13s %7.1f us frac %.3f  ref %8.1f us" % (r["op"], r["flow"], r["us"], r["frac"], r.get("ref_us") or float('nan')))
PY
