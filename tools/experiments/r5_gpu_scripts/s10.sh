#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r5_s10; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 --tb=short -p no:cacheprovider -k "resample or correctness or big_plane or config2 or bench_shape" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
python tools/opbench.py --only rs_fwd,rs_bwd --iters 20 --out $O/opbench_rs.jsonl > $O/opbench_rs.log 2>&1
run() { python tools/bench_config2.py --tag "$1" --tuning "$2" --no-ref --flows smooth --out $O/config2.jsonl > /dev/null 2>&1; }
for t in "39=0" "10=20" "10=24" "10=40" "37=6" "37=12" "35=6,36=32" "35=10,36=32" "35=12,36=32"; do run v $t; done
python - <<'PY'
import json
for l in open("gpurun_out/r5_s10/config2.jsonl"):
    r = json.loads(l)
    if "fwd" in r["op"]:
        print("%-26s %-38s %-8s %7.1f us  frac %.3f" % (r["tuning"], r["op"], r["flow"], r["us"], r["frac"]))
for l in open("gpurun_out/r5_s10/opbench_rs.jsonl"):
    r = json.loads(l)
    print("opbench %-60s %7.1f us frac %.3f ref %s" % (r["case"], r["us"], r["frac_peak"], r["ref_us"]))
PY
python bench.py --no-cpu-baseline --no-variants --no-legs --steps 20 --warmup 5 > $O/bench_quick.json 2> $O/bench_quick.err; python -c "
import json; d=json.load(open('$O/bench_quick.json')); print('step ms', d['ms_per_step'], 'value', d['value'])
for r in d['kernels'][:14]: print(r['entry'], r['dims'], r['avg_us'])"
