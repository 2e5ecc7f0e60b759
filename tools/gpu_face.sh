#!/bin/bash
# config-5 leg on the GPU box: bf16 tests, then the FaceGenerator-shaped bench and its kernel trace
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fc_mfma_gpu.py tests/test_gpu_parity.py -q -k "bf16" --timeout=600 2>&1 | tail -5 > gpurun_out/face_tests.log
timeout 600 python bench.py --workload face_bf16 --batch 8 --steps 10 --warmup 3 > gpurun_out/face_bench.json 2> gpurun_out/face_bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_face -o face -- python /root/repo/bench.py --workload face_bf16 --batch 8 --steps 5 --warmup 2 > /dev/null 2>&1)
# 8 steps (1 priming + 2 warm-up + 5 timed) + 5 instrumented = 13 steps in the trace
python - > gpurun_out/face_kernel_stats.txt 2>&1 <<'PY'
import csv
rows = list(csv.DictReader(open("/tmp/prof_face/face_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("rocprofv3 --kernel-trace --stats: bench.py --workload face_bf16 --batch 8 --steps 5 --warmup 2 (13 steps in the trace)")
print("%10s %8s %10s %6s  %s" % ("total_ms", "calls", "avg_us", "%", "kernel"))
for r in rows[:28]:
    print("%10.2f %8s %10.1f %6.1f  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3,
                                         100 * float(r["TotalDurationNs"]) / tot, r["Name"][:110]))
PY
tail -3 gpurun_out/face_tests.log; tail -c 600 gpurun_out/face_bench.err; head -c 1500 gpurun_out/face_bench.json; echo; head -20 gpurun_out/face_kernel_stats.txt
