"""The LAST stdout line of bench.py must be one short JSON object the driver can parse (round 5's was 22 KB and was not:
BENCH_r05.json `parsed: null`).  The reference's own report is one short line per iteration (train.py:46-48)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "detail_file")


def test_compact_line_of_a_full_record_fits_4k():
    """A real full record (round 5's final run, 22 KB) through compact_line: < 4 KB, every contract key, roofline +
    north_star + oracle_check + cpu_baseline carried."""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r5_final_bench.json")))
    assert len(json.dumps(full)) > 16000
    out, text = bench.compact_line(full, "gpurun_out/bench_detail.json")
    assert len(text) < 4096 and "\n" not in text
    back = json.loads(text)
    for k in REQUIRED + ("roofline", "north_star", "oracle_check", "cpu_baseline", "vendor_fallback_calls"):
        assert k in back, k
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "avg_us", "traffic"):
        assert k in back["roofline"], k
    assert back["roofline"]["frac"] == full["roofline"]["frac"] and back["value"] == full["value"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"], k
    assert set(back["config"]) >= {"workload", "batch_per_gpu", "global_batch", "parallelism"}
    assert set(back["north_star"]["layers"]) == {"attn3", "attn2"}
    assert "kernels" not in back and "fc_kernels" not in back and "variants" not in back


def test_compact_line_sheds_optional_objects_before_it_overflows():
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r5_final_bench.json")))
    full["legs"] = {"leg%03d" % i: {"ms_per_step": float(i)} for i in range(400)}     # a leg table that would not fit
    out, text = bench.compact_line(full, "x.json")
    assert len(text) < 4096 and "legs" not in out and "roofline" in out and "cpu_baseline" in out


def test_last_stdout_line_of_an_n2_run_is_short_and_detail_goes_to_a_file(tmp_path):
    """`python bench.py --gpus 2` on two gloo ranks (GFLA_BENCH_CPU_STUB): exactly one JSON line on stdout, < 4 KB, with
    the `dist` fingerprint; the full record is in the detail file."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    detail = str(tmp_path / "detail.json")
    env.update(GFLA_BENCH_CPU_STUB="1", OMP_NUM_THREADS="1", BENCH_DETAIL=detail)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    last = res.stdout.strip().splitlines()[-1]
    assert len(last) < 4096
    line = json.loads(last)
    for k in REQUIRED + ("dist",):
        assert k in line, k
    assert line["n_gpus"] == 2 and line["dist"]["world_size_seen_by_group"] == 2
    full = json.load(open(detail))
    assert full["value"] == line["value"] and "kernels" in full
