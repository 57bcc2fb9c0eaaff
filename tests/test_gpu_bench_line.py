"""The driver's command, shortened: `python bench.py --steps 2 --warmup 1 --no-cpu-baseline` must print ONE compact JSON line of at
most bench.COMPACT_LINE_MAX_BYTES bytes (the driver reads a bounded tail of stdout: BENCH_r04.parsed was null with a 24 KB line) and
write the full block to bench_workloads.json, whose `workloads` block holds every sub-line of bench.SUB_WORKLOADS and
bench.PASS_WORKLOADS with a value (no {"error": ...} entry), the whole-pass lines with K14 / K15 inside their timed region and a
roofline from their committed counter profiles; every compact value equals the full block's."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_default_line_carries_every_sub_line():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]   # stdout is the one JSON line
    assert len(lines[0].encode()) <= bench.COMPACT_LINE_MAX_BYTES <= 2000, len(lines[0])
    c = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "workloads"):
        assert k in c, k
    assert "truncated" not in c and c["value_config_iters"] > 0 and c["whole_pass"][0] > 0 and c["roofline"]["frac_kind"] == c["roofline"]["bound"]
    assert c["roofline"]["frac"] > 0 and c["roofline"]["bound"] and c["roofline"]["pmc_source"] and c["roofline"]["traffic"] > 0
    d = json.load(open(os.path.join(ROOT, bench.FULL_BLOCK_FILE)))
    assert "bench.py full block: " in r.stderr
    assert c["value"] == d["value"] and c["ms_per_step"] == d["ms_per_step"] and c["roofline"]["frac"] == d["roofline"]["frac"]
    assert c["roofline"]["avg_launch_ms"] == d["roofline"]["avg_launch_ms"] and c["config"]["workload"] == d["config"]["workload"]
    assert set(c["workloads"]) == set(d["workloads"])
    for key, (value, ms, frac) in c["workloads"].items():
        full = d["workloads"][key]
        assert value == full["value"] and ms == full.get("ms_per_pass", full.get("ms_per_step")), key
    assert d["metric"].startswith("Mpix*iterations/sec") and d["value"] > 0 and d["steps"] == 2 and d["n_gpus"] == 1
    assert d["config"]["workload"] == bench.DEFAULT_WORKLOAD and d["roofline"]["avg_launch_ms"] > 0
    w = d["workloads"]
    for key, name, steps, warmup, exch, vpg in bench.SUB_WORKLOADS:
        assert key in w and "error" not in w[key] and w[key]["value"] > 0, (key, w.get(key))
        assert w[key]["steps"] == steps and w[key]["config"]["workload"] == name and w[key]["config"]["views_per_gpu"] == vpg
        assert w[key]["pass_allgather_inside_timed_region"] is bool(exch)
    for key, name, passes, warm, kind in bench.PASS_WORKLOADS:
        p = w[key]
        assert "error" not in p and p["value"] > 0 and p["passes"] == passes and p["iterations_per_pass"] == bench.PASS_ITERATIONS
        assert ("geom" in p["config"]["state"]) == (kind == "geometric")
        k = p["kernel_ms_per_pass"]
        assert k["DepthToWeak"] > 0 and k["LocalRefine"] > 0 and k["BlackPixelUpdateWeak"] > 0 and k["BlackPixelUpdateStrong"] > 0
        assert sum(k.values()) <= p["ms_per_pass"] * 1.001   # the kernels lie inside the timed call
        for kn in ("K14", "K15"):
            roof = p["pass_kernels"][kn]
            assert roof["launches"] == passes and roof["pmc_source"] and 0.3 < roof["frac"] < 1.0 and 0.0 < roof["hbm"]["frac"] < 1.0
    # the geometric pass costs more than the photometric one (consistency term in K9/K10, K14, K15), both well under two seconds here
    assert w["configs2_pipes_apd_whole_pass"]["ms_per_pass"] < w["configs2_pipes_apd_geometric_pass"]["ms_per_pass"] < 2000.0
