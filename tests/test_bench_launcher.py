"""bench.py's own launcher: `python bench.py --gpus N` must start N ranks itself (the driver calls it exactly like that) and
must refuse to run with fewer devices than asked for.  Covered on CPU through --selftest-cpu (gloo, no PatchMatch work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=600, env=env)


def test_gpus_2_spawns_two_ranks_and_reports_the_slowest():
    r = _run("--gpus", "2", "--steps", "3", "--warmup", "0", "--selftest-cpu")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE line, from rank 0: the compact one (what the driver parses), at most 2,000 bytes
    assert len(lines[0].encode()) <= 2000
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] is None and line["selftest"] is True
    assert line["config"]["backend"] == "gloo" and line["workloads"]["configs3_tt1080p_pass_with_exchange"][0] is None
    out = json.loads(open(os.path.join(ROOT, "bench_workloads.json")).readline())   # the full block of the same run
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] is None and "selftest" in out
    assert out["config"]["backend"] == "gloo" and out["ms_per_step"] == line["ms_per_step"]
    per_rank = out["rank_ms_per_step"]
    assert len(per_rank) == 2 and per_rank[1] > 1.5 * per_rank[0]           # rank 1 sleeps twice as long
    assert out["ms_per_step"] >= per_rank[1] * 0.99                            # MAX over ranks, not rank 0's own time
    assert out["allgather_ms"] is not None
    # the C4-shaped sub-line: two views per rank, the per-pass depth all-gather inside the timed region
    sub = out["workloads"]["configs3_tt1080p_pass_with_exchange"]
    assert sub["n_gpus"] == 2 and sub["config"]["views_per_gpu"] == 2 and sub["config"]["views"] == 4
    assert sub["pass_allgather_inside_timed_region"] is True and sub["pass_allgather_ms"] > 0
    assert len(sub["rank_ms_per_step"]) == 2 and sub["timed_region_ms"] >= max(sub["rank_ms_per_step"])
    assert sub["timed_region_ms"] >= sub["pass_allgather_ms"]


def test_gpus_8_spawns_eight_ranks():
    """The driver's 8-GPU command line, on CPU: rank 7 exists before the 8-GPU node does.  Eight processes under gloo:
    barrier-bracketed region, MAX over ranks, the padded all-gather of two views per rank (16 views) in view order on every rank."""
    r = _run("--gpus", "8", "--steps", "2", "--warmup", "0", "--selftest-cpu")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0].encode()) <= 2000, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["value"] is None and line["selftest"] is True and line["scaling"] == "weak"
    out = json.loads(open(os.path.join(ROOT, "bench_workloads.json")).readline())
    per_rank = out["rank_ms_per_step"]
    assert len(per_rank) == 8 and per_rank[7] > 4 * per_rank[0] and out["ms_per_step"] >= per_rank[7] * 0.99
    sub = out["workloads"]["configs3_tt1080p_pass_with_exchange"]
    assert sub["n_gpus"] == 8 and sub["config"]["views"] == 16 and len(sub["rank_ms_per_step"]) == 8


def test_sub_workload_table_names_every_baseline_config():
    """The `workloads` block of the default line: configs[1] at 6 and 3 iterations, configs[2] (APD), configs[4] shape, configs[3]
    frame size alone and as a sharded pass with its exchange; names resolve, the shared-scene lines share a workload."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    keys = [s[0] for s in bench.SUB_WORKLOADS]
    assert len(set(keys)) == len(keys) >= 6
    # configs[0] (half resolution, 2 source views, 3 iterations): on the clock too, with the oracle timed on the same shape
    c0 = [s for s in bench.SUB_WORKLOADS if s[0] == bench.CONFIGS0_KEY]
    assert len(c0) == 1 and bench.resolve_workload(c0[0][1])[0] == (3100, 2065, 2) and c0[0][2] == 3
    by_wl = {}
    for key, name, steps, warmup, exch, vpg in bench.SUB_WORKLOADS:
        (w, h, n), apd = bench.resolve_workload(name)
        assert steps >= 1 and warmup >= 0 and vpg >= 1 and w * h > 0 and n >= 1
        by_wl.setdefault(name, []).append((steps, exch))
    assert (6, False) in by_wl[bench.DEFAULT_WORKLOAD] and (3, False) in by_wl[bench.DEFAULT_WORKLOAD]
    assert bench.resolve_workload("eth3d_pipes_fullres_10src_apd")[1] is True and "eth3d_pipes_fullres_10src_apd" in by_wl
    assert (8, False) in by_wl["synthetic_4096x3072_16src"]
    assert any(exch for _, exch in by_wl["tt_family_1080p_10src"])
    # whole passes (apd_run, K1..K15): measured on the resident handle of a sweep sub-line of the same workload at the reference's 3 iterations
    assert bench.PASS_ITERATIONS == 3 and len(bench.PASS_WORKLOADS) >= 1
    assert {e[4] for e in bench.PASS_WORKLOADS} == {"photometric", "geometric"}   # both pass kinds of a level (main.cpp:172-213)
    for key, name, passes, warm, kind in bench.PASS_WORKLOADS:
        assert key not in keys and passes >= 1 and warm >= 0
        assert bench.resolve_workload(name)[1] is True   # the states the reference runs at the full frame size: REFINE_INIT / REFINE_ITER + APD
        assert any(steps == bench.PASS_ITERATIONS for steps, _ in by_wl[name])
        assert max(steps for steps, _ in by_wl[name]) == bench.PASS_ITERATIONS   # the handle's max_iterations is the pass's


def test_refuses_more_gpus_than_visible():
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", env_extra={"HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert r.returncode != 0
    assert "refusing" in r.stderr and "--gpus 2" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_must_match_gpus():
    r = _run("--gpus", "1", "--selftest-cpu", env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_committed_counter_profiles_give_roofline_fractions_below_one():
    """bench.py's roofline = live launch time + the committed counter profile of the same command line.  With the launch times the
    profiles themselves recorded, every fraction the line can print must be a fraction: VALU issue against the 2-cycle peak, the
    mix-weighted busy estimate, memory-side bytes against 8 TB/s -- for every profile under profiles/, per launch and per dispatch."""
    import glob
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    mix = bench.load_valu_mix()
    assert mix is not None and 2.0 <= mix["mean_cycles_per_inst"] <= 4.2
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02", "pmc_bench_*.json")))
    assert len(paths) >= 3
    seen_default = False
    for path in paths:
        rec = json.load(open(path))
        cfg = rec["config"]
        for name, k in rec["kernels"].items():
            t = k["launch_ms"] * 1e-3
            valu = k["valu_insts_per_launch"] / t / 1e9 / bench.VALU_PEAK_GINST
            busy = k["valu_insts_per_launch"] * mix["mean_cycles_per_inst"] / (bench.NUM_SIMDS * bench.MAX_CLOCK_GHZ * 1e9 * t)
            hbm = k["hbm_bytes_per_launch"] / t / 1e9 / bench.HBM_PEAK_GBPS
            assert 0.05 < valu <= 1.0 and 0.0 < hbm <= 1.0, (path, name, valu, hbm)
            if name == "k67":
                assert busy <= 1.0, (path, busy)
            pd = k["per_dispatch_timed"]
            assert len(pd["SQ_INSTS_VALU"]) == 2 * cfg["steps"] == len(pd["duration_ns@trace"])
            for insts, ns, fe in zip(pd["SQ_INSTS_VALU"], pd["duration_ns@sq"], pd["FETCH_SIZE"]):
                assert insts / ns / bench.VALU_PEAK_GINST <= 1.0
        if cfg["workload"] == "eth3d_office_fullres_8src" and cfg["steps"] == 6 and cfg["warmup"] == 1:
            seen_default = True
            got = bench.load_pmc_profile(cfg["workload"], 6, 1, "k67")
            assert got is not None and got["source"].startswith("profiles/")
    assert seen_default, "the default bench command line needs a committed profile"
    # another --steps of the same workload is the same launches, one by one (tests/test_profiles_consistent.py); what is
    # never borrowed is a profile of another workload, of other options or of another seed
    assert bench.load_pmc_profile("eth3d_pipes_fullres_10src", 6, 1, "k67") is None
    assert bench.load_pmc_profile("eth3d_office_fullres_8src", 6, 1, "k67", options=["early_out=0"]) is None


def test_compact_line_fits_a_bounded_reader_and_equals_the_full_block():
    """The stdout line of bench.py is a digest of the full block (bench_workloads.json): at most 2,000 bytes whatever the block holds,
    every number equal to the block's.  Checked on the committed full blocks of rounds 4 and 5 (24 KB each)."""
    import glob
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[45]", "bench_*.json")))
    assert len(paths) >= 2
    for path in paths:
        full = json.loads(open(path).readline())
        if "workloads" not in full:
            continue
        text = bench.compact_line(full)
        assert len(text.encode()) <= bench.COMPACT_LINE_MAX_BYTES <= 2000 and "\n" not in text
        c = json.loads(text)
        assert c["value"] == full["value"] and c["steps"] == full["steps"] and c["ms_per_step"] == full["ms_per_step"]
        assert c["metric"] == full["metric"] and c["config"]["workload"] == full["config"]["workload"]
        r, fr = c["roofline"], full["roofline"]
        assert (r["frac"], r["achieved"], r["peak"], r["avg_launch_ms"], r["launches"]) == (fr["frac"], fr["achieved"], fr["peak"], fr["avg_launch_ms"], fr["launches"])
        assert r["traffic"] == int(fr["traffic"]) and r["hbm_frac"] == fr["hbm"]["frac"] and r["pmc_source"] == fr["pmc_source"]
        if full.get("cpu_baseline"):
            assert c["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and c["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
        assert set(c["workloads"]) == set(full["workloads"])
        for key, (value, ms, frac) in c["workloads"].items():
            w = full["workloads"][key]
            assert value == w["value"] and ms == w.get("ms_per_pass", w.get("ms_per_step")), (path, key)
    # a block that cannot fit is shortened, never refused (an assertion here cost a measured run its line): the headline,
    # the roofline and the cpu baseline survive, the line says what it dropped
    fat = json.loads(open(paths[0]).readline())
    fat["workloads"] = {"k%03d_%s" % (i, "x" * 40): {"value": 1.0, "ms_per_step": 1.0} for i in range(60)}
    text = bench.compact_line(fat)
    assert len(text.encode()) <= bench.COMPACT_LINE_MAX_BYTES
    c = json.loads(text)
    assert c["truncated"] and c["value"] == fat["value"] and c["roofline"]["frac"] == fat["roofline"]["frac"]
    assert c["cpu_baseline"]["value"] == fat["cpu_baseline"]["value"]
    # the fields that say what the headline is not
    full = json.loads(open(paths[-1]).readline())
    c = json.loads(bench.compact_line(full))
    assert c["value_config_iters"] == full["workloads"][bench.CONFIG_ITERS_KEY]["value"]
    assert c["whole_pass"] == [full["workloads"][bench.WHOLE_PASS_KEY]["value"], full["workloads"][bench.WHOLE_PASS_KEY]["ms_per_pass"]]
    assert c["roofline"]["frac_kind"] == "valu-issue" and c["roofline"]["algorithmic_over_hbm_peak"] > 1.0
