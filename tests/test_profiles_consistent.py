"""The committed bench lines and the committed counter profiles must tell the same story: every roofline field of
profiles/r02/bench_*.json is re-derived from the per-dispatch counter values of the profile of the same command line
(tools/recompute_roofline.py) -- a stale profile, a line made before its profile, or a fraction above 1 fails here."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


import pytest


@pytest.mark.parametrize("round_dir", ["r02", "r03", "r04", "r05", "r06"])
def test_roofline_fields_follow_from_the_committed_counters(round_dir):
    d = os.path.join(ROOT, "profiles", round_dir)
    import glob
    if not glob.glob(os.path.join(d, "pmc_bench_*.json")):
        pytest.skip("no counter profiles committed under profiles/%s yet" % round_dir)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "recompute_roofline.py"), d],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert r.stdout.count("frac") >= 3, r.stdout  # default line, driver line, APD line


def test_one_profile_serves_every_steps_and_warmup_of_its_workload():
    """bench.py's timed launch j is the same work whatever --steps / --warmup say (the pass is re-initialised after the warm-up):
    a shorter run reads a prefix of the profiled launches, a longer one repeats the last profiled iteration and reports how
    many launches it extrapolated."""
    sys.path.insert(0, ROOT)
    import bench
    steps = 4
    rec = {"per_dispatch_timed": {c: [100.0 * (j + 1) for j in range(2 * steps)] for c in bench.PMC_COUNTERS.values()}}
    series, extra = bench.pmc_timed_series(rec, steps, 2 * steps)
    assert extra == 0 and series["valu_insts_per_launch"] == [100.0 * (j + 1) for j in range(8)]
    series, extra = bench.pmc_timed_series(rec, steps, 4)       # --steps 2: the first two iterations
    assert extra == 0 and series["fetch_kib"] == [100.0, 200.0, 300.0, 400.0]
    series, extra = bench.pmc_timed_series(rec, steps, 12)      # --steps 6: two iterations beyond the profile
    assert extra == 4 and series["write_kib"][8:] == [700.0, 800.0, 700.0, 800.0]
    # warm-up launches recorded before the timed ones are ignored (the timed launches are the last 2 * steps)
    rec2 = {"per_dispatch_timed": {c: [-1.0, -1.0] + v for c, v in rec["per_dispatch_timed"].items()}}
    assert bench.pmc_timed_series(rec2, steps, 8)[0] == bench.pmc_timed_series(rec, steps, 8)[0]
    # a counter the profile lacks is reported as missing, not as zero
    del rec["per_dispatch_timed"]["FETCH_SIZE"]
    assert bench.pmc_timed_series(rec, steps, 8)[0]["fetch_kib"] is None


def test_profiles_of_other_options_or_seeds_are_never_borrowed():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.load_pmc_profile("eth3d_office_fullres_8src", 6, 1, "k67", options=["fast_rcp=1"]) is None
    assert bench.load_pmc_profile("eth3d_office_fullres_8src", 6, 1, "k67", seed=99) is None
    assert bench.load_pmc_profile("no_such_workload", 6, 1, "k67") is None
    got = bench.load_pmc_profile("eth3d_office_fullres_8src", 6, 1, "k67")
    assert got is not None and got["extrapolated_launches"] == 0 and got["valu_insts_per_launch"] > 1e9


def test_whole_pass_profile_is_found_and_gives_fractions_below_one():
    """K14 / K15 of bench.py's whole-pass sub-line: the committed pmc_pass_*.json of the workload, per-launch means over its timed
    passes; with the launch time the profile itself recorded both fractions are fractions.  Nothing is borrowed across options / seeds."""
    sys.path.insert(0, ROOT)
    import bench
    for key, name, passes, _, kind in bench.PASS_WORKLOADS:
        for kk in ("k14", "k15"):
            got = bench.load_pass_profile(name, kk, kind=kind)
            assert got is not None, (name, kk, kind)
            assert got["launches"] == passes and got["launch_ms"] > 0
            t = got["launch_ms"] * 1e-3
            assert 0.2 < got["valu_insts_per_launch"] / t / 1e9 / bench.VALU_PEAK_GINST < 1.0
            assert 0.0 < got["hbm_bytes_per_launch"] / t / 1e9 / bench.HBM_PEAK_GBPS < 1.0
        assert bench.load_pass_profile(name, "k14", options=["fast_rcp=1"]) is None
        assert bench.load_pass_profile(name, "k14", seed=99) is None
    assert bench.load_pass_profile("no_such_workload", "k14") is None


def test_pass_kernel_roofline_without_a_profile_reports_null_fields():
    sys.path.insert(0, ROOT)
    import bench

    class Pkg:
        K14 = 14
        KERNEL_NAMES = {14: "DepthToWeak"}
    live = {14: (600.0, 2)}   # two launches, 300 ms each
    r = bench.pass_kernel_roofline(Pkg, live, Pkg.K14, "k14", "no_such_workload", "photometric", (), 12345)
    assert r["avg_launch_ms"] == 300.0 and r["achieved"] is None and r["frac"] is None and r["traffic"] is None and "pmc_note" in r
    name = bench.PASS_WORKLOADS[0][1]
    r = bench.pass_kernel_roofline(Pkg, live, Pkg.K14, "k14", name, "photometric", (), 12345)
    assert r["pmc_source"].startswith("profiles/") and 0.0 < r["frac"] < 1.0 and 0.0 < r["hbm"]["frac"] < 1.0
    assert abs(r["achieved"] - r["valu_insts_per_launch"] / 0.3 / 1e9) < 0.1


def test_whole_kernel_valu_busy_comes_from_the_class_counters():
    """Round 6: valu_busy_estimate = the launch's VALU instructions by class (SQ_INSTS_VALU_*: every basic block at its execution count)
    x measured issue cycles, as a bracket; the committed round-6 profile of the default workload gives a bracket inside (frac, 1)."""
    sys.path.insert(0, ROOT)
    import bench
    got = bench.load_pmc_profile("eth3d_office_fullres_8src", 20, 5, "k67")
    assert got is not None and got["source"].startswith("profiles/r06/") and got["valu_classes"] is not None
    t_ms = got["launch_ms"]
    b = bench.valu_busy_from_classes(got["valu_insts_per_launch"], got["valu_classes"], t_ms)
    issue = got["valu_insts_per_launch"] / (t_ms * 1e-3) / 1e9 / bench.VALU_PEAK_GINST
    assert issue < b["frac"] < b["frac_hi"] < 1.0, (issue, b)      # 2.2 cycles per instruction at least, never more than the pipe has
    assert abs(sum(b["class_share"].values()) - 1.0) < 0.02
    assert bench.valu_busy_from_classes(1e9, None, 1.0) is None     # a profile without the class pass: no estimate (the static mix of rounds 2-5 is used)
    # a synthetic check of the arithmetic: 100 instructions, 50 fast, 10 transcendental, 10 conversions, 30 of unknown class
    c = {"fma": 30.0, "add": 10.0, "mul": 10.0, "trans": 10.0, "cvt": 10.0, "int32": 20.0, "int64": 0.0}
    r = bench.valu_busy_from_classes(100.0, c, 1.0)
    simd = bench.NUM_SIMDS * bench.MAX_CLOCK_GHZ * 1e9 * 1e-3
    base = 2.2 * 50 + 8.108 * 10 + 4.067 * 10
    assert abs(r["frac"] * simd - (base + 2.2 * 30)) < 1e-3 * simd * r["frac"] + 1e-9 or abs(r["frac"] - round((base + 2.2 * 30) / simd, 4)) <= 1e-4
