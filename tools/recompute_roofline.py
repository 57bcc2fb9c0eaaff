#!/usr/bin/env python3
"""Recomputes every `roofline` field of the committed bench lines from the committed counter files, nothing else.

    python tools/recompute_roofline.py [profiles/rNN]      (default: the newest round that holds bench lines)

For each bench line under the directory (bench_default.json, bench_driver_s20_w5.json, bench_apd_s3_w1.json, ...) it finds the
counter profile of the same workload / --steps / --warmup (pmc_bench_<workload>_s<steps>_w<warmup>.json, written by
tools/profile_bench.py from separate rocprofv3 --pmc passes), re-derives

    per-launch means   = mean of the per-dispatch values over the timed launches (the .csv holds the same numbers)
    achieved           = SQ_INSTS_VALU per launch / live launch time
    frac               = achieved / (1024 SIMDs x 2.4 GHz / 2 cycles)
    traffic            = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB per launch
    hbm.frac           = traffic / live launch time / 8 TB/s
    valu_busy_estimate = SQ_INSTS_VALU x mean issue cycles of the window body (valu_mix_k67w.json) / (1024 x 2.4 GHz x time)
    profile vs live    = launch duration in the rocprofv3 kernel trace against the HIP-event duration of the bench line

(for an APD workload, whose `roofline` is the weak sweep's against the L1 tag pipeline: TCP_TOTAL_CACHE_ACCESSES per launch / live
launch time / (256 CUs x 2.4 GHz x 1.85 accesses per clock), its `strong_path` block like a K6/K7 roofline) and compares them
with what the line says.  Since round 5 bench.py's stdout is a compact line (line_<name>.json, at most 2,000 bytes: what the driver
parses) and the full block lies beside it (bench_<name>.json = bench_workloads.json of the same run): every number of the compact
line must be the full block's.  Exits non-zero on any mismatch; tests/test_profiles_consistent.py runs it."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK = 1024 * 2.4 / 2.0  # G wave-instructions / s
HBM = 8000.0             # GB/s


def close(a, b, rel):
    return abs(a - b) <= rel * max(abs(a), abs(b), 1e-30)


TAG_PEAK = 256 * 2.4 * 1.85  # G L1 tag accesses / s (bench.py: TCP_ACCESSES_PER_CLOCK)


def load_mix(directory):
    """valu_mix_k67w.json of the directory the counter profile lies in (the static mix of that round's kernel), or None: a line
    without `valu_busy_estimate` is then expected (rounds 2 and 3 shared round 2's file; since round 4 a mix is never borrowed)."""
    legacy = os.path.basename(os.path.normpath(directory)) in ("r02", "r03")
    cands = [os.path.join(directory, "valu_mix_k67w.json")]
    if legacy:
        cands += sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[23]", "valu_mix_k67w.json")), reverse=True)
    for c in cands:
        if os.path.exists(c):
            return json.load(open(c))["window_body"]["mean_cycles_per_inst"]
    return None


def timed_mean(k, profiled_steps, launches):
    """Mean of a counter over the first `launches` timed launches.  Timed launch j of any command line of a workload is launch j
    of the profiled one (bench.py re-initialises the pass after the warm-up); launches beyond the profile repeat its last
    iteration (black, red).  The same rule as bench.py's pmc_timed_series, written a second time."""
    pd, have = k["per_dispatch_timed"], 2 * profiled_steps

    def mean(c):
        v = pd[c][-have:]
        assert len(v) == have, (c, len(v), have)
        vals = [v[j] if j < have else v[have - 2 + (j - have) % 2] for j in range(launches)]
        return sum(vals) / launches
    return mean


def from_classes(roof):
    """round 6: valu_busy_estimate comes from the whole-kernel class counters of the counter profile, not from a static block mix"""
    return "frac_hi" in (roof.get("valu_busy_estimate") or {})


FAST_C, SLOW_C, TRANS_C = 2.2, 4.067, 8.108   # bench.py: VALU_COST_*


def class_busy_rows(roof, mean, insts, t):
    """valu_busy_estimate.frac / frac_hi re-derived: sum over the SQ_INSTS_VALU_* classes of instructions x measured issue cycles."""
    if not from_classes(roof):
        return ()
    fast = mean("SQ_INSTS_VALU_FMA_F32") + mean("SQ_INSTS_VALU_ADD_F32") + mean("SQ_INSTS_VALU_MUL_F32")
    trans, cvt = mean("SQ_INSTS_VALU_TRANS_F32"), mean("SQ_INSTS_VALU_CVT")
    other = max(insts - fast - trans - cvt, 0.0)
    base = FAST_C * fast + TRANS_C * trans + SLOW_C * cvt
    simd = 1024 * 2.4e9 * t
    return (("valu_busy_estimate.frac", roof["valu_busy_estimate"]["frac"], (base + FAST_C * other) / simd, 2e-3),
            ("valu_busy_estimate.frac_hi", roof["valu_busy_estimate"]["frac_hi"], (base + SLOW_C * other) / simd, 2e-3))


def check_k67(tag, roof, k, mix, problems, profiled_steps):
    n = roof["launches"]
    mean = timed_mean(k, profiled_steps, n)
    if roof.get("pmc_extrapolated_launches", 0) != max(0, n - 2 * profiled_steps):
        problems.append("%s: pmc_extrapolated_launches is %r, the profile covers %d of %d launches" % (
            tag, roof.get("pmc_extrapolated_launches"), 2 * profiled_steps, n))
    insts = mean("SQ_INSTS_VALU")
    traffic = mean("FETCH_SIZE") * 1024 * 2 + mean("WRITE_SIZE") * 1024
    t = roof["avg_launch_ms"] * 1e-3
    achieved = insts / t / 1e9
    for what, got, want, rel in ((
            ("valu_insts_per_launch", roof["valu_insts_per_launch"], insts, 1e-9),
            ("achieved", roof["achieved"], achieved, 1e-3),
            ("frac", roof["frac"], achieved / PEAK, 1e-3),
            ("peak", roof["peak"], PEAK, 1e-6),
            ("traffic", roof["traffic"], traffic, 1e-9),
            ("hbm.frac", roof["hbm"]["frac"], traffic / t / 1e9 / HBM, 2e-3),
            ) + class_busy_rows(roof, mean, insts, t) + ((("valu_busy_estimate", roof["valu_busy_estimate"]["frac"], insts * mix / (1024 * 2.4e9 * t), 2e-3),) if (mix and "valu_busy_estimate" in roof and not from_classes(roof)) else ()) + (
            ("profile launch time vs live launch time", mean("duration_ns@trace") * 1e-9, t, 0.03),)):
        ok = got == want if rel == 0 else close(got, want, rel)
        if not ok:
            problems.append("%s: %s is %r in the line, %r from the counters" % (tag, what, got, want))
    if mix is None and "valu_busy_estimate" in roof and not from_classes(roof):
        problems.append("%s: valu_busy_estimate without a valu_mix_k67w.json of the profile's own round" % tag)
    if roof["frac"] > 1 or roof["hbm"]["frac"] > 1 or ("valu_busy_estimate" in roof and roof["valu_busy_estimate"]["frac"] > 1):
        problems.append("%s: a fraction above 1" % tag)
    rows = class_busy_rows(roof, mean, insts, t)
    busy = ("%.4f..%.4f" % (rows[0][2], rows[1][2])) if rows else (("%.4f" % (insts * mix / (1024 * 2.4e9 * t))) if mix else "n/a")
    return "k67 frac %.4f  hbm %.4f  busy %s  (%d launches, %.3f ms live, %.3f ms in the trace)" % (
        achieved / PEAK, traffic / t / 1e9 / HBM, busy, n, t * 1e3, mean("duration_ns@trace") * 1e-6)


def check_k910(tag, roof, k, problems, profiled_steps):
    """APD workloads: the line's roofline is the weak sweep's, against the L1 tag pipeline."""
    n = roof["launches"]
    mean = timed_mean(k, profiled_steps, n)
    acc = mean("TCP_TOTAL_CACHE_ACCESSES_sum")
    insts = mean("SQ_INSTS_VALU")
    traffic = mean("FETCH_SIZE") * 1024 * 2 + mean("WRITE_SIZE") * 1024
    t = roof["avg_launch_ms"] * 1e-3
    achieved = acc / t / 1e9
    for what, got, want, rel in (
            ("tag_accesses_per_launch", roof["tag_accesses_per_launch"], acc, 1e-9),
            ("achieved", roof["achieved"], achieved, 1e-3),
            ("frac", roof["frac"], achieved / TAG_PEAK, 1e-3),
            ("peak", roof["peak"], TAG_PEAK, 1e-3),
            ("traffic", roof["traffic"], traffic, 1e-9),
            ("hbm.frac", roof["hbm"]["frac"], traffic / t / 1e9 / HBM, 2e-3),
            ("valu.frac", roof["valu"]["frac"], insts / t / 1e9 / PEAK, 2e-3),
            ("profile launch time vs live launch time", mean("duration_ns@trace") * 1e-9, t, 0.03)):
        ok = got == want if rel == 0 else close(got, want, rel)
        if not ok:
            problems.append("%s: %s is %r in the line, %r from the counters" % (tag, what, got, want))
    if roof["frac"] > 1 or roof["hbm"]["frac"] > 1 or roof["valu"]["frac"] > 1:
        problems.append("%s: a fraction above 1" % tag)
    return "k910 tag frac %.4f  hbm %.4f  valu %.4f  (%d launches, %.3f ms live, %.3f ms in the trace)" % (
        achieved / TAG_PEAK, traffic / t / 1e9 / HBM, insts / t / 1e9 / PEAK, n, t * 1e3, mean("duration_ns@trace") * 1e-6)


def check_pass_kernel(tag, roof, k, problems):
    """K14 / K15 of a whole-pass sub-line: one launch per timed pass, every pass the same work."""
    pd = k["per_dispatch_timed"]
    mean = lambda c: sum(pd[c]) / len(pd[c])
    insts = mean("SQ_INSTS_VALU")
    traffic = mean("FETCH_SIZE") * 1024 * 2 + mean("WRITE_SIZE") * 1024
    t = roof["avg_launch_ms"] * 1e-3
    achieved = insts / t / 1e9
    for what, got, want, rel in (
            ("valu_insts_per_launch", roof["valu_insts_per_launch"], insts, 1e-9),
            ("achieved", roof["achieved"], achieved, 1e-3),
            ("frac", roof["frac"], achieved / PEAK, 1e-3),
            ("traffic", roof["traffic"], traffic, 1e-9),
            ("hbm.frac", roof["hbm"]["frac"], traffic / t / 1e9 / HBM, 2e-3),
            ("profile launch time vs live launch time", mean("duration_ns@trace") * 1e-9, t, 0.03)) + class_busy_rows(roof, mean, insts, t):
        if not close(got, want, rel):
            problems.append("%s: %s is %r in the line, %r from the counters" % (tag, what, got, want))
    if roof["frac"] > 1 or roof["hbm"]["frac"] > 1:
        problems.append("%s: a fraction above 1" % tag)
    return "frac %.4f hbm %.4f (%.3f ms live, %.3f ms in the trace)" % (achieved / PEAK, traffic / t / 1e9 / HBM, t * 1e3, mean("duration_ns@trace") * 1e-6)


def sub_lines(name, line):
    """The line itself and every sub-line of its `workloads` block (round 4: every BASELINE config in one process)."""
    yield name, line
    for key, sub in (line.get("workloads") or {}).items():
        yield "%s[%s]" % (name, key), sub


def check(directory):
    problems, lines = [], 0
    mix = load_mix(directory)
    for path in sorted(glob.glob(os.path.join(directory, "bench_*.json"))):
        with open(path) as f:
            first = f.readline()
        try:
            top = json.loads(first)
        except ValueError:
            continue
        if "full_block" in top:   # a compact stdout line (tools/profile_bench.py keeps the line of every counter pass): checked as line_*.json only
            continue
        for tag, line in sub_lines(os.path.basename(path), top):
            for kname, kroof in sorted((line.get("pass_kernels") or {}).items()):   # whole-pass sub-lines: K14 / K15 against their own profile
                if kroof.get("achieved") is None:
                    continue
                prof_path = os.path.join(ROOT, kroof.get("pmc_source") or "")
                if not kroof.get("pmc_source") or not os.path.exists(prof_path):
                    problems.append("%s %s: cites counters but %r is not committed" % (tag, kname, kroof.get("pmc_source")))
                    continue
                prof = json.load(open(prof_path))
                want_kind = "geometric" if "geom" in line["config"].get("state", "") else "photometric"
                if prof["config"]["workload"] != line["config"]["workload"] or list(prof["config"].get("options", [])) != list(line["config"].get("options", [])) \
                        or prof["config"].get("pass_kind", "photometric") != want_kind:
                    problems.append("%s %s: profile %s is of another workload or other options" % (tag, kname, kroof["pmc_source"]))
                    continue
                lines += 1
                print("%-36s %s  %s %s" % (tag, line["config"]["workload"], kname, check_pass_kernel(tag + " " + kname, kroof, prof["kernels"][kname.lower()], problems)))
            roof = line.get("roofline")
            if not roof or roof.get("achieved") is None:
                continue
            cfg = line["config"]
            # the profile the line names (any --steps / --warmup of a workload is served by one profile of that workload, see timed_mean)
            prof_path = os.path.join(ROOT, roof.get("pmc_source") or "")
            if not roof.get("pmc_source") or not os.path.exists(prof_path):
                problems.append("%s: cites counters but %r is not committed" % (tag, roof.get("pmc_source")))
                continue
            prof = json.load(open(prof_path))
            kernels, pcfg = prof["kernels"], prof["config"]
            if pcfg["workload"] != cfg["workload"] or list(pcfg.get("options", [])) != list(cfg.get("options", [])):
                problems.append("%s: profile %s is of workload %s options %s" % (tag, roof["pmc_source"], pcfg["workload"], pcfg.get("options")))
                continue
            if line["steps"] * 2 != roof["launches"]:
                problems.append("%s: %d launches for %d steps" % (tag, roof["launches"], line["steps"]))
            lines += 1
            line_mix = load_mix(os.path.dirname(prof_path)) if os.path.dirname(prof_path) != os.path.normpath(directory) else mix
            if roof.get("bound") == "l1-tag-pipeline":
                msg = check_k910(tag, roof, kernels["k910"], problems, pcfg["steps"])
                strong = line.get("strong_path")
                if strong and strong.get("achieved") is not None:
                    msg += " | " + check_k67(tag + " strong_path", strong, kernels["k67"], line_mix, problems, pcfg["steps"])
            else:
                msg = check_k67(tag, roof, kernels["k67"], line_mix, problems, pcfg["steps"])
            print("%-36s %s  %s" % (tag, cfg["workload"], msg))
    return problems, lines


def check_compact_lines(directory, problems):
    """line_<name>.json (bench.py's stdout) against bench_<name>.json (the full block of the same run)."""
    n = 0
    for path in sorted(glob.glob(os.path.join(directory, "line_*.json"))):
        text = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
        tag = os.path.basename(path)
        full_path = os.path.join(directory, "bench_" + tag[len("line_"):])
        if len(text) != 1 or not os.path.exists(full_path):
            problems.append("%s: %d JSON lines, full block %s" % (tag, len(text), "present" if os.path.exists(full_path) else "missing"))
            continue
        if len(text[0].encode()) > 2000:
            problems.append("%s: %d bytes (the driver's reader is bounded: <= 2000)" % (tag, len(text[0].encode())))
        c, full = json.loads(text[0]), json.loads(open(full_path).readline())
        pairs = [(k, c.get(k), full.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "scaling")]
        pairs += [("config." + k, c["config"].get(k), full["config"].get(k)) for k in ("workload", "width", "height", "num_src", "state")]
        r, fr = c.get("roofline") or {}, full.get("roofline") or {}
        pairs += [("roofline." + k, r.get(k), fr.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches", "pmc_source")]
        pairs += [("roofline.traffic", r.get("traffic"), None if fr.get("traffic") is None else int(fr["traffic"])),
                  ("roofline.hbm_frac", r.get("hbm_frac"), (fr.get("hbm") or {}).get("frac")),
                  ("roofline.algorithmic_GBps", r.get("algorithmic_GBps"), (fr.get("algorithmic") or {}).get("GBps"))]
        if "valu_busy" in r:
            pairs += [("roofline.valu_busy", r.get("valu_busy"), (fr.get("valu_busy_estimate") or {}).get("frac"))]
        if "valu_busy_hi" in r:
            pairs += [("roofline.valu_busy_hi", r.get("valu_busy_hi"), (fr.get("valu_busy_estimate") or {}).get("frac_hi"))]
        if "frac_kind" in r:   # round 6 on: the line says which bound `frac` is a fraction of, and carries SURVEY 8(d)'s byte ratio beside it
            gb = (fr.get("algorithmic") or {}).get("GBps", (fr.get("algorithmic") or {}).get("GBps_nominal_max"))
            pairs += [("roofline.frac_kind", r.get("frac_kind"), fr.get("bound")),
                      ("roofline.algorithmic_over_hbm_peak", r.get("algorithmic_over_hbm_peak"), None if gb is None else round(gb / 8000.0, 2))]
        fw = full.get("workloads") or {}
        if "value_config_iters" in c:
            it6 = fw.get("configs1_office_6iter") or {}
            pairs += [("value_config_iters", c["value_config_iters"],
                       full["value"] if full.get("steps") == 6 and full["config"].get("workload") == "eth3d_office_fullres_8src" else it6.get("value"))]
        if "whole_pass" in c:
            wp = fw.get("configs2_pipes_apd_whole_pass") or {}
            pairs += [("whole_pass", c["whole_pass"], None if wp.get("value") is None else [wp["value"], wp.get("ms_per_pass")])]
        if "configs0_cpu" in c:
            c0 = (fw.get("configs0_office_halfres_2src_3iter") or {}).get("cpu_baseline")
            pairs += [("configs0_cpu", c["configs0_cpu"], None if not c0 else [c0["value"], c0["cores"]])]
        cb, fcb = c.get("cpu_baseline") or {}, full.get("cpu_baseline") or {}
        pairs += [("cpu_baseline." + k, cb.get(k), fcb.get(k)) for k in ("value", "cores", "kind")]
        if set(c.get("workloads") or {}) != set(full.get("workloads") or {}):
            problems.append("%s: the compact line's sub-lines are not the full block's" % tag)
        for key, triple in (c.get("workloads") or {}).items():
            w = (full.get("workloads") or {}).get(key) or {}
            roof = w.get("roofline") or (w.get("pass_kernels") or {}).get("K14") or (w.get("weak_path") or {}).get("roofline") or {}
            pairs += [("workloads.%s.value" % key, triple[0], w.get("value")),
                      ("workloads.%s.ms" % key, triple[1], w.get("ms_per_pass", w.get("ms_per_step"))),
                      ("workloads.%s.frac" % key, triple[2], roof.get("frac"))]
        bad = [(k, a, b) for k, a, b in pairs if a != b]
        for k, a, b in bad:
            problems.append("%s: %s is %r in the compact line, %r in the full block" % (tag, k, a, b))
        n += 1
        print("%-36s %d bytes, %d fields equal to %s" % (tag, len(text[0].encode()), len(pairs) - len(bad), os.path.basename(full_path)))
    return n


def newest_round():
    rounds = sorted(d for d in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]")) if glob.glob(os.path.join(d, "bench_*.json")))
    return rounds[-1] if rounds else os.path.join(ROOT, "profiles", "r02")


def main():
    directory = sys.argv[1] if len(sys.argv) > 1 else newest_round()   # the newest round that holds bench lines, like bench.py's profile choice
    problems, lines = check(directory)
    check_compact_lines(directory, problems)
    for p in problems:
        print("MISMATCH " + p)
    if lines == 0:
        print("no bench line with counter fields under " + directory)
        return 1
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
