#!/usr/bin/env python3
"""bench.py -- Mpix*iterations/s of the PatchMatch sweep (BASELINE.json metric) on 1..8 MI355X.

One "step" = one iteration of the loop body APD.cu:2443-2457 (K6 black + K7 red strong update,
K8 fit-plane, K9/K10 weak update when WEAK pixels exist) over one reference view.  The K timed steps are iterations
0..K-1 of a freshly initialised pass (the warm-up iterations run first, then the state is reset with the same seed).  N=1 workload =
BASELINE.json configs[1]: ETH3D-office-shaped stand-in, 6200x4130, 8 source views (datasets are not
in the image: SURVEY.md 8d synthetic generator).  With --gpus N each rank sweeps its own reference
view (views shard, no collective in the data path: weak scaling) and the final depth/normal maps are
all-gathered over RCCL after the timed region, as they would be before fusion.

`python bench.py --gpus N` with N > 1 starts the N ranks itself (one process per GPU through torch.distributed.run on
127.0.0.1) unless it is already running under a launcher (WORLD_SIZE set); it refuses to run when fewer than N devices are
visible instead of quietly measuring fewer.

The same line carries a `workloads` block: every other BASELINE.json config timed in this process after the headline (SUB_WORKLOADS:
configs[1] at 6 and 3 iterations, configs[2] with adaptive patches and K9/K10's roofline, the configs[4] shape, 1080p frames, a
pass of the sharded scheduler with its depth exchange inside the timed region) and whole `apd_run` passes (PASS_WORKLOADS: K1..K15
with K14 / K15 against their own counter profile).

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Handles launch on their own HIP streams; the runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two views
# in flight whose streams share a queue take turns.  Eight queues, as the drop-in binary sets for itself (host/main.cpp,
# profiles/r06/ab_hw_queues.txt); only the lines with several views in flight per GPU can notice.  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
# Vector-ALU issue ceiling of the chip: 256 CUs x 4 SIMDs at the 2.4 GHz maximum clock, one wave64 binary32 multiply / add /
# FMA per 2 cycles per SIMD (MI355X_MICROARCH.md "v_fma_f32 (wave64) 2 cyc"; tools/valu_issue.hip measures a plateau of
# 2.2 cycles, profiles/r02/valu_issue.csv).  That is the fastest instruction class: v_fract / conversions / v_fma_mix /
# integer multiply-add issue in 4 cycles, v_rcp_f32 in 8, packed binary32 in 4 -- so `frac` against this peak is a LOWER
# bound of how busy the pipe is; `valu_busy_estimate` weights the instruction count with the mean issue cost of the
# kernel's own 36-sample body (tools/valu_mix.py -> profiles/r02/valu_mix_k67w.json).
NUM_SIMDS = 1024
MAX_CLOCK_GHZ = 2.4
VALU_CYCLES_PER_WAVE_INST = 2.0
VALU_PEAK_GINST = NUM_SIMDS * MAX_CLOCK_GHZ / VALU_CYCLES_PER_WAVE_INST  # G wave64 instructions / s
TCP_ACCESSES_PER_CLOCK = 1.85  # L1 tag accesses per clock and CU with every access a hit (profiles/r02/unaligned_gather.txt: 59 in 32 cycles)

DEFAULT_WORKLOAD = "eth3d_office_fullres_8src"
WORKLOADS = {
    # name: (width, height, num_src)
    "eth3d_office_fullres_8src": (6200, 4130, 8),    # BASELINE.json configs[1]
    "eth3d_office_halfres_2src": (3100, 2065, 2),    # configs[0]: half resolution, 2 source views (the reference's CPU-path case)
    "eth3d_pipes_fullres_10src": (6200, 4130, 10),   # configs[2] shape (strong pixels only here)
    "synthetic_4096x3072_16src": (4096, 3072, 16),   # configs[4]
    "tt_family_1080p_10src": (1920, 1080, 10),       # configs[3] shape
    "synthetic_4096x3072_8src": (4096, 3072, 8),     # tuning workload
    "small": (640, 480, 8),
}
# BASELINE.json configs[2] ("adaptive-patch on"): the timed sweep is the REFINE_INIT pass with use_APD, whose WEAK map
# comes from a complete FIRST_INIT pass (K1..K15) run before the timed region; 20 % of the scene is textureless.
APD_WORKLOADS = {"eth3d_pipes_fullres_10src_apd": "eth3d_pipes_fullres_10src", "synthetic_4096x3072_8src_apd": "synthetic_4096x3072_8src",
                 "small_apd": "small"}
WORKLOADS.update({k: WORKLOADS[v] for k, v in APD_WORKLOADS.items()})
# <name>_hard: the same shape on synth.HARD -- slabs in front of the planes (depth steps, occlusions), per-view gain / offset, a wider
# camera ring whose sources aim off the target (parts of the frame project outside a source).  The default scene is the best case of
# the LDS windows and the refinement early-outs; these lines put the other end on the same clock.
HARD_SUFFIX = "_hard"


def resolve_workload(name):
    """Named workloads above, or an ad-hoc shape for tuning runs: custom_<W>x<H>_<N>src[_apd]."""
    import re
    if name.endswith(HARD_SUFFIX):
        name = name[:-len(HARD_SUFFIX)]
    if name in WORKLOADS:
        return WORKLOADS[name], name in APD_WORKLOADS
    m = re.match(r"^custom_(\d+)x(\d+)_(\d+)src(_apd)?$", name)
    if not m:
        raise SystemExit("bench.py: unknown workload %r (named: %s; or custom_<W>x<H>_<N>src[_apd]; any of them with the suffix _hard)"
                         % (name, ", ".join(sorted(WORKLOADS))))
    return (int(m.group(1)), int(m.group(2)), int(m.group(3))), m.group(4) is not None


def scene_kwargs(synth, name):
    """Scene preset of a workload name: synth.HARD for <name>_hard, the default two-plane scene otherwise."""
    return dict(synth.HARD) if name.endswith(HARD_SUFFIX) else {}


def algorithmic_bytes_per_weak_pixel(num_src):
    """SURVEY.md 8(d), nominal: 15*N NCCNew of 108 samples + N NCCOld of 36 samples, 20 B per sample, + 176 B of state."""
    return 33120 * num_src + 176


def algorithmic_bytes_per_strong_pixel(num_src):
    """14*N NCCs x 36 samples x (4 B ref texel + 16 B bilinear taps) + 176 B of state."""
    return 10080 * num_src + 176


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD,
                    help="one of %s, or custom_<W>x<H>_<N>src[_apd]" % ", ".join(sorted(WORKLOADS)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="1024x768", help="WxH of the CPU-baseline sample")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="apd_set_option on the handle (A/B runs), e.g. --opt k67_windows=0; names: fast_rcp early_out source_quads "
                         "tiled_copy k67_windows k1415_windows.  Reported in config.options")
    ap.add_argument("--no-workloads", action="store_true", help="headline only: skip the `workloads` block (the other BASELINE configs)")
    ap.add_argument("--only-workloads", action="append", default=[], metavar="KEY", help="restrict the `workloads` block to these keys")
    ap.add_argument("--full-line", action="store_true",
                    help="tooling only (tools/ab.sh, tools/profile_round.sh): print the full block as the stdout line instead of the compact one")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="launcher / collective plumbing only, on CPU with gloo: no PatchMatch work is done or reported "
                         "(value is null); used by tests/test_bench_launcher.py to cover the --gpus N spawn path without GPUs")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: one process per GPU via torch.distributed.run, rendezvous on 127.0.0.1."""
    if not args.selftest_cpu:
        import torch
        found = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if found < args.gpus:
            sys.stderr.write("bench.py: --gpus %d requested, %d HIP device(s) visible: refusing to measure fewer ranks than asked for\n"
                             % (args.gpus, found))
            return 3
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class Ctx:
    """What every measurement of one process shares: package, device, rank layout, process group."""

    def __init__(self, pkg, synth, np, torch, dev, rank, world, distributed, dist):
        self.pkg, self.synth, self.np, self.torch = pkg, synth, np, torch
        self.dev, self.rank, self.world, self.distributed, self.dist = dev, rank, world, distributed, dist

    def barrier(self, handles=()):
        if self.distributed:
            self.dist.barrier()
        self.torch.cuda.synchronize()
        for h in handles:
            h.synchronize()

    def max_over_ranks(self, values):
        if not self.distributed:
            return [float(v) for v in values]
        t = self.torch.tensor(list(values), device=self.dev, dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    def gather_scalar(self, value):
        if not self.distributed:
            return [float(value)]
        out = self.torch.zeros(self.world, device=self.dev, dtype=self.torch.float64)
        self.dist.all_gather_into_tensor(out, self.torch.tensor([value], device=self.dev, dtype=self.torch.float64))
        return [float(v) for v in out.tolist()]


class SweepWorkload:
    """One reference view per owned slot of a named workload, inputs resident in HBM: scene, handle, (APD: the prior of an
    untimed FIRST_INIT pass).  measure() times iterations 0..K-1 of a freshly initialised pass, any number of times."""

    def __init__(self, ctx, name, max_iters, opts=(), seed=12345, views_per_gpu=1):
        pkg, torch, np = ctx.pkg, ctx.torch, ctx.np
        self.ctx, self.name, self.opts, self.seed, self.max_iters = ctx, name, list(opts), seed, max_iters
        (self.W, self.H, self.N), self.apd_mode = resolve_workload(name)
        W, H, N = self.W, self.H, self.N
        t0 = time.perf_counter()
        self.handles, self.priors, self.gt = [], [], []
        self.weak_fraction = 0.0
        self.pass_inputs, self.geom = None, None
        for slot in range(views_per_gpu):
            # every rank owns different reference views of the same camera ring (round-robin, as the schedulers shard them)
            view = ctx.rank + slot * ctx.world
            keep = views_per_gpu == 1 and any(e[1] == name and e[4] == "geometric" for e in PASS_WORKLOADS)   # a geometric whole pass follows
            sc = ctx.synth.make_scene(W, H, N, seed=0, ref_view=view, device=ctx.dev, textureless=0.2 if self.apd_mode else 0.0,
                                      keep_view_depths=keep, **scene_kwargs(ctx.synth, name))
            cams = [pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
            dmin, dmax = 0.6 * sc.depth_min, 1.2 * sc.depth_max
            if keep:
                self.pass_inputs = {"cams": cams, "images": list(sc.images), "depths": list(sc.view_depths), "dmin": dmin, "dmax": dmax}
            prior = None
            if not self.apd_mode:
                params = pkg.default_params(num_images=N + 1, depth_min=dmin, depth_max=dmax, use_APD=0,
                                            state=pkg.FIRST_INIT, max_iterations=max_iters, seed=seed)
                h = pkg.Handle(W, H, params, device=ctx.dev.index)
                apply_options(h, self.opts)
                h.upload_views(cams, sc.images)  # device->device copies: inputs are resident in HBM before timing
            else:
                # untimed: the photometric FIRST_INIT pass of main.cpp:169-190 (3 iterations, weak_peak_radius 6) that
                # classifies pixels, then ProcessProblem's post-processing (main.cpp:105-115)
                p0 = pkg.default_params(num_images=N + 1, depth_min=dmin, depth_max=dmax, use_APD=0, state=pkg.FIRST_INIT,
                                        max_iterations=3, weak_peak_radius=6, seed=seed)
                h0 = pkg.Handle(W, H, p0, device=ctx.dev.index)
                h0.upload_views(cams, sc.images)
                h0.run()
                planes, weak, views = h0.download()
                h0.close()
                bad = (planes[..., 3] < np.float32(dmin)) | (planes[..., 3] > np.float32(dmax))
                planes[..., 3][bad] = 0
                weak[bad] = pkg.UNKNOWN
                params = pkg.default_params(num_images=N + 1, depth_min=dmin, depth_max=dmax, use_APD=1, state=pkg.REFINE_INIT,
                                            max_iterations=max_iters, weak_peak_radius=6, rotate_time=4,
                                            ransac_threshold=0.01 - 0.00125 * 3, seed=seed + 1)
                h = pkg.Handle(W, H, params, device=ctx.dev.index)
                apply_options(h, self.opts)
                h.upload_views(cams, sc.images)
                prior = (planes, views, weak)
            self.handles.append(h)
            self.priors.append(prior)
            self.gt.append(sc.gt_depth)
            del sc.images[:]
            del sc
        torch.cuda.empty_cache()
        self.init_pass()
        if self.apd_mode:
            self.weak_fraction = self.handles[0].weak_count / float(W * H)
        self.setup_s = time.perf_counter() - t0

    def init_pass(self):
        """Everything before the loop of APD.cu:2443 (not timed): prior state, K1..K5."""
        pkg = self.ctx.pkg
        for h, prior in zip(self.handles, self.priors):
            if self.apd_mode:
                h.upload_prior(*prior)
            h.run_kernel(pkg.K1)
            h.run_kernel(pkg.K2)
            if self.apd_mode and h.weak_count > 0:
                h.run_kernel(pkg.K3)
                h.run_kernel(pkg.K4)
            h.run_kernel(pkg.K5)

    def close(self):
        for h in self.handles:
            h.close()
        if self.geom is not None:
            self.geom[0].close()
        self.handles = []
        self.priors = []
        self.pass_inputs, self.geom = None, None
        self.ctx.torch.cuda.empty_cache()

    def measure(self, steps, warmup, pass_exchange=False):
        """Times `steps` iterations of the sweep on every owned view (barrier + synchronize on both sides, MAX over ranks).
        pass_exchange: the timed region also holds what ends a pass of the sharded scheduler -- K11..K13, the depth export and
        the all-gather of every view's depth map (the exchange the reference does through depths.dmb, APD.cpp:497-500)."""
        ctx, pkg, torch = self.ctx, self.ctx.pkg, self.ctx.torch
        W, H, N = self.W, self.H, self.N
        hs = self.handles
        # Warm-up: W iterations of the same sweep (clocks, caches, code objects).  Then the state is re-initialised with the
        # same seed (K1 + K5 again, untimed) so that the timed region is exactly what the config names: the first K
        # iterations of a pass, INCLUDING iteration 0, whose random planes scatter the gathers over the whole source images
        # and which costs about twice a later iteration.
        if warmup > 0:
            for h in hs:
                h.run_sweeps(0, warmup)
        self.init_pass()
        for h in hs:
            h.profile_enable(True)
            h.profile_reset()
        depth = [torch.empty((H, W), device=ctx.dev, dtype=torch.float32) for _ in hs] if pass_exchange else None
        gathered = None
        pass_allgather_ms = None

        gc.collect()   # a collection of the interpreter inside a 40 ms region would be timed as this rank's work
        gc.disable()
        ctx.barrier(hs)
        t0 = time.perf_counter()
        if len(hs) == 1:
            hs[0].run_sweeps(0, 1, sync=False)
            hs[0].synchronize()     # one view: iteration 0 (random planes) is reported on its own
            t_first = time.perf_counter()
            if steps > 1:
                hs[0].run_sweeps(1, steps - 1, sync=False)
        else:                       # several views in flight: every view's sweep is queued at once, each on its handle's stream
            for h in hs:
                h.run_sweeps(0, steps, sync=False)
            t_first = t0
        if pass_exchange:
            for h, d in zip(hs, depth):
                for kid in (pkg.K11, pkg.K12, pkg.K13):
                    h.run_kernel(kid)
                h.export_depth_normal(d, None)
        for h in hs:
            h.synchronize()
        torch.cuda.synchronize()
        t_rank = time.perf_counter() - t0  # this rank's own work, before waiting for the others
        if pass_exchange:
            ta = time.perf_counter()
            gathered = self.exchange_depths(depth)
            torch.cuda.synchronize()
            pass_allgather_ms = (time.perf_counter() - ta) * 1e3
        if ctx.distributed:
            ctx.dist.barrier()
        elapsed = time.perf_counter() - t0
        first_iter_s = t_first - t0
        elapsed, first_iter_s = ctx.max_over_ranks([elapsed, first_iter_s])
        rank_ms_per_step = ctx.gather_scalar(t_rank / steps * 1e3)
        gc.enable()
        prof = hs[0].profile()
        for h in hs:
            h.profile_enable(False)

        # after the timed region: post-loop kernels + all-gather of depth/normal maps (before fusion)
        allgather_ms = None
        t_post0 = time.perf_counter()
        if not pass_exchange:
            for kid in (pkg.K11, pkg.K12, pkg.K13):
                hs[0].run_kernel(kid)
            d0 = torch.empty((H, W), device=ctx.dev, dtype=torch.float32)
            n0 = torch.empty((H, W, 3), device=ctx.dev, dtype=torch.float32)
            hs[0].export_depth_normal(d0, n0)
            torch.cuda.synchronize()
            if ctx.distributed and len(hs) == 1:
                from apd_mvs_amd import sharding
                ta = time.perf_counter()
                gathered_d = sharding.allgather_maps({ctx.rank: d0.view(H, W, 1)}, ctx.world)   # view index == rank here
                torch.cuda.synchronize()
                pass_allgather_ms = (time.perf_counter() - ta) * 1e3   # what every pass ends with: the sources' depth maps for the next one
                gathered_n = sharding.allgather_maps({ctx.rank: n0}, ctx.world)
                torch.cuda.synchronize()
                allgather_ms = (time.perf_counter() - ta) * 1e3
                assert gathered_d.shape[0] == ctx.world and torch.equal(gathered_d[ctx.rank, :, :, 0], d0)
                assert gathered_n.shape[0] == ctx.world
                del gathered_d, gathered_n
            del n0
        else:
            d0 = depth[0]
            views = len(hs) * ctx.world
            assert gathered.shape[0] == views
            for slot, d in enumerate(depth):   # every rank holds every view's map, its own ones bit for bit
                assert torch.equal(gathered[ctx.rank + slot * ctx.world, :, :, 0], d)
        gt = self.gt[0]
        err = (d0 - gt).abs() / gt
        within = float((err[8:-8, 8:-8] < 0.01).float().mean().item())
        post_ms = (time.perf_counter() - t_post0) * 1e3
        del d0, depth, gathered

        mpix = W * H / 1e6
        views_total = ctx.world * len(hs)
        value = views_total * mpix * steps / elapsed
        kernel_ms = {pkg.KERNEL_NAMES[k]: round(v[0], 3) for k, v in sorted(prof.items())}
        roofline = strong_roofline(pkg, prof, W, H, N, self.weak_fraction, self.name, steps, warmup, self.opts, self.seed)
        weak_path = weak_roofline = None
        if self.apd_mode:
            weak_path, weak_roofline = weak_rooflines(pkg, prof, W, H, N, self.weak_fraction, self.name, steps, warmup, self.opts, self.seed)
        later_s = elapsed - first_iter_s
        return {
            "value": round(value, 4), "unit": "Mpix*iter/s", "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "timed_region_ms": round(elapsed * 1e3, 3),
            "config": {"workload": self.name, "width": W, "height": H, "num_src": N,
                       "state": "REFINE_INIT+APD" if self.apd_mode else "FIRST_INIT",
                       "views_per_gpu": len(hs), "parallelism": "views sharded, %d rank(s)" % ctx.world,
                       "backend": "nccl" if ctx.distributed else "single process", "options": self.opts,
                       "timed_region": ("%d sweep iterations per view + K11..K13 + depth export + all-gather of every view's depth map"
                                        % steps) if pass_exchange else "%d sweep iterations (APD.cu:2443-2457)" % steps},
            # two views in flight on one device share the CUs: per-launch times of such a line are not a kernel's own
            "roofline": None if len(hs) > 1 else (weak_roofline if self.apd_mode else roofline),
            "strong_path": roofline if (self.apd_mode and len(hs) == 1) else None,
            "weak_path": weak_path if len(hs) == 1 else None,   # per-launch event times of views sharing the CUs are not a kernel's own
            "iterations": {"first_ms": round(first_iter_s * 1e3, 3) if len(hs) == 1 else None,   # several views in flight: iteration 0 is not timed on its own
                           "later_ms_per_step": round(later_s / max(steps - 1, 1) * 1e3, 3) if steps > 1 and len(hs) == 1 and not pass_exchange else None,
                           "later_value": round(views_total * mpix * (steps - 1) / later_s, 4) if steps > 1 and len(hs) == 1 and not pass_exchange else None,
                           "note": "timed region = iterations 0..K-1 of a freshly initialised pass; iteration 0 starts from random planes"},
            "rank_ms_per_step": [round(v, 3) for v in rank_ms_per_step],
            "kernel_ms_timed_region": kernel_ms,
            "post_loop_ms": round(post_ms, 1),
            "allgather_ms": None if allgather_ms is None else round(allgather_ms, 3),
            "pass_allgather_ms": None if pass_allgather_ms is None else round(pass_allgather_ms, 3),
            "pass_allgather_inside_timed_region": bool(pass_exchange),
            "quality_within_1pct_depth": round(within, 4),
            "setup_s": round(self.setup_s, 2),
        }

    def geometric_handle(self):
        """The pass that follows the photometric one at a level (main.cpp:191-213): REFINE_ITER + APD + geometric consistency,
        weak_peak_radius 4.  Prior = the state one (untimed) photometric pass leaves, post-processed as ProcessProblem does
        (main.cpp:105-115); depth maps of the sources = the analytic depth of every source view, i.e. what converged neighbours hold."""
        if self.geom is None:
            ctx, pkg, np = self.ctx, self.ctx.pkg, self.ctx.np
            pi = self.pass_inputs
            h = self.handles[0]
            h.upload_prior(*self.priors[0])
            h.run()
            planes, weak, views = h.download()
            bad = (planes[..., 3] < np.float32(pi["dmin"])) | (planes[..., 3] > np.float32(pi["dmax"]))
            planes[..., 3][bad] = 0
            weak[bad] = pkg.UNKNOWN
            params = pkg.default_params(num_images=self.N + 1, depth_min=pi["dmin"], depth_max=pi["dmax"], use_APD=1, state=pkg.REFINE_ITER,
                                        geom_consistency=1, max_iterations=PASS_ITERATIONS, weak_peak_radius=4, rotate_time=4,
                                        ransac_threshold=0.01 - 0.00125 * 3, seed=self.seed + 2)
            hg = pkg.Handle(self.W, self.H, params, device=ctx.dev.index)
            apply_options(hg, self.opts)
            hg.upload_views(pi["cams"], pi["images"], pi["depths"])
            self.geom = (hg, (planes, views, weak), float((weak == pkg.WEAK).mean()))
        return self.geom

    def measure_whole_pass(self, passes, warmup, kind="photometric"):
        """Times whole `apd_run` calls = APD::RunPatchMatch (APD.cu:2408-2470) on the first owned view: K1..K5, the three sweep
        iterations, K11..K13, K14 (DepthToWeak) and K15 (LocalRefine) of a REFINE_INIT + APD pass (kind "photometric") or of the
        REFINE_ITER + APD + geometric-consistency pass that follows it three times per level (kind "geometric") -- the states the
        reference's schedule runs at the full frame size (main.cpp:172-213).  The prior state is uploaded before every pass, outside
        the timed region (the in-memory scheduler hands it over device to device)."""
        ctx, pkg, torch = self.ctx, self.ctx.pkg, self.ctx.torch
        assert self.apd_mode and self.max_iters == PASS_ITERATIONS and len(self.handles) == 1
        W, H, N = self.W, self.H, self.N
        if kind == "geometric":
            h, prior, weak_fraction = self.geometric_handle()
        else:
            h, prior, weak_fraction = self.handles[0], self.priors[0], self.weak_fraction
        for _ in range(warmup):
            h.upload_prior(*prior)
            h.run()
        h.profile_enable(True)
        h.profile_reset()
        per_pass = []
        gc.collect()
        gc.disable()
        ctx.barrier([h])
        t_region = time.perf_counter()
        t_run = 0.0
        for _ in range(passes):
            h.upload_prior(*prior)   # not timed
            ctx.barrier([h])
            t0 = time.perf_counter()
            h.run()
            h.synchronize()
            torch.cuda.synchronize()
            t_own = time.perf_counter() - t0
            if ctx.distributed:
                ctx.dist.barrier()
            per_pass.append(ctx.max_over_ranks([time.perf_counter() - t0])[0])
            t_run += t_own
        region_s = time.perf_counter() - t_region
        gc.enable()
        prof = h.profile()
        h.profile_enable(False)
        elapsed = sum(per_pass)
        rank_ms = ctx.gather_scalar(t_run / passes * 1e3)
        d0 = torch.empty((H, W), device=ctx.dev, dtype=torch.float32)   # after the region: what the pass left, against the analytic depth
        h.export_depth_normal(d0, None)
        torch.cuda.synchronize()
        gt = self.gt[0]
        within = float((((d0 - gt).abs() / gt)[8:-8, 8:-8] < 0.01).float().mean().item())
        del d0
        kernel_ms = {pkg.KERNEL_NAMES[k]: round(v[0] / passes, 3) for k, v in sorted(prof.items())}
        total_kernel = sum(kernel_ms.values())
        mpix = W * H / 1e6
        return {
            "value": round(ctx.world * mpix * PASS_ITERATIONS * passes / elapsed, 4), "unit": "Mpix*iter/s",
            "passes": passes, "warmup_passes": warmup, "iterations_per_pass": PASS_ITERATIONS,
            "ms_per_pass": round(elapsed / passes * 1e3, 3), "timed_region_ms": round(elapsed * 1e3, 3),
            "wall_with_prior_uploads_ms": round(region_s * 1e3, 1),
            "Mpix_per_s_whole_pass": round(ctx.world * mpix * passes / elapsed, 3),
            "config": {"workload": self.name, "width": W, "height": H, "num_src": N,
                       "state": "REFINE_ITER+APD+geom_consistency" if kind == "geometric" else "REFINE_INIT+APD", "views_per_gpu": 1,
                       "weak_fraction": round(weak_fraction, 4),
                       "source_depth_maps": "analytic depth of every source view (converged neighbours)" if kind == "geometric" else None,
                       "parallelism": "views sharded, %d rank(s)" % ctx.world, "backend": "nccl" if ctx.distributed else "single process",
                       "options": self.opts,
                       "timed_region": "apd_run = APD::RunPatchMatch (APD.cu:2408-2470): K1..K5, %d sweep iterations, K11..K13, K14, K15; "
                                       "prior state uploaded before each pass, outside the region" % PASS_ITERATIONS},
            "roofline": None,
            "kernel_ms_per_pass": kernel_ms,
            "share_of_kernel_time": {k: round(v / total_kernel, 4) for k, v in sorted(kernel_ms.items(), key=lambda kv: -kv[1])[:6]} if total_kernel > 0 else None,
            "pass_kernels": {"K14": pass_kernel_roofline(pkg, prof, pkg.K14, "k14", self.name, kind, self.opts, self.seed),
                             "K15": pass_kernel_roofline(pkg, prof, pkg.K15, "k15", self.name, kind, self.opts, self.seed)},
            "rank_ms_per_pass": [round(v, 3) for v in rank_ms],
            "quality_within_1pct_depth": round(within, 4),
            "setup_s": round(self.setup_s, 2),
        }

    def exchange_depths(self, depth):
        """Every rank's depth maps to every rank: one padded equal-count all-gather (sharding.allgather_maps) under a process
        group, the identity for a single process."""
        ctx, torch = self.ctx, self.ctx.torch
        views = len(depth) * ctx.world
        if ctx.distributed:
            from apd_mvs_amd import sharding
            return sharding.allgather_maps({ctx.rank + s * ctx.world: d.view(self.H, self.W, 1) for s, d in enumerate(depth)}, views)
        out = torch.empty((views, self.H, self.W, 1), device=ctx.dev, dtype=torch.float32)
        for s, d in enumerate(depth):
            out[s, :, :, 0].copy_(d)
        return out


def strong_roofline(pkg, prof, W, H, N, weak_fraction, workload, steps, warmup, opts, seed):
    """Roofline of K6/K7 (strong update).  Live in this run: the kernel's average launch duration (HIP events on the handle's
    stream).  From the committed rocprofv3 counter passes of this workload (tools/profile_bench.py): VALU instructions and
    memory-side bytes per launch, reduced over the launches this command line times.  The kernel is limited by vector-ALU issue
    (DESIGN.md 6), so that is the bound the fraction is taken against; the memory-side rate and the SURVEY 8(d) algorithmic
    count are reported beside it."""
    k6 = prof.get(pkg.K6, (0.0, 0))
    k7 = prof.get(pkg.K7, (0.0, 0))
    launches = k6[1] + k7[1]
    avg_ms = (k6[0] + k7[0]) / max(launches, 1)
    # K6/K7 skip WEAK pixels; the weak fraction is the one at upload time (K4 only ever lowers it)
    bytes_per_launch = (W * H / 2.0) * (1.0 - weak_fraction) * algorithmic_bytes_per_strong_pixel(N)
    alg_gbps = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    pmc = load_pmc_profile(workload, steps, warmup, "k67", opts, seed)
    roofline = {
        "bound": "valu-issue", "kernel": "k67w_update_strong (Black/RedPixelUpdateStrong, LDS source windows)",
        "achieved": None, "peak": round(VALU_PEAK_GINST, 1), "unit": "Gwave-inst/s", "frac": None, "traffic": None,
        "avg_launch_ms": round(avg_ms, 3), "launches": launches,
        "peak_note": "%d SIMDs x %.1f GHz / %.0f cycles per wave64 VALU instruction (tools/valu_issue.hip)"
                     % (NUM_SIMDS, MAX_CLOCK_GHZ, VALU_CYCLES_PER_WAVE_INST),
        "hbm": None,
        "algorithmic": {"bytes_per_launch": bytes_per_launch, "bytes_per_pixel_iter": algorithmic_bytes_per_strong_pixel(N),
                        "GBps": round(alg_gbps, 1),
                        "note": "SURVEY 8(d) nominal count (every tap of every nominal NCC priced as 20 B of HBM traffic); it counts "
                                "L1/L2/LDS hits and early-outed NCCs, so it exceeds any memory roofline and is NOT one"},
        "pmc_source": None,
    }
    if pmc is not None:
        insts = pmc["valu_insts_per_launch"]
        achieved = insts / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = pmc["hbm_bytes_per_launch"]
        hbm_gbps = traffic / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        roofline.update({
            "achieved": round(achieved, 1), "frac": round(achieved / VALU_PEAK_GINST, 4), "traffic": traffic,
            "valu_insts_per_launch": insts,
            "hbm": {"achieved": round(hbm_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(hbm_gbps / HBM_PEAK_GBPS, 4),
                    "bytes_per_launch": traffic, "note": "2 x FETCH_SIZE + WRITE_SIZE per launch (gfx950 correction of the guide)"},
            "pmc_source": pmc["source"], "pmc_launch_ms": pmc.get("launch_ms"), "pmc_profile_steps": pmc["profile_steps"],
            "pmc_extrapolated_launches": pmc["extrapolated_launches"],
        })
        busy = valu_busy_from_classes(insts, pmc.get("valu_classes"), avg_ms)
        if busy is not None:
            roofline["valu_busy_estimate"] = busy
        else:   # profiles of rounds 2-5 hold no class counters: the static mix of the window body, as those rounds priced it
            mix = load_valu_mix(os.path.dirname(pmc["source"]))
            if mix is not None:
                b = insts * mix["mean_cycles_per_inst"] / (NUM_SIMDS * MAX_CLOCK_GHZ * 1e9 * avg_ms * 1e-3)
                roofline["valu_busy_estimate"] = {
                    "frac": round(b, 4), "mean_issue_cycles_per_inst": mix["mean_cycles_per_inst"], "source": mix["source"],
                    "note": "VALU instructions per launch x mean issue cycles of the kernel's LDS-window body (static mix of ONE basic block); "
                            "superseded in round 6 by the whole-kernel class counters"}
    else:
        roofline["pmc_note"] = ("no committed rocprofv3 counter profile for workload=%s options=%s seed=%d under profiles/: "
                                "achieved / frac / traffic are null rather than borrowed from another configuration"
                                % (workload, list(opts), seed))
    return roofline


# measured issue cost per wave64 instruction and SIMD (tools/valu_issue.hip, profiles/r02/valu_issue.csv): plain binary32 add / mul / fma (and
# v_mov, v_and, v_add_u32) 2.2 cycles; conversions, v_fract, v_fma_mix, 24-bit multiply-adds, shift-adds, v_med3 4.07; v_rcp / v_sqrt 8.1
VALU_COST_FAST, VALU_COST_SLOW, VALU_COST_TRANS = 2.2, 4.067, 8.108


def valu_busy_from_classes(insts, classes, avg_ms):
    """How busy the vector ALU is over the WHOLE kernel: the launch's VALU instructions by class (SQ_INSTS_VALU_* counters, every basic
    block at its real execution count) priced with the measured issue costs.  FMA / ADD / MUL are the 2.2-cycle class, TRANS 8.1, CVT
    4.07; INT32 / INT64 and the instructions no class counter names (v_fract, v_mov, v_cndmask, compares, v_fma_mix ...) hold members of
    both the 2.2- and the 4.07-cycle class, so they are priced both ways: `frac` is the LOWER bound (all of them fast), `frac_hi` the
    upper one.  VERDICT r05 weak #6: rounds 2-5 applied the mix of one basic block to every instruction of the launch."""
    if not classes or avg_ms <= 0 or not insts:
        return None
    fast = classes["fma"] + classes["add"] + classes["mul"]
    other = max(insts - fast - classes["trans"] - classes["cvt"], 0.0)   # INT32 + INT64 + unclassified
    base = VALU_COST_FAST * fast + VALU_COST_TRANS * classes["trans"] + VALU_COST_SLOW * classes["cvt"]
    cyc_lo, cyc_hi = base + VALU_COST_FAST * other, base + VALU_COST_SLOW * other
    simd_cycles = NUM_SIMDS * MAX_CLOCK_GHZ * 1e9 * avg_ms * 1e-3
    return {"frac": round(cyc_lo / simd_cycles, 4), "frac_hi": round(cyc_hi / simd_cycles, 4),
            "mean_issue_cycles_per_inst": [round(cyc_lo / insts, 3), round(cyc_hi / insts, 3)],
            "class_share": {k: round(v / insts, 4) for k, v in (("fma_add_mul_f32", fast), ("trans_f32", classes["trans"]), ("cvt", classes["cvt"]),
                                                                 ("int32", classes["int32"]), ("int64", classes["int64"]),
                                                                 ("unclassified", max(other - classes["int32"] - classes["int64"], 0.0)))},
            "source": "SQ_INSTS_VALU_* of the same counter profile (whole kernel, per launch)",
            "note": "sum over classes of instructions x measured issue cycles / (1024 SIMDs x 2.4 GHz x launch time); frac = lower bound (the classes the "
                    "counters do not split priced at 2.2 cycles), frac_hi = upper bound (at 4.07)"}


def weak_rooflines(pkg, prof, W, H, N, weak_fraction, workload, steps, warmup, opts, seed):
    """K9/K10 (weak update): `weak_path` (nominal bytes, fabric) and the roofline against the L1 tag pipeline."""
    k9, k10 = prof.get(pkg.K9, (0.0, 0)), prof.get(pkg.K10, (0.0, 0))
    wl = k9[1] + k10[1]
    wms = (k9[0] + k10[0]) / max(wl, 1)
    wbytes = (W * H / 2.0) * weak_fraction * algorithmic_bytes_per_weak_pixel(N)
    weak_path = {"kernel": "k910_update_weak (Black/RedPixelUpdateWeak)", "weak_fraction": round(weak_fraction, 4),
                 "avg_launch_ms": round(wms, 3), "launches": wl, "algorithmic_bytes_per_launch_nominal_max": wbytes,
                 "achieved_GBps_nominal_max": round(wbytes / (wms * 1e-3) / 1e9, 1) if wms > 0 else 0.0}
    # what bounds K9/K10: its scattered sub-patch gathers -- L1 tag look-ups per gather and the bytes the
    # misses pull through the fabric; counters from the profile of this same command line, time measured live
    wp = load_pmc_profile(workload, steps, warmup, "k910", opts, seed)
    # K9/K10 owns most of an APD iteration: it is the dominant kernel of this workload and the line's `roofline`; the
    # strong sweep's block moves to `strong_path`.  Bound: the L1 (TCP) tag pipeline -- a scattered dword gather costs one
    # tag access per lane whatever the lines (tools/tcp_patterns.hip, profiles/r03/tcp_patterns.txt), and the pipeline
    # sustains 1.85 accesses per clock and CU (tools/unaligned_gather.hip, profiles/r02/unaligned_gather.txt).
    tag_peak = 256 * MAX_CLOCK_GHZ * TCP_ACCESSES_PER_CLOCK   # G accesses / s
    weak_roofline = {"bound": "l1-tag-pipeline", "kernel": "k910_update_weak (Black/RedPixelUpdateWeak)", "achieved": None,
                     "peak": round(tag_peak, 1), "unit": "Gaccess/s", "frac": None, "traffic": None, "avg_launch_ms": round(wms, 3),
                     "launches": wl,
                     "peak_note": "256 CUs x %.1f GHz x %.2f L1 tag accesses per clock (all-hit dword gathers, tools/unaligned_gather.hip)"
                                  % (MAX_CLOCK_GHZ, TCP_ACCESSES_PER_CLOCK),
                     "algorithmic": {"bytes_per_launch_nominal_max": wbytes, "bytes_per_weak_pixel_iter_nominal_max": algorithmic_bytes_per_weak_pixel(N),
                                     "GBps_nominal_max": round(wbytes / (wms * 1e-3) / 1e9, 1) if wms > 0 else 0.0,
                                     "note": "SURVEY 8(d) nominal maximum (15 N NCCNew of 108 samples + N NCCOld, 20 B per sample); counts cache "
                                             "hits and early-outed hypotheses: NOT a roofline"},
                     "pmc_source": None}
    if wp and wms > 0 and wp.get("tcp_tag_accesses_per_launch"):
        acc = wp["tcp_tag_accesses_per_launch"] / (wms * 1e-3) / 1e9
        hbm_b = wp["hbm_bytes_per_launch"]
        weak_roofline.update({"achieved": round(acc, 1), "frac": round(acc / tag_peak, 4), "traffic": hbm_b,
                              "tag_accesses_per_launch": wp["tcp_tag_accesses_per_launch"],
                              # the pipeline's rate depends on where the lanes of a gather go: 1.85 per clock when they share
                              # lines, 0.95 when every lane reads its own line (tools/tcp_mix.hip, profiles/r03/tcp_mix.txt:
                              # 32 accesses in 33.6 clocks; tcp_patterns 3, 7, 30 agree); `frac` is against the former
                              "tag_rate": {"achieved_per_clock_and_cu": round(acc / (256 * MAX_CLOCK_GHZ), 3),
                                           "peak_lanes_sharing_lines": TCP_ACCESSES_PER_CLOCK, "peak_lanes_on_distinct_lines": 0.95,
                                           "note": "more resident waves do not shorten the launch (profiles/r03/ab_k910_split.txt: 8 to 16 "
                                                   "workgroups per CU, same time); L1 misses served by the L2 hide behind the tag "
                                                   "accesses up to one miss per ~2.5 accesses (tcp_mix: 2.4 clocks per 128-byte line)"},
                              "hbm": {"achieved": round(hbm_b / (wms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                      "frac": round(hbm_b / (wms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "bytes_per_launch": hbm_b,
                                      "note": "2 x FETCH_SIZE + WRITE_SIZE per launch"},
                              "valu": {"achieved": round(wp["valu_insts_per_launch"] / (wms * 1e-3) / 1e9, 1), "peak": round(VALU_PEAK_GINST, 1),
                                       "unit": "Gwave-inst/s", "frac": round(wp["valu_insts_per_launch"] / (wms * 1e-3) / 1e9 / VALU_PEAK_GINST, 4),
                                       "busy_estimate": valu_busy_from_classes(wp["valu_insts_per_launch"], wp.get("valu_classes"), wms)},
                              "pmc_source": wp["source"], "pmc_launch_ms": wp.get("launch_ms"), "pmc_profile_steps": wp["profile_steps"],
                              "pmc_extrapolated_launches": wp["extrapolated_launches"]})
    else:
        weak_roofline["pmc_note"] = ("no committed rocprofv3 counter profile for workload=%s options=%s seed=%d under profiles/"
                                     % (workload, list(opts), seed))
    if wp and wms > 0 and wp.get("fetch_bytes_per_launch") and wp.get("vmem_rd_insts_per_launch"):
        fabric = wp["fetch_bytes_per_launch"] / (wms * 1e-3) / 1e9
        gathers = wp["vmem_rd_insts_per_launch"]
        cyc = wms * 1e-3 * 2.4e9 * 256 / gathers
        weak_path["bound"] = {"kind": "l1-tag pipeline + fabric", "fabric_GBps": round(fabric, 1), "fabric_peak_GBps": 8000.0,
                              "fabric_frac": round(fabric / 8000.0, 4), "fabric_frac_of_achievable_6290": round(fabric / 6290.0, 4),
                              "wave_gathers_per_launch": gathers,
                              "tag_lookups_per_gather": round(wp["tcp_tag_accesses_per_launch"] / gathers, 1) if wp.get("tcp_tag_accesses_per_launch") else None,
                              "cu_cycles_per_gather": round(cyc, 1),
                              "cu_cycles_per_gather_all_hits": 32.0,
                              "note": "an all-hit wave-level dword gather occupies a CU's L1 for 32 cycles (tools/unaligned_gather.hip); "
                                      "FETCH_SIZE x 2 = bytes requested from the fabric (Infinity Cache + HBM)",
                              "pmc_source": wp["source"]}
    return weak_path, weak_roofline


# The sub-lines of the `workloads` block: every BASELINE.json config on the clock of the same process, after the headline.
# (key, workload, steps, warmup, pass_exchange, views_per_gpu)
CONFIGS0_KEY = "configs0_office_halfres_2src_3iter"
SUB_WORKLOADS = [
    (CONFIGS0_KEY, "eth3d_office_halfres_2src", 3, 1, False, 1),   # configs[0]: the shape BASELINE.json runs on the CPU path; the oracle
    # is timed on the SAME shape and iterations right after (run_cpu_configs0), so the config has its HIP and its CPU number side by side
    ("configs1_office_6iter", "eth3d_office_fullres_8src", 6, 1, False, 1),        # configs[1] at its own six iterations
    ("configs1_office_ref_pass_3iter", "eth3d_office_fullres_8src", 3, 1, False, 1),  # ... at the reference's default pass (main.cpp:183)
    ("configs2_pipes_apd_3iter", "eth3d_pipes_fullres_10src_apd", 3, 1, False, 1),  # configs[2]: adaptive patches on (K9/K10 roofline)
    ("configs1_office_hard_6iter", "eth3d_office_fullres_8src_hard", 6, 1, False, 1),  # configs[1] on the hard scene (occlusions, gain, lost overlap)
    ("configs2_pipes_hard_apd_3iter", "eth3d_pipes_fullres_10src_apd_hard", 3, 1, False, 1),  # configs[2] on the hard scene
    ("configs4_synthetic_16src_8iter", "synthetic_4096x3072_16src", 8, 1, False, 1),  # configs[4] shape, one replica per GPU
    ("configs3_tt1080p_20iter", "tt_family_1080p_10src", 20, 5, False, 1),          # configs[3] frame size, sweep only
    ("configs3_tt1080p_6views_20iter", "tt_family_1080p_10src", 20, 5, False, 6),    # ... with six views in flight per GPU, each on its own stream:
    # how the schedulers fill a device with small frames (host/multi_device.cpp: DefaultLanes)
    ("configs3_tt1080p_pass_with_exchange", "tt_family_1080p_10src", 3, 1, True, 2),  # configs[3] as the sharded scheduler runs it:
    # two views per GPU, the reference's three iterations, then K11..K13 + export + all-gather of all depth maps inside the timed region
]


# Whole passes: (key, workload, passes, warm-up passes).  Measured right after the sweep sub-lines of the same workload, on the same
# resident handle; K14 / K15 -- which the sweep metric never times and which are two fifths of an end-to-end run -- get the driver's
# clock and a VALU-issue roofline from their own counter profile (tools/profile_bench.py, APD_PROFILE_PASS_KEY).
PASS_ITERATIONS = 3   # PatchMatchParams::max_iterations of the reference (main.h:85)
PASS_WORKLOADS = [   # (key, workload, timed passes, warm-up passes, kind)
    ("configs2_pipes_apd_whole_pass", "eth3d_pipes_fullres_10src_apd", 2, 1, "photometric"),
    ("configs2_pipes_apd_geometric_pass", "eth3d_pipes_fullres_10src_apd", 2, 1, "geometric"),
    ("configs2_pipes_hard_whole_pass", "eth3d_pipes_fullres_10src_apd_hard", 2, 1, "photometric"),
]


def pass_kernel_roofline(pkg, prof, kid, kernel_key, workload, kind, opts, seed):
    """K14 / K15 inside a whole pass: live launch time (HIP events on the handle's stream) and, from the committed counter profile
    of the same whole-pass command (profiles/rNN/pmc_pass_<workload>_p<passes>.json), VALU instructions and memory-side bytes per
    launch.  Both kernels are vector-ALU issue bound: the fraction is taken against that peak."""
    ms, n = prof.get(kid, (0.0, 0))
    avg_ms = ms / max(n, 1)
    out = {"bound": "valu-issue", "kernel": pkg.KERNEL_NAMES[kid], "avg_launch_ms": round(avg_ms, 3), "launches": n, "achieved": None,
           "peak": round(VALU_PEAK_GINST, 1), "unit": "Gwave-inst/s", "frac": None, "traffic": None, "hbm": None, "pmc_source": None}
    pmc = load_pass_profile(workload, kernel_key, opts, seed, kind)
    if pmc is None or avg_ms <= 0:
        out["pmc_note"] = "no committed whole-pass counter profile (pmc_pass_*.json) for workload=%s kind=%s options=%s seed=%d" % (
            workload, kind, list(opts), seed)
        return out
    achieved = pmc["valu_insts_per_launch"] / (avg_ms * 1e-3) / 1e9
    hbm_gbps = pmc["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
    out.update({"achieved": round(achieved, 1), "frac": round(achieved / VALU_PEAK_GINST, 4), "traffic": pmc["hbm_bytes_per_launch"],
                "valu_insts_per_launch": pmc["valu_insts_per_launch"],
                "hbm": {"achieved": round(hbm_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(hbm_gbps / HBM_PEAK_GBPS, 4),
                        "bytes_per_launch": pmc["hbm_bytes_per_launch"], "note": "2 x FETCH_SIZE + WRITE_SIZE per launch"},
                "pmc_source": pmc["source"], "pmc_launch_ms": pmc["launch_ms"], "pmc_launches": pmc["launches"]})
    busy = valu_busy_from_classes(pmc["valu_insts_per_launch"], pmc.get("valu_classes"), avg_ms)
    if busy is not None:
        out["valu_busy_estimate"] = busy
    return out


def load_pass_profile(workload, kernel_key, options=(), seed=12345, kind="photometric"):
    """Newest profiles/rNN/pmc_pass_<workload>_p*.json (tools/profile_bench.py with APD_PROFILE_PASS_KEY): per-launch means of K14 /
    K15 over the timed passes.  Every timed pass starts from the same prior and seed, so its launches are the same work."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_pass_*.json"))):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        cfg = rec.get("config", {})
        k = rec.get("kernels", {}).get(kernel_key)
        if cfg.get("workload") != workload or list(cfg.get("options", [])) != list(options) or cfg.get("seed", 12345) != seed or not k:
            continue
        if cfg.get("pass_kind", "photometric") != kind:
            continue
        if k.get("valu_insts_per_launch") is None or k.get("hbm_bytes_per_launch") is None:
            continue
        pd = k.get("per_dispatch_timed") or {}
        classes = {c: (sum(pd[n]) / len(pd[n]) if pd.get(n) else None) for c, n in VALU_CLASS_COUNTERS.items()}
        best = {"valu_insts_per_launch": k["valu_insts_per_launch"], "hbm_bytes_per_launch": k["hbm_bytes_per_launch"],
                "valu_classes": None if any(v is None for v in classes.values()) else classes,
                "launch_ms": k.get("launch_ms"), "launches": k.get("launches_timed"), "source": os.path.relpath(path, ROOT)}
    return best


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when its first communicator is built ("RCCL version : ...", "Librccl path : ...").
    This program's stdout is ONE JSON line: file descriptor 1 points at stderr while the process group is set up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def init_distributed(args, torch, local_rank):
    """RCCL process group with a bounded set-up: a communicator that cannot be built must end the run with a diagnosis, not hang
    the node (the driver's 8-GPU run would otherwise sit until its own limit)."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local_rank)
    try:
        with stdout_to_stderr():
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=int(os.environ.get("APD_BENCH_PG_TIMEOUT_S", "300"))))
            assert dist.get_world_size() == args.gpus and dist.get_backend() == "nccl", (dist.get_world_size(), dist.get_backend())
            probe = torch.ones(1, device=torch.device("cuda", local_rank))
            dist.all_reduce(probe)   # builds the communicator now: a failure shows up here, before any timed region
            objs = [None] * args.gpus
            dist.all_gather_object(objs, int(os.environ.get("RANK", "0")))   # ... and the object path sharding.allgather_maps uses
            torch.cuda.synchronize()
        assert int(probe.item()) == args.gpus and objs == list(range(args.gpus))
    except Exception as e:  # noqa: BLE001 -- whatever RCCL raises
        sys.stderr.write("bench.py: RCCL process group set-up failed on rank %s: %r\n"
                         "bench.py: re-run with NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,ENV for RCCL's own log (NCCL_DEBUG=%s in this run)\n"
                         % (os.environ.get("RANK", "?"), e, os.environ.get("NCCL_DEBUG", "unset")))
        sys.stderr.flush()
        os._exit(4)
    return dist


def main():
    t_main = time.perf_counter()
    args = parse_args()
    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        return 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: launched with WORLD_SIZE=%d but --gpus %d\n" % (world, args.gpus))
        return 2
    if args.selftest_cpu:
        return selftest_cpu(args, world, rank)

    import numpy as np
    import torch

    import __graft_entry__ as ge
    pkg = ge.load_package()
    from apd_mvs_amd import synth

    distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ  # under torchrun even one rank takes the RCCL path
    if torch.cuda.device_count() < (local_rank + 1):
        sys.stderr.write("bench.py: rank %d needs HIP device %d, %d visible\n" % (rank, local_rank, torch.cuda.device_count()))
        return 3
    dist = None
    if distributed:
        dist = init_distributed(args, torch, local_rank)   # NCCL_DEBUG stays as the caller set it: RCCL logs to STDOUT, which is one JSON line here
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if distributed else 0)
    ctx = Ctx(pkg, synth, np, torch, dev, rank, world, distributed, dist)

    # ---- headline: BASELINE.json configs[1] (or --workload) at the command line's --steps / --warmup ----
    # the `workloads` block belongs to the default line (the driver's command); tuning runs (--workload X, --opt) time one thing
    subs = [] if (args.no_workloads or args.opt or (args.workload != DEFAULT_WORKLOAD and not args.only_workloads)) else \
        [s for s in SUB_WORKLOADS if not args.only_workloads or s[0] in args.only_workloads]
    same = [s for s in subs if s[1] == args.workload and s[5] == 1]
    wl = SweepWorkload(ctx, args.workload, max([max(args.warmup, args.steps)] + [max(s[2], s[3]) for s in same]), args.opt, args.seed)
    head = wl.measure(args.steps, args.warmup)
    head_apd = wl.apd_mode
    N_head = wl.N

    # ---- every other BASELINE config, in the same process, on the same clock ----
    workloads = {}
    t_subs = time.perf_counter()
    cache = {(args.workload, 1): wl}
    def guarded(key, fn):
        """A sub-line that fails must not take the headline with it: in a single process its entry becomes {"error": ...} and the
        block goes on.  Under a process group every rank must reach the same collectives, so there a failure stays fatal."""
        try:
            line = fn()
        except Exception as e:  # noqa: BLE001 -- reported in the line
            if distributed:
                raise
            gc.enable()
            sys.stderr.write("bench.py: sub-line %s failed: %r\n" % (key, e))
            workloads[key] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
            return False
        line["n_gpus"] = world
        line["scaling"] = "weak"
        workloads[key] = line
        return True

    for key, name, steps, warmup, pass_exchange, vpg in subs:
        w = cache.get((name, vpg))
        if w is None:
            for old in cache.values():   # one workload resident at a time: the next one gets the whole device
                old.close()
            cache.clear()
            try:
                w = SweepWorkload(ctx, name, max([max(s[2], s[3]) for s in subs if s[1] == name and s[5] == vpg]), (), args.seed, views_per_gpu=vpg)
            except Exception as e:  # noqa: BLE001
                if distributed:
                    raise
                sys.stderr.write("bench.py: workload %s could not be set up: %r\n" % (name, e))
                workloads[key] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
                continue
            cache[(name, vpg)] = w
        if not guarded(key, lambda: w.measure(steps, warmup, pass_exchange=pass_exchange)):
            cache.pop((name, vpg), None)   # its state is unknown: the next sub-line of this workload builds a fresh one
            try:
                w.close()
            except Exception:  # noqa: BLE001
                pass
            continue
        for pkey, pname, passes, pwarm, pkind in PASS_WORKLOADS:
            if pname == name and vpg == 1 and pkey not in workloads and w.max_iters == PASS_ITERATIONS and \
                    (not args.only_workloads or pkey in args.only_workloads):
                guarded(pkey, lambda: w.measure_whole_pass(passes, pwarm, pkind))
    for old in cache.values():
        old.close()
    cache.clear()
    subs_s = time.perf_counter() - t_subs

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # reported at N=1 only
        cpu_baseline = run_cpu_baseline(args, N_head, np)
        if workloads.get(CONFIGS0_KEY, {}).get("value") is not None:   # configs[0] names the CPU path: its own shape on the CPU, same line
            c0 = run_cpu_configs0(args)
            workloads[CONFIGS0_KEY]["cpu_baseline"] = c0
            workloads[CONFIGS0_KEY]["hip_over_cpu"] = round(workloads[CONFIGS0_KEY]["value"] / c0["value"], 1) if c0["value"] > 0 else None

    if rank == 0:
        out = {
            "metric": "Mpix*iterations/sec (PatchMatch sweep)",
            "value": head["value"],
            "unit": "Mpix*iter/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": head["config"],
            "roofline": head["roofline"],
            "strong_path": head["strong_path"] if head_apd else None,
            "cpu_baseline": cpu_baseline,
            "weak_path": head["weak_path"],
            "iterations": head["iterations"],
            "rank_ms_per_step": head["rank_ms_per_step"],
            "kernel_ms_timed_region": head["kernel_ms_timed_region"],
            "post_loop_ms": head["post_loop_ms"],
            "allgather_ms": head["allgather_ms"],
            "pass_allgather_ms": head["pass_allgather_ms"],
            "allgather_note": "RCCL all-gather over the ranks after the timed region: depth maps (the exchange that ends every pass, "
                              "pass_allgather_ms) + normal maps (before fusion; allgather_ms is both); the sub-line "
                              "configs3_tt1080p_pass_with_exchange has the per-pass exchange INSIDE its timed region",
            "quality_within_1pct_depth": head["quality_within_1pct_depth"],
            "workloads": workloads,
            "workloads_note": "every BASELINE.json config timed in this process after the headline, same contract (barrier + synchronize on "
                              "both sides, MAX over ranks, whole-job value): ms_per_step x steps of all lines lies inside this process's wall time",
            "wall_s": {"sub_workloads": round(subs_s, 1), "process": round(time.perf_counter() - t_main, 1)},
        }
        emit(out, full_line=args.full_line)
    if distributed:
        dist.destroy_process_group()
    return 0


COMPACT_LINE_MAX_BYTES = 1990   # the driver keeps a bounded tail of stdout (BENCH_r04.parsed was null with a 24 KB line)
FULL_BLOCK_FILE = "bench_workloads.json"


def compact_roofline(r):
    """The roofline object of the stdout line: numbers only, notes stay in the full block.  `frac` is the fraction of the bound the
    kernel actually runs into (`frac_kind`); SURVEY 8(d)'s algorithmic-byte ratio is carried beside it as `algorithmic_over_hbm_peak`
    (it prices LDS / L1 / L2 hits as HBM bytes and exceeds 1: not a roofline)."""
    if not r:
        return None
    alg = r.get("algorithmic") or {}
    hbm = r.get("hbm") or {}
    gbps = alg.get("GBps", alg.get("GBps_nominal_max"))
    return {"bound": r.get("bound"), "frac_kind": r.get("bound"), "kernel": str(r.get("kernel", "")).split(" ")[0], "achieved": r.get("achieved"),
            "peak": r.get("peak"), "unit": r.get("unit"), "frac": r.get("frac"), "traffic": None if r.get("traffic") is None else int(r["traffic"]),
            "avg_launch_ms": r.get("avg_launch_ms"), "launches": r.get("launches"), "hbm_frac": hbm.get("frac"),
            "valu_busy": (r.get("valu_busy_estimate") or {}).get("frac"),   # whole-kernel class counters x measured issue costs: lower bound ...
            "valu_busy_hi": (r.get("valu_busy_estimate") or {}).get("frac_hi"),   # ... and upper bound
            "algorithmic_GBps": gbps, "algorithmic_over_hbm_peak": None if gbps is None else round(gbps / HBM_PEAK_GBPS, 2),
            "pmc_source": r.get("pmc_source")}


def compact_workloads(workloads):
    """{key: [value, ms_per_step or ms_per_pass, roofline fraction of the line's dominant kernel]}; a failed sub-line is [null, null, null]."""
    out = {}
    for key, w in workloads.items():
        if w.get("value") is None:
            out[key] = [None, None, None]
            continue
        ms = w.get("ms_per_pass", w.get("ms_per_step"))
        roof = w.get("roofline")
        if roof is None and w.get("pass_kernels"):   # whole passes: K14, the largest kernel of the pass
            roof = w["pass_kernels"].get("K14")
        if roof is None and w.get("weak_path"):
            roof = (w["weak_path"] or {}).get("roofline")
        out[key] = [w["value"], ms, None if not roof else roof.get("frac")]
    return out


CONFIG_ITERS_KEY = "configs1_office_6iter"          # configs[1] at the six iterations the config names
WHOLE_PASS_KEY = "configs2_pipes_apd_whole_pass"    # apd_run = RunPatchMatch, K1..K15: what a real schedule runs at full size


def compact_line(out):
    """The ONE stdout line: headline + roofline + cpu_baseline + one triple per sub-workload, at most COMPACT_LINE_MAX_BYTES bytes.
    Everything else of `out` is in FULL_BLOCK_FILE and on stderr.  A line that would not fit is shortened (cpu sample text, then the
    per-workload triples, then pmc_source) and says so in `truncated` -- it never costs a measured run its line."""
    cfg = out["config"]
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    line["config"] = {k: cfg.get(k) for k in ("workload", "width", "height", "num_src", "state")}
    if "backend" in cfg and out.get("selftest"):
        line["config"]["backend"] = cfg["backend"]
    line["roofline"] = compact_roofline(out.get("roofline"))
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = None if not cb else {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                                "sample": cb.get("sample_short", cb["sample"])[:96]}
    wl = out.get("workloads") or {}
    # what the headline is NOT: the config at its own iteration count, and a whole pass of the real schedule
    it6 = wl.get(CONFIG_ITERS_KEY) or {}
    line["value_config_iters"] = out["value"] if out.get("steps") == 6 and cfg.get("workload") == DEFAULT_WORKLOAD else it6.get("value")
    wp = wl.get(WHOLE_PASS_KEY) or {}
    line["whole_pass"] = None if wp.get("value") is None else [wp["value"], wp.get("ms_per_pass")]
    c0 = (wl.get(CONFIGS0_KEY) or {}).get("cpu_baseline")
    line["configs0_cpu"] = None if not c0 else [c0["value"], c0["cores"]]   # configs[0] on the CPU (oracle, same shape); its HIP value is in `workloads`
    line["workloads"] = compact_workloads(wl)
    line["workloads_fields"] = ["value", "ms_per_step|ms_per_pass", "frac"]
    line["full_block"] = FULL_BLOCK_FILE
    if out.get("selftest"):
        line["selftest"] = True

    def size():
        return len(json.dumps(line, separators=(",", ":")).encode())

    steps_taken = []
    if size() > COMPACT_LINE_MAX_BYTES and line["cpu_baseline"]:
        line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:24]
        steps_taken.append("cpu_baseline.sample")
    if size() > COMPACT_LINE_MAX_BYTES:
        line["workloads"] = {k: v[0] for k, v in line["workloads"].items()}
        line["workloads_fields"] = ["value"]
        steps_taken.append("workloads: values only")
    if size() > COMPACT_LINE_MAX_BYTES and line["roofline"]:
        line["roofline"]["pmc_source"] = None
        steps_taken.append("roofline.pmc_source")
    if size() > COMPACT_LINE_MAX_BYTES:
        line["workloads"] = None
        steps_taken.append("workloads")
    if steps_taken:
        line["truncated"] = steps_taken
    return json.dumps(line, separators=(",", ":"))


def emit(out, full_line=False):
    """Full block -> FULL_BLOCK_FILE (next to bench.py, and under gpurun_out/ when that exists) and stderr FIRST; then the compact line
    -> the LAST and only line of stdout."""
    full = json.dumps(out)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, FULL_BLOCK_FILE), "w") as f:
                    f.write(full + "\n")
            except OSError as e:
                sys.stderr.write("bench.py: could not write %s: %r\n" % (os.path.join(d, FULL_BLOCK_FILE), e))
    sys.stderr.write("bench.py full block: " + full + "\n")
    sys.stderr.flush()
    text = compact_line(out)
    print(full if full_line else text, flush=True)


def apply_options(h, opts):
    for o in opts:
        name, _, value = o.partition("=")
        h.set_option(name, int(value))


def selftest_cpu(args, world, rank):
    """Launcher / collective plumbing without GPUs (gloo): barrier-bracketed timed region, MAX over ranks, per-rank times,
    the padded all-gather of per-view maps.  No PatchMatch work: `value` is null and the line says so."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    assert dist.get_world_size() == args.gpus == world
    import __graft_entry__ as ge
    ge.load_package()
    from apd_mvs_amd import sharding
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1) * args.steps)  # stand-in for this rank's sweep: rank r is (r + 1) x slower
    t_rank = time.perf_counter() - t0
    dist.barrier()
    elapsed = sharding.timed_region_max(time.perf_counter() - t0, torch.device("cpu"))
    per_rank = torch.zeros(world, dtype=torch.float64)
    dist.all_gather_into_tensor(per_rank, torch.tensor([t_rank / args.steps * 1e3], dtype=torch.float64))
    depth = torch.full((8, 12, 1), float(rank), dtype=torch.float32)
    ta = time.perf_counter()
    g = sharding.allgather_maps({rank: depth}, world)
    allgather_ms = (time.perf_counter() - ta) * 1e3
    assert g.shape[0] == world and all(float(g[r].mean()) == float(r) for r in range(world))
    # the pass-with-exchange sub-line of the `workloads` block: two views per rank, the all-gather of every view's depth map INSIDE
    # the barrier-bracketed region, gathered maps in view order on every rank
    vpg, hh, ww = 2, 6, 10
    mine = {rank + s * world: torch.full((hh, ww, 1), float(rank + s * world), dtype=torch.float32) for s in range(vpg)}
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.005 * (rank + 1))
    t_rank2 = time.perf_counter() - t0
    ta = time.perf_counter()
    g2 = sharding.allgather_maps(mine, vpg * world)
    pass_ms = (time.perf_counter() - ta) * 1e3
    dist.barrier()
    elapsed2 = sharding.timed_region_max(time.perf_counter() - t0, torch.device("cpu"))
    assert g2.shape[0] == vpg * world and all(float(g2[v].mean()) == float(v) for v in range(vpg * world))
    per_rank2 = torch.zeros(world, dtype=torch.float64)
    dist.all_gather_into_tensor(per_rank2, torch.tensor([t_rank2 * 1e3], dtype=torch.float64))
    assert elapsed2 * 1e3 >= float(per_rank2.max()) + 0.0   # the exchange is inside: the region cannot be shorter than the slowest rank
    if rank == 0:
        # through the same emit() as a measured line: compact stdout line + full block in bench_workloads.json
        emit({"metric": "Mpix*iterations/sec (PatchMatch sweep)", "value": None, "unit": "Mpix*iter/s", "n_gpus": world,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
              "higher_is_better": True, "vs_baseline": None, "dtype": "f32", "data": "synthetic", "roofline": None, "cpu_baseline": None,
              "selftest": "launcher and collectives only, CPU/gloo, no PatchMatch work", "scaling": "weak",
              "config": {"workload": args.workload, "backend": dist.get_backend()},
              "rank_ms_per_step": [round(float(v), 3) for v in per_rank.tolist()],
              "allgather_ms": round(allgather_ms, 3),
              "workloads": {"configs3_tt1080p_pass_with_exchange": {
                  "value": None, "n_gpus": world, "steps": 1, "ms_per_step": round(elapsed2 * 1e3, 3),
                  "timed_region_ms": round(elapsed2 * 1e3, 3), "pass_allgather_ms": round(pass_ms, 3),
                  "pass_allgather_inside_timed_region": True, "rank_ms_per_step": [round(float(v), 3) for v in per_rank2.tolist()],
                  "config": {"views_per_gpu": vpg, "backend": dist.get_backend(), "views": vpg * world}}}}, full_line=args.full_line)
    dist.destroy_process_group()
    return 0


VALU_CLASS_COUNTERS = {"fma": "SQ_INSTS_VALU_FMA_F32", "add": "SQ_INSTS_VALU_ADD_F32", "mul": "SQ_INSTS_VALU_MUL_F32", "trans": "SQ_INSTS_VALU_TRANS_F32",
                       "cvt": "SQ_INSTS_VALU_CVT", "int32": "SQ_INSTS_VALU_INT32", "int64": "SQ_INSTS_VALU_INT64"}
PMC_COUNTERS = {  # field of the returned record -> counter of tools/profile_bench.py
    "valu_insts_per_launch": "SQ_INSTS_VALU", "vmem_rd_insts_per_launch": "SQ_INSTS_VMEM_RD",
    "valu_fma": "SQ_INSTS_VALU_FMA_F32", "valu_add": "SQ_INSTS_VALU_ADD_F32", "valu_mul": "SQ_INSTS_VALU_MUL_F32", "valu_trans": "SQ_INSTS_VALU_TRANS_F32",
    "valu_cvt": "SQ_INSTS_VALU_CVT", "valu_int32": "SQ_INSTS_VALU_INT32", "valu_int64": "SQ_INSTS_VALU_INT64",
    "tcp_tag_accesses_per_launch": "TCP_TOTAL_CACHE_ACCESSES_sum", "launch_ns": "duration_ns@trace",
    "fetch_kib": "FETCH_SIZE", "write_kib": "WRITE_SIZE"}


def pmc_timed_series(kernel_rec, profiled_steps, launches):
    """Per-dispatch counter values of the first `launches` timed launches, from a profile that timed 2 * profiled_steps of
    them.  Timed launch j of any command line of a workload is launch j of the profiled one (the pass is re-initialised
    after the warm-up, no kernel reads max_iterations), so a shorter run reads a prefix.  A longer run repeats the last
    profiled iteration (black, red) for the launches beyond the profile -- converged launches, whose counters are flat -- and
    the number of such launches is returned so that the line can say so."""
    pd = kernel_rec.get("per_dispatch_timed") or {}
    have = 2 * profiled_steps
    out, extrapolated = {}, max(0, launches - have)
    for field, cname in PMC_COUNTERS.items():
        v = pd.get(cname)
        if not v or len(v) < have:
            out[field] = None
            continue
        t = list(v[-have:])
        while len(t) < launches:
            t += t[have - 2:have]
        out[field] = t[:launches]
    return out, extrapolated


def load_pmc_profile(workload, steps, warmup, kernel, options=(), seed=12345):
    """Counter profile of THIS workload from the newest profiles/rNN/pmc_bench_*.json written by tools/profile_bench.py:
    separate rocprofv3 --pmc passes (SQ_*; TCP_* / TCC_*; FETCH_SIZE; WRITE_SIZE) with per-dispatch values, reduced here over
    the launches this command line times (pmc_timed_series).  PMC counters cannot be read inside this process.  Never
    substituted: a profile of another workload, of other --opt options or of another seed (returns None)."""
    import glob
    best, best_rank = None, None
    launches = 2 * steps
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_bench_*.json"))):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        cfg = rec.get("config", {})
        if cfg.get("workload") != workload or list(cfg.get("options", [])) != list(options) or cfg.get("seed", 12345) != seed:
            continue
        k = rec.get("kernels", {}).get(kernel)
        if not k or not cfg.get("steps"):
            continue
        series, extrapolated = pmc_timed_series(k, cfg["steps"], launches)
        if series["valu_insts_per_launch"] is None or series["fetch_kib"] is None or series["write_kib"] is None:
            continue
        mean = lambda f_: None if series[f_] is None else sum(series[f_]) / len(series[f_])
        fe, wr = mean("fetch_kib"), mean("write_kib")
        cand = {"valu_insts_per_launch": mean("valu_insts_per_launch"), "hbm_bytes_per_launch": fe * 1024 * 2 + wr * 1024,
                "launch_ms": None if mean("launch_ns") is None else mean("launch_ns") / 1e6, "source": os.path.relpath(path, ROOT),
                "fetch_bytes_per_launch": fe * 1024 * 2, "vmem_rd_insts_per_launch": mean("vmem_rd_insts_per_launch"),
                "tcp_tag_accesses_per_launch": mean("tcp_tag_accesses_per_launch"),
                "valu_classes": None if any(series["valu_" + c] is None for c in VALU_CLASS_COUNTERS) else {c: mean("valu_" + c) for c in VALU_CLASS_COUNTERS},
                "profile_steps": cfg["steps"], "profile_warmup": cfg.get("warmup"), "extrapolated_launches": extrapolated}
        # the newest round's profiles describe today's kernels; within a round: covered without extrapolation, then the exact command line
        rank = (os.path.basename(os.path.dirname(path)), extrapolated == 0, cfg["steps"] == steps and cfg.get("warmup") == warmup)
        if best is None or rank >= best_rank:
            best, best_rank = cand, rank
    return best


def load_valu_mix(prefer_dir=None):
    """Static instruction mix of the K6/K7 window body (tools/valu_mix.py).  Only the file of the SAME round directory as the
    counter profile describes the kernel that profile measured: with `prefer_dir` nothing else is used (a
    round-2 mix priced a round-3 kernel)."""
    import glob
    paths = [os.path.join(ROOT, prefer_dir, "valu_mix_k67w.json")] if prefer_dir else sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "valu_mix_k67w.json")))
    best = None
    for path in paths:
        try:
            with open(path) as f:
                rec = json.load(f)
            best = {"mean_cycles_per_inst": rec["window_body"]["mean_cycles_per_inst"], "source": os.path.relpath(path, ROOT)}
        except (OSError, ValueError, KeyError):
            continue
    return best


def time_oracle_sweeps(width, height, num_src, steps, seed, budget_s):
    """Iterations 0..steps-1 of a fresh FIRST_INIT pass on the CPU oracle (K1, K2, K5 untimed, like the GPU lines), stopped early
    once `budget_s` is spent.  Returns (iterations done, seconds, threads)."""
    import __graft_entry__ as ge
    ge.load_package()
    from apd_mvs_amd import synth
    from oracle import binding as ob

    def make(w, h):
        sc = synth.make_scene(w, h, num_src, seed=0)
        imgs = sc.images_numpy()
        cams = [ob.make_camera(sc.K[i], sc.R[i], sc.t[i], w, h, sc.depth_min, sc.depth_max) for i in range(num_src + 1)]
        p = ob.default_params(num_images=num_src + 1, depth_min=0.6 * sc.depth_min, depth_max=1.2 * sc.depth_max, use_APD=0,
                              state=ob.FIRST_INIT, max_iterations=steps, seed=seed)
        o = ob.Oracle(w, h, p, cams, imgs)
        for kid in (1, 2, 5):
            o.run_kernel(kid)
        return o

    tiny = make(160, 120)  # thread pool and code warm-up on a throw-away frame
    tiny.run_sweeps(0, 1)
    tiny.close()
    o = make(width, height)
    iters = 0
    t0 = time.perf_counter()
    while True:
        o.run_sweeps(iters, 1)
        iters += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or iters >= steps:
            break
    cores = int(ob.lib().orc_get_threads())
    o.close()
    return iters, dt, cores


def run_cpu_baseline(args, num_src, np):
    """The oracle (kind "port": plain-C restatement of the reference path, OpenMP over pixels) timed on
    this host's cores on a bounded sample of the same workload: same generator, same N, smaller frame, and the same
    iterations the GPU line times (iteration 0 on random planes included)."""
    w, hgt = [int(v) for v in args.cpu_sample.lower().split("x")]
    iters, dt, cores = time_oracle_sweeps(w, hgt, num_src, min(args.steps, 8), args.seed, 12.0)
    return {"value": round(w * hgt * iters / dt / 1e6, 5), "unit": "Mpix*iter/s", "cores": cores, "kind": "port",
            "sample": "%dx%d frame of the same synthetic scene, %d src views, iterations 0..%d of a fresh pass (iteration 0 included, "
                      "as in the GPU line), %.1f s" % (w, hgt, num_src, iters - 1, dt),
            "sample_short": "%dx%d, %d src, iterations 0..%d, %.1f s" % (w, hgt, num_src, iters - 1, dt)}


def run_cpu_configs0(args):
    """BASELINE.json configs[0] on the CPU, beside its HIP sub-line: the oracle on the SAME shape (3100 x 2065, 2 source views, no crop)
    and the same three iterations of a fresh FIRST_INIT pass; a host too slow to finish inside the budget stops after fewer iterations and
    the sample says so."""
    (w, hgt, n), _ = resolve_workload("eth3d_office_halfres_2src")
    iters, dt, cores = time_oracle_sweeps(w, hgt, n, 3, args.seed, 25.0)
    return {"value": round(w * hgt * iters / dt / 1e6, 5), "unit": "Mpix*iter/s", "cores": cores, "kind": "port",
            "sample": "the whole %dx%d frame, %d src views, iterations 0..%d of 0..2, %.1f s" % (w, hgt, n, iters - 1, dt)}


if __name__ == "__main__":
    sys.exit(main())
