#!/usr/bin/env python3
"""bench.py -- Mpix*iterations/s of the PatchMatch sweep (BASELINE.json metric) on 1..8 MI355X.

One "step" = one iteration of the loop body APD.cu:2443-2457 (K6 black + K7 red strong update,
K8 fit-plane, K9/K10 weak update when WEAK pixels exist) over one reference view.  The K timed steps are iterations
0..K-1 of a freshly initialised pass (the warm-up iterations run first, then the state is reset with the same seed).  N=1 workload =
BASELINE.json configs[1]: ETH3D-office-shaped stand-in, 6200x4130, 8 source views (datasets are not
in the image: SURVEY.md 8d synthetic generator).  With --gpus N each rank sweeps its own reference
view (views shard, no collective in the data path: weak scaling) and the final depth/normal maps are
all-gathered over RCCL after the timed region, as they would be before fusion.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)

WORKLOADS = {
    # name: (width, height, num_src)
    "eth3d_office_fullres_8src": (6200, 4130, 8),    # BASELINE.json configs[1]
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


def algorithmic_bytes_per_weak_pixel(num_src):
    """SURVEY.md 8(d), nominal: 15*N NCCNew of 108 samples + N NCCOld of 36 samples, 20 B per sample, + 176 B of state."""
    return 33120 * num_src + 176


def algorithmic_bytes_per_strong_pixel(num_src):
    """14*N NCCs x 36 samples x (4 B ref texel + 16 B bilinear taps) + 176 B of state."""
    return 10080 * num_src + 176


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="eth3d_office_fullres_8src", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="1024x768", help="WxH of the CPU-baseline sample")
    ap.add_argument("--seed", type=int, default=12345)
    args = ap.parse_args()

    import numpy as np
    import torch

    import __graft_entry__ as ge
    pkg = ge.load_package()
    from apd_mvs_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ  # under torchrun even one rank takes the RCCL path
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if distributed else 0)

    W, H, N = WORKLOADS[args.workload]
    apd_mode = args.workload in APD_WORKLOADS
    # every rank owns a different reference view of the same camera ring
    sc = synth.make_scene(W, H, N, seed=0, ref_view=rank, device=dev, textureless=0.2 if apd_mode else 0.0)
    cams = [pkg.make_camera(sc.K[i], sc.R[i], sc.t[i], W, H, sc.depth_min, sc.depth_max) for i in range(N + 1)]
    total_iters = max(args.warmup, args.steps)
    dmin, dmax = 0.6 * sc.depth_min, 1.2 * sc.depth_max
    weak_fraction = 0.0
    if not apd_mode:
        params = pkg.default_params(num_images=N + 1, depth_min=dmin, depth_max=dmax, use_APD=0,
                                    state=pkg.FIRST_INIT, max_iterations=total_iters, seed=args.seed)
        h = pkg.Handle(W, H, params, device=dev.index)
        h.upload_views(cams, sc.images)  # device->device copies: inputs are resident in HBM before timing
    else:
        # untimed: the photometric FIRST_INIT pass of main.cpp:169-190 (3 iterations, weak_peak_radius 6) that
        # classifies pixels, then ProcessProblem's post-processing (main.cpp:105-115)
        p0 = pkg.default_params(num_images=N + 1, depth_min=dmin, depth_max=dmax, use_APD=0, state=pkg.FIRST_INIT,
                                max_iterations=3, weak_peak_radius=6, seed=args.seed)
        h0 = pkg.Handle(W, H, p0, device=dev.index)
        h0.upload_views(cams, sc.images)
        h0.run()
        planes, weak, views = h0.download()
        h0.close()
        bad = (planes[..., 3] < np.float32(dmin)) | (planes[..., 3] > np.float32(dmax))
        planes[..., 3][bad] = 0
        weak[bad] = pkg.UNKNOWN
        params = pkg.default_params(num_images=N + 1, depth_min=dmin, depth_max=dmax, use_APD=1, state=pkg.REFINE_INIT,
                                    max_iterations=total_iters, weak_peak_radius=6, rotate_time=4,
                                    ransac_threshold=0.01 - 0.00125 * 3, seed=args.seed + 1)
        h = pkg.Handle(W, H, params, device=dev.index)
        h.upload_views(cams, sc.images)
        prior = (planes, views, weak)
    del sc.images[:]
    torch.cuda.empty_cache()

    def init_pass():
        """Everything before the loop of APD.cu:2443 (not timed): prior state, K1..K5."""
        if apd_mode:
            h.upload_prior(*prior)
        h.run_kernel(pkg.K1)
        h.run_kernel(pkg.K2)
        if apd_mode and h.weak_count > 0:
            h.run_kernel(pkg.K3)
            h.run_kernel(pkg.K4)
        h.run_kernel(pkg.K5)

    init_pass()
    if apd_mode:
        weak_fraction = h.weak_count / float(W * H)
    # Warm-up: W iterations of the same sweep (clocks, caches, code objects).  Then the state is re-initialised with the
    # same seed (K1 + K5 again, untimed) so that the timed region is exactly what the config names: the first K
    # iterations of a pass, INCLUDING iteration 0, whose random planes scatter the gathers over the whole source images
    # and which costs about twice a later iteration.
    if args.warmup > 0:
        h.run_sweeps(0, args.warmup)
        init_pass()
    h.profile_enable(True)
    h.profile_reset()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        h.synchronize()

    barrier()
    t0 = time.perf_counter()
    h.run_sweeps(0, 1, sync=False)
    h.synchronize()
    t_first = time.perf_counter()
    if args.steps > 1:
        h.run_sweeps(1, args.steps - 1, sync=False)
        h.synchronize()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    first_iter_s = t_first - t0
    if distributed:
        tt = torch.tensor([elapsed, first_iter_s], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, first_iter_s = float(tt[0].item()), float(tt[1].item())
    prof = h.profile()
    h.profile_enable(False)

    # after the timed region: post-loop kernels + all-gather of depth/normal maps (before fusion)
    allgather_ms = None
    t_post0 = time.perf_counter()
    for kid in (pkg.K11, pkg.K12, pkg.K13):
        h.run_kernel(kid)
    depth = torch.empty((H, W), device=dev, dtype=torch.float32)
    normal = torch.empty((H, W, 3), device=dev, dtype=torch.float32)
    h.export_depth_normal(depth, normal)
    torch.cuda.synchronize()
    if distributed:
        from apd_mvs_amd import sharding
        torch.cuda.synchronize()
        ta = time.perf_counter()
        gathered_d = sharding.allgather_maps({rank: depth.view(H, W, 1)}, world)   # view index == rank here
        gathered_n = sharding.allgather_maps({rank: normal}, world)
        torch.cuda.synchronize()
        allgather_ms = (time.perf_counter() - ta) * 1e3
        assert gathered_d.shape[0] == world and torch.equal(gathered_d[rank, :, :, 0], depth)
    gt = sc.gt_depth
    err = (depth - gt).abs() / gt
    within = float((err[8:-8, 8:-8] < 0.01).float().mean().item())
    post_ms = (time.perf_counter() - t_post0) * 1e3

    mpix = W * H / 1e6
    value = world * mpix * args.steps / elapsed

    # roofline of the dominant kernel (K6/K7 strong update), from HIP events on the handle's stream
    k6 = prof.get(pkg.K6, (0.0, 0))
    k7 = prof.get(pkg.K7, (0.0, 0))
    launches = k6[1] + k7[1]
    avg_ms = (k6[0] + k7[0]) / max(launches, 1)
    # K6/K7 skip WEAK pixels; the weak fraction is the one at upload time (K4 only ever lowers it)
    bytes_per_launch = (W * H / 2.0) * (1.0 - weak_fraction) * algorithmic_bytes_per_strong_pixel(N)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, traffic_src, valu = load_pmc_traffic(args.workload)
    roofline = {
        "bound": "hbm", "kernel": "k67w_update_strong (Black/RedPixelUpdateStrong, LDS source windows)", "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
        "traffic_source": traffic_src,
        "avg_launch_ms": round(avg_ms, 3), "launches": launches,
        "algorithmic_bytes_per_launch": bytes_per_launch,
        "bytes_per_pixel_iter": algorithmic_bytes_per_strong_pixel(N),
        # what actually limits the kernel: wave64 VALU instructions per launch and the fraction of the launch
        # the 16-lane VALU pipes are busy with them, from the SQ pass of the same committed profile
        "valu": valu,
    }
    kernel_ms = {pkg.KERNEL_NAMES[k]: round(v[0], 3) for k, v in sorted(prof.items())}
    weak_path = None
    if apd_mode:
        k9, k10 = prof.get(pkg.K9, (0.0, 0)), prof.get(pkg.K10, (0.0, 0))
        wl = k9[1] + k10[1]
        wms = (k9[0] + k10[0]) / max(wl, 1)
        wbytes = (W * H / 2.0) * weak_fraction * algorithmic_bytes_per_weak_pixel(N)
        weak_path = {"kernel": "k910_update_weak (Black/RedPixelUpdateWeak)", "weak_fraction": round(weak_fraction, 4),
                     "avg_launch_ms": round(wms, 3), "launches": wl, "algorithmic_bytes_per_launch_nominal_max": wbytes,
                     "achieved_GBps_nominal_max": round(wbytes / (wms * 1e-3) / 1e9, 1) if wms > 0 else 0.0}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # reported at N=1 only
        cpu_baseline = run_cpu_baseline(args, N, np)

    if rank == 0:
        out = {
            "metric": "Mpix*iterations/sec (PatchMatch sweep)",
            "value": round(value, 4),
            "unit": "Mpix*iter/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": args.workload, "width": W, "height": H, "num_src": N,
                       "state": "REFINE_INIT+APD" if apd_mode else "FIRST_INIT",
                       "views_per_gpu": 1, "parallelism": "views sharded, %d rank(s)" % world},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "weak_path": weak_path,
            "iterations": {"first_ms": round(first_iter_s * 1e3, 3),
                           "later_ms_per_step": round((elapsed - first_iter_s) / max(args.steps - 1, 1) * 1e3, 3) if args.steps > 1 else None,
                           "later_value": round(world * mpix * (args.steps - 1) / (elapsed - first_iter_s), 4) if args.steps > 1 else None,
                           "note": "timed region = iterations 0..K-1 of a freshly initialised pass; iteration 0 starts from random planes"},
            "kernel_ms_timed_region": kernel_ms,
            "post_loop_ms": round(post_ms, 1),
            "allgather_ms": None if allgather_ms is None else round(allgather_ms, 3),
            "quality_within_1pct_depth": round(within, 4),
        }
        print(json.dumps(out), flush=True)
    h.close()
    if distributed:
        dist.destroy_process_group()


def load_pmc_traffic(workload):
    """HBM bytes per K6/K7 launch from the newest committed rocprofv3 PMC summary of this workload
    (profiles/rNN/pmc_traffic.json, written by tools/profile.sh: separate --pmc passes, FETCH_SIZE doubled
    as MI355X_MICROARCH.md prescribes for gfx950).  PMC counters cannot be read inside this process, so the
    field is null when no such profile exists."""
    import glob
    best = (None, None, None)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json"))):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get("workload") == workload and rec.get("hbm_bytes_per_launch"):
            valu = None
            if rec.get("valu_insts_per_launch"):
                valu = {"insts_per_launch": rec["valu_insts_per_launch"], "pipe_busy_frac": round(rec.get("valu_pipe_busy_frac", 0.0), 3)}
            best = (rec["hbm_bytes_per_launch"], os.path.relpath(path, ROOT), valu)
    return best


def run_cpu_baseline(args, num_src, np):
    """The oracle (kind "port": plain-C restatement of the reference path, OpenMP over pixels) timed on
    this host's cores on a bounded sample of the same workload: same generator, same N, smaller frame."""
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from apd_mvs_amd import synth
    from oracle import binding as ob

    w, hgt = [int(v) for v in args.cpu_sample.lower().split("x")]
    sc = synth.make_scene(w, hgt, num_src, seed=0)
    imgs = sc.images_numpy()
    cams = [ob.make_camera(sc.K[i], sc.R[i], sc.t[i], w, hgt, sc.depth_min, sc.depth_max) for i in range(num_src + 1)]
    p = ob.default_params(num_images=num_src + 1, depth_min=0.6 * sc.depth_min, depth_max=1.2 * sc.depth_max, use_APD=0,
                          state=ob.FIRST_INIT, max_iterations=2, seed=args.seed)
    o = ob.Oracle(w, hgt, p, cams, imgs)
    for kid in (1, 2, 5):
        o.run_kernel(kid)
    o.run_sweeps(0, 1)  # warm caches / thread pool
    iters = 0
    t0 = time.perf_counter()
    while True:
        o.run_sweeps(1 + iters, 1)
        iters += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or iters >= 8:
            break
    cores = ob.lib().orc_get_threads()
    o.close()
    return {"value": round(w * hgt * iters / dt / 1e6, 5), "unit": "Mpix*iter/s", "cores": int(cores), "kind": "port",
            "sample": "%dx%d frame of the same synthetic scene, %d src views, %d iterations, %.1f s" % (w, hgt, num_src, iters, dt)}


if __name__ == "__main__":
    main()
