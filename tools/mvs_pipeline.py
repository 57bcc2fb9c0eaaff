#!/usr/bin/env python3
"""Multi-GPU driver of the PatchMatch passes over a dense folder (the reference's layout: pair.txt, cams/, images/).

  one GPU :  python tools/mvs_pipeline.py <dense_folder> [--iters 3] [--seed 12345] [--single-level]
  N GPUs  :  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/mvs_pipeline.py <dense_folder>

Reference views are sharded round-robin over the ranks (one process per GPU, RCCL over xGMI); depth maps are all-gathered
after every pass, depth + normal + weak maps after the last one, and rank 0 writes <dense>/APD/<id>/{depths.dmb,
normals.dmb, weak.bin, selected_views.bin} for fusion.  With one rank the files equal the drop-in binary's bit for bit."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dense_folder")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--single-level", action="store_true")
    ap.add_argument("--max-src", type=int, default=0, help="keep only the first N source views of pair.txt")
    ap.add_argument("--no-fusion", action="store_true")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from apd_mvs_amd import pipeline
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    scene = pipeline.load_dense_folder(args.dense_folder, pkg.Camera)
    if args.max_src > 0:
        scene.pairs = [p[:args.max_src] for p in scene.pairs]
    if rank == 0:
        print("%d views, %dx%d, %d rank(s)" % (scene.num_views, scene.images[0].shape[1], scene.images[0].shape[0], world), flush=True)
    t0 = time.time()
    results = pipeline.run_pipeline(scene, pipeline.HipBackend(pkg, device=local_rank), iters=args.iters, seed=args.seed,
                                    single_level=args.single_level, log=(print if rank == 0 else None))
    if rank == 0:
        pipeline.save_results(args.dense_folder, scene, results)
        if not args.no_fusion:
            tf = time.time()
            colour = pipeline.load_colour_images(args.dense_folder, getattr(scene, "ids", list(range(scene.num_views)))[:scene.num_views])
            n = pipeline.fuse(scene, results, os.path.join(args.dense_folder, "APD", "APD.ply"), device=local_rank, colour_images=colour)
            print("fused %d points into APD/APD.ply in %.2f s" % (n, time.time() - tf), flush=True)
        print("PatchMatch passes done in %.1f s; maps written under %s" % (time.time() - t0, os.path.join(args.dense_folder, "APD")), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
