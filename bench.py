#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X path-tracing core.

Metric (BASELINE.json): Mrays/s at 1920x1080, 8 bounces, on the 1,000,000-triangle scene (config 3:
`configs[2]`, the configuration the metric is quoted on; it fits one GPU).  One "step" = one frame =
one sample per pixel through PathTracer::path_trace (streaming mode).  1 ray = 1 closest-hit query,
primary rays included.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: the frame's rows are dealt to the N ranks in blocks of 8 rows (round-robin: sky rows are cheap, terrain
rows expensive, so contiguous bands would be unbalanced), one process per GPU; no data-path collective during
tracing; the ranks' radiance is gathered to rank 0 over RCCL once, at present time, inside the timed region,
and scattered back into frame order there.  Fixed total work -> "scaling": "strong".

Prints ONE JSON line on rank 0 with the contract fields plus:
  roofline      dominant kernel = the closest-hit (trace) kernel; achieved = algorithmic bytes of all its
                launches in the timed region / their summed duration (HIP events on the kernel's stream);
                launches of several frames in flight overlap, so achieved_chip (bytes / wall time) is given too.
                Algorithmic bytes per launch = n*52 + box_tests*32 + tri_tests*48 (the
                reference's node / triangle sizes), with the test counts taken from an instrumented, untimed
                re-run of the same frames; achieved_own_layout prices a box test at the 16 bytes this
                implementation's 64-byte four-box nodes cost.  Most of these bytes are served by L2 / Infinity
                Cache (roofline.traffic), so achieved can exceed the HBM peak.
  cpu_baseline  the CPU oracle (oracle/, kind "port": the reference has no CPU path) timed on this
                box's cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

# Batches of frames run on separate HIP streams; ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4), and streams that share a queue serialise.  Must be set before the HIP runtime initialises
# (torch starts it here, before libptcore.so gets the chance to ask).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--max-bounces", type=int, default=8)
    ap.add_argument("--grid", type=str, default="1001x501", help="heightfield vertex grid (1001x501 = 1,000,000 triangles)")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="0 = streams x the batch size")
    ap.add_argument("--batch-frames", type=int, default=0,
                    help="frames traced per launch; 0 = up to 32, chosen so that the timed steps split evenly over the streams")
    ap.add_argument("--streams", type=int, default=0, help="batches in flight; 0 = 2 per rank on one or two GPUs, 4 per rank on more")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="all ranks share cuda:0 and talk over gloo (to rehearse the N>1 code path on a 1-GPU box)")
    ap.add_argument("--no-events", action="store_true", help="do not time the trace kernel with HIP events (diagnostic)")
    ap.add_argument("--cpu-sample", type=str, default="1920x1080", help="resolution of the CPU-oracle sample frame")
    ap.add_argument("--cpu-threads", type=int, default=16, help="oracle threads (the GPU box's CPU share for one GPU is 16)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    rehearse = args.rehearse_on_one_gpu
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    coll_dev = "cpu" if rehearse else "cuda"

    pkg = graft.load_package()
    W, H, MB = args.width, args.height, args.max_bounces
    nx, nz = (int(v) for v in args.grid.split("x"))
    scene = pkg.scenes.heightfield_scene((W, H), nx=nx, nz=nz)
    flat = scene.build_scene()
    mesh = list(scene.mesh_map_.values())[0]
    t0 = time.perf_counter()
    flat.bvh, bvh_depth = pkg.bvh_from_mesh(mesh)
    bvh_build_s = time.perf_counter() - t0

    # rows in blocks of BLOCK_ROWS dealt round-robin over the ranks (cuda-path-tracer_amd/bands.py)
    BLOCK_ROWS = 8
    rank_rows = pkg.bands.interleaved_rows(H, world, BLOCK_ROWS)

    pt = pkg.PathTracer(device=local_rank, max_bounces=MB)
    # Schedule: `batch` consecutive frames share every launch (the latency tail of a bounce -- a few long rays --
    # and the drain of the persistent wavefronts are paid once per batch) and `streams` batches are in flight so
    # that the small kernels and the tail of one overlap the bulk of another.  A rank that owns 1/N of the rows
    # has 1/N of the rays per launch and, from N = 4 on, keeps 4 smaller batches in flight (measured on 1/2, 1/4 and
    # 1/8 shares of the rows on one GPU).  Results do not depend on any of it
    # (tests/test_gpu_schedules.py).
    streams = args.streams or (2 if world <= 2 else 4)
    if args.batch_frames:
        batch = args.batch_frames
    else:
        rounds = max(1, -(-args.steps // (streams * 32)))
        batch = max(1, min(32, -(-args.steps // (streams * rounds))))
    args.batch_frames = batch
    args.frames_in_flight = args.frames_in_flight or streams * batch
    pt.set_param("frames_in_flight", args.frames_in_flight)
    pt.set_param("batch_frames", batch)
    # persistent traversal wavefronts per launch: what is resident at 5 per SIMD on one GPU (a second launch's
    # wavefronts move in as the first one's drain); half of that for the smaller launches of a rank among 4 or 8
    pt.set_param("traverse_waves", 5120 if world <= 2 else 2560)
    pt.create_buffers((W, H), flat)
    pt.set_stream(torch.cuda.current_stream().cuda_stream)
    if world > 1:
        pt.set_interleave(rank, world, BLOCK_ROWS)
    pt.max_iterations = 1 << 30
    # gather needs equal sizes on every rank: pad each rank's rows to the largest share
    max_rows = max(len(rr) for rr in rank_rows)
    band_color = torch.zeros((max_rows, W, 3), dtype=torch.float32, device="cuda")
    gathered = frame = row_index = None
    if world > 1 and rank == 0:
        gathered = [torch.empty((max_rows, W, 3), dtype=torch.float32, device="cuda") for r in range(world)]
        frame = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
        row_index = [torch.tensor(rank_rows[r], dtype=torch.long, device="cuda") for r in range(world)]

    def present():
        """gather of per-rank radiance at present time (the only inter-GPU traffic) + back into frame order"""
        pt.download_to_device("color", band_color.data_ptr())
        if world > 1:
            if rehearse:
                parts = [torch.empty(g.shape, dtype=g.dtype) for g in gathered] if rank == 0 else None
                dist.gather(band_color.cpu(), parts, dst=0)
                if rank == 0:
                    for r in range(world):
                        gathered[r].copy_(parts[r])
            else:
                dist.gather(band_color, gathered if rank == 0 else None, dst=0)
            if rank == 0:
                for r in range(world):
                    frame.index_copy_(0, row_index[r], gathered[r][: len(rank_rows[r])])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pt.path_trace(scene.camera)
    present()
    fence()
    pt.reset_profile()
    pt.set_profiling(time_trace_kernel=not args.no_events, count_tests=False)
    rays0 = pt.stats()["rays_total"]
    first_iter = pt.iteration()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pt.path_trace(scene.camera)
    present()
    fence()
    elapsed = time.perf_counter() - t0

    rays = pt.stats()["rays_total"] - rays0
    prof = pt.profile()
    last_live = pt.stats()["last_live"]

    # max over ranks of the elapsed time, sum of rays
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([rays], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays = int(r.item())

    # instrumented, untimed re-run of the same frames: BVH box / triangle test counts per bounce
    pt.set_profiling(time_trace_kernel=False, count_tests=True)
    paths_timed = list(prof["paths"])
    pt.reset_profile()
    pt.set_iteration(first_iter)
    count_steps = min(args.steps, 4)
    for _ in range(count_steps):
        pt.path_trace(scene.camera)
    counted = pt.profile()
    pt.set_profiling(False, False)

    # roofline of the dominant kernel (trace): algorithmic bytes / summed launch duration
    scale = [(paths_timed[b] / counted["paths"][b]) if counted["paths"][b] else 0.0 for b in range(MB)]
    alg_bytes = sum(paths_timed[b] * 52 + scale[b] * (counted["box_tests"][b] * 32 + counted["tri_tests"][b] * 48)
                    for b in range(MB))
    # the same sum with what THIS implementation fetches per test (16 B per box: four boxes in a 64-byte node)
    own_bytes = sum(paths_timed[b] * 52 + scale[b] * (counted["box_tests"][b] * 16 + counted["tri_tests"][b] * 48)
                    for b in range(MB))
    trace_ms = sum(prof["trace_ms"])
    launches = sum(prof["trace_launches"])
    achieved = alg_bytes / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
    traffic = None
    pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_file):
        try:
            traffic = json.load(open(pmc_file)).get("trace_kernel_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    # Frames in flight overlap launches of this kernel on several streams: `achieved` is what ONE launch gets
    # while sharing the chip (bytes of a launch / its own duration); achieved_chip is all launches' bytes over
    # the wall time of the timed region.
    achieved_chip = alg_bytes / elapsed / 1e9
    roofline = {
        "bound": "hbm", "kernel": "k_traverse4 (closest hit: persistent wavefronts over the 4-wide BVH, %d frames per launch)" % args.batch_frames, "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
        "launches": launches, "avg_launch_us": round(trace_ms * 1e3 / max(launches, 1), 2),
        "alg_bytes_per_launch": round(alg_bytes / max(launches, 1)),
        "concurrent_launches": round(trace_ms * 1e-3 / elapsed, 3),
        "achieved_chip": round(achieved_chip, 2), "frac_chip": round(achieved_chip / HBM_PEAK_GBS, 5),
        "achieved_own_layout": round(own_bytes / (trace_ms * 1e-3) / 1e9, 2) if trace_ms > 0 else 0.0,
        "box_tests_per_ray": round(sum(counted["box_tests"]) / max(sum(counted["paths"]), 1), 2),
        "tri_tests_per_ray": round(sum(counted["tri_tests"]) / max(sum(counted["paths"]), 1), 2),
    }

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        orc = graft.load_oracle()
        cw, ch = (int(v) for v in args.cpu_sample.split("x"))
        sh = orc.SceneHandle(flat)
        cores = max(1, min(args.cpu_threads, orc.lib().orc_hardware_threads()))
        t0 = time.perf_counter()
        ref = orc.render_streaming(flat, scene.camera, cw, ch, 0, 1, MB, nthreads=cores, scene_handle=sh)
        dt = time.perf_counter() - t0
        cpu_baseline = {"value": round(ref["rays"] / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                        "sample": f"1 frame of the same scene (same {len(flat.indices) // 3}-triangle BVH, {MB} bounces) at "
                                  f"{cw}x{ch}: {ref['rays']} rays in {dt:.2f} s; oracle/liboracle.so (CPU restatement; "
                                  "the reference has no CPU path)"}

    if rank == 0:
        value = rays / elapsed / 1e6
        line = {
            "metric": "Mrays/s at 1920\u00d71080, 8 bounces", "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"config 3: procedural {len(flat.indices) // 3}-triangle heightfield + 3 spheres, "
                                   f"{W}x{H}, {MB} bounces, 1 spp/step, streaming mode",
                       "triangles": len(flat.indices) // 3, "bvh_nodes": int(len(flat.bvh)), "bvh_depth": int(bvh_depth),
                       "resolution": [W, H], "max_bounces": MB, "frames_in_flight": args.frames_in_flight, "frames_per_launch": args.batch_frames, "rays_per_step": round(rays / args.steps),
                       "live_per_bounce_last_frame_rank0": last_live, "partition": "full frame" if world == 1 else f"rows in blocks of {BLOCK_ROWS} dealt round-robin over {world} ranks",
                       "bvh_build_s": round(bvh_build_s, 3)},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    pt.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
