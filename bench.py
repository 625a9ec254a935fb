#!/usr/bin/env python3
"""bench.py -- benchmark of the MI355X path-tracing core.

Metric (BASELINE.json): Mrays/s at 1920x1080, 8 bounces, on the 1,000,000-triangle scene (config 3:
`configs[2]`, the configuration the metric is quoted on; it fits one GPU).  One "step" = one frame =
one sample per pixel through PathTracer::path_trace (streaming mode).  1 ray = 1 closest-hit query,
primary rays included.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3|2|5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset) starts its own N ranks: a child
`torch.distributed.run` (one rank per GPU, RCCL), spawned before this process has touched the GPU, whose JSON line and
exit code it hands on.  Fewer than N devices is an error unless --rehearse-on-one-gpu is given (all ranks on cuda:0,
gloo: the N > 1 code path on a one-GPU box, not a scaling number).

--config 3 (default)  the headline: 1M-triangle heightfield + 3 spheres, 1920x1080, 8 bounces
--config 2            Cornell box + two instances of the 69,984-triangle mesh, 1280x720, 8 bounces (N=1)
--config 5            config 3's scene, 1 spp + the A-Trous denoiser after EVERY frame, presented every frame
                      (interactive mode; metric ms per frame; roofline of k_denoise) (N=1)

N > 1 (config 3): the frame's rows are dealt to the N ranks in blocks of 8 rows (round-robin: sky rows are
cheap, terrain rows expensive, so contiguous bands would be unbalanced), one process per GPU; no data-path
collective during tracing; the ranks' radiance is gathered to rank 0 over RCCL once, at present time, inside the
timed region, and scattered back into frame order there.  Fixed total work -> "scaling": "strong".

Prints ONE JSON line on rank 0 with the contract fields plus:
  roofline      dominant kernel = the closest-hit (trace) kernel k_traverse4.
                achieved = SURVEY 8(d)'s ALGORITHMIC bytes of all its launches in the timed region / their summed
                duration (HIP events on the kernel's stream, recorded inside libptcore around every launch).
                Algorithmic bytes of a launch = rays*52 + node_visits*32 + tri_tests*48 (32 B per BVH node visited,
                12 B indices + 36 B vertices per triangle test, 32 B ray read + 20 B compact hit written per ray:
                layout-independent minimums taken from the reference's node / triangle sizes), with the counts from
                an instrumented, untimed re-run of the same frames.  `requested` prices what THIS layout asks the
                memory system for (64-byte four-child node records, 48-byte triangle records).  `traffic` = HBM-side
                bytes per launch from separate rocprofv3 --pmc passes (profiles/pmc_traffic_<frames per launch>.json), given only when
                that record was taken on the same launch shape (frames per launch), else null.  The kernel is not
                HBM-bound -- its working set lives in L2 / Infinity Cache -- so `valu` carries the roofline that
                does bind it: VALU issue (busy %, lanes per instruction, instructions per ray; SQ counter passes,
                profiles/pmc_sq_<frames per launch>.json).
  parity        the tracer that was just timed, restarted, against the CPU oracle's frames (the ones rendered for
                cpu_baseline): mse, bit_exact, live counts and ray count equal.  N > 1 (and --share-of N): EVERY rank's
                rows against the oracle's rendering of that rank (orc_render_streaming_interleaved: interleaved blocks,
                per-rank numbering, slot_offset = rank * W * H), rank 0's first (timed: cpu_baseline), the others' after it.
  rccl_ranks    N > 1: the sum of a device all-reduce of ones over the process group whose backend reports "nccl"
                (asserted == N; 0 in a gloo rehearsal); distinct_devices = distinct PCI identities among the ranks' GPUs
                (asserted == N outside a rehearsal).
  steady_state  the same workload with 32 frames per launch and 256 steps (the tuned schedule; the timed region
                above runs exactly --steps frames, which the driver sets to 20).
  latency       ms per frame when frames are strictly serial (one frame in flight) and in the viewer pattern
                (present after every iteration; eight one-frame slots).
  cpu_baseline  the CPU oracle (oracle/, kind "port": the reference has no CPU path) timed on this box's cores on
                a bounded sample of the same workload (rank 0, N=1 only).
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
BLOCK_ROWS = 8         # multi-GPU: rows are dealt to the ranks in blocks of this many


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--config", type=int, default=3, choices=(2, 3, 5))
    ap.add_argument("--width", type=int, default=0, help="0 = the configuration's resolution")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--max-bounces", type=int, default=8)
    ap.add_argument("--grid", type=str, default="1001x501", help="heightfield vertex grid (1001x501 = 1,000,000 triangles)")
    ap.add_argument("--batch-frames", type=int, default=0,
                    help="frames traced per launch; 0 = 32, or an even split of --steps over the streams when --steps is "
                         "smaller than one round of full batches")
    ap.add_argument("--streams", type=int, default=0, help="batches in flight; 0 = 2 per rank on one or two GPUs, 4 per rank on more")
    ap.add_argument("--traverse-waves", type=int, default=0, help="persistent wavefronts of a full-size traversal launch (0 = tuned default)")
    ap.add_argument("--ray-sort", type=int, default=0, help="1: direction-octant ray sorting of the pick-up order (config.ray_sort)")
    ap.add_argument("--trace-variant", type=int, default=-1, help="closest-hit kernel (-1 = the library's default; 0 reference order, 1 culled two-wide, 3 persistent four-wide)")
    ap.add_argument("--param", action="append", default=[], metavar="NAME=VALUE", help="extra ptc_set_param before the scene upload (A/B runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the steady-state and latency measurements")
    ap.add_argument("--gather", choices=("rccl", "ipc"), default="rccl",
                    help="N > 1, present time: rccl = torch.distributed gather of the ranks' rows over RCCL (default); ipc = the "
                         "library's inter-process gather (ptc_band_* / ptc_gather_frame, what hip_pt --gpus N uses)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="all ranks share cuda:0 and talk over gloo (to rehearse the N>1 code path on a 1-GPU box)")
    ap.add_argument("--share-of", type=int, default=0, metavar="N",
                    help="diagnostic, one process: trace only rank 0's rows of an N-way split with the schedule an N-GPU run "
                         "would pick (how long does ONE rank of --gpus N take?  no gather, not a whole-job number)")
    ap.add_argument("--no-events", action="store_true", help="do not time the trace kernel with HIP events (diagnostic)")
    ap.add_argument("--cpu-sample", type=str, default="", help="resolution of the CPU-oracle sample frames (default: the full frame)")
    ap.add_argument("--cpu-frames", type=int, default=8, help="iterations the CPU oracle renders (cpu_baseline sample and parity frame): "
                                                               "8 full frames are about 15 s on 16 threads")
    ap.add_argument("--cpu-threads", type=int, default=16, help="oracle threads (the GPU box's CPU share for one GPU is 16)")
    return ap.parse_args()


def pmc_record(name, frames_per_launch):
    """profiles/<name>: counter figures of the dominant kernel from separate rocprofv3 --pmc passes."""
    path = os.path.join(ROOT, "profiles", name % frames_per_launch)
    if not os.path.exists(path):   # no record for this launch shape: the 32-frame record, marked as not matching
        path = os.path.join(ROOT, "profiles", name % 32)
    if not os.path.exists(path):
        return None
    try:
        rec = json.load(open(path))
    except Exception:
        return None
    rec["matches_this_run"] = rec.get("frames_per_launch") == frames_per_launch
    return rec


def trace_roofline(prof, counted, paths_timed, MB, elapsed, frames_per_launch, pmc_scene=True, pixels=0, frames=0, pmc_tag=""):
    """SURVEY 8(d) pricing of the closest-hit launches of the timed region.  pmc_scene: the counter records under
    profiles/ (pmc_traffic_<tag><frames per launch>.json, pmc_sq_...: tag "" = config 3, "c2_" = config 2) were taken on THIS
    scene; otherwise traffic / valu are null."""
    scale = [(paths_timed[b] / counted["paths"][b]) if counted["paths"][b] else 0.0 for b in range(MB)]
    rays = sum(paths_timed)
    # "filter_rays": a bounce whose traversal launch fetched through a work list walked only the listed rays; the others'
    # closest-hit queries were answered by the kernel that built the list (their world-box test).  The launch is priced with
    # the rays it read and wrote, the frame with all queries.
    listed = prof.get("listed_rays") or [0] * MB
    walked = [int(listed[b]) if listed[b] else int(paths_timed[b]) for b in range(MB)]
    nodes = sum(scale[b] * counted["node_visits"][b] for b in range(MB))
    boxes = sum(scale[b] * counted["box_tests"][b] for b in range(MB))
    tris = sum(scale[b] * counted["tri_tests"][b] for b in range(MB))
    alg_bytes = sum(walked) * 52 + nodes * 32 + tris * 48   # 32 B per node visit
    req_bytes = sum(walked) * 52 + nodes * 64 + tris * 48   # this layout: 64-byte node records
    trace_ms = sum(prof["trace_ms"])
    launches = sum(prof["trace_launches"])
    achieved = alg_bytes / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
    traffic_rec = pmc_record("pmc_traffic_" + pmc_tag + "%d.json", frames_per_launch) if pmc_scene else None
    traffic = None
    note = None if pmc_scene else "null: no counter record for this scene (profiles/pmc_*.json are configs 2 and 3 at their default sizes)"
    if traffic_rec is not None:
        if traffic_rec["matches_this_run"]:
            traffic = traffic_rec.get("trace_kernel_hbm_bytes_per_launch")
            note = "profiles/pmc_traffic_<frames per launch>.json (%s)" % traffic_rec.get("source", "separate --pmc passes")
        else:
            note = ("null: the PMC record at hand was measured on %s-frame launches (%.3g GB per launch), this run used %d"
                    % (traffic_rec.get("frames_per_launch"), (traffic_rec.get("trace_kernel_hbm_bytes_per_launch") or 0) / 1e9,
                       frames_per_launch))
    roof = {
        "bound": "hbm", "priced_against": "hbm",
        "kernel": "k_traverse4 (closest hit: persistent wavefronts over the 4-wide quantised BVH, %d frames per launch)" % frames_per_launch,
        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
        "traffic": traffic, "traffic_note": note,
        "pricing": "SURVEY 8(d): rays*52 + node_visits*32 + tri_tests*48 bytes",
        "launches": launches, "avg_launch_us": round(trace_ms * 1e3 / max(launches, 1), 2),
        "alg_bytes_per_launch": round(alg_bytes / max(launches, 1)),
        "alg_bytes_per_ray": round(alg_bytes / max(rays, 1), 1),
        "rays_walked_by_the_launches": sum(walked), "rays_answered_by_the_list_builder": int(rays - sum(walked)),
        "requested_bytes_per_launch": round(req_bytes / max(launches, 1)),
        "achieved_requested": round(req_bytes / (trace_ms * 1e-3) / 1e9, 2) if trace_ms > 0 else 0.0,
        "frac_requested": round(req_bytes / (trace_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if trace_ms > 0 else 0.0,
        "concurrent_launches": round(trace_ms * 1e-3 / elapsed, 3),
        "node_visits_per_ray": round(nodes / max(rays, 1), 2),
        "box_tests_per_ray": round(boxes / max(rays, 1), 2),
        "tri_tests_per_ray": round(tris / max(rays, 1), 2),
        "note": "the kernel's working set (27 MB of nodes + 48 MB of triangles + path state of the batch) is served "
                "by L2 / Infinity Cache; it is bound by VALU issue at about half lane utilisation, see valu",
    }
    # where the launch time goes, bounce by bounce (primary rays = bounce 0): rays, launch time (HIP events), node visits
    # and triangle tests per ray, and the same SURVEY 8(d) pricing per bounce
    per_bounce = []
    for b in range(MB):
        if not paths_timed[b]:
            continue
        nb, tb = scale[b] * counted["node_visits"][b], scale[b] * counted["tri_tests"][b]
        bytes_b = walked[b] * 52 + nb * 32 + tb * 48
        ms_b = prof["trace_ms"][b]
        per_bounce.append({"bounce": b, "rays": int(paths_timed[b]), "rays_walked": walked[b], "trace_ms": round(ms_b, 3), "launches": int(prof["trace_launches"][b]),
                           "node_visits_per_ray": round(nb / paths_timed[b], 2), "tri_tests_per_ray": round(tb / paths_timed[b], 2),
                           "ns_per_ray": round(ms_b * 1e6 / paths_timed[b], 3) if ms_b > 0 else None,
                           "frac": round(bytes_b / (ms_b * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_b > 0 else None})
    roof["per_bounce"] = per_bounce
    # SURVEY 8(d)'s whole-frame figure: B = sum_b n_b * 168 + node visits * 32 + triangle tests * 48 + P * (65 + 56) bytes per
    # frame over the wall time of a frame (all kernels, not only the closest-hit launches)
    if pixels and frames and elapsed > 0:
        frame_bytes = (rays * 168 + nodes * 32 + tris * 48) / frames + pixels * 121
        roof["frame_level"] = {"bytes_per_frame": round(frame_bytes), "ms_per_frame": round(elapsed / frames * 1e3, 4),
                               "achieved": round(frame_bytes * frames / elapsed / 1e9, 2),
                               "frac": round(frame_bytes * frames / elapsed / 1e9 / HBM_PEAK_GBS, 4),
                               "pricing": "SURVEY 8(d): sum_b n_b*168 + node_visits*32 + tri_tests*48 + P*(65+56) bytes per frame / wall time per frame"}
    sq = pmc_record("pmc_sq_" + pmc_tag + "%d.json", frames_per_launch) if pmc_scene else None
    if sq is not None:
        roof["valu"] = {k: sq.get(k) for k in ("valu_pipe_frac", "lanes_per_valu_inst", "valu_inst_per_ray", "vmem_inst_per_ray",
                                                "salu_inst_per_ray", "l2_hit_frac", "wave_occupancy_frac", "frames_per_launch",
                                                "matches_this_run", "source")}
        raw = sq.get("raw_means_per_launch") or {}
        us, clk = sq.get("mean_launch_us_serialised_by_the_profiler"), sq.get("shader_clock_ghz")
        if raw.get("TCP_TOTAL_CACHE_ACCESSES_sum") and us and clk:
            # vector-L1 tag look-ups per shader cycle and CU (256 CUs): a 64-byte node record is four of them
            roof["valu"]["l1_tag_accesses_per_cycle_per_cu"] = round(raw["TCP_TOTAL_CACHE_ACCESSES_sum"] / (us * 1e-6 * clk * 1e9 * 256), 3)
        # what the counters say binds the kernel: VALU issue at under half the lanes, and the vector L1's
        # tag rate; the fabric carries a quarter of its peak.  The SURVEY 8(d) HBM pricing stays in achieved / peak / frac.
        l1 = roof["valu"].get("l1_tag_accesses_per_cycle_per_cu") or 0.0
        if (sq.get("valu_pipe_frac") or 0) >= 0.45 or l1 >= 0.6:
            roof["bound"] = "valu-issue / vL1D"
            roof["bound_note"] = ("from counters (profiles/pmc_sq_*.json): VALU pipe %.2f at %.1f of 64 lanes, %s L1 tag accesses per cycle and CU, "
                                  "L2 hit rate %.2f; fabric traffic is a quarter of the HBM peak -- achieved / peak / frac price the kernel "
                                  "against HBM as SURVEY 8(d) prescribes (priced_against), they do not name what binds it"
                                  % (sq.get("valu_pipe_frac") or 0, sq.get("lanes_per_valu_inst") or 0, l1, sq.get("l2_hit_frac") or 0))
        else:
            # (config 2's launch over the two instances: few listed rays, each with its fixed round trips)
            roof["bound"] = "latency (dependent round trips per ray)"
            roof["bound_note"] = ("from counters (profiles/pmc_sq_%s*.json): VALU pipe %.2f at %.1f of 64 lanes, %s L1 tag accesses per cycle and CU, L2 hit rate "
                                  "%.2f, wavefront-slot occupancy %.2f -- no unit is busy half the time; a listed ray's fetch, root tests, winner check and "
                                  "store are dependent round trips that %d wavefronts per SIMD do not cover.  achieved / peak / frac price the kernel "
                                  "against HBM as SURVEY 8(d) prescribes (priced_against)"
                                  % (pmc_tag, sq.get("valu_pipe_frac") or 0, sq.get("lanes_per_valu_inst") or 0, l1, sq.get("l2_hit_frac") or 0,
                                     sq.get("wave_occupancy_frac") or 0, round((raw.get("SQ_WAVES") or 0) / 1024.0)))
    return roof


def _gather_floats(dist, group, world, value):
    """every rank's value (control plane: objects over the gloo group, outside any timed region)"""
    out = [None] * world
    dist.all_gather_object(out, float(value), group=group)
    return out


_SCENES = {}


def config3_scene(pkg, W, H, grid):
    """The benchmark scene, its flattened arrays and the host-built reference BVH (the tree the CPU oracle walks), built once
    per process: the extra legs of a run (rank_share, reference_order_gpu, config 5) reuse them."""
    key = (W, H, grid)
    if key not in _SCENES:
        nx, nz = (int(v) for v in grid.split("x"))
        scene = pkg.scenes.heightfield_scene((W, H), nx=nx, nz=nz)
        flat = scene.build_scene()
        mesh = list(scene.mesh_map_.values())[0]
        t0 = time.perf_counter()
        flat.bvh, bvh_depth = pkg.bvh_from_mesh(mesh)       # host builder: the tree the CPU oracle walks (parity, cpu_baseline)
        _SCENES[key] = (scene, flat, bvh_depth, time.perf_counter() - t0)
    return _SCENES[key]


def run_config3(args, pkg, torch, dist, world, rank, local_rank, rehearse, ctl=None):
    W, H, MB = args.width or 1920, args.height or 1080, args.max_bounces
    scene, flat, bvh_depth, bvh_build_s = config3_scene(pkg, W, H, args.grid)
    import copy
    bare = copy.copy(flat)
    bare.bvh = None   # the tracer gets the scene as the reference's front end hands it over: the library builds BVH and layouts (on the GPU)
    startup = {}
    split = args.share_of if (args.share_of > 1 and world == 1) else world   # ranks the frame's rows are dealt to
    rank_rows = pkg.bands.interleaved_rows(H, split, BLOCK_ROWS)

    # Schedule: `batch` consecutive frames share every launch (the latency tail of a bounce -- a few long rays --
    # and the drain of the persistent wavefronts are paid once per batch) and `streams` batches are in flight so
    # that the small kernels and the tail of one overlap the bulk of another.  32 frames per launch is the tuned
    # value; a timed region shorter than one round of full batches (the driver's --steps 20) is split evenly over
    # the streams instead, and the tuned schedule is reported beside it (steady_state).  Results do not depend on
    # any of it (tests/test_gpu_schedules.py).
    streams = args.streams or (2 if split <= 2 else 4)
    if args.batch_frames:
        batch = args.batch_frames
    elif args.steps >= streams * 32:
        batch = 32
    elif args.steps <= 32 and split < 4:
        batch, streams = args.steps, 1      # one launch sequence carries them all (measured: 1 x 20 beats 2 x 10 and 4 x 5)
    elif args.steps <= 32:
        # a rank of four or eight: its launches are small enough to be bound by the tail of their longest rays, and two
        # half-size sequences overlap those tails (measured with --share-of: 1/8 of the frame 0.159 vs 0.169 ms per step,
        # 1/4 0.257 vs 0.281; 4 x 5 loses again)
        batch, streams = -(-args.steps // 2), 2
    else:
        batch = max(1, min(32, -(-args.steps // streams)))

    def make_tracer(batch_frames, n_streams):
        pt = pkg.PathTracer(device=local_rank, max_bounces=MB)
        pt.set_param("frames_in_flight", n_streams * batch_frames)
        pt.set_param("batch_frames", batch_frames)
        # persistent traversal wavefronts per launch: what is resident at 5 per SIMD on one GPU; half of that for
        # the smaller launches of a rank among 4 or 8
        pt.set_param("traverse_waves", args.traverse_waves or (5120 if split <= 2 else 2560))
        pt.set_param("ray_sort", args.ray_sort)
        if args.trace_variant >= 0:
            pt.set_trace_variant(args.trace_variant)
        for kv in args.param:
            name, value = kv.split("=")
            pt.set_param(name, int(value))
        pt.create_buffers((W, H), bare)
        startup.update(pt.upload_times())
        pt.set_stream(torch.cuda.current_stream().cuda_stream)
        if split > 1:
            pt.set_interleave(rank, split, BLOCK_ROWS)
            # every rank numbers its compacted paths from 0 (no collective while tracing): an offset keeps the
            # ranks' random streams apart (the material RNG is keyed on the slot index, path_tracer.cu:300)
            pt.set_param("slot_offset", rank * W * H)
        pt.max_iterations = 1 << 30
        return pt

    # N > 1: who is really there.  rccl_ranks = what a device all-reduce of ones over the process group sums to, counted only
    # when the group's backend for CUDA tensors is nccl (= RCCL); distinct_devices = distinct PCI identities of the ranks' GPUs.
    rccl_ranks = distinct_devices = rccl_error = None
    data_ok = [True]            # the default process group (RCCL; gloo in a rehearsal) carries collectives
    gather_fallback = [None]
    if world > 1:
        props = torch.cuda.get_device_properties(local_rank)
        ident = tuple(str(getattr(props, k)) for k in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id") if hasattr(props, k)) or None
        idents = [None] * world
        dist.all_gather_object(idents, ident, group=ctl)
        distinct_devices = len(set(idents)) if all(i is not None for i in idents) else None
        if rehearse:
            rccl_ranks = 0
        else:
            # the first collective over RCCL.  If it throws (no peer access, an IPC failure at communicator set-up) the run goes
            # on: frames are traced without any collective, control messages travel over the gloo group, the present-time gather
            # falls back to the library's own (HIP IPC), and the line says so (rccl_error, gather.fallback) -- a scaling number
            # with rccl_ranks 0 is still a measurement of the tracing, not of RCCL.
            err = None
            try:
                ones = torch.ones(1, dtype=torch.int32, device="cuda")
                dist.all_reduce(ones)
                torch.cuda.synchronize()
                rccl_ranks = int(ones.item()) if "nccl" in str(dist.get_backend()) else 0
            except Exception as exc:   # noqa: BLE001
                err, rccl_ranks = repr(exc), 0
            errs = [None] * world
            dist.all_gather_object(errs, err, group=ctl)
            if any(errs):
                rccl_error = next(e for e in errs if e)
                data_ok[0] = False
                rccl_ranks = 0
                args.gather = "ipc"
            elif rccl_ranks != world:
                raise SystemExit(f"bench.py: --gpus {world}, but the RCCL all-reduce over the process group sums to {rccl_ranks} "
                                 f"(backend {dist.get_backend()})")
            if distinct_devices is not None and distinct_devices != world:
                raise SystemExit(f"bench.py: --gpus {world}, but the ranks sit on {distinct_devices} distinct device(s): {idents}")

    pt = make_tracer(batch, streams)
    # gather needs equal sizes on every rank: pad each rank's rows to the largest share
    max_rows = max(len(rr) for rr in rank_rows)
    band_color = torch.zeros((max_rows, W, 3), dtype=torch.float32, device="cuda")
    gathered = frame = row_index = gathered_all = frame_pad = None
    if world > 1 and rank == 0:
        # the ranks' (padded) bands land in ONE tensor and go into row order with ONE index_copy: row H of frame_pad takes
        # the padding rows of the ranks that own fewer rows than the largest share
        gathered_all = torch.empty((world, max_rows, W, 3), dtype=torch.float32, device="cuda")
        gathered = [gathered_all[r] for r in range(world)]
        frame_pad = torch.empty((H + 1, W, 3), dtype=torch.float32, device="cuda")
        frame = frame_pad[:H]
        row_index = torch.tensor([rr[k] if k < len(rr) else H for rr in rank_rows for k in range(max_rows)], dtype=torch.long, device="cuda")

    ipc_gathers = {}

    def present(tracer):
        """gather of per-rank radiance at present time (the only inter-GPU traffic) + back into frame order"""
        if world > 1 and args.gather == "ipc":
            if id(tracer) not in ipc_gathers:
                ipc_gathers[id(tracer)] = pkg.bands.BandGather(tracer, rank, world, dist, group=ctl)
            ipc_gathers[id(tracer)].gather("color", frame.data_ptr() if rank == 0 else None)
            return
        tracer.download_to_device("color", band_color.data_ptr())
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
                frame_pad.index_copy_(0, row_index, gathered_all.view(world * max_rows, W, 3))

    def fence():
        if world > 1:
            if data_ok[0]:
                dist.barrier()
            else:
                torch.cuda.synchronize()
                dist.barrier(group=ctl)
        torch.cuda.synchronize()

    per_rank = []   # N > 1: every rank's own rays and elapsed time of the last timed() region

    def timed(tracer, steps, warmup):
        for _ in range(warmup):
            tracer.path_trace(scene.camera)
        present(tracer)
        fence()
        tracer.reset_profile()
        tracer.set_profiling(time_trace_kernel=not args.no_events, count_tests=False)
        rays0 = tracer.stats()["rays_total"]
        first_iter = tracer.iteration()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            tracer.path_trace(scene.camera)
        present(tracer)
        fence()
        elapsed = time.perf_counter() - t0
        rays = tracer.stats()["rays_total"] - rays0
        prof = tracer.profile()
        tracer.set_profiling(False, False)
        per_rank.clear()
        if world > 1:   # max over ranks of the elapsed time, sum of rays
            mine = [None] * world
            dist.all_gather_object(mine, {"rank": rank, "rays": int(rays), "elapsed_ms": round(elapsed * 1e3, 4)}, group=ctl)
            per_rank.extend(mine)
            elapsed = max(_gather_floats(dist, ctl, world, elapsed))   # the MAX over ranks
            rays = sum(int(m["rays"]) for m in mine)
        return elapsed, rays, prof, first_iter

    def count_tests(tracer, first_iter, steps):
        """instrumented, untimed re-run of the same frames: node visits / box tests / triangle tests per bounce"""
        tracer.set_profiling(time_trace_kernel=False, count_tests=True)
        tracer.reset_profile()
        tracer.set_iteration(first_iter)
        for _ in range(min(steps, 4)):
            tracer.path_trace(scene.camera)
        counted = tracer.profile()
        tracer.set_profiling(False, False)
        return counted

    # N > 1: one present over RCCL before anything is timed.  A gather that throws here (on any rank) does not end the run: every
    # rank switches to the library's gather (--gather ipc: HIP IPC mapping, control messages over the gloo group) and the line
    # carries the error in gather.fallback -- as the IPC leg below already reports ITS failures beside the RCCL figure.
    if world > 1 and args.gather == "rccl" and data_ok[0]:
        err = None
        try:
            present(pt)
            torch.cuda.synchronize()
        except Exception as exc:   # noqa: BLE001
            err = repr(exc)
        errs = [None] * world
        dist.all_gather_object(errs, err, group=ctl)
        if any(errs):
            gather_fallback[0] = "the RCCL gather failed before the timed region (%s): the timed region uses --gather ipc" % next(e for e in errs if e)
            args.gather = "ipc"
    elif world > 1 and not data_ok[0]:
        gather_fallback[0] = "RCCL unavailable (%s): the timed region uses --gather ipc, control messages over gloo" % rccl_error

    elapsed, rays, prof, first_iter = timed(pt, args.steps, args.warmup)
    last_live = pt.stats()["last_live"]
    ranks_detail = list(per_rank)

    # N > 1: the present-time gather on its own (SURVEY config 4: "RCCL gather time reported separately").  Both
    # transports, five presents each after the timed region: the torch.distributed gather (RCCL; gloo in a rehearsal)
    # timed with events on rank 0's stream from the first byte leaving to the frame being in row order, and the
    # library's one-kernel pull over mapped peer buffers (ptc_gather_frame, device time from ptc_gather_last_us).
    gather = None
    if world > 1:
        gather = {"bytes_per_present": int(sum(len(rr) for rr in rank_rows[1:]) * W * 12),
                  "transport_in_timed_region": args.gather, "backend": "gloo (rehearsal)" if rehearse else "nccl (RCCL)",
                  "fallback": gather_fallback[0]}
        saved = args.gather
        rccl_frame = None
        if gather_fallback[0] is None:
            args.gather = "rccl"
            us, err = [], None
            try:
                for _ in range(5):
                    fence()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    present(pt)
                    e1.record()
                    torch.cuda.synchronize()
                    us.append(e0.elapsed_time(e1) * 1e3)
            except Exception as exc:   # noqa: BLE001 -- reported; the IPC leg follows
                err = repr(exc)
            errs = [None] * world
            dist.all_gather_object(errs, err, group=ctl)
            if any(errs):
                gather["rccl_us"], gather["rccl_error"] = None, next(e for e in errs if e)
            else:
                rccl_frame = frame.clone() if rank == 0 else None
                gather["rccl_us"] = round(sorted(us)[len(us) // 2], 1) if rank == 0 else None
            gather["rccl_note"] = "pack + dist.gather + index_copy into row order, rank 0's stream, median of 5"
        else:
            gather["rccl_us"], gather["rccl_note"] = None, "not measured: " + gather_fallback[0]
        # the library's gather: handles once, then publish / barrier / pull
        try:
            handle, err = (pt.band_export() if rank else b""), None
        except Exception as exc:   # noqa: BLE001 -- reported, not fatal: the RCCL figure stands
            handle, err = None, repr(exc)
        # (the pull kernel reads the peers' buffers in place: where the root's GPU cannot address a peer's memory the leg is
        # skipped and says so, rather than finding out inside a kernel)
        if rank and not rehearse and err is None:
            try:
                if not torch.cuda.can_device_access_peer(0, local_rank):
                    err = f"cuda:0 cannot access cuda:{local_rank} (no peer access)"
            except Exception as exc:   # noqa: BLE001
                err = "peer-access query failed: " + repr(exc)
        handles = [None] * world
        dist.all_gather_object(handles, (handle, err), group=ctl)
        state = [None]
        if rank == 0:
            try:
                for r in range(1, world):
                    if handles[r][1]:
                        raise RuntimeError(f"rank {r}: {handles[r][1]}")
                    pt.band_import(r, handles[r][0])
            except Exception as exc:   # noqa: BLE001
                state = [repr(exc)]
        dist.broadcast_object_list(state, src=0, group=ctl)
        if state[0] is None:
            us = []
            for _ in range(5):
                if rank:
                    pt.band_publish("color")
                dist.barrier(group=ctl)
                if rank == 0:
                    try:
                        pt.gather_frame("color", frame.data_ptr())
                        us.append(pt.gather_last_us())
                    except Exception as exc:   # noqa: BLE001 -- reported in the line, the RCCL figure stands
                        state[0] = repr(exc)
                dist.barrier(group=ctl)
            gather["ipc_us"] = round(sorted(us)[len(us) // 2], 1) if rank == 0 and us else None
            gather["transports_agree"] = (bool(torch.equal(rccl_frame, frame)) if rank == 0 and state[0] is None and rccl_frame is not None
                                          else None)   # the same assembled frame, bit for bit
            if rank == 0 and state[0] is not None:
                gather["ipc_error"] = state[0]
            gather["ipc_note"] = ("ptc_gather_frame: one kernel on the root reads every rank's band where it lies (HIP IPC "
                                  "mapping; xGMI between GPUs) and writes row order; device time, median of 5")
        else:
            gather["ipc_us"], gather["ipc_note"] = None, "library gather unavailable here: " + state[0]
        args.gather = saved
    counted = count_tests(pt, first_iter, args.steps)
    # parity leg, GPU side: THE TRACER THAT WAS JUST TIMED (same context, slots, streams, launch plan) restarted and run
    # for the iterations the CPU oracle renders below; compared bit for bit there
    timed_frames = None
    cw, ch = (int(v) for v in args.cpu_sample.split("x")) if args.cpu_sample else (W, H)
    cpu_frames = max(1, min(args.cpu_frames, args.steps))
    if (rank == 0 or split > 1) and not args.no_cpu_baseline and (cw, ch) == (W, H):
        pt.restart()
        r0 = pt.stats()["rays_total"]
        for _ in range(cpu_frames):
            pt.path_trace(scene.camera)
        timed_frames = {k: pt.download(k) for k in ("color", "normal", "depth")}
        st = pt.stats()
        timed_frames["rays"] = st["rays_total"] - r0
        timed_frames["last_live"] = st["last_live"]
    # (the counter records under profiles/ were taken on the default workload with the default kernel)
    pmc_scene = args.grid == "1001x501" and (W, H, MB) == (1920, 1080, 8) and args.trace_variant in (-1, 3) and split == 1
    roofline = trace_roofline(prof, counted, list(prof["paths"]), MB, elapsed, batch, pmc_scene=pmc_scene,
                              pixels=W * H // split, frames=args.steps)
    slow_rays = sum(prof["slow_rays"])
    walked_total = roofline["rays_walked_by_the_launches"]
    if world > 1:   # every rank's traversal launches
        walked_total = int(sum(_gather_floats(dist, ctl, world, walked_total)))

    # the tuned schedule on the same workload (only when the timed region above was too short to show it)
    steady = None
    if not args.no_extras and (batch != 32 or args.steps < 256):
        tuned_streams = args.streams or (2 if split <= 2 else 4)   # (not the single stream of a short timed region)
        if batch != 32 or streams != tuned_streams:
            pt.close()
            pt = make_tracer(32, tuned_streams)
        s_el, s_rays, s_prof, s_first = timed(pt, 256, 64)
        s_counted = count_tests(pt, s_first, 4)
        s_roof = trace_roofline(s_prof, s_counted, list(s_prof["paths"]), MB, s_el, 32, pmc_scene=pmc_scene, pixels=W * H // split, frames=256)
        steady = {"value": round(s_rays / s_el / 1e6, 3), "unit": "Mrays/s", "steps": 256, "warmup": 64, "frames_per_launch": 32,
                  "frames_in_flight": tuned_streams * 32, "ms_per_step": round(s_el / 256 * 1e3, 4),
                  "roofline_frac": s_roof["frac"], "roofline_achieved": s_roof["achieved"],
                  "avg_launch_us": s_roof["avg_launch_us"], "traffic": s_roof["traffic"],
                  "frac_requested": s_roof["frac_requested"], "concurrent_launches": s_roof["concurrent_launches"],
                  "frame_level_frac": (s_roof.get("frame_level") or {}).get("frac"), "per_bounce": s_roof["per_bounce"],
                  "mrays_per_s_walked_rank0": round(s_roof["rays_walked_by_the_launches"] / s_el / 1e6, 3)}
    pt.close()

    # single-frame latency: strictly serial frames, and the viewer pattern (app.cpp:141-170 presents every frame)
    latency = None
    if not args.no_extras and split == 1:
        latency = {}
        rgba = torch.empty((H, W), dtype=torch.int32, device="cuda")
        for key, fif, every in (("serial_frame_ms", 1, False), ("present_every_frame_ms", 4, True)):
            lp = pkg.PathTracer(device=local_rank, max_bounces=MB)
            lp.set_param("frames_in_flight", fif)
            lp.set_param("batch_frames", min(fif, 2))
            lp.create_buffers((W, H), flat)
            lp.set_stream(torch.cuda.current_stream().cuda_stream)
            lp.max_iterations = 1 << 30
            for _ in range(8):
                lp.path_trace(scene.camera)
                if every:
                    lp.send_to_preview(rgba.data_ptr())
            lp.synchronize()
            t0 = time.perf_counter()
            n = 32
            for _ in range(n):
                lp.path_trace(scene.camera)
                if every:
                    lp.send_to_preview(rgba.data_ptr())
            lp.synchronize()
            latency[key] = round((time.perf_counter() - t0) / n * 1e3, 4)
            lp.close()

    cpu_baseline = parity = None

    def compare(got, ref):
        d = got["color"].astype(np.float64) - ref["color"].astype(np.float64)
        return {"mse": float(np.mean(np.sum(d * d, axis=-1))),
                "bit_exact": bool(all(np.array_equal(got[k], ref[k]) for k in ("color", "normal", "depth"))),
                "live_equal": bool(got["last_live"][:MB] == [int(v) for v in ref["live"][-1][:MB]]),
                "rays_equal": bool(got["rays"] == ref["rays"])}

    if rank == 0 and split == 1 and not args.no_cpu_baseline:
        orc = graft.load_oracle()
        sh = orc.SceneHandle(flat)
        cores = max(1, min(args.cpu_threads, orc.lib().orc_hardware_threads()))
        t0 = time.perf_counter()
        ref = orc.render_streaming(flat, scene.camera, cw, ch, 0, cpu_frames, MB, nthreads=cores, scene_handle=sh)
        dt = time.perf_counter() - t0
        cpu_baseline = {"value": round(ref["rays"] / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                        "sample": f"{cpu_frames} accumulated iterations of the same scene (same {len(flat.indices) // 3}-triangle BVH, "
                                  f"{MB} bounces) at {cw}x{ch}: {ref['rays']} rays in {dt:.2f} s; oracle/liboracle.so (CPU "
                                  "restatement; the reference has no CPU path)"}
        if timed_frames is not None:
            got, what = timed_frames, "the timed tracer itself (same context and schedule: %d frames per launch, %d stream(s)), restarted" % (batch, streams)
        else:   # a reduced --cpu-sample resolution: a tracer of that size with the library's default schedule
            with pkg.PathTracer(device=local_rank, max_bounces=MB) as cp:
                cp.create_buffers((cw, ch), flat)
                cp.max_iterations = cpu_frames
                for _ in range(cpu_frames):
                    cp.path_trace(scene.camera)
                got = {k: cp.download(k) for k in ("color", "normal", "depth")}
                st = cp.stats()
                got["rays"], got["last_live"] = st["rays_total"], st["last_live"]
            what = "a tracer of the sample's size, default schedule"
        parity = {"against": "the CPU oracle's %d accumulated iterations at %dx%d (the frames of cpu_baseline)" % (cpu_frames, cw, ch),
                  "gpu_side": what, **compare(got, ref), "tolerance_mse": 1e-4}
    elif split > 1 and not args.no_cpu_baseline and timed_frames is not None:
        # The split that was timed (interleaved row blocks, per-rank numbering, slot_offset = rank * W * H) against the oracle's
        # rendering of the same rank's rows.  Rank 0 goes first and alone -- its oracle run is the cpu_baseline sample -- then
        # the other ranks check their own rows side by side on what is left of the host's cores.
        orc = graft.load_oracle()
        sh = orc.SceneHandle(flat)
        hw = orc.lib().orc_hardware_threads()
        cores = max(1, min(args.cpu_threads, hw))
        mine = None
        if rank == 0:
            t0 = time.perf_counter()
            ref = orc.render_interleaved(flat, scene.camera, W, H, 0, split, BLOCK_ROWS, 0, 0, cpu_frames, MB, nthreads=cores, scene_handle=sh)
            dt = time.perf_counter() - t0
            cpu_baseline = {"value": round(ref["rays"] / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                            "sample": f"rank 0's rows of the {split}-way split ({len(rank_rows[0])} of {H} rows in blocks of {BLOCK_ROWS}), "
                                      f"{cpu_frames} accumulated iterations, same {len(flat.indices) // 3}-triangle BVH, {MB} bounces: "
                                      f"{ref['rays']} rays in {dt:.2f} s; oracle/liboracle.so (CPU restatement; the reference has no CPU path)"}
            mine = {"rank": 0, **compare(timed_frames, ref)}
        if world > 1:
            dist.barrier(group=ctl)   # (the gloo group: its timeout is set for a rank that waits for rank 0's oracle leg, see main)
        if rank != 0:
            share = max(1, min(args.cpu_threads, hw // max(world - 1, 1)))
            ref = orc.render_interleaved(flat, scene.camera, W, H, rank, split, BLOCK_ROWS, rank * W * H, 0, cpu_frames, MB,
                                         nthreads=share, scene_handle=sh)
            mine = {"rank": rank, **compare(timed_frames, ref)}
        every = [mine]
        if world > 1:
            every = [None] * world
            dist.all_gather_object(every, mine, group=ctl)
        if rank == 0:
            parity = {"against": "the CPU oracle's rendering of each rank's rows (orc_render_streaming_interleaved: blocks of %d rows dealt over %d "
                                 "ranks, paths numbered per rank, slot_offset = rank * W * H), %d accumulated iterations at %dx%d"
                                 % (BLOCK_ROWS, split, cpu_frames, W, H),
                      "gpu_side": "every rank's timed tracer itself (same context and schedule: %d frames per launch, %d stream(s)), restarted"
                                  % (batch, streams) + ("" if world > 1 else "; --share-of: rank 0's rows only"),
                      "mse": max(e["mse"] for e in every), "bit_exact": all(e["bit_exact"] for e in every),
                      "live_equal": all(e["live_equal"] for e in every), "rays_equal": all(e["rays_equal"] for e in every),
                      "ranks": every, "tolerance_mse": 1e-4}

    if rank != 0:
        return None
    value = rays / elapsed / 1e6
    roofline["frame_level_frac"] = (roofline.get("frame_level") or {}).get("frac")
    if steady is not None:
        roofline["steady_state_frame_level_frac"] = steady["frame_level_frac"]
    return {
        "metric": "Mrays/s at 1920×1080, 8 bounces", "value": round(value, 3), "unit": "Mrays/s",
        "mrays_per_s_walked": round(walked_total / elapsed / 1e6, 3),
        "value_note": "value counts every closest-hit query (SURVEY 8d); mrays_per_s_walked counts only the rays the traversal launches "
                      "fetched: primary rays whose three world-box tests in k_raygen already say 'sky' are answered there "
                      "(roofline.rays_answered_by_the_list_builder) and never walk the tree",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"config 3: procedural {len(flat.indices) // 3}-triangle heightfield + 3 spheres, "
                               f"{W}x{H}, {MB} bounces, 1 spp/step, streaming mode",
                   "triangles": len(flat.indices) // 3, "bvh_nodes": int(len(flat.bvh)), "bvh_depth": int(bvh_depth),
                   "resolution": [W, H], "max_bounces": MB, "frames_in_flight": streams * batch,
                   "frames_per_launch": batch, "ray_sort": args.ray_sort,
                   "schedule_note": ("tuned schedule (32 frames per launch)" if batch == 32 else
                                     f"--steps {args.steps} is less than one round of full batches: {streams} launch sequence(s) of "
                                     f"{batch} frames; the tuned 32-frame schedule is reported under steady_state"),
                   "rays_per_step": round(rays / args.steps),
                   "live_per_bounce_last_frame_rank0": last_live, "exact_redo_rays": slow_rays,
                   "diagnostic": (f"--share-of {split}: ONE rank's rows of a {split}-way split traced on one GPU (no gather); value is that "
                                  "rank's own rate, not a job's") if split != world else None,
                   "partition": "full frame" if split == 1 else f"rows in blocks of {BLOCK_ROWS} dealt round-robin over {split} ranks; "
                                f"present-time gather: {args.gather}",
                   "startup": {"scene_upload_ms": startup, "note": "ptc_upload_scene of the scene without BVH: reference BVH (bit-identical "
                               "to the host builder's) and traversal layouts built on the GPU", "host_bvh_builder_s": round(bvh_build_s, 3)}},
        "roofline": roofline,
        "gather": gather,
        "ranks": ranks_detail or None,
        "rccl_ranks": rccl_ranks, "distinct_devices": distinct_devices, "rccl_error": rccl_error,
        "parity": parity,
        "steady_state": steady,
        "latency": latency,
        "cpu_baseline": cpu_baseline,
    }


def run_config2(args, pkg, torch, local_rank):
    """Config 2: Cornell box + two instances of the 69,984-triangle stand-in mesh, 1280x720."""
    W, H, MB = args.width or 1280, args.height or 720, args.max_bounces
    scene = pkg.scenes.cornell_bunny((W, H))
    flat = scene.build_scene()
    mesh = list(scene.mesh_map_.values())[0]
    flat.bvh, bvh_depth = pkg.bvh_from_mesh(mesh)
    batch = args.batch_frames or 32
    streams = args.streams or 2
    pt = pkg.PathTracer(device=local_rank, max_bounces=MB)
    pt.set_param("frames_in_flight", streams * batch)
    pt.set_param("batch_frames", batch)
    for kv in args.param:
        name, value = kv.split("=")
        pt.set_param(name, int(value))
    pt.create_buffers((W, H), flat)
    pt.set_stream(torch.cuda.current_stream().cuda_stream)
    pt.max_iterations = 1 << 30
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    for _ in range(args.warmup):
        pt.path_trace(scene.camera)
    pt.download_to_device("color", out.data_ptr())
    torch.cuda.synchronize()
    pt.reset_profile()
    pt.set_profiling(time_trace_kernel=not args.no_events, count_tests=False)
    rays0 = pt.stats()["rays_total"]
    first_iter = pt.iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pt.path_trace(scene.camera)
    pt.download_to_device("color", out.data_ptr())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rays = pt.stats()["rays_total"] - rays0
    prof = pt.profile()
    last_live = pt.stats()["last_live"]
    pt.set_profiling(False, True)
    pt.reset_profile()
    pt.set_iteration(first_iter)
    for _ in range(min(args.steps, 4)):
        pt.path_trace(scene.camera)
    counted = pt.profile()
    pt.set_profiling(False, False)
    # (counter records of this scene: profiles/pmc_*_c2_32.json, taken on the default workload with the default parameters)
    pmc_scene = (W, H, MB) == (1280, 720, 8) and not args.param and args.trace_variant in (-1, 3)
    roofline = trace_roofline(prof, counted, list(prof["paths"]), MB, elapsed, batch, pmc_scene=pmc_scene, pixels=W * H, frames=args.steps, pmc_tag="c2_")
    roofline["kernel"] = roofline["kernel"].replace("k_traverse4 (", "k_traverse4m (both instances of the mesh in one launch per bounce; ")
    roofline["note"] = ("%.1f node visits per ray: the fixed round trips of a ray (fetch, root, winner's parent box and normal, store) "
                        "weigh more than its walk" % roofline["node_visits_per_ray"])
    roofline["frame_level_frac"] = (roofline.get("frame_level") or {}).get("frac")
    cpu_baseline = parity = None
    cpu_frames = max(1, min(args.cpu_frames, 4, args.steps))
    if not args.no_cpu_baseline:
        # the tracer that was just timed, restarted, for the iterations the oracle renders below
        pt.restart()
        r0 = pt.stats()["rays_total"]
        for _ in range(cpu_frames):
            pt.path_trace(scene.camera)
        got = {k: pt.download(k) for k in ("color", "normal", "depth")}
        st = pt.stats()
        got["rays"], got["last_live"] = st["rays_total"] - r0, st["last_live"]
    pt.close()
    if not args.no_cpu_baseline:
        orc = graft.load_oracle()
        cores = max(1, min(args.cpu_threads, orc.lib().orc_hardware_threads()))
        t0 = time.perf_counter()
        ref = orc.render_streaming(flat, scene.camera, W, H, 0, cpu_frames, MB, nthreads=cores)
        dt = time.perf_counter() - t0
        cpu_baseline = {"value": round(ref["rays"] / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                        "sample": f"{cpu_frames} accumulated iterations of the same scene at {W}x{H}: {ref['rays']} rays in {dt:.2f} s; "
                                  "oracle/liboracle.so (CPU restatement; the reference has no CPU path)"}
        d = got["color"].astype(np.float64) - ref["color"].astype(np.float64)
        parity = {"against": "the CPU oracle's %d accumulated iterations at %dx%d (the frames of cpu_baseline)" % (cpu_frames, W, H),
                  "gpu_side": "the timed tracer itself (same context and schedule: %d frames per launch, %d streams), restarted" % (batch, streams),
                  "mse": float(np.mean(np.sum(d * d, axis=-1))),
                  "bit_exact": bool(all(np.array_equal(got[k], ref[k]) for k in ("color", "normal", "depth"))),
                  "live_equal": bool(got["last_live"][:MB] == [int(v) for v in ref["live"][-1][:MB]]),
                  "rays_equal": bool(got["rays"] == ref["rays"]), "tolerance_mse": 1e-4}
    return {
        "metric": "Mrays/s at 1280×720, 8 bounces (config 2)", "value": round(rays / elapsed / 1e6, 3), "unit": "Mrays/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"config 2: Cornell box (5 wall spheres + 3 balls) + two instances of a {len(flat.indices) // 3}-triangle "
                               f"displaced sphere (stand-in for bunny.obj, an LFS pointer in the reference), {W}x{H}, {MB} bounces",
                   "triangles": len(flat.indices) // 3, "instances": 2, "bvh_depth": int(bvh_depth), "resolution": [W, H],
                   "max_bounces": MB, "frames_in_flight": streams * batch, "frames_per_launch": batch,
                   "rays_per_step": round(rays / args.steps), "live_per_bounce_last_frame_rank0": last_live},
        "roofline": roofline, "parity": parity, "cpu_baseline": cpu_baseline,
    }


def run_config5(args, pkg, torch, local_rank):
    """Config 5: 1 spp interactive mode + Edge-Avoiding A-Trous denoise after every frame, 1920x1080."""
    W, H, MB = args.width or 1920, args.height or 1080, args.max_bounces
    scene, flat, bvh_depth, _ = config3_scene(pkg, W, H, args.grid)
    P = W * H
    rgba = torch.empty((H, W), dtype=torch.int32, device="cuda")

    checked = {}

    def loop(denoise, steps, warmup, events, keep=False):
        pt = pkg.PathTracer(device=local_rank, max_bounces=MB)
        pt.set_param("frames_in_flight", 4)   # a present after every iteration leaves no room for more
        pt.set_param("batch_frames", 2)
        for kv in args.param:
            name, value = kv.split("=")
            pt.set_param(name, int(value))
        pt.create_buffers((W, H), flat)
        pt.set_stream(torch.cuda.current_stream().cuda_stream)
        pt.max_iterations = 1 << 30
        for _ in range(warmup):
            pt.restart()                      # a moving camera: every frame starts a new image (app.cpp:112,130)
            pt.path_trace(scene.camera)
            if denoise:
                pt.denoise()
            pt.send_to_preview(rgba.data_ptr())
        torch.cuda.synchronize()
        pt.reset_profile()
        pt.set_profiling(time_trace_kernel=events, count_tests=False)
        rays0 = pt.stats()["rays_total"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pt.restart()
            pt.path_trace(scene.camera)
            if denoise:
                pt.denoise()
            pt.send_to_preview(rgba.data_ptr())
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        rays = pt.stats()["rays_total"] - rays0
        prof = pt.profile()
        if keep:   # what the last timed frame left on the device: iteration 0, denoised -- compared with the oracle below
            pt.set_profiling(False, False)
            for k in ("color", "normal", "depth", "final"):
                checked[k] = pt.download(k)
            checked["rgba"] = pt.send_to_preview()
            checked["last_live"] = pt.stats()["last_live"]
        pt.close()
        return el, rays, prof

    el, rays, prof = loop(True, args.steps, args.warmup, not args.no_events, keep=not args.no_cpu_baseline)
    el_nd, _, _ = loop(False, args.steps, args.warmup, False)
    passes = max(prof["denoise_passes"], 1)
    den_ms_pass = prof["denoise_ms"] / passes
    # HBM-side algorithmic bytes of one pass: colour 16 + normal/depth 16 + position 16 read, 16 written per pixel
    pass_bytes = 64 * P
    achieved = pass_bytes / (den_ms_pass * 1e-3) / 1e9 if den_ms_pass > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": "k_denoise (A-Trous pass, 5x5 taps at stride 1/2/4/8)",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": None, "pricing": "(48 B read + 16 B written) per pixel and pass", "launches": prof["denoise_passes"],
                "avg_launch_us": round(den_ms_pass * 1e3, 2), "alg_bytes_per_launch": pass_bytes}
    # what binds the kernel, from counters (tools/pmc_denoise.sh -> profiles/pmc_denoise.json; every `pixel` there is one output
    # pixel of one pass): the HBM pricing stays in achieved / peak / frac
    den_rec = None
    try:
        den_rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_denoise.json")))
    except Exception:   # noqa: BLE001
        pass
    if den_rec is not None and (W, H) == (1920, 1080):
        raw = den_rec.get("raw_means_per_launch") or {}
        lds_conf = raw.get("SQ_LDS_BANK_CONFLICT")
        lds_act = raw.get("SQ_LDS_IDX_ACTIVE") or raw.get("SQ_ACTIVE_INST_LDS")
        roofline["priced_against"] = "hbm"
        roofline["bound"] = den_rec.get("bound", "valu-issue")
        roofline["counters"] = {"valu_pipe_frac": den_rec.get("valu_pipe_frac"), "lanes_per_valu_inst": den_rec.get("lanes_per_valu_inst"),
                                "valu_inst_per_pixel_and_pass": den_rec.get("valu_inst_per_ray"),
                                "lds_inst_per_pixel_and_pass": den_rec.get("lds_inst_per_ray"),
                                "lds_bank_conflict_cycles_over_lds_active": round(lds_conf / lds_act, 4) if lds_conf is not None and lds_act else None,
                                "wave_occupancy_frac": den_rec.get("wave_occupancy_frac"), "l2_hit_frac": den_rec.get("l2_hit_frac"),
                                "mean_launch_us_serialised_by_the_profiler": den_rec.get("mean_launch_us_serialised_by_the_profiler"),
                                "source": den_rec.get("source")}
        roofline["bound_note"] = den_rec.get("bound_note")
    cpu_baseline = parity = None
    if not args.no_cpu_baseline:
        # the same frame on the host: one iteration, the four A-Trous passes, the tonemap -- timed (cpu_baseline) and compared
        # with what the timed loop's LAST frame left on the device (every frame of the loop is iteration 0 of the same camera)
        orc = graft.load_oracle()
        cores = max(1, min(args.cpu_threads, orc.lib().orc_hardware_threads()))
        t0 = time.perf_counter()
        ref = orc.render_streaming(flat, scene.camera, W, H, 0, 1, MB, nthreads=cores)
        t1 = time.perf_counter()
        den, touched = orc.denoise(scene.camera, W, H, ref["color"], ref["normal"], ref["depth"], nthreads=cores)
        t2 = time.perf_counter()
        img = orc.preview(den, W, H, 0)
        t3 = time.perf_counter()
        cpu_baseline = {"value": round((t3 - t0) * 1e3, 1), "unit": "ms/frame", "cores": cores, "kind": "port",
                        "sample": f"ONE frame of the same workload at {W}x{H}: 1 spp ({ref['rays']} rays, {t1 - t0:.2f} s) + 4-pass A-Trous "
                                  f"({t2 - t1:.2f} s) + tonemap ({t3 - t2:.2f} s); oracle/liboracle.so (CPU restatement; the reference has no CPU path)"}
        ok = ~touched     # pixels that depend on the reference's out-of-bounds taps are excluded
        den_err = float(np.max(np.abs(checked["final"][ok] - den[ok])))
        lsb = int(np.max(np.abs(checked["rgba"][ok].astype(np.int32) - img[ok].astype(np.int32))))
        parity = {"against": "the CPU oracle's frame: 1 iteration, orc_denoise, orc_preview (the frame of cpu_baseline)",
                  "gpu_side": "the last frame of the timed loop itself",
                  "bit_exact": bool(all(np.array_equal(checked[k], ref[k]) for k in ("color", "normal", "depth"))),
                  "live_equal": bool(checked["last_live"][:MB] == [int(v) for v in ref["live"][-1][:MB]]),
                  "denoised_max_abs_err": den_err, "tolerance_denoised": 1e-5, "denoised_within_tolerance": bool(den_err <= 1e-5),
                  "rgba_max_lsb": lsb, "tolerance_rgba_lsb": 1, "pixels_excluded_oob_taps": int(touched.sum()),
                  "note": "bit_exact: the undenoised colour / normal / depth planes; expf (denoiser) and powf (tonemap) use the platform "
                          "libraries on both sides, hence tolerances there"}
    return {
        "metric": "ms per frame: 1 spp + A-Trous denoise + present at 1920×1080, 8 bounces (config 5)",
        "value": round(el / args.steps * 1e3, 4), "unit": "ms/frame", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"config 5: config 3's scene ({len(flat.indices) // 3} triangles), {W}x{H}, {MB} bounces, every frame = "
                               "restart + 1 spp + 4-pass A-Trous (filter_size 10, .45/.30/.25) + present; G-buffer in HBM",
                   "resolution": [W, H], "max_bounces": MB, "rays_per_step": round(rays / args.steps),
                   "ms_per_frame_without_denoise": round(el_nd / args.steps * 1e3, 4),
                   "denoise_ms_per_frame_kernel_time": round(prof["denoise_ms"] / args.steps, 4),
                   "denoise_ms_per_pass": round(den_ms_pass, 4), "mrays_per_s": round(rays / el / 1e6, 1)},
        "roofline": roofline, "parity": parity, "cpu_baseline": cpu_baseline,
    }


def reference_order_gpu(args, pkg, torch, local_rank, frames=3):
    """The reference's ALGORITHM restated in HIP on this GPU -- not the CUDA build, which cannot run here: one thread per path
    walking the 32-byte two-child nodes in the reference's order without culling (trace_variant 0 = k_trace,
    path_tracer.cu:36-76), intersect -> material -> partition as three kernels per bounce (fused_shade 0), every ray in every
    launch (filter_rays 0, beam 0), one frame at a time (frames_in_flight 1), and the live count read back by the host after
    every bounce (ptc_read_live_count: the reference's thrust result, path_tracer.cu:454-457).  Same scene, resolution,
    bounce cap and ray definition as the headline; the denominator for "x times the reference's way of doing it" on the
    same silicon."""
    W, H, MB = args.width or 1920, args.height or 1080, args.max_bounces
    scene, flat, _, _ = config3_scene(pkg, W, H, args.grid)

    def tracer(reference_order):
        pt = pkg.PathTracer(device=local_rank, max_bounces=MB)
        if reference_order:
            pt.set_param("frames_in_flight", 1)
            pt.set_param("batch_frames", 1)
            pt.set_trace_variant(0)
            for name in ("fused_shade", "filter_rays", "beam"):
                pt.set_param(name, 0)
        pt.create_buffers((W, H), flat)
        pt.set_stream(torch.cuda.current_stream().cuda_stream)
        pt.max_iterations = 1 << 30
        return pt

    def frame(pt):
        pt.trace_begin(scene.camera)
        n, rays = W * H, 0
        for b in range(MB):
            if n == 0:                     # path_tracer.cu:418: the loop ends with the last live path
                break
            rays += n
            pt.trace_bounce(b)
            n = pt.read_live_count(b + 1)  # host read-back per bounce, as the reference's partition result
        pt.trace_end()
        return rays

    pt = tracer(True)
    frame(pt)
    pt.synchronize()
    t0 = time.perf_counter()
    rays = sum(frame(pt) for _ in range(frames))
    pt.synchronize()
    dt = time.perf_counter() - t0
    # its frames against the default schedule's (which the parity leg of this line holds against the oracle)
    pt.restart()
    for _ in range(2):
        frame(pt)
    got = {k: pt.download(k) for k in ("color", "normal", "depth")}
    pt.close()
    with tracer(False) as dp:
        for _ in range(2):
            dp.path_trace(scene.camera)
        want = {k: dp.download(k) for k in ("color", "normal", "depth")}
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "frames": frames, "ms_per_frame": round(dt / frames * 1e3, 3),
            "rays": int(rays),
            "what": "the reference's algorithm restated in HIP on this GPU -- NOT the CUDA build: one thread per path, 32-byte "
                    "two-child nodes in the reference's order, no culling (trace_variant 0), three kernels per bounce "
                    "(fused_shade 0), every ray in every launch (filter_rays 0, beam 0), one frame in flight, live count read "
                    "back by the host after every bounce (path_tracer.cu:36-76,413-458)",
            "equals_default_schedule_bit_for_bit": bool(all(np.array_equal(got[k], want[k]) for k in got))}


def extra_legs(args, pkg, torch, dist, local_rank, line):
    """What only builder-run logs showed until round 5, attached to the driver's line (N = 1, config 3; --no-extras skips it):
    other_configs (config 2 and config 5 with their parity), rank_share (rank 0's rows of a 2 / 4 / 8-way split traced alone
    on this GPU with the schedule such a run would pick, parity against the oracle's rendering of those rows),
    scaling_projection (what rank_share predicts for --gpus N before any gather or barrier) and reference_order_gpu."""
    import copy

    def sub(**over):
        a = copy.copy(args)
        a.param = list(args.param)
        for k, v in over.items():
            setattr(a, k, v)
        return a

    t_begin = time.perf_counter()
    out = {}
    shares = {}
    whole_ms = line["ms_per_step"] * args.steps
    for n in (2, 4, 8):
        l = run_config3(sub(share_of=n, no_extras=True, cpu_frames=min(args.cpu_frames, 2)), pkg, torch, dist, 1, 0, local_rank, False)
        ms = l["ms_per_step"] * args.steps
        shares[str(n)] = {"ms_per_%d_steps" % args.steps: round(ms, 4), "whole_frame_over_share": round(whole_ms / ms, 3),
                          "mrays_per_s_of_the_rank": l["value"], "frames_per_launch": l["config"]["frames_per_launch"],
                          "frames_in_flight": l["config"]["frames_in_flight"],
                          "parity_bit_exact": (l.get("parity") or {}).get("bit_exact")}
    out["rank_share"] = {"what": "rank 0's rows of an N-way split (blocks of %d rows round-robin) traced alone on this GPU with the "
                                 "schedule a --gpus N run picks; no gather, no barrier; parity = those rows against the CPU oracle's "
                                 "rendering of that rank" % BLOCK_ROWS,
                         "whole_frame_ms_per_%d_steps" % args.steps: round(whole_ms, 4), "by_ranks": shares}
    rays_job = line["config"]["rays_per_step"] * args.steps
    out["scaling_projection"] = {
        "what": "a one-GPU projection of --gpus N, to read a measured SCALE value against: the job's rays / rank 0's time for its "
                "share (interleaved blocks make the ranks' shares alike to a few percent); the present-time gather (%.1f MB into rank 0, "
                "tens of microseconds over xGMI) and two barriers come on top" % (1920 * 1080 * 12 * 7 / 8 / 1e6),
        "by_gpus": {k: {"predicted_mrays_per_s": round(rays_job / (v["ms_per_%d_steps" % args.steps] * 1e-3) / 1e6, 1),
                        "predicted_speedup": v["whole_frame_over_share"],
                        "schedule": "%d x %d frames, %d persistent wavefronts" % (v["frames_in_flight"] // max(v["frames_per_launch"], 1),
                                                                                 v["frames_per_launch"], 5120 if int(k) <= 2 else 2560)}
                    for k, v in shares.items()}}
    out["reference_order_gpu"] = reference_order_gpu(args, pkg, torch, local_rank)
    ref = out["reference_order_gpu"]["value"]
    out["vs_reference_order_gpu"] = {"ratio": round(line["value"] / ref, 2) if ref else None,
                                     "note": "value / reference_order_gpu.value: this library's schedule against the reference's "
                                             "algorithm restated in HIP on the same GPU (not the CUDA build)"}
    c2 = run_config2(sub(config=2, steps=64, warmup=32, width=0, height=0, cpu_frames=min(args.cpu_frames, 2), batch_frames=0, streams=0,
                         param=[]), pkg, torch, local_rank)
    c5 = run_config5(sub(config=5, steps=32, warmup=8, param=[]), pkg, torch, local_rank)
    out["other_configs"] = {
        "config2": {"value": c2["value"], "unit": c2["unit"], "steps": 64, "ms_per_step": c2["ms_per_step"],
                    "frame_level_frac": c2["roofline"].get("frame_level_frac"), "kernel_frac": c2["roofline"]["frac"],
                    "parity": {k: (c2.get("parity") or {}).get(k) for k in ("bit_exact", "live_equal", "rays_equal", "mse")},
                    "workload": c2["config"]["workload"]},
        "config5": {"ms_per_frame": c5["value"], "unit": c5["unit"], "steps": 32,
                    "denoise_ms_per_pass": c5["config"]["denoise_ms_per_pass"], "denoise_bound": c5["roofline"]["bound"],
                    "ms_per_frame_without_denoise": c5["config"]["ms_per_frame_without_denoise"],
                    "parity": {k: (c5.get("parity") or {}).get(k) for k in ("bit_exact", "live_equal", "denoised_within_tolerance",
                                                                              "denoised_max_abs_err", "rgba_max_lsb")},
                    "workload": c5["config"]["workload"]}}
    out["extra_legs_s"] = round(time.perf_counter() - t_begin, 2)
    return out


def count_gpus_without_touching_them():
    """GPUs this process could use, counted WITHOUT starting a GPU runtime here (the parent of the ranks must stay clean: it
    spawns the launcher).  A short-lived child asks the library (ptc_device_count: hipGetDeviceCount, which honours
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES); if that child cannot run, the KFD topology is read instead (nodes with
    SIMDs are GPUs)."""
    import subprocess
    code = ("import ctypes, sys; lib = ctypes.CDLL(sys.argv[1]); n = ctypes.c_int(0); "
            "rc = lib.ptc_device_count(ctypes.byref(n)); print(n.value if rc == 0 else 0)")
    try:
        out = subprocess.run([sys.executable, "-c", code, os.path.join(ROOT, "cuda-path-tracer_amd", "libptcore.so")],
                             capture_output=True, text=True, timeout=120)
        if out.returncode == 0:
            return int(out.stdout.strip().splitlines()[-1])
    except Exception:   # noqa: BLE001
        pass
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            for line in open(os.path.join(base, node, "properties")):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except OSError:
        pass
    return n


def spawn_ranks(args):
    """`bench.py --gpus N` without a launcher: run the same command line under torch.distributed.run, N ranks on this
    node, as a CHILD process (this one never initialises HIP: the devices are counted by a short-lived child), and hand
    its exit code on.  The ranks print the JSON line themselves (rank 0)."""
    import socket
    import subprocess
    have = count_gpus_without_touching_them()
    if have < args.gpus and not args.rehearse_on_one_gpu:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but this node has {have} GPU(s); pass --rehearse-on-one-gpu to run "
                         f"the {args.gpus}-rank code path on one device (not a scaling measurement)\n")
        return 2
    if args.rehearse_on_one_gpu and have < 1:
        sys.stderr.write("bench.py: no GPU\n")
        return 2
    with socket.socket() as sock:   # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1 and args.config != 3:
        raise SystemExit("--config 2 and 5 are single-GPU workloads")
    rehearse = args.rehearse_on_one_gpu
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    ctl = None
    if world > 1:
        import datetime
        long_wait = datetime.timedelta(minutes=30)   # ranks wait at a barrier while rank 0 renders its oracle rows (seconds; never minutes)
        if rehearse:
            dist.init_process_group(backend="gloo", timeout=long_wait)
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=long_wait)
            # control plane: handles, flags, per-rank figures and the barriers outside the data path travel over gloo, so that
            # a failing RCCL call can be reported and worked around instead of ending (or hanging) the run
            ctl = dist.new_group(backend="gloo", timeout=long_wait)

    pkg = graft.load_package()
    if args.config == 3:
        line = run_config3(args, pkg, torch, dist, world, rank, local_rank, rehearse, ctl)
        if line is not None and world == 1 and args.share_of <= 1 and not args.no_extras:
            line.update(extra_legs(args, pkg, torch, dist, local_rank, line))
    elif args.config == 2:
        line = run_config2(args, pkg, torch, local_rank)
    else:
        line = run_config5(args, pkg, torch, local_rank)
    if rank == 0 and line is not None:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier(group=ctl)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
