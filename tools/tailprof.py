#!/usr/bin/env python3
"""Where does the end of a traversal launch go?  Needs a -DPT_TAILPROF build of the library:

    make -C cuda-path-tracer_amd/csrc variant TAG=tail EXTRA=-DPT_TAILPROF
    PTCORE_LIB=$PWD/cuda-path-tracer_amd/libptcore_w_tail.so python3 tools/tailprof.py [--share-of N] [--frames F] [--fif K]

Every wavefront of a k_traverse4 launch records (100 MHz wall clock) when it started, when it first found the launch's
feed exhausted and when it left the loop, plus its loop iterations and split rounds.  One batch of F frames of config 3
is traced, then the table of its eight launches is printed: the span of the launch, the moment the first / median / last
wavefront ran out of rays to fetch, and how the wavefronts' exits are spread after that."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import __graft_entry__ as graft  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--share-of", type=int, default=1)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--fif", type=int, default=0, help="frames in flight (0 = --frames: one batch)")
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--param", action="append", default=[])
    ap.add_argument("--by-xcd", action="store_true", help="per XCD (workgroup id mod 8): when its wavefronts ran out of rays and left")
    ap.add_argument("--repeat", type=int, default=3, help="batches traced before the one that is read")
    args = ap.parse_args()
    pkg = graft.load_package()
    lib = pkg.lib()
    W, H, MB = 1920, 1080, 8
    scene = pkg.scenes.heightfield_scene((W, H), nx=1001, nz=501)
    flat = scene.build_scene()
    pt = pkg.PathTracer(device=0, max_bounces=MB)
    pt.set_param("frames_in_flight", args.fif or args.frames)
    pt.set_param("batch_frames", min(args.frames, 32))
    pt.set_param("traverse_waves", args.waves or (5120 if args.share_of <= 2 else 2560))
    for kv in args.param:
        k, v = kv.split("=")
        pt.set_param(k, int(v))
    pt.create_buffers((W, H), flat)
    if args.share_of > 1:
        pt.set_interleave(0, args.share_of, 8)
    pt.max_iterations = 1 << 30
    lib.ptc_debug_tailprof.restype = C.c_int
    lib.ptc_debug_tailprof.argtypes = [C.c_void_p, C.c_size_t]
    for rep in range(args.repeat + 1):
        if rep == args.repeat:
            assert lib.ptc_debug_tailprof(None, 0) == 0     # clear: only the last batch is read
        for _ in range(args.frames):
            pt.path_trace(scene.camera)
        pt.synchronize()
    buf = np.zeros((16, 8192, 8), dtype=np.uint64)
    assert lib.ptc_debug_tailprof(buf.ctypes.data, buf.nbytes) == 0
    live = pt.stats()["last_live"]
    pt.close()
    print(f"share 1/{args.share_of}, {args.frames} frames per launch; times in us from the first wavefront's start")
    print("bounce  waves  rays(last frame)  span   first_exh  med_exh  last_exh | exits after first_exh: p50   p90   p99   last | iters mean  max | "
          "own tail (exit - own exhaustion): us p50 p90 max | iterations p50 p90 max | us/iter before, in the tail | lanes busy at exhaustion p50 | waves that split")
    for b in range(MB):
        rec = buf[b]
        ran = rec[:, 2] != 0
        if not ran.any():
            continue
        r = rec[ran]
        t0 = r[:, 0].min()
        start = (r[:, 0] - t0) / 100.0
        exh = np.where(r[:, 1] != 0, (r[:, 1].astype(np.int64) - np.int64(t0)) / 100.0, np.nan)
        end = (r[:, 2] - t0) / 100.0
        iters = (r[:, 3] >> np.uint64(40)).astype(np.int64)
        iters_exh = ((r[:, 3] >> np.uint64(16)) & np.uint64(0xffffff)).astype(np.int64)
        lanes_exh = ((r[:, 3] >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64)
        splits = (r[:, 3] & np.uint64(0xff)).astype(np.int64)
        own = end - exh
        tail_it = iters - iters_exh
        ok = ~np.isnan(exh)
        fe = np.nanmin(exh)
        after = end - fe
        print(f"{b:5d} {ran.sum():6d} {live[b]:12d}   {end.max():7.1f} {fe:9.1f} {np.nanmedian(exh):8.1f} {np.nanmax(exh):8.1f} |"
              f" {np.percentile(after, 50):21.1f} {np.percentile(after, 90):5.1f} {np.percentile(after, 99):5.1f} {after.max():6.1f} |"
              f" {iters.mean():8.1f} {iters.max():5d} | {np.nanpercentile(own, 50):6.1f} {np.nanpercentile(own, 90):6.1f} {np.nanmax(own):6.1f} |"
              f" {np.percentile(tail_it[ok], 50):5.0f} {np.percentile(tail_it[ok], 90):5.0f} {tail_it[ok].max():5d} |"
              f" {np.nansum(exh - start) / max(iters_exh[ok].sum(), 1):5.2f} {np.nansum(own) / max(tail_it[ok].sum(), 1):5.2f} |"
              f" {np.percentile(lanes_exh[ok], 50):3.0f} | {(splits > 0).sum():6d}   late starts (>5us): {(start > 5).sum()}")
        if args.by_xcd:
            blk = np.nonzero(ran)[0]
            for x in range(8):
                m = (blk % 8) == x
                print(f"        xcd {x}: waves {m.sum():5d}  exhaustion p10 {np.nanpercentile(exh[m], 10):7.1f} p50 {np.nanpercentile(exh[m], 50):7.1f} p90 {np.nanpercentile(exh[m], 90):7.1f}"
                      f" | exit p50 {np.percentile(end[m], 50):7.1f} p90 {np.percentile(end[m], 90):7.1f} last {end[m].max():7.1f} | iterations mean {iters[m].mean():7.1f} | tail iterations p50 {np.percentile(tail_it[m & ok], 50):4.0f}"
                      f" | us per tail iteration {np.nansum(own[m]) / max(tail_it[m & ok].sum(), 1):5.2f}")
        ti = max(int(tail_it[ok].sum()), 1)
        print(f"        shader-clock cycles per tail iteration: retire {r[ok, 4].sum() / ti:7.0f}  split {r[ok, 5].sum() / ti:7.0f}  step {r[ok, 6].sum() / ti:7.0f}"
              f"   active lanes per tail iteration {(r[ok, 7] & np.uint64(0xffffff)).sum() / ti:5.1f}"
              f"   rays handed over: given {int(((r[:, 7] >> np.uint64(24)) & np.uint64(0xfff)).sum())} taken {int(((r[:, 7] >> np.uint64(36)) & np.uint64(0xfff)).sum())}"
              f" by {int((((r[:, 7] >> np.uint64(36)) & np.uint64(0xfff)) > 0).sum())} wavefronts")


if __name__ == "__main__":
    main()
