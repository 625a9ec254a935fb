"""Multi-GPU partition logic on the CPU: row bands + the per-bounce live-count exchange that keeps the
reference's GLOBAL slot numbering (material RNG key, path_tracer.cu:297-301), over torch.distributed `gloo`
with world_size 2, with the CPU oracle standing in for each rank's GPU.  The union of the bands must be the
single-process full-frame render, bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_rows_and_slot_base(pkg):
    b = pkg.bands
    for h, world in ((1080, 1), (1080, 8), (7, 3), (5, 5), (1081, 4)):
        rows = b.split_rows(h, world)
        assert rows[0][0] == 0 and rows[-1][1] == h and len(rows) == world
        assert all(r0 < r1 for r0, r1 in rows) and all(a[1] == c[0] for a, c in zip(rows, rows[1:]))
        assert max(r1 - r0 for r0, r1 in rows) - min(r1 - r0 for r0, r1 in rows) <= 1
    assert b.slot_base_from_counts([5, 7, 9], 0) == 0 and b.slot_base_from_counts([5, 7, 9], 2) == 12


def test_bands_with_exchange_equal_full_frame_single_process(pkg, orc):
    """3 bands simulated in one process (lock-step over bounces is not needed: counts of a bounce depend only on
    earlier bounces, so the exchange can be answered from a first pass that records every band's counts)."""
    scene = pkg.scenes.cornell_bunny((48, 30), n_lat=8, n_lon=16)
    flat = scene.build_scene()
    w, h, mb = 48, 30, 6
    full = orc.render_streaming(flat, scene.camera, w, h, 0, 1, mb)
    rows = pkg.bands.split_rows(h, 3)
    # pass 1: per-band live counts with global numbering require lower bands' counts -> iterate rank by rank:
    # rank r's base only needs counts of ranks < r, and those ranks' renders do not depend on rank r.
    counts, bands = [], []
    for r, rr in enumerate(rows):
        def exchange(bounce, mine, r=r):
            return rr[0] * w if bounce == 0 else sum(int(c[bounce]) for c in counts[:r])
        out = orc.render_band(flat, scene.camera, w, h, rr, 0, mb, exchange=exchange)
        counts.append(out["live"])
        bands.append(out)
    for k in ("color", "normal", "depth"):
        assert np.array_equal(pkg.bands.assemble([b[k] for b in bands]), full[k])
    assert sum(b["rays"] for b in bands) == full["rays"]
    # band-local numbering is a different (but valid) image
    local = [orc.render_band(flat, scene.camera, w, h, rr, 0, mb) for rr in rows]
    assert np.array_equal(local[0]["color"], bands[0]["color"])          # band 0 has base 0 either way
    assert not np.array_equal(local[1]["color"], bands[1]["color"])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg, orc = graft.load_package(), graft.load_oracle()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = pkg.scenes.cornell_bunny((48, 30), n_lat=8, n_lon=16)
    flat = scene.build_scene()
    w, h, mb, iters = 48, 30, 6, 2
    rows = pkg.bands.split_rows(h, world)[rank]

    def exchange(bounce, mine):
        mine_t = torch.tensor([mine], dtype=torch.int64)
        all_t = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(all_t, mine_t)
        return pkg.bands.slot_base_from_counts([int(t.item()) for t in all_t], rank)

    prev = None
    for it in range(iters):
        prev = orc.render_band(flat, scene.camera, w, h, rows, it, mb, exchange=exchange, prev=prev, nthreads=1)
    # gather of per-band radiance at present time
    mine = torch.from_numpy(prev["color"].copy())
    shapes = [(r1 - r0, w, 3) for r0, r1 in pkg.bands.split_rows(h, world)]
    gathered = [torch.zeros(s, dtype=torch.float32) for s in shapes] if rank == 0 else None
    dist.gather(mine, gathered, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "frame.npy"), pkg.bands.assemble([g.numpy() for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_gloo(pkg, orc, tmp_path):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "frame.npy")
    scene = pkg.scenes.cornell_bunny((48, 30), n_lat=8, n_lon=16)
    full = orc.render_streaming(scene.build_scene(), scene.camera, 48, 30, 0, 2, 6)
    assert np.array_equal(got, full["color"])
