"""The multi-GPU split that bench.py --gpus N and hip_pt --gpus N run -- interleaved blocks of 8 rows, paths numbered per
rank, "slot_offset" = rank * W * H (the reference keys the material RNG on the compacted slot, path_tracer.cu:297-301, so
the numbering decides every random number) -- against the CPU oracle's rendering of the same rank
(orc_render_streaming_interleaved), bit for bit: colour, first-hit normal and depth, ray totals and per-bounce live
counts of every rank of worlds 2, 3 and 8 (contexts one after another on the one GPU of the test box), the assembled
frames against tests/golden/interleaved.npz.  Round 3 checked this mode only for an equal G-buffer and a close mean."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


@pytest.fixture(scope="module")
def mk():
    import make_golden
    return make_golden


def rank_frames(pkg, scene, flat, w, h, rank, world, block, iters, mb, params=(), variant=None, one_by_one=False):
    lives = []
    with pkg.PathTracer(device=0, max_bounces=mb) as pt:
        for k, v in params:
            pt.set_param(k, v)
        pt.create_buffers((w, h), flat)
        if variant is not None:
            pt.set_trace_variant(variant)
        pt.set_interleave(rank, world, block)
        pt.set_param("slot_offset", rank * w * h)
        pt.max_iterations = iters
        for _ in range(iters):
            pt.path_trace(scene.camera)
            if one_by_one:
                lives.append(pt.stats()["last_live"])
        out = {k: pt.download(k) for k in ("color", "normal", "depth")}
        out["stats"] = pt.stats()
        out["lives"] = lives
    return out


@pytest.mark.parametrize("world", [2, 3, 8])
def test_every_rank_against_the_oracle_and_the_fixture(pkg, orc, mk, golden_dir, world):
    c = mk.INTERLEAVED
    scene = mk.interleaved_scene()
    flat = scene.build_scene()
    w, h, block, mb, iters = c["w"], c["h"], c["block_rows"], c["max_bounces"], c["iterations"]
    gold = np.load(os.path.join(golden_dir, "interleaved.npz"))
    sh = orc.SceneHandle(flat)
    parts = []
    for rank in range(world):
        want = orc.render_interleaved(flat, scene.camera, w, h, rank, world, block, rank * w * h, 0, iters, mb, scene_handle=sh)
        got = rank_frames(pkg, scene, flat, w, h, rank, world, block, iters, mb)            # the default schedule, batched
        for k in ("color", "normal", "depth"):
            assert got[k].shape == want[k].shape and np.array_equal(got[k], want[k]), (world, rank, k)
        assert got["stats"]["rays_total"] == want["rays"] == int(gold[f"w{world}_rays"][rank])
        assert got["stats"]["last_live"] == [int(x) for x in want["live"][-1]]
        # one frame at a time: the live counts of every iteration
        serial = rank_frames(pkg, scene, flat, w, h, rank, world, block, iters, mb, params=(("frames_in_flight", 1),),
                             one_by_one=True)
        assert np.array_equal(serial["color"], want["color"])
        assert np.array_equal(np.array(serial["lives"], dtype=np.uint32), want["live"]), (world, rank)
        assert np.array_equal(want["live"], gold[f"w{world}_live"][rank])
        parts.append(got)
    for k in ("color", "normal", "depth"):
        frame = pkg.bands.assemble_interleaved([p[k] for p in parts], h, world, block)
        assert np.array_equal(frame, gold[f"w{world}_{k}"]), (world, k)


def test_other_schedules_and_a_frame_whose_height_is_no_multiple_of_the_block(pkg, orc):
    """Reference-order kernel, three-kernel end of a bounce, unfiltered rays, a mesh-instance scene; 100 x 67 pixels in
    blocks of 8 over 3 ranks: the last block has 3 rows."""
    scene = pkg.scenes.cornell_bunny((100, 67), n_lat=12, n_lon=24)
    flat = scene.build_scene()
    w, h, world, block, mb, iters = 100, 67, 3, 8, 6, 2
    sh = orc.SceneHandle(flat)
    for rank in range(world):
        want = orc.render_interleaved(flat, scene.camera, w, h, rank, world, block, rank * w * h, 0, iters, mb, scene_handle=sh)
        for kw in (dict(), dict(variant=0, params=(("frames_in_flight", 1),)), dict(params=(("fused_shade", 0),)),
                   dict(params=(("filter_rays", 0), ("batch_frames", 2), ("frames_in_flight", 4)))):
            got = rank_frames(pkg, scene, flat, w, h, rank, world, block, iters, mb, **kw)
            for k in ("color", "normal", "depth"):
                assert np.array_equal(got[k], want[k]), (rank, kw, k)
            assert got["stats"]["rays_total"] == want["rays"]
