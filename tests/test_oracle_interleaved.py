"""The oracle's rendering of ONE RANK of the benchmarked multi-GPU split (interleaved row blocks, paths numbered per rank,
"slot_offset" = rank * W * H; orc_render_streaming_interleaved), pinned on the CPU three independent ways before the GPU
tests (tests/test_gpu_interleaved.py) compare ptc_set_interleave with it:

  * one rank of one, offset 0, IS the single-GPU frame (orc_render_streaming);
  * when every rank owns exactly one block, a rank is a contiguous band: the older band function
    (orc_render_streaming_band, its slot base supplied per bounce by a callback) must give the same rows;
  * whatever the split, the first-hit G-buffer of the assembled frame is the single-GPU one (ray generation is keyed on
    the frame's pixel index, ray_gen.cu:17-22; only the material RNG sees the numbering, path_tracer.cu:297-301), and
    rank 0 -- offset 0 -- keeps the single-GPU streams for its primary-ray slots of its first row only (slot == pixel).

The reference has no multi-GPU mode: these results are parity unpinned by construction."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


@pytest.fixture(scope="module")
def mk():
    import make_golden
    return make_golden


def test_row_counts_agree_with_the_host_side(pkg, orc):
    for h, world, block in ((72, 2, 8), (72, 3, 8), (72, 8, 8), (1080, 8, 8), (1080, 3, 8), (67, 4, 5), (5, 2, 8), (9, 8, 1)):
        rows = pkg.bands.interleaved_rows(h, world, block)
        assert sorted(y for r in rows for y in r) == list(range(h))
        for r in range(world):
            assert orc.lib().orc_interleaved_rows(h, r, world, block) == len(rows[r])
    assert orc.lib().orc_interleaved_rows(72, 2, 2, 8) == 0 and orc.lib().orc_interleaved_rows(72, 0, 0, 8) == 0


def test_one_rank_of_one_is_the_single_gpu_frame(pkg, orc, mk):
    scene = mk.interleaved_scene()
    flat = scene.build_scene()
    full = orc.render_streaming(flat, scene.camera, 48, 72, 0, 2, 6)
    one = orc.render_interleaved(flat, scene.camera, 48, 72, 0, 1, 8, 0, 0, 2, 6)
    for k in ("color", "normal", "depth", "live"):
        assert np.array_equal(one[k], full[k]), k
    assert one["rays"] == full["rays"]
    # iterations accumulate like the full frame's (running means carried in `prev`)
    a = orc.render_interleaved(flat, scene.camera, 48, 72, 0, 1, 8, 0, 0, 1, 6)
    b = orc.render_interleaved(flat, scene.camera, 48, 72, 0, 1, 8, 0, 1, 1, 6, prev=a)
    assert np.array_equal(b["color"], full["color"])


def test_one_block_per_rank_is_a_contiguous_band(pkg, orc, mk):
    scene = mk.interleaved_scene()
    flat = scene.build_scene()
    w, h, world, mb = 48, 72, 3, 6
    block = h // world
    for rank in range(world):
        offset = rank * w * h
        got = orc.render_interleaved(flat, scene.camera, w, h, rank, world, block, offset, 2, 1, mb)
        # iteration 2 on its own: the band function renders one iteration into running means, so give both zeros as the
        # previous state and compare the raw sample through the same (old * 2 + new) / 3
        want = orc.render_band(flat, scene.camera, w, h, (rank * block, (rank + 1) * block), 2, mb,
                               exchange=lambda bounce, mine, offset=offset: offset)
        for k in ("color", "normal", "depth"):
            assert np.array_equal(got[k], want[k]), (rank, k)
        assert np.array_equal(got["live"][0], want["live"]) and got["rays"] == want["rays"]


def test_assembled_gbuffer_is_the_single_gpu_one_and_the_fixture_reproduces(pkg, orc, mk, golden_dir):
    c = mk.INTERLEAVED
    scene = mk.interleaved_scene()
    flat = scene.build_scene()
    w, h, block, mb, iters = c["w"], c["h"], c["block_rows"], c["max_bounces"], c["iterations"]
    full = orc.render_streaming(flat, scene.camera, w, h, 0, iters, mb)
    gold = np.load(os.path.join(golden_dir, "interleaved.npz"))
    for world in c["worlds"]:
        parts = [orc.render_interleaved(flat, scene.camera, w, h, r, world, block, r * w * h, 0, iters, mb)
                 for r in range(world)]
        frame = {k: pkg.bands.assemble_interleaved([p[k] for p in parts], h, world, block) for k in ("color", "normal", "depth")}
        assert np.array_equal(frame["normal"], full["normal"]) and np.array_equal(frame["depth"], full["depth"])
        assert not np.array_equal(frame["color"], full["color"])            # another noise realisation ...
        assert abs(float(frame["color"].mean()) - float(full["color"].mean())) < 0.02   # ... of the same image
        # every rank starts every iteration with all its pixels; together they start with the frame's
        live = np.stack([p["live"] for p in parts])
        assert int(live[:, :, 0].sum()) == iters * w * h
        assert np.array_equal(live[:, :, 1].sum(axis=0), full["live"][:, 1])    # bounce 0 hits do not depend on the RNG
        for k in ("color", "normal", "depth"):
            assert np.array_equal(frame[k], gold[f"w{world}_{k}"]), (world, k)
        assert np.array_equal(live, gold[f"w{world}_live"])
        assert [p["rays"] for p in parts] == [int(x) for x in gold[f"w{world}_rays"]]
    # the ranks' streams are kept apart by the offset: without it rank 1 of 2 would replay rank 0's draws
    same = orc.render_interleaved(flat, scene.camera, w, h, 1, 2, block, 0, 0, 1, mb)
    apart = orc.render_interleaved(flat, scene.camera, w, h, 1, 2, block, w * h, 0, 1, mb)
    assert not np.array_equal(same["color"], apart["color"])
