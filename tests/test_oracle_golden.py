"""The oracle reproduces the committed golden frames bit for bit (tests/golden/frames.npz, written by
tests/golden/make_golden.py).  The reference has no golden images or result tests for this path, so these
pin the oracle against regressions, not against the CUDA renderer (parity unpinned there)."""
import importlib.util
import os

import numpy as np
import pytest


def _golden_scenes(golden_dir):
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(golden_dir, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.golden_scenes()


@pytest.fixture(scope="module")
def frames(golden_dir):
    return np.load(os.path.join(golden_dir, "frames.npz"))


@pytest.mark.parametrize("name", ["spheres", "mesh", "heightfield"])
@pytest.mark.parametrize("mb", [4, 8, 50])
def test_streaming_frames(orc, golden_dir, frames, name, mb):
    scene, w, h = _golden_scenes(golden_dir)[name]
    flat = scene.build_scene()
    r = orc.render_streaming(flat, scene.camera, w, h, 0, 4, mb, nthreads=3)
    for k in ("color", "normal", "depth", "live"):
        assert np.array_equal(r[k], frames[f"{name}_mb{mb}_{k}"]), k
    assert r["rays"] == int(frames[f"{name}_mb{mb}_rays"][0]) == int(r["live"].sum())
    # thread count must not matter
    r1 = orc.render_streaming(flat, scene.camera, w, h, 0, 1, mb, nthreads=1)
    assert np.array_equal(r1["color"], frames[f"{name}_mb{mb}_color_it0"])


def test_streaming_properties(orc, golden_dir, frames):
    scene, w, h = _golden_scenes(golden_dir)["mesh"]
    flat = scene.build_scene()
    full = orc.render_streaming(flat, scene.camera, w, h, 0, 4, 8)
    # running mean: continuing from the state after 2 iterations gives the 4-iteration result
    first = orc.render_streaming(flat, scene.camera, w, h, 0, 2, 8)
    cont = orc.render_streaming(flat, scene.camera, w, h, 2, 2, 8, prev=first)
    for k in ("color", "normal", "depth"):
        assert np.array_equal(cont[k], full[k])
    # live counts never grow, bounce 0 sees every pixel
    assert np.all(full["live"][:, 0] == w * h) and np.all(np.diff(full["live"].astype(np.int64), axis=1) <= 0)
    # sky-lit scene: every value finite, radiance within [0, 1]
    assert np.isfinite(full["color"]).all() and full["color"].min() >= 0 and full["color"].max() <= 1.0 + 1e-6


def test_megakernel_frames(orc, golden_dir, frames):
    for name, (scene, w, h) in _golden_scenes(golden_dir).items():
        flat = scene.build_scene()
        m = orc.render_megakernel(flat, scene.camera, w, h, 0, 2, 8)
        assert np.array_equal(m["color"], frames[f"{name}_mega_color"])
        assert m["rays"] == int(frames[f"{name}_mega_rays"][0])


def test_denoise_and_preview(orc, pkg, frames):
    scene = pkg.scenes.heightfield_scene((64, 64), nx=33, nz=17)
    den, touched = orc.denoise(scene.camera, 64, 64, frames["denoise_in_color"], frames["denoise_in_normal"],
                               frames["denoise_in_depth"])
    assert np.array_equal(den, frames["denoise_out"])
    assert np.array_equal(touched, frames["denoise_touched_oob"].astype(bool))
    # only the bottom rows depend on the reference's out-of-bounds row H (steps 1,2,4,8 -> 2*15 rows)
    assert not touched[: 64 - 31].any() and touched[-1].all()
    assert np.array_equal(orc.preview(frames["denoise_in_color"], 64, 64, 0), frames["preview_rgba"])
    assert np.all(frames["preview_rgba"][..., 3] == 255)
