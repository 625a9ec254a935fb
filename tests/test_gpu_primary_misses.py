"""Primary rays that hit nothing end in ray generation ("filter_rays": k_raygen<kFilter, kFinish>) and bounce 0's shade
kernel walks the traversal launch's work list instead of all slots.  That path is taken when the bounce's one traversal
launch covers the scene's whole mesh part (spheres may only END the object list; their world boxes join the filter).
Checked against the CPU oracle (raygen_kernel ray_gen.cu:11-32, material_kernel's miss branch path_tracer.cu:304-307):
images, per-bounce live counts and ray totals, staged and unstaged, and against the same run with the filter off."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _terrain(pkg, spheres):
    """A small heightfield under a wide sky; spheres: none, or two -- one on the ground, one high up in the sky part of
    the image, far outside the terrain's world box (only its own box lists the rays that hit it)."""
    glm = pkg.glmlite
    s = pkg.SceneDescription()
    s.add_material("ground", pkg.DiffuseMateral((0.7, 0.6, 0.5)))
    s.add_material("metal", pkg.MetalMaterial((0.8, 0.8, 0.9), 0.05))
    s.add_material("glass", pkg.DielectricMaterial(1.5))
    mesh = s.add_mesh("terrain", pkg.scenes.heightfield_mesh(49, 33, 3.0, 2.0, seed=11))
    s.add_object(mesh, glm.identity(), "ground")
    if spheres:
        s.add_object(pkg.Sphere((0, 0, 0), 0.4), glm.translate((0.3, 0.6, 0.2)), "glass")
        s.add_object(pkg.Sphere((0, 0, 0), 0.5), glm.translate((-0.8, 3.2, -1.0)), "metal")
    s.camera = pkg.scenes._camera_from_look_at((0.0, 1.6, 5.0), (0.0, 1.2, 0.0), vfov_deg=55.0)
    return s


def _run(pkg, scene, flat, w, h, iters, mb, params):
    with pkg.PathTracer(device=0, max_bounces=mb) as pt:
        for k, v in params:
            pt.set_param(k, v)
        pt.create_buffers((w, h), flat)
        pt.max_iterations = iters
        for _ in range(iters):
            pt.path_trace(scene.camera)
        out = {k: pt.download(k) for k in ("color", "normal", "depth")}
        out["stats"] = pt.stats()
        out["profile"] = pt.profile()
    return out


@pytest.mark.parametrize("spheres", [False, True])
def test_sky_pixels_finished_by_raygen_match_the_oracle(pkg, orc, spheres):
    w, h, iters, mb = 160, 96, 5, 6
    scene = _terrain(pkg, spheres)
    flat = scene.build_scene()
    ref = orc.render_streaming(flat, scene.camera, w, h, 0, iters, mb)
    hit_first = int(ref["live"][0][1]) if mb > 1 else 0
    assert 0 < hit_first < w * h * 0.7          # a good part of the image is sky
    off = _run(pkg, scene, flat, w, h, iters, mb, (("filter_rays", 0), ("frames_in_flight", 1)))
    for params in ((("frames_in_flight", 1),),                              # unstaged: straight into the framebuffers
                   (("frames_in_flight", 8), ("batch_frames", 4)),          # staged, a batch and a ragged one
                   (("frames_in_flight", 3),)):
        got = _run(pkg, scene, flat, w, h, iters, mb, params)
        for k in ("color", "normal", "depth"):
            assert np.array_equal(got[k], ref[k]), (k, params)
            assert np.array_equal(got[k], off[k]), (k, params)
        assert got["stats"]["rays_total"] == ref["rays"] == off["stats"]["rays_total"]
        # the path under test really ran: bounce 0's launch fetched a list, and that list left the sky out
        listed = got["profile"]["listed_rays"][0]
        assert 0 < listed < iters * w * h * 0.8, listed
        assert listed >= int(ref["live"][:, 1].sum())      # every ray that survives bounce 0 was on it
    assert off["profile"]["listed_rays"][0] == 0


def test_frame_that_is_all_sky(pkg, orc):
    """The camera looks away from everything: every list is empty, k_shade_fused has no tile to take, and the frame is
    the sky (live[1] == 0, the rays still count)."""
    w, h, iters, mb = 96, 64, 3, 4
    scene = _terrain(pkg, True)
    scene.camera = pkg.scenes._camera_from_look_at((0.0, 6.0, 5.0), (0.0, 12.0, 0.0), vfov_deg=40.0)
    flat = scene.build_scene()
    ref = orc.render_streaming(flat, scene.camera, w, h, 0, iters, mb)
    assert ref["rays"] == iters * w * h
    for params in ((("frames_in_flight", 1),), (("frames_in_flight", 4), ("batch_frames", 2))):
        got = _run(pkg, scene, flat, w, h, iters, mb, params)
        for k in ("color", "normal", "depth"):
            assert np.array_equal(got[k], ref[k]), (k, params)
        assert got["stats"]["rays_total"] == ref["rays"]
        assert got["profile"]["listed_rays"][0] == 0


@pytest.mark.parametrize("size", [(33, 17), (130, 70), (64, 16), (257, 4), (1030, 3)])
def test_odd_frame_sizes_through_the_lists(pkg, orc, size):
    """Frames that are not a whole number of 1024-slot tiles, of 64-slot wavefronts, or smaller than one tile: the
    look-back scan of the work list, the ticketed tiles of the shade kernel and the feed's regions at their edges."""
    w, h = size
    iters, mb = 3, 5
    scene = _terrain(pkg, True)
    flat = scene.build_scene()
    ref = orc.render_streaming(flat, scene.camera, w, h, 0, iters, mb)
    for params in ((("frames_in_flight", 1),), (("frames_in_flight", 6), ("batch_frames", 3)), (("frames_in_flight", 2),)):
        got = _run(pkg, scene, flat, w, h, iters, mb, params)
        for k in ("color", "normal", "depth"):
            assert np.array_equal(got[k], ref[k]), (k, size, params)
        assert got["stats"]["rays_total"] == ref["rays"]


@pytest.mark.parametrize("size", [(37, 29), (129, 65)])
def test_odd_frame_sizes_with_a_sphere_run_in_front(pkg, orc, size):
    """The other list builder: k_spheres (a Cornell box of wall spheres in front of two mesh instances, a glass sphere
    behind them) at sizes that leave ragged tiles at every bounce."""
    w, h = size
    iters, mb = 3, 6
    scene = pkg.scenes.cornell_bunny((w, h), n_lat=12, n_lon=24)
    flat = scene.build_scene()
    ref = orc.render_streaming(flat, scene.camera, w, h, 0, iters, mb)
    off = _run(pkg, scene, flat, w, h, iters, mb, (("filter_rays", 0), ("frames_in_flight", 1)))
    for params in ((("frames_in_flight", 1),), (("frames_in_flight", 6), ("batch_frames", 3)), (("frames_in_flight", 4), ("fused_shade", 0))):
        got = _run(pkg, scene, flat, w, h, iters, mb, params)
        for k in ("color", "normal", "depth"):
            assert np.array_equal(got[k], ref[k]), (k, size, params)
            assert np.array_equal(got[k], off[k]), (k, size, params)
        assert got["stats"]["rays_total"] == ref["rays"]
        assert sum(got["profile"]["listed_rays"]) > 0
