"""The bounce-spanning persistent launch (k_persist, round 5): bounces >= 1 of a batch of frames and every
shade pass as ONE launch -- four of five wavefronts walk rays that PersistFeed deals frame by frame, one shades tiles of
whichever frame's traversal phase is complete; phases are opened by whoever finishes the previous phase's last piece of
work, all of it through agent-scope atomics and write-through (sc1) hand-overs.  It is a schedule: images, G-buffers, live
counts and ray counts must be the oracle's bits, whatever the mix of roles, the number of wavefronts, the frames per batch --
and the per-bounce launches ("persist" 0) must give the same."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(pkg, scene, flat, w, h, iters, mb, params=(), batch=None):
    with pkg.PathTracer(device=0, max_bounces=mb) as pt:
        pt.set_param("persist", 1)   # (not the default schedule: measured slower)
        for k, v in params:
            pt.set_param(k, v)
        if batch:
            pt.set_param("frames_in_flight", batch[0] * batch[1])
            pt.set_param("batch_frames", batch[1])
        pt.create_buffers((w, h), flat)
        pt.max_iterations = iters
        pt.reset_profile()
        for _ in range(iters):
            pt.path_trace(scene.camera)
        out = {k: pt.download(k) for k in ("color", "normal", "depth")}
        st = pt.stats()
        out["rays"], out["live"] = st["rays_total"], st["last_live"]
        out["persist_launches"] = pt.profile()["persist_launches"]
    return out


def _same(got, ref, what):
    for k in ("color", "normal", "depth"):
        assert np.array_equal(got[k], ref[k]), (what, k, int(np.sum(got[k] != ref[k])))
    assert got["rays"] == ref["rays"], what
    assert got["live"][:len(ref["live"][-1])] == [int(x) for x in ref["live"][-1]], what


@pytest.mark.parametrize("spheres", [True, False])
def test_persistent_launch_against_the_oracle(pkg, orc, spheres):
    w, h, iters, mb = 160, 96, 12, 6
    scene = pkg.scenes.heightfield_scene((w, h), nx=65, nz=33)
    if not spheres:   # no sphere run behind the mesh: the service wavefronts' tiles without the trailing spheres
        glm = pkg.glmlite
        s = pkg.SceneDescription()
        s.resolution, s.camera = (w, h), scene.camera
        s.add_material("ground", pkg.DiffuseMateral((0.7, 0.7, 0.7)))
        mesh = list(scene.mesh_map_.values())[0]
        s.add_mesh("ground", mesh)
        s.add_object(mesh, glm.translate((0.0, 0.0, 0.0)), "ground")
        scene = s
    flat = scene.build_scene()
    ref = orc.render_streaming(flat, scene.camera, w, h, 0, iters, mb)
    on = _render(pkg, scene, flat, w, h, iters, mb, batch=(1, 12))
    assert on["persist_launches"] >= 1, "the batch did not take the persistent launch"
    _same(on, ref, "persist")
    off = _render(pkg, scene, flat, w, h, iters, mb, params=(("persist", 0),), batch=(1, 12))
    assert off["persist_launches"] == 0
    _same(off, ref, "per-bounce launches")
    # other mixes of roles and launch sizes: every second wavefront shading; one in nine; a launch of few wavefronts (more
    # rays than lanes: the feed's cursors are what limits it) and the full-size one; three-frame batches on two streams
    for params, batch in (((("persist_service_every", 2),), (1, 12)), ((("persist_service_every", 9), ("traverse_waves", 256)), (1, 12)),
                          ((("traverse_waves", 64),), (1, 6)), ((), (2, 3)), ((("beam", 0), ("filter_rays", 0)), (1, 4))):
        got = _render(pkg, scene, flat, w, h, iters, mb, params=params, batch=batch)
        assert got["persist_launches"] >= 1, (params, batch)
        _same(got, ref, (params, batch))


def test_persistent_launch_with_set_aside_rays(pkg, orc):
    """force_slow 2 sets EVERY winner aside for the exact redo: the R phases of the persistent launch (one frame at a time,
    under the launch's lock) carry the whole image."""
    w, h, iters, mb = 96, 64, 4, 4
    scene = pkg.scenes.heightfield_scene((w, h), nx=33, nz=17)
    flat = scene.build_scene()
    ref = orc.render_streaming(flat, scene.camera, w, h, 0, iters, mb)
    for fs in (1, 2):
        got = _render(pkg, scene, flat, w, h, iters, mb, params=(("debug_force_slow", fs),), batch=(1, 4))
        assert got["persist_launches"] >= 1
        _same(got, ref, ("force_slow", fs))


def test_persistent_launch_at_benchmark_size(pkg, orc):
    """Config 3 itself, 1920x1080 over the 1,000,000-triangle mesh, six frames in one batch: uneven load for real (the sky rows
    of a frame are done in raygen, the horizon rows walk hundreds of nodes), 5120 wavefronts, every XCD."""
    w, h, iters, mb = 1920, 1080, 6, 8
    scene = pkg.scenes.heightfield_scene((w, h))
    flat = scene.build_scene()
    flat.bvh, _ = pkg.bvh_from_mesh(list(scene.mesh_map_.values())[0])
    ref = orc.render_streaming(flat, scene.camera, w, h, 0, iters, mb, nthreads=16)
    got = _render(pkg, scene, flat, w, h, iters, mb, batch=(1, 6))
    assert got["persist_launches"] >= 1
    _same(got, ref, "1080p")


def test_paired_batches_take_turns(pkg, orc):
    """ "pair_batches": a full batch is held until the next one is full, the two are enqueued bounce by bounce on two slots and
    events make their traversal launches alternate.  A schedule: same bits; a lone batch (the last, odd one) goes out alone, and
    anything that looks at the context flushes what is held."""
    w, h, mb = 128, 80, 5
    scene = pkg.scenes.heightfield_scene((w, h), nx=33, nz=17)
    flat = scene.build_scene()
    for iters, batch in ((12, 3), (9, 3), (7, 2)):
        ref = orc.render_streaming(flat, scene.camera, w, h, 0, iters, mb)
        with pkg.PathTracer(device=0, max_bounces=mb) as pt:
            pt.set_param("pair_batches", 1)
            pt.set_param("frames_in_flight", 2 * batch)
            pt.set_param("batch_frames", batch)
            pt.create_buffers((w, h), flat)
            pt.max_iterations = iters
            for k in range(iters):
                pt.path_trace(scene.camera)
                if k == 4:
                    assert pt.iteration() == 5      # (held or queued iterations count as rendered, as in the reference)
            got = {k: pt.download(k) for k in ("color", "normal", "depth")}
            st = pt.stats()
            got["rays"], got["live"], got["persist_launches"] = st["rays_total"], st["last_live"], 0
        _same(got, ref, ("pair_batches", iters, batch))
