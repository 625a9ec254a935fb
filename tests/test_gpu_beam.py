"""Entry points for primary rays ("beam", round 4): a pre-pass hands the primary rays of an 8 x 8-pixel tile the deepest
nodes of the launch's tree its frustum overlaps, and they start there instead of at the root.  Only the start of a walk
changes, so every frame must stay the oracle's bits -- checked here where the pre-pass has something to get wrong: the
entries themselves against the host's (pt_beam_rules.hpp runs on both sides: same bits), a mesh under a rotation and a
non-uniform scale, a camera inside the mesh, frames of one batch with different cameras (one set of entries per distinct
camera), frame sizes that are no multiple of the tile, the interleaved multi-GPU rows (a slot is another pixel there)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cam_c(pkg, camera):
    cam = pkg._capi.ptc_camera()
    cam.position[:] = [float(x) for x in camera.position]
    cam.rotation_wxyz[:] = [float(x) for x in camera.rotation]
    cam.vfov = float(camera.vfov)
    return cam


def test_device_entries_are_the_host_entries(pkg):
    glm = pkg.glmlite
    lib = pkg.lib()
    for w, h, transform in ((160, 96, None), (101, 67, glm.compose([glm.rotate(np.float32(0.5), (0.2, 1.0, 0.1)), glm.scale((1.1, 0.7, 0.9)),
                                                                      glm.translate((0.2, -0.1, 0.3))]))):
        scene = pkg.SceneDescription()
        scene.resolution = (w, h)
        scene.camera = pkg.scenes._camera_from_look_at((0.0, 2.0, 4.5), (0.0, 0.0, 0.0), vfov_deg=50.0)
        mesh = pkg.scenes.heightfield_mesh(97, 49, 8.0, 4.0, seed=7)
        scene.add_mesh("m", mesh)
        scene.add_material("white", pkg.DiffuseMateral((0.7, 0.7, 0.7)))
        scene.add_object(mesh, glm.translate((0.0, 0.0, 0.0)) if transform is None else transform, "white")
        flat = scene.build_scene()
        cam = _cam_c(pkg, scene.camera)
        tiles = ((w + 7) // 8) * ((h + 7) // 8)
        host = np.zeros(tiles * 32, dtype=np.float32)
        pos = np.ascontiguousarray(mesh.positions, dtype=np.float32)
        idx = np.ascontiguousarray(mesh.indices, dtype=np.uint32)
        m = np.ascontiguousarray(np.array(flat.objects[0]["m"], dtype=np.float32).reshape(16))
        stats = (C.c_uint64 * 5)()
        assert lib.ptc_check_beam(pos.ctypes.data, len(pos), idx.ctypes.data, len(idx), m.ctypes.data, C.byref(cam), w, h, 4, stats, host.ctypes.data) == 0
        with pkg.PathTracer(device=0, max_bounces=4) as pt:
            pt.create_buffers((w, h), flat)
            dev = np.zeros(tiles * 32, dtype=np.float32)
            assert lib.ptc_debug_beam_entries(pt._ctx, C.byref(cam), dev.ctypes.data, dev.size) == 0
        assert np.array_equal(host.view(np.uint32), dev.view(np.uint32)), int(np.sum(host.view(np.uint32) != dev.view(np.uint32)))
        assert stats[2] > 0 and stats[1] < stats[0]


def _frames(pkg, flat, cameras, w, h, mb, params=()):
    with pkg.PathTracer(device=0, max_bounces=mb) as pt:
        for k, v in params:
            pt.set_param(k, v)
        pt.create_buffers((w, h), flat)
        pt.max_iterations = len(cameras)
        for cam in cameras:
            pt.path_trace(cam)
        out = {k: pt.download(k) for k in ("color", "normal", "depth")}
        out["rays"] = pt.stats()["rays_total"]
    return out


def test_frames_with_and_without_entry_points(pkg, orc):
    glm = pkg.glmlite
    look = pkg.scenes._camera_from_look_at
    w, h, mb = 120, 75, 6
    scene = pkg.SceneDescription()
    scene.resolution = (w, h)
    mesh = pkg.scenes.heightfield_mesh(65, 33, 8.0, 4.0, seed=3)
    scene.add_mesh("m", mesh)
    for name, mat in (("white", pkg.DiffuseMateral((0.7, 0.7, 0.7))), ("steel", pkg.MetalMaterial((0.8, 0.8, 0.9), 0.1)), ("glass", pkg.DielectricMaterial(1.5))):
        scene.add_material(name, mat)
    scene.add_object(mesh, glm.compose([glm.rotate(np.float32(0.4), (0.1, 1.0, 0.0)), glm.scale((1.0, 1.5, 0.8)), glm.translate((0.0, -0.2, 0.0))]), "white")
    scene.add_object(pkg.Sphere((0, 0, 0), 0.4), glm.translate((0.5, 0.6, 0.5)), "glass")
    scene.add_object(pkg.Sphere((0, 0, 0), 0.3), glm.translate((-0.8, 0.5, 0.2)), "steel")
    flat = scene.build_scene()
    # one camera for three iterations (accumulated), then a batch whose frames all have different cameras -- the way the
    # accumulation is defined a new camera would restart it, so those are compared frame by frame below
    cam = look((0.0, 2.2, 4.0), (0.0, 0.0, 0.0), vfov_deg=50.0)
    ref = orc.render_streaming(flat, cam, w, h, 0, 3, mb)
    for params in ((), (("beam", 0),), (("frames_in_flight", 1),), (("batch_frames", 2), ("frames_in_flight", 4))):
        got = _frames(pkg, flat, [cam] * 3, w, h, mb, params)
        for k in ("color", "normal", "depth"):
            assert np.array_equal(got[k], ref[k]), (params, k)
        assert got["rays"] == ref["rays"]
    cams = [look((0.0, 2.2, 4.0), (0.0, 0.0, 0.0), vfov_deg=50.0), look((1.5, 0.3, 0.2), (0.0, 0.1, 0.0), vfov_deg=65.0),   # the second: inside the mesh's box
            look((-2.0, 1.0, -3.0), (0.3, 0.0, 0.0), vfov_deg=40.0)]
    # iteration i with camera i: the running mean of three different views (what ptc_trace does when nobody restarts)
    prev = None
    for i, c in enumerate(cams):
        prev = orc.render_streaming(flat, c, w, h, i, 1, mb, prev=prev)
    for params in ((), (("beam", 0),)):
        got = _frames(pkg, flat, cams, w, h, mb, params)      # one batch of three frames, three cameras
        for k in ("color", "normal", "depth"):
            assert np.array_equal(got[k], prev[k]), (params, k)


def test_interleaved_rows_with_entry_points(pkg, orc):
    """a rank's slot s of bounce 0 holds pixel band_pixel(s): the tile of a ray comes from there"""
    scene = pkg.scenes.heightfield_scene((96, 72), nx=65, nz=33)
    flat = scene.build_scene()
    w, h, world, block, mb = 96, 72, 3, 8, 6
    for rank in range(world):
        want = orc.render_interleaved(flat, scene.camera, w, h, rank, world, block, rank * w * h, 0, 2, mb)
        with pkg.PathTracer(device=0, max_bounces=mb) as pt:
            pt.create_buffers((w, h), flat)
            pt.set_interleave(rank, world, block)
            pt.set_param("slot_offset", rank * w * h)
            pt.max_iterations = 2
            for _ in range(2):
                pt.path_trace(scene.camera)
            for k in ("color", "normal", "depth"):
                assert np.array_equal(pt.download(k), want[k]), (rank, k)
