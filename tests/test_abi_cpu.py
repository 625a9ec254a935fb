"""CPU-side checks of the product library: it loads, exports exactly what include/ptcore.h declares,
its GPU-free host functions (BVH builder, object flattening) agree with the oracle bit for bit, and
GPU entry points fail loudly (no CPU fallback) when there is no device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptcore.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ptcore.h but not exported by libptcore.so"
    assert set(names) == set(pkg._capi.SIGNATURES), "ctypes binding and header disagree"
    assert lib.ptc_abi_version() == 3


def test_struct_layouts_match_reference_sizes(pkg):
    c = pkg._capi
    assert C.sizeof(c.ptc_object) == 160      # GPUObject, scene.hpp:16-22
    assert C.sizeof(c.ptc_sphere) == 16       # Sphere, sphere.hpp:8-11
    assert C.sizeof(c.ptc_material) == 20     # Material, material.hpp:19-38
    assert C.sizeof(c.ptc_bvh_node) == 32     # static_assert in bvh.hpp:30
    assert C.sizeof(c.ptc_camera) == 32


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device path")
def test_no_device_fails_loudly(pkg):
    with pytest.raises(pkg.PtcError) as e:
        pkg.PathTracer()
    assert e.value.code == pkg._capi.PTC_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


@pytest.mark.parametrize("mesh_name", ["grid9x5", "grid33x17", "sphere8x16", "sphere24x48", "grid_ties"])
def test_host_bvh_builder_equals_oracle(pkg, orc, mesh_name):
    mesh = {
        "grid9x5": lambda: pkg.scenes.heightfield_mesh(9, 5, 2.0, 1.0, seed=1),
        "grid33x17": lambda: pkg.scenes.heightfield_mesh(33, 17, 8.0, 4.0, seed=7),
        "sphere8x16": lambda: pkg.scenes.displaced_sphere_mesh(8, 16),
        "sphere24x48": lambda: pkg.scenes.displaced_sphere_mesh(24, 48),
        # flat regular grid: many exactly equal centroids along the split axis (tie rule)
        "grid_ties": lambda: pkg.Mesh(*_flat_grid(17, 9)),
    }[mesh_name]()
    got, got_depth = pkg.bvh_from_mesh(mesh)
    want, want_depth = orc.build_bvh(mesh.positions, mesh.indices)
    assert got_depth == want_depth
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


@pytest.mark.parametrize("mesh_name", ["grid33x17", "sphere24x48", "grid_ties", "grid257x129", "one_triangle", "thin",
                                       "benchmark_1M"])
def test_traversal_layout_is_sound(pkg, mesh_name):
    """The default traversal walks a four-wide collapse of the reference BVH with 64-byte quantised nodes.  It only
    has to be conservative, so the invariant is containment: every quantised child box contains the exact box of the
    reference node it stands for (checked in double precision), the children of a node tile its leaves in
    depth-first order, every triangle hangs under exactly one node, and the recorded parent boxes are the
    reference's.  Host-side, no GPU."""
    import ctypes as C
    mesh = {
        "grid33x17": lambda: pkg.scenes.heightfield_mesh(33, 17, 8.0, 4.0, seed=7),
        "sphere24x48": lambda: pkg.scenes.displaced_sphere_mesh(24, 48),
        "grid_ties": lambda: pkg.Mesh(*_flat_grid(17, 9)),
        "grid257x129": lambda: pkg.scenes.heightfield_mesh(257, 129, 8.0, 4.0, seed=3),
        "benchmark_1M": lambda: list(pkg.scenes.heightfield_scene((64, 64)).mesh_map_.values())[0],  # bench.py's mesh
        "one_triangle": lambda: pkg.Mesh(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32),
                                         np.array([0, 1, 2], dtype=np.uint32)),
        # extents that differ by 1e6 between the axes, far from the origin: a different grid step per axis
        "thin": lambda: pkg.Mesh((pkg.scenes.heightfield_mesh(33, 17, 8.0, 4.0, seed=5).positions
                                  * np.array([1000.0, 0.001, 1.0], dtype=np.float32)
                                  + np.array([5000.0, -3.0, 77.0], dtype=np.float32)).astype(np.float32),
                                 pkg.scenes.heightfield_mesh(33, 17, 8.0, 4.0, seed=5).indices),
    }[mesh_name]()
    nodes, _ = pkg.bvh_from_mesh(mesh)
    checked = C.c_uint64(0)
    arr = np.ascontiguousarray(nodes)
    bad = pkg.lib().ptc_check_traversal_layout(arr.ctypes.data_as(C.POINTER(pkg._capi.ptc_bvh_node)), len(arr),
                                                C.byref(checked))
    assert bad == 0
    tris = len(mesh.indices) // 3
    assert checked.value >= tris or tris == 1   # every leaf is one of the child boxes (a lone triangle is the root)


def _flat_grid(nx, nz):
    m_x, m_z = np.meshgrid(np.arange(nx, dtype=np.float32), np.arange(nz, dtype=np.float32), indexing="xy")
    pos = np.stack([m_x, 0.01 * ((m_x * 7 + m_z * 3) % 5), m_z], axis=-1).reshape(-1, 3)
    v = (np.arange(nz - 1)[:, None] * nx + np.arange(nx - 1)[None, :]).astype(np.uint32)
    idx = np.stack([np.stack([v, v + nx, v + 1], -1), np.stack([v + 1, v + nx, v + nx + 1], -1)], axis=2).reshape(-1)
    return pos, idx


def test_host_bvh_builder_errors(pkg):
    lib = pkg.lib()
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    idx = np.tile(np.array([0, 1, 2], dtype=np.uint32), 6)
    nodes = np.zeros(11, dtype=pkg.scene_description.BVH_NODE_DTYPE)
    fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    rc = lib.ptc_build_bvh(pos.ctypes.data_as(fp), 3, idx.ctypes.data_as(up), len(idx),
                           nodes.ctypes.data_as(C.POINTER(pkg._capi.ptc_bvh_node)), None)
    assert rc == pkg._capi.PTC_ERR_BVH       # coincident centroids (reference panics, bvh.cpp:84-85)
    bad = np.array([0, 1, 7], dtype=np.uint32)
    rc = lib.ptc_build_bvh(pos.ctypes.data_as(fp), 3, bad.ctypes.data_as(up), 3,
                           nodes.ctypes.data_as(C.POINTER(pkg._capi.ptc_bvh_node)), None)
    assert rc == pkg._capi.PTC_ERR_INVALID   # vertex index out of range


def test_make_object_equals_oracle(pkg, orc):
    """scene_description.cpp:17-52 on both sides: glm::inverse, sphere AABB, transform_aabb"""
    glm = pkg.glmlite
    lib, L = pkg.lib(), orc.lib()
    transforms = [
        glm.identity(), glm.translate((1.0, -0.5, -2.0)),
        glm.compose([glm.scale(0.5), glm.translate((-1.0, -0.5, -2.0))]),
        glm.compose([glm.translate((-0.053126335, 0.030193329, 17.283958)), glm.scale(0.2),
                     glm.rotate(np.float32(np.radians(150.0)), (0, 1, 0)), glm.translate((0, 0, -0.25))]),
        glm.compose([glm.rotate(np.float32(0.7), (1, 2, 3)), glm.scale((1.5, 0.25, 3.0)), glm.translate((3, 4, 5))]),
    ]
    fp = C.POINTER(C.c_float)
    for m in transforms:
        m = np.ascontiguousarray(m, dtype=np.float32)
        sp = pkg._capi.ptc_sphere()
        sp.center[:] = [0.1, -0.2, 0.3]
        sp.radius = 0.75
        box = np.array([-1, -2, -3, 1.5, 2.5, 3.5], dtype=np.float32)
        # extension (ptc_mesh_range): a mesh object's index names its mesh and is kept
        got = pkg._capi.ptc_object()
        assert lib.ptc_make_object(1, 3, m.ctypes.data_as(fp), None, box.ctypes.data_as(fp), C.byref(got)) == 0
        assert got.index == 3 and got.type == 1
        for typ in (0, 1):
            index = 4 if typ == 0 else 0   # the reference's mesh objects all carry index 0 (one mesh per scene)
            got = pkg._capi.ptc_object()
            assert lib.ptc_make_object(typ, index, m.ctypes.data_as(fp), C.byref(sp) if typ == 0 else None,
                                       box.ctypes.data_as(fp) if typ == 1 else None, C.byref(got)) == 0
            want = np.zeros(160, dtype=np.uint8)
            L.orc_make_object(typ, index, m.ctypes.data, C.addressof(sp) if typ == 0 else None,
                              box.ctypes.data if typ == 1 else None, want.ctypes.data)
            assert bytes(got) == want.tobytes()
    # the inverse really is one
    inv = np.frombuffer(bytes(got), dtype=np.float32)[18:34].reshape(4, 4)
    assert np.allclose(pkg.glmlite.matmul(m, inv), np.eye(4), atol=1e-5)


def test_invalid_arguments_return_error_codes(pkg):
    lib = pkg.lib()
    assert lib.ptc_create(None, None) == pkg._capi.PTC_ERR_INVALID
    assert lib.ptc_restart(None) == pkg._capi.PTC_ERR_INVALID
    assert lib.ptc_iteration(None) == pkg._capi.PTC_ERR_INVALID
    assert lib.ptc_make_object(2, 0, None, None, None, None) == pkg._capi.PTC_ERR_INVALID
    lib.ptc_destroy(None)  # no-op


def _beam_check(pkg, mesh, camera, w, h, m=None, stride=3):
    import ctypes as C
    capi, lib = pkg._capi, pkg.lib()
    pos = np.ascontiguousarray(mesh.positions, dtype=np.float32)
    idx = np.ascontiguousarray(mesh.indices, dtype=np.uint32)
    cam = capi.ptc_camera()
    cam.position[:] = [float(x) for x in camera.position]
    cam.rotation_wxyz[:] = [float(x) for x in camera.rotation]
    cam.vfov = float(camera.vfov)
    stats = (C.c_uint64 * 5)()
    mm = None if m is None else np.ascontiguousarray(np.asarray(m, dtype=np.float32).reshape(16))
    bad = lib.ptc_check_beam(pos.ctypes.data, len(pos), idx.ctypes.data, len(idx), None if mm is None else mm.ctypes.data, C.byref(cam), w, h,
                             stride, stats, None)
    return bad, [int(x) for x in stats]


def test_entry_points_of_primary_rays_are_sound(pkg):
    """"beam" (pt_beam_rules.hpp, the functions k_beam runs on the GPU), on the host: for every 8 x 8-pixel tile the
    entries its frustum keeps, and for the corners and the centre of the jitter range of sample pixels the closest hit of
    a walk from those entries against the walk from the root -- cameras outside, above, inside the mesh's box, looking
    away from it, a rotated and scaled object, frames that are no multiple of the tile, a single triangle."""
    glm = pkg.glmlite
    hf = pkg.scenes.heightfield_mesh(65, 33, 8.0, 4.0, seed=7)
    ball = pkg.scenes.displaced_sphere_mesh(16, 32)
    scene = pkg.scenes.heightfield_scene((96, 64), nx=65, nz=33)
    cam0 = scene.camera
    look = pkg.scenes._camera_from_look_at
    cases = [
        (hf, cam0, 96, 64, None),
        (hf, cam0, 101, 67, None),                                              # ragged last tiles
        (hf, look((0.0, 0.05, 0.0), (1.0, 0.0, 0.3), vfov_deg=70.0), 80, 48, None),     # inside the box, skimming the terrain
        (hf, look((0.0, 3.0, 0.0), (0.0, 0.0, 0.01), vfov_deg=40.0), 64, 64, None),     # straight down
        (hf, look((0.0, 2.0, 6.0), (0.0, 3.0, 12.0), vfov_deg=50.0), 64, 40, None),     # looking away: nothing in reach
        (ball, look((0.0, 0.0, 0.0), (0.3, 0.2, -1.0), vfov_deg=90.0), 72, 56, None),   # from inside a closed mesh
        (ball, look((0.5, 0.4, 2.5), (0.0, 0.0, 0.0), vfov_deg=35.0), 64, 64,
         glm.compose([glm.rotate(np.float32(0.7), (0.3, 1.0, 0.2)), glm.scale((0.8, 0.5, 1.2)), glm.translate((0.1, -0.2, 0.3))])),
        (pkg.scenes.heightfield_mesh(2, 2, 1.0, 1.0, seed=1), look((0.2, 1.0, 0.9), (0.0, 0.0, 0.0)), 32, 24, None),   # two triangles
    ]
    reached = 0
    for mesh, cam, w, h, m in cases:
        bad, st = _beam_check(pkg, mesh, cam, w, h, m)
        assert bad == 0, (w, h, st)
        tiles, empty, entries, rays, hits = st
        assert tiles == ((w + 7) // 8) * ((h + 7) // 8) and entries <= 4 * (tiles - empty) and rays > 0
        reached += hits
    assert reached > 10000
    # the case that looks away keeps nothing at all
    _, st = _beam_check(pkg, hf, look((0.0, 2.0, 6.0), (0.0, 3.0, 12.0), vfov_deg=50.0), 64, 40)
    assert st[1] == st[0] and st[4] == 0


def test_the_ray_feed_hands_every_ray_out_once(pkg):
    """ptc_check_feed (host, no GPU): the geometry of the traversal launches' ray feed (csrc/pt_feed_rules.hpp: eight
    regions of interleaved blocks, static batches then dynamic ones) on frame sizes around every boundary it has -- a
    batch, a block, eight blocks, the benchmark's per-bounce live counts -- for every static share and both dynamic batch
    sizes.  Found the hard way in round 4: a batch of two that straddled two blocks traced 64 rays of another region
    twice and skipped 64 of its own; only the 643 M-ray soak test noticed."""
    lib = pkg.lib()
    rng = np.random.default_rng(3)
    sizes = [0, 1, 63, 64, 65, 127, 128, 129, 511, 512, 513, 1023, 1024, 1025, 8191, 8192, 8193, 16383, 16384, 16385,
             65097, 65536, 89175, 126854, 131072, 131073, 183471, 281220, 425894, 769669, 924531, 2073600]
    sizes += [int(x) for x in rng.integers(1, 3_000_000, size=24)]
    for n in sizes:
        for eighths in (0, 1, 3, 7, 8):
            for dyn in (64, 128):
                assert lib.ptc_check_feed(n, eighths, dyn) == 0, (n, eighths, dyn)
    assert lib.ptc_check_feed(100, 9, 64) < 0 and lib.ptc_check_feed(100, 3, 96) < 0
