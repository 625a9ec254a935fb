"""Pins the CPU oracle against every known answer available for this path:
 - SURVEY.md section 8(a) KATs for hash / seed / jitter (computed from the reference's hash.cuh formula),
 - rocThrust 7.2's minstd_rand + uniform_real_distribution run on the host (tests/golden/rng_kat.json,
   generator oracle/tools/thrust_rng_kat.cpp) -- the algorithm behind thrust::default_random_engine,
 - the values of the reference's own Catch2 tests (test/aabb_test.cpp:6-59, test/transform_test.cpp:8-45,
   tolerance 100*FLT_EPSILON as in test/glm_test_helper.hpp:53-60).
Everything else on the path has no reference-side vector: parity unpinned."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

EPS = 100 * np.finfo(np.float32).eps


def test_hash_kats(orc):
    L = orc.lib()
    assert L.orc_hash(0) == 0x6B4ED927
    assert L.orc_hash(1) == 0xB48681B6
    assert L.orc_path_seed(0, 0) == 0x2B4F8145
    # the XOR happens in 64 bits, then truncates: high iteration bits must not leak in
    assert L.orc_path_seed(5, (1 << 32) | 3) == L.orc_path_seed(5, 3)


def test_rng_matches_rocthrust(orc, golden_dir):
    L = orc.lib()
    cases = json.load(open(os.path.join(golden_dir, "rng_kat.json")))["cases"]
    assert len(cases) >= 50
    for c in cases:
        st = C.c_uint32(L.orc_rng_seed(c["seed"]))
        L.orc_rng_discard(C.byref(st), c["discard"])
        raw_state = C.c_uint32(st.value)
        raw = [L.orc_rng_next(C.byref(raw_state)) for _ in range(4)]
        assert raw == c["raw"], c
        uni = [np.float32(L.orc_rng_uniform(C.byref(st))).view(np.uint32) for _ in range(4)]
        assert [int(u) for u in uni] == c["uniform_bits"], c


def test_rng_survey_values(orc):
    L = orc.lib()
    st = C.c_uint32(L.orc_rng_seed(12345))
    got = [L.orc_rng_uniform(C.byref(st)) for _ in range(3)]
    assert np.allclose(got, [0.277490109, 0.725584686, 0.697912633], rtol=0, atol=1e-9)
    st = C.c_uint32(L.orc_rng_seed(L.orc_path_seed(0, 0)))
    jitter = [L.orc_rng_uniform(C.byref(st)) for _ in range(2)]
    assert np.allclose(jitter, [0.158939525, 0.169658437], rtol=0, atol=1e-9)
    # discard(z) == z single steps
    a = C.c_uint32(L.orc_rng_seed(777))
    b = C.c_uint32(a.value)
    L.orc_rng_discard(C.byref(a), 9)
    for _ in range(9):
        L.orc_rng_next(C.byref(b))
    assert a.value == b.value


def _aabb(lo, hi):
    return np.array([*lo, *hi], dtype=np.float32)


def test_aabb_reference_values(orc):
    """test/aabb_test.cpp:6-59"""
    L = orc.lib()
    box = _aabb((1, 2, 3), (7, 6, 5))
    ext = np.zeros(3, dtype=np.float32)
    L.orc_aabb_extent(box.ctypes.data, ext.ctypes.data)          # SECTION("extent"), aabb_test.cpp:8-15
    assert np.all(np.abs(ext - np.float32([6, 4, 2])) <= EPS)
    assert L.orc_aabb_surface_area(box.ctypes.data) == 88
    assert L.orc_aabb_max_extent(_aabb((1, 2, 3), (100, 6, 5)).ctypes.data) == 0
    assert L.orc_aabb_max_extent(_aabb((1, 2, 3), (7, 100, 5)).ctypes.data) == 1
    assert L.orc_aabb_max_extent(_aabb((1, 2, 3), (7, 6, 100)).ctypes.data) == 2
    out = np.zeros(3, dtype=np.float32)
    for p, want in (((1, 2, 3), (0, 0, 0)), ((7, 6, 5), (1, 1, 1)), ((4, 4, 4), (.5, .5, .5))):
        pt = np.array(p, dtype=np.float32)
        L.orc_aabb_offset(box.ctypes.data, pt.ctypes.data, out.ctypes.data)
        assert np.all(np.abs(out - np.array(want, dtype=np.float32)) <= EPS)


def _inverse_ray(orc, pkg, m):
    L = orc.lib()
    m = np.ascontiguousarray(m, dtype=np.float32)
    inv = np.zeros((4, 4), dtype=np.float32)
    L.orc_mat4_inverse(m.ctypes.data, inv.ctypes.data)
    ray = orc.ORay()
    ray.origin[:] = [1, 2, 3]
    ray.t_min = 0
    ray.direction[:] = [1, 0, 0]
    ray.t_max = 100
    out = orc.ORay()
    L.orc_inverse_transform_ray(m.ctypes.data, inv.ctypes.data, C.byref(ray), C.byref(out))
    return np.array(out.origin[:]), np.array(out.direction[:]), out.t_min, out.t_max


def test_transform_reference_values(orc, pkg):
    """test/transform_test.cpp:8-45"""
    glm = pkg.glmlite
    o, d, t0, t1 = _inverse_ray(orc, pkg, glm.translate((1, 1, 1)))
    assert np.all(np.abs(o - [0, 1, 2]) <= EPS) and np.all(np.abs(d - [1, 0, 0]) <= EPS) and (t0, t1) == (0, 100)
    o, d, t0, t1 = _inverse_ray(orc, pkg, glm.scale((2, 2, 2)))
    assert np.all(np.abs(o - [0.5, 1, 1.5]) <= EPS) and np.all(np.abs(d - [1, 0, 0]) <= EPS) and (t0, t1) == (0, 100)
    o, d, t0, t1 = _inverse_ray(orc, pkg, glm.rotate(np.float32(math.pi), (0, 1, 0)))
    assert np.all(np.abs(o - [-1, 2, -3]) <= EPS) and np.all(np.abs(d - [-1, 0, 0]) <= EPS) and (t0, t1) == (0, 100)


def test_deterministic_sincos_is_close_to_libm(orc):
    """orc_sincos replaces sinf/cosf (documented deviation); it must stay within ~1 ulp of the true value
    on the range random_in_unit_sphere uses, [0, 2 pi]."""
    L = orc.lib()
    xs = np.linspace(0, 2 * math.pi, 20001).astype(np.float32)
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for x in xs:
        L.orc_sincos(float(x), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - math.sin(float(x))), abs(c.value - math.cos(float(x))))
    assert worst < 1.2e-7


def test_generate_ray_golden(orc, golden_dir):
    kat = np.load(os.path.join(golden_dir, "kat.npz"))
    cam = orc.OCamera()
    cam.position[:] = [0, 0, 0]
    cam.rotation_wxyz[:] = [1, 0, 0, 0]
    cam.vfov = float(np.float32(np.radians(60.0)))
    g = orc.OGPUCamera()
    orc.lib().orc_to_gpu_camera(C.byref(cam), 4, 3, C.byref(g))
    want = kat["generate_ray_4x3"]
    for y in range(3):
        for x in range(4):
            ray = orc.ORay()
            orc.lib().orc_generate_ray(C.byref(g), x + 0.5, y + 0.5, C.byref(ray))
            got = np.frombuffer(bytes(ray), dtype=np.float32)
            assert np.array_equal(got.view(np.uint32), want[y, x].view(np.uint32))
    # geometry of ray_gen.cu:34-61: row 0 is the top of the view, rays look down -z, unit length
    assert want[0, 0, 5] > 0 > want[2, 0, 5] and np.all(want[..., 6] < 0)
    assert np.allclose(np.linalg.norm(want[..., 4:7], axis=-1), 1, atol=1e-6)
    assert np.all(want[..., 3] == np.float32(1e-4)) and np.all(want[..., 7] == np.finfo(np.float32).max)


def _ray(orc, o, d, tmin=1e-4, tmax=np.finfo(np.float32).max):
    r = orc.ORay()
    r.origin[:] = o
    r.t_min = tmin
    r.direction[:] = d
    r.t_max = float(tmax)
    return r


def test_sphere_cases(orc):
    """intersections.cuh:7-41: front hit, back hit from inside, t range, miss"""
    L = orc.lib()
    sph = np.array([0, 0, 0, 1], dtype=np.float32)
    rec = orc.OIntersection()
    assert L.orc_ray_sphere(C.byref(_ray(orc, (0, 0, 3), (0, 0, -1))), sph.ctypes.data, C.byref(rec)) == 1
    assert rec.t == 2 and rec.side == 0 and list(rec.normal) == [0, 0, 1]
    assert L.orc_ray_sphere(C.byref(_ray(orc, (0, 0, 0), (0, 0, -1))), sph.ctypes.data, C.byref(rec)) == 1
    assert rec.t == 1 and rec.side == 1 and list(rec.normal) == [0, 0, 1]  # flipped to face the ray
    assert L.orc_ray_sphere(C.byref(_ray(orc, (0, 0, 3), (0, 0, -1), tmax=1.5)), sph.ctypes.data, C.byref(rec)) == 0
    assert L.orc_ray_sphere(C.byref(_ray(orc, (0, 2, 3), (0, 0, -1))), sph.ctypes.data, C.byref(rec)) == 0
    # un-normalised direction: t scales with 1/|d|
    assert L.orc_ray_sphere(C.byref(_ray(orc, (0, 0, 3), (0, 0, -2))), sph.ctypes.data, C.byref(rec)) == 1
    assert rec.t == 1


def test_triangle_cases(orc):
    """intersections.cuh:49-85: edges u=0 / u+v=1 accepted, parallel rejected, t == t_max accepted"""
    L = orc.lib()
    p = [np.array(v, dtype=np.float32) for v in ((0, 0, 0), (1, 0, 0), (0, 1, 0))]
    rec = orc.OIntersection()

    def hit(o, d, **kw):
        return L.orc_ray_triangle(C.byref(_ray(orc, o, d, **kw)), p[0].ctypes.data, p[1].ctypes.data, p[2].ctypes.data,
                                  C.byref(rec))

    assert hit((0.25, 0.25, 1), (0, 0, -1)) == 1 and rec.t == 1 and rec.side == 0 and list(rec.normal) == [0, 0, 1]
    assert rec.material_id == 1
    assert hit((0.25, 0.25, -1), (0, 0, 1)) == 1 and rec.side == 1 and list(rec.normal) == [0, 0, -1]
    assert hit((0.0, 0.5, 1), (0, 0, -1)) == 1      # u == 0 edge
    assert hit((0.5, 0.5, 1), (0, 0, -1)) == 1      # u + v == 1 edge
    assert hit((0.75, 0.75, 1), (0, 0, -1)) == 0
    assert hit((0.25, 0.25, 1), (1, 0, 0)) == 0     # parallel
    assert hit((0.25, 0.25, 1), (0, 0, -1), tmax=1.0) == 1   # t == t_max accepted
    assert hit((0.25, 0.25, 1), (0, 0, -1), tmax=0.999) == 0
    assert hit((0.25, 0.25, 1), (0, 0, -1), tmin=1.5) == 0


def test_aabb_test_quirks(orc):
    """intersections.cuh:87-103: ignores the t range, accepts boxes behind the origin, empty box rejected"""
    L = orc.lib()
    box = _aabb((-1, -1, -1), (1, 1, 1))
    assert L.orc_ray_aabb(C.byref(_ray(orc, (0, 0, 5), (0, 0, -1))), box.ctypes.data) == 1
    assert L.orc_ray_aabb(C.byref(_ray(orc, (0, 0, 5), (0, 0, 1))), box.ctypes.data) == 1    # behind the ray
    assert L.orc_ray_aabb(C.byref(_ray(orc, (0, 0, 5), (0, 0, -1), tmax=0.1)), box.ctypes.data) == 1
    assert L.orc_ray_aabb(C.byref(_ray(orc, (3, 0, 5), (0, 0, -1))), box.ctypes.data) == 0
    assert L.orc_ray_aabb(C.byref(_ray(orc, (0.5, 0, 5), (0, 0, -1))), box.ctypes.data) == 1  # zero direction components
    empty = _aabb((np.finfo(np.float32).max,) * 3, (-np.finfo(np.float32).max,) * 3)
    assert L.orc_ray_aabb(C.byref(_ray(orc, (0, 0, 5), (0, 0, -1))), empty.ctypes.data) == 0


def test_bvh_grid_golden_and_structure(orc, golden_dir):
    kat = np.load(os.path.join(golden_dir, "kat.npz"))
    nodes, depth = orc.build_bvh(kat["bvh_grid_positions"], kat["bvh_grid_indices"])
    assert np.array_equal(nodes.view(np.uint8).reshape(-1, 32), kat["bvh_grid_nodes"])
    assert depth == int(kat["bvh_grid_depth"][0])
    t = len(kat["bvh_grid_indices"]) // 3
    assert len(nodes) == 2 * t - 1
    leaves = nodes[nodes["primitive_count"] != 0]
    assert len(leaves) == t and np.all(leaves["primitive_count"] == 1)
    assert sorted(leaves["first_child_or_primitive"].tolist()) == [3 * i for i in range(t)]
    # breadth-first: children adjacent and after their parent; parent box encloses both
    for i, n in enumerate(nodes):
        if n["primitive_count"] == 0:
            l = n["first_child_or_primitive"]
            assert l > i and l + 1 < len(nodes)
            for c in (nodes[l], nodes[l + 1]):
                assert np.all(n["aabb_min"] <= c["aabb_min"]) and np.all(n["aabb_max"] >= c["aabb_max"])


def test_bvh_errors(orc):
    pos = np.zeros((3, 3), dtype=np.float32)
    with pytest.raises(ValueError):
        orc.build_bvh(pos, np.zeros(0, dtype=np.uint32))          # empty mesh: bvh.cpp:200
    # coincident centroids: SAH side empty, bvh.cpp:84-85
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    with pytest.raises(ValueError):
        orc.build_bvh(pos, np.tile(np.array([0, 1, 2], dtype=np.uint32), 6))
