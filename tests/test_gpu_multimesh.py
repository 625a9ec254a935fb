"""Several distinct meshes in one scene (ptc_mesh_range; SURVEY section 8 f2 -- the reference keeps one mesh per scene,
scene_description.cpp:42,95, so there is no reference behaviour to compare with: parity unpinned).  What is checked:
a scene of several meshes returns, per ray, the closest hit over its objects in the reference's object order
(ray_scene_intersection_test, path_tracer.cu:110-128) -- computed from single-object scenes through the tested
one-mesh path -- and every traversal schedule renders the same image."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(pkg, size=(96, 64)):
    glm = pkg.glmlite
    sc = pkg.scenes.cornell_spheres(size)            # walls and three small spheres
    a = pkg.scenes.displaced_sphere_mesh(16, 32)
    b = pkg.scenes.heightfield_mesh(33, 17, 2.0, 1.0, seed=4)
    sc.add_mesh("a", a)
    sc.add_mesh("b", b)
    sc.add_material("ma", pkg.DiffuseMateral((0.8, 0.3, 0.2)))
    sc.add_material("mb", pkg.MetalMaterial((0.7, 0.7, 0.9), 0.1))
    sc.add_object(a, glm.compose([glm.scale(0.5), glm.translate((-0.7, 0.2, 0.4))]), "ma")
    sc.add_object(b, glm.compose([glm.rotate(np.float32(0.4), (0.0, 1.0, 0.0)), glm.translate((0.2, -0.9, 0.0))]), "mb")
    sc.add_object(a, glm.compose([glm.rotate(np.float32(0.6), (0.3, 1.0, 0.2)), glm.scale((0.4, 0.25, 0.5)),
                                  glm.translate((0.8, 0.5, -0.3))]), "mb")
    return sc, a, b


def _frames(pkg, scene, flat, variant, iters=3, mb=6, params=()):
    w, h = scene.resolution if scene.resolution[0] else (96, 64)
    with pkg.PathTracer(device=0, max_bounces=mb) as pt:
        for k, v in params:
            pt.set_param(k, v)
        pt.create_buffers((96, 64), flat)
        pt.set_trace_variant(variant)
        for _ in range(iters):
            pt.path_trace(scene.camera)
        return {k: pt.download(k) for k in ("color", "normal", "depth")}, pt.stats()


def test_every_schedule_renders_the_same_image(pkg):
    scene, a, b = _scene(pkg)
    flat = scene.build_scene(distinct_meshes=True)
    assert flat.mesh_ranges is not None and len(flat.mesh_ranges) == 2
    assert [int(o["index"]) for o in flat.objects if o["type"] == 1] == [0, 1, 0]
    base, base_stats = _frames(pkg, scene, flat, 0)
    assert base_stats["triangle_count"] == a.triangle_count() + b.triangle_count()
    for variant, params in ((1, ()), (3, ()), (3, (("frames_in_flight", 6), ("batch_frames", 3))),
                            (3, (("layout_on_device", 0), ("bvh_build_on_device", 0)))):
        got, stats = _frames(pkg, scene, flat, variant, params=params)
        for k in base:
            assert np.array_equal(got[k], base[k]), (variant, params, k)
        assert stats["rays_total"] == base_stats["rays_total"]
    # the two meshes are both seen: the image differs from the reference's one-mesh reading of the same description
    one, _ = _frames(pkg, scene, scene.build_scene(), 3)
    assert not np.array_equal(one["color"], base["color"])


def test_one_mesh_through_the_table_equals_the_plain_scene(pkg):
    """a table of one mesh is the reference's scene: same image as without a table"""
    scene = pkg.scenes.cornell_bunny((96, 64), n_lat=12, n_lon=24)
    plain = scene.build_scene()
    table = scene.build_scene(distinct_meshes=True)
    assert len(table.mesh_ranges) == 1
    want, _ = _frames(pkg, scene, plain, 3)
    got, _ = _frames(pkg, scene, table, 3)
    for k in want:
        assert np.array_equal(got[k], want[k]), k


def test_rays_against_single_object_scenes(pkg):
    """closest hit over the objects in object order == the single-mesh path applied object by object"""
    glm = pkg.glmlite
    a = pkg.scenes.displaced_sphere_mesh(16, 32)
    b = pkg.scenes.heightfield_mesh(33, 17, 2.0, 1.0, seed=4)
    placements = [(a, glm.compose([glm.scale(0.5), glm.translate((-0.7, 0.2, 0.4))])),
                  (b, glm.compose([glm.rotate(np.float32(0.4), (0.0, 1.0, 0.0)), glm.translate((0.2, -0.9, 0.0))])),
                  (a, glm.compose([glm.scale(0.5), glm.translate((-0.55, 0.25, 0.4))])),   # overlaps the first: near ties
                  (b, glm.compose([glm.rotate(np.float32(0.4), (0.0, 1.0, 0.0)), glm.translate((0.2, -0.9, 0.0))]))]  # coincident: exact ties

    def scene_of(items):
        sc = pkg.SceneDescription()
        for k in range(4):
            sc.add_material(f"m{k}", pkg.DiffuseMateral((0.1 * (k + 1), 0.5, 0.5)))
        for k, (mesh, tr) in items:
            sc.add_object(mesh, tr, f"m{k}")
        return sc

    rng = np.random.default_rng(5)
    n = 60_000
    origin = rng.uniform(-3, 3, size=(n, 3)).astype(np.float32)
    target = rng.uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32)
    d = target - origin
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = origin; rays[:, 3] = 1e-4; rays[:, 4:7] = d; rays[:, 7] = np.finfo(np.float32).max

    def shoot(flat, variant=3):
        with pkg.PathTracer() as pt:
            pt.create_buffers((32, 32), flat)
            pt.set_trace_variant(variant)
            return pt.intersect_rays(rays)

    full = scene_of(list(enumerate(placements))).build_scene(distinct_meshes=True)
    t, nrm, mat, side = shoot(full)
    # object by object, each alone in a one-mesh scene; a later object wins when its hit is not farther (t <= t_max,
    # intersections.cuh:80-81)
    best_t = np.full(n, -1.0, dtype=np.float32)
    best_n = np.zeros((n, 3), dtype=np.float32); best_m = np.zeros(n, dtype=np.uint32); best_s = np.zeros(n, dtype=np.uint8)
    for k, item in enumerate(placements):
        sc = scene_of([(k, item)])
        sc.add_mesh("only", item[0])
        tk, nk, mk, sk = shoot(sc.build_scene())
        take = (tk >= 0) & ((best_t < 0) | (tk <= best_t))
        best_t[take] = tk[take]; best_n[take] = nk[take]; best_m[take] = mk[take]; best_s[take] = sk[take]
    hit = best_t >= 0
    assert 0.2 < hit.mean() < 0.95
    assert np.array_equal(t >= 0, hit)
    assert np.array_equal(t[hit], best_t[hit]) and np.array_equal(nrm[hit], best_n[hit])
    assert np.array_equal(mat[hit], best_m[hit]) and np.array_equal(side[hit], best_s[hit])
    assert (best_m[hit] == 3).sum() > 100 and (best_m[hit] == 1).sum() == 0   # the coincident copy wins every tie
    for variant in (0, 1):
        tv, nv, mv, sv = shoot(full, variant)
        assert np.array_equal(tv, t) and np.array_equal(nv[hit], nrm[hit]) and np.array_equal(mv[hit], mat[hit])


def test_mesh_table_validation(pkg):
    scene, a, b = _scene(pkg)
    flat = scene.build_scene(distinct_meshes=True)
    bad = copy.copy(flat)
    bad.objects = flat.objects.copy()
    mesh_rows = [i for i, o in enumerate(bad.objects) if o["type"] == 1]
    bad.objects["index"][mesh_rows[0]] = 7
    with pkg.PathTracer() as pt:
        with pytest.raises(pkg.PtcError) as e:
            pt.create_buffers((32, 32), bad)
        assert e.value.code == pkg._capi.PTC_ERR_INVALID and "mesh index" in str(e.value)
        bad2 = copy.copy(flat)
        bad2.mesh_ranges = flat.mesh_ranges.copy()
        bad2.mesh_ranges[1, 1] += 5          # vertex range beyond the array
        with pytest.raises(pkg.PtcError) as e:
            pt.create_buffers((32, 32), bad2)
        assert e.value.code == pkg._capi.PTC_ERR_INVALID
        pt.create_buffers((32, 32), flat)    # and the context still takes a good scene


def test_run_of_instances_in_one_launch(pkg):
    """five instances of one mesh, two of them coincident (exact ties) and two overlapping: walked by ONE launch per
    bounce (k_traverse4m) == one launch per instance == the single-object scenes merged in object order"""
    glm = pkg.glmlite
    a = pkg.scenes.displaced_sphere_mesh(16, 32)
    t0 = glm.compose([glm.scale(0.5), glm.translate((-0.7, 0.2, 0.4))])
    placements = [t0, glm.compose([glm.scale(0.5), glm.translate((-0.55, 0.25, 0.4))]), t0,
                  glm.compose([glm.rotate(np.float32(0.6), (0.3, 1.0, 0.2)), glm.scale((0.4, 0.25, 0.5)), glm.translate((0.8, 0.5, -0.3))]),
                  glm.compose([glm.scale(0.3), glm.translate((0.1, -0.4, 0.9))])]

    def scene_of(items):
        sc = pkg.SceneDescription()
        sc.add_mesh("a", a)
        for k in range(len(placements)):
            sc.add_material(f"m{k}", pkg.DiffuseMateral((0.15 * (k + 1), 0.5, 0.5)))
        for k, tr in items:
            sc.add_object(a, tr, f"m{k}")
        return sc

    rng = np.random.default_rng(9)
    n = 80_000
    origin = rng.uniform(-3, 3, size=(n, 3)).astype(np.float32)
    target = rng.uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32)
    d = target - origin
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # a share of axis-parallel rays: set aside by the fast walk, redone for all instances by the launch's epilogue
    d[: n // 20] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, n // 20)] * rng.choice([-1.0, 1.0], (n // 20, 1)).astype(np.float32)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = origin; rays[:, 3] = 1e-4; rays[:, 4:7] = d; rays[:, 7] = np.finfo(np.float32).max
    full = scene_of(list(enumerate(placements)))
    flat = full.build_scene()

    # per-ray results of the frame pipeline are not exposed; the image is: render with both launch plans and the
    # reference-order kernel, and compare ray by ray through ptc_intersect_rays (one launch per object there)
    def render(params, variant=3):
        with pkg.PathTracer(device=0, max_bounces=5) as pt:
            for k, v in params:
                pt.set_param(k, v)
            pt.create_buffers((128, 96), flat)
            pt.set_trace_variant(variant)
            cam = pkg.Camera(position=(0.0, 0.0, 4.0), vfov=0.8)
            for _ in range(3):
                pt.path_trace(cam)
            out = {k: pt.download(k) for k in ("color", "normal", "depth")}
            redone = sum(pt.profile()["slow_rays"])
            return out, pt.stats(), redone

    base, base_stats, _ = render((), variant=0)
    for params in ((), (("merge_instances", 0),), (("frames_in_flight", 6), ("batch_frames", 3)),
                   (("frames_in_flight", 2), ("batch_frames", 1), ("split_idle", 1)), (("debug_force_slow", 1),),
                   (("debug_force_slow", 2),)):
        got, stats, redone = render(params)
        for k in base:
            assert np.array_equal(got[k], base[k]), (params, k)
        assert stats["rays_total"] == base_stats["rays_total"], params
        if params and params[0][0] == "debug_force_slow":
            assert redone > 0
    with pkg.PathTracer() as pt:
        pt.create_buffers((32, 32), flat)
        t, nrm, mat, side = pt.intersect_rays(rays)
    best_t = np.full(n, -1.0, dtype=np.float32); best_m = np.zeros(n, dtype=np.uint32)
    for k, tr in enumerate(placements):
        with pkg.PathTracer() as pt:
            pt.create_buffers((32, 32), scene_of([(k, tr)]).build_scene())
            tk, nk, mk, sk = pt.intersect_rays(rays)
        take = (tk >= 0) & ((best_t < 0) | (tk <= best_t))
        best_t[take] = tk[take]; best_m[take] = mk[take]
    hit = best_t >= 0
    assert np.array_equal(t >= 0, hit) and np.array_equal(t[hit], best_t[hit]) and np.array_equal(mat[hit], best_m[hit])
    assert (best_m[hit] == 0).sum() == 0 and (best_m[hit] == 2).sum() > 100   # the coincident later copy wins every tie


def test_more_traversal_launches_per_frame_than_cursor_sets(pkg):
    """Three meshes that cannot share a launch x 50 bounces = 150 traversal launches per frame (the reference's own
    bounce cap, path_tracer.cu:27).  Round 2 gave each launch of a frame its own fetch-cursor set out of 128 and
    recycled stale ones beyond that: the dynamic share of the rays was then never walked (advisor finding).  Now a
    launch's last wavefront zeroes the set it used and launch n takes set n % kWorkSlots, so the count is unbounded.
    Closed Cornell box: most paths live for all 50 bounces."""
    scene, a, b = _scene(pkg)
    flat = scene.build_scene(distinct_meshes=True)
    base, base_stats = _frames(pkg, scene, flat, 0, iters=2, mb=50)
    assert base_stats["last_live"][10] > 100   # rays are still alive when the cursor sets have gone round several times
    for params in ((), (("frames_in_flight", 6), ("batch_frames", 3)), (("merge_instances", 0),),
                   (("traverse_waves", 8),)):   # 8 wavefronts: no static share, every ray comes from a cursor
        got, stats = _frames(pkg, scene, flat, 3, iters=2, mb=50, params=params)
        for k in base:
            assert np.array_equal(got[k], base[k]), (params, k)
        assert stats["rays_total"] == base_stats["rays_total"] and stats["last_live"] == base_stats["last_live"], params


def test_intersect_rays_with_many_mesh_objects(pkg):
    """ptc_intersect_rays launches one traversal per mesh object: 150 instances (round 2 stopped silently after 128
    launches and returned the closest hit over a truncated object list)."""
    glm = pkg.glmlite
    mesh = pkg.scenes.displaced_sphere_mesh(6, 12)
    sc = pkg.SceneDescription()
    sc.add_mesh("m", mesh)
    sc.add_material("first", pkg.DiffuseMateral((0.5, 0.5, 0.5)))
    sc.add_material("last", pkg.DiffuseMateral((0.9, 0.1, 0.1)))
    count = 150
    for k in range(count):
        x, y = (k % 15) - 7.0, (k // 15) - 4.5
        sc.add_object(mesh, glm.compose([glm.scale(0.9), glm.translate((x, y, 0.0))]), "last" if k >= 128 else "first")
    sc.add_object(pkg.Sphere((0, 0, 0), 0.2), glm.translate((0.0, 0.0, 2.0)), "first")   # a sphere run behind them all
    flat = sc.build_scene()
    rng = np.random.default_rng(11)
    n = 30_000
    origin = np.stack([rng.uniform(-8, 8, n), rng.uniform(-5.5, 5.5, n), np.full(n, 6.0)], axis=1).astype(np.float32)
    d = np.stack([rng.uniform(-0.2, 0.2, n), rng.uniform(-0.2, 0.2, n), np.full(n, -1.0)], axis=1).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = origin; rays[:, 3] = 1e-4; rays[:, 4:7] = d; rays[:, 7] = np.finfo(np.float32).max
    with pkg.PathTracer() as pt:
        pt.create_buffers((32, 32), flat)
        got = pt.intersect_rays(rays)
        pt.set_trace_variant(0)
        want = pt.intersect_rays(rays)
    hit = want[0] >= 0
    assert 0.2 < hit.mean() < 0.9
    mats = {name: i for i, name in enumerate(sorted(["first", "last"]))}
    assert (want[2][hit] == mats["last"]).sum() > 500    # objects beyond the 128th are hit
    assert np.array_equal(got[0], want[0])
    for k in (1, 2, 3):
        assert np.array_equal(got[k][hit], want[k][hit]), k


def _golden_multimesh(golden_dir):
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(golden_dir, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.multimesh_scenes(), np.load(os.path.join(golden_dir, "multimesh.npz"))


@pytest.mark.parametrize("name", ["two_meshes", "three_meshes_ties"])
def test_multimesh_frames_and_rays_against_the_oracle(pkg, orc, golden_dir, name):
    """Round 2 checked multi-mesh scenes only against the HIP path itself (the oracle had no mesh table).  Now: frames
    of every schedule == the oracle's frames (tests/golden/multimesh.npz, regenerated by tests/test_oracle_multimesh.py
    on the CPU) bit for bit, and ptc_intersect_rays == orc.intersect_rays on the committed probe rays."""
    scenes, golden = _golden_multimesh(golden_dir)
    scene, w, h, mb = scenes[name]
    flat = scene.build_scene(distinct_meshes=True)
    for variant, params in ((3, ()), (3, (("frames_in_flight", 6), ("batch_frames", 3))), (3, (("merge_instances", 0),)),
                            (3, (("fused_shade", 0),)), (3, (("layout_on_device", 0), ("bvh_build_on_device", 0))), (1, ()), (0, ())):
        with pkg.PathTracer(device=0, max_bounces=mb) as pt:
            for k, v in params:
                pt.set_param(k, v)
            pt.create_buffers((w, h), flat)
            pt.set_trace_variant(variant)
            pt.max_iterations = 3
            for _ in range(3):
                pt.path_trace(scene.camera)
            for k in ("color", "normal", "depth"):
                assert np.array_equal(pt.download(k), golden[f"{name}_{k}"]), (variant, params, k)
            st = pt.stats()
            assert st["rays_total"] == int(golden[f"{name}_rays"][0]), (variant, params)
            assert st["last_live"][:mb] == [int(v) for v in golden[f"{name}_live"][-1]], (variant, params)
    rays = golden[f"{name}_probe_rays"]
    m = golden[f"{name}_probe_hit"].astype(bool)
    with pkg.PathTracer() as pt:
        pt.create_buffers((32, 32), flat)
        for variant in (3, 1, 0):
            pt.set_trace_variant(variant)
            t, nrm, mat, side = pt.intersect_rays(rays)
            assert np.array_equal(t >= 0, m), variant
            assert np.array_equal(t[m], golden[f"{name}_probe_t"][m]) and np.array_equal(nrm[m], golden[f"{name}_probe_normal"][m]), variant
            assert np.array_equal(mat[m], golden[f"{name}_probe_material"][m]) and np.array_equal(side[m], golden[f"{name}_probe_side"][m]), variant
    # and against a fresh oracle run (the fixture is not stale)
    ref = orc.render_streaming(flat, scene.camera, w, h, 0, 3, mb)
    assert np.array_equal(ref["color"], golden[f"{name}_color"])


def test_run_of_instances_against_the_oracle(pkg, orc):
    """the five-instance scene of test_run_of_instances_in_one_launch (ONE launch per bounce walks all five: k_traverse4m)
    against the oracle itself: frames, and 80,000 rays one by one"""
    glm = pkg.glmlite
    a = pkg.scenes.displaced_sphere_mesh(16, 32)
    t0 = glm.compose([glm.scale(0.5), glm.translate((-0.7, 0.2, 0.4))])
    placements = [t0, glm.compose([glm.scale(0.5), glm.translate((-0.55, 0.25, 0.4))]), t0,
                  glm.compose([glm.rotate(np.float32(0.6), (0.3, 1.0, 0.2)), glm.scale((0.4, 0.25, 0.5)), glm.translate((0.8, 0.5, -0.3))]),
                  glm.compose([glm.scale(0.3), glm.translate((0.1, -0.4, 0.9))])]
    sc = pkg.SceneDescription()
    sc.add_mesh("a", a)
    for k, tr in enumerate(placements):
        sc.add_material(f"m{k}", pkg.DiffuseMateral((0.15 * (k + 1), 0.5, 0.5)))
        sc.add_object(a, tr, f"m{k}")
    flat = sc.build_scene()
    cam = pkg.Camera(position=(0.0, 0.0, 4.0), vfov=0.8)
    ref = orc.render_streaming(flat, cam, 128, 96, 0, 3, 5)
    for params in ((), (("merge_instances", 0),), (("frames_in_flight", 6), ("batch_frames", 3))):
        with pkg.PathTracer(device=0, max_bounces=5) as pt:
            for k, v in params:
                pt.set_param(k, v)
            pt.create_buffers((128, 96), flat)
            pt.max_iterations = 3
            for _ in range(3):
                pt.path_trace(cam)
            for k in ("color", "normal", "depth"):
                assert np.array_equal(pt.download(k), ref[k]), (params, k)
            assert pt.stats()["rays_total"] == ref["rays"]
    rng = np.random.default_rng(9)
    n = 80_000
    origin = rng.uniform(-3, 3, size=(n, 3)).astype(np.float32)
    d = rng.uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32) - origin
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[: n // 20] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, n // 20)] * rng.choice([-1.0, 1.0], (n // 20, 1)).astype(np.float32)
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = origin; rays[:, 3] = 1e-4; rays[:, 4:7] = d; rays[:, 7] = np.finfo(np.float32).max
    recs, hit = orc.intersect_rays(flat, rays)
    m = hit.astype(bool)
    with pkg.PathTracer() as pt:
        pt.create_buffers((32, 32), flat)
        t, nrm, mat, side = pt.intersect_rays(rays)
    assert np.array_equal(t >= 0, m) and 0.05 < m.mean() < 0.95
    assert np.array_equal(t[m], recs["t"][m]) and np.array_equal(nrm[m], recs["normal"][m])
    assert np.array_equal(mat[m], recs["material_id"][m].astype(np.uint32)) and np.array_equal(side[m], recs["side"][m])
