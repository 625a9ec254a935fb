"""The GPU BVH builder (pt_bvh_gpu.hip, ptc_build_bvh_device) against the host builder and the oracle: the same
nodes, bit for bit -- same split decisions (pt_bvh_rules.hpp), level order = the reference's breadth-first numbering
(accelerators/bvh.cpp:228-250)."""
import copy
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _flat_grid(nx, nz):
    m_x, m_z = np.meshgrid(np.arange(nx, dtype=np.float32), np.arange(nz, dtype=np.float32), indexing="xy")
    pos = np.stack([m_x, 0.01 * ((m_x * 7 + m_z * 3) % 5), m_z], axis=-1).reshape(-1, 3)
    v = (np.arange(nz - 1)[:, None] * nx + np.arange(nx - 1)[None, :]).astype(np.uint32)
    idx = np.stack([np.stack([v, v + nx, v + 1], -1), np.stack([v + 1, v + nx, v + nx + 1], -1)], axis=2).reshape(-1)
    return pos, idx


def _soup(pkg, n, seed):
    rng = np.random.default_rng(seed)
    centre = rng.uniform(-50, 50, size=(n, 1, 3))
    pos = (centre + rng.normal(scale=0.7, size=(n, 3, 3))).astype(np.float32).reshape(-1, 3)
    return pkg.Mesh(pos, np.arange(3 * n, dtype=np.uint32))


def _meshes(pkg):
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 0, 1], [3, 1, 0], [2, 2, 2], [5, 5, 5], [6, 5, 5], [5, 6, 7]], dtype=np.float32)
    yield "one_triangle", pkg.Mesh(tri[:3].copy(), np.array([0, 1, 2], dtype=np.uint32))
    for n in (2, 3, 4, 5, 6, 9):
        yield f"soup{n}", _soup(pkg, n, n)
    yield "grid9x5", pkg.scenes.heightfield_mesh(9, 5, 2.0, 1.0, seed=1)
    yield "grid33x17", pkg.scenes.heightfield_mesh(33, 17, 8.0, 4.0, seed=7)
    yield "sphere24x48", pkg.scenes.displaced_sphere_mesh(24, 48)
    yield "grid_ties", pkg.Mesh(*_flat_grid(17, 9))          # equal centroids along the split axis: the tie rule
    yield "grid_ties_large", pkg.Mesh(*_flat_grid(129, 65))
    yield "soup20k", _soup(pkg, 20_000, 3)
    yield "grid257x129", pkg.scenes.heightfield_mesh(257, 129, 8.0, 4.0, seed=3)


def test_device_builder_equals_host_builder_and_oracle(pkg, orc):
    with pkg.PathTracer() as pt:
        for name, mesh in _meshes(pkg):
            got, got_depth = pt.build_bvh(mesh)
            want, want_depth = pkg.bvh_from_mesh(mesh)
            assert len(got) == len(want) == 2 * mesh.triangle_count() - 1, name
            assert got_depth == want_depth, name
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), name
            if mesh.triangle_count() <= 25_000:
                ref, ref_depth = orc.build_bvh(mesh.positions, mesh.indices)
                assert ref_depth == got_depth and np.array_equal(got.view(np.uint8), ref.view(np.uint8)), name


def test_device_builder_benchmark_mesh(pkg):
    """bench.py's 1,000,000-triangle mesh: 1,999,999 nodes, depth 25, the host builder's bytes"""
    mesh = list(pkg.scenes.heightfield_scene((64, 64)).mesh_map_.values())[0]
    want, want_depth = pkg.bvh_from_mesh(mesh)
    with pkg.PathTracer() as pt:
        got, got_depth = pt.build_bvh(mesh)
    assert len(got) == 1_999_999 and got_depth == want_depth
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


def test_device_builder_errors(pkg):
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    with pkg.PathTracer() as pt:
        with pytest.raises(pkg.PtcError) as e:   # coincident centroids: the reference panics (bvh.cpp:84-85)
            pt.build_bvh(pkg.Mesh(pos, np.tile(np.array([0, 1, 2], dtype=np.uint32), 6)))
        assert e.value.code == pkg._capi.PTC_ERR_BVH
        with pytest.raises(pkg.PtcError) as e:   # the same through the many-triangle path (bins by atomics)
            pt.build_bvh(pkg.Mesh(pos, np.tile(np.array([0, 1, 2], dtype=np.uint32), 100)))
        assert e.value.code == pkg._capi.PTC_ERR_BVH
        # ... and deep inside an otherwise fine mesh: 40 coincident triangles among 2000 others
        soup = _soup(pkg, 2000, 11)
        dup = np.concatenate([soup.indices, np.tile(soup.indices[:3], 40)])
        with pytest.raises(pkg.PtcError) as e:
            pt.build_bvh(pkg.Mesh(soup.positions, dup))
        assert e.value.code == pkg._capi.PTC_ERR_BVH
        rc = pkg.lib().ptc_build_bvh(soup.positions.ctypes.data_as(C.POINTER(C.c_float)), len(soup.positions),
                                     dup.ctypes.data_as(C.POINTER(C.c_uint32)), len(dup),
                                     np.zeros(2 * (len(dup) // 3), dtype=pkg.scene_description.BVH_NODE_DTYPE).ctypes.data_as(
                                         C.POINTER(pkg._capi.ptc_bvh_node)), None)
        assert rc == pkg._capi.PTC_ERR_BVH       # the host builder fails the same way
        with pytest.raises(pkg.PtcError) as e:
            pt.build_bvh(pkg.Mesh(pos, np.array([0, 1, 7], dtype=np.uint32)))
        assert e.value.code == pkg._capi.PTC_ERR_INVALID
        # the context still works
        nodes, depth = pt.build_bvh(pkg.Mesh(pos, np.array([0, 1, 2], dtype=np.uint32)))
        assert len(nodes) == 1 and depth == 0 and nodes["primitive_count"][0] == 1


@pytest.mark.parametrize("on_device", [1, 0])
def test_scene_without_bvh_renders_the_same(pkg, on_device):
    """ptc_upload_scene builds the BVH itself when the scene brings none -- on the GPU by default, on the host with
    bvh_build_on_device = 0: the image is the one of the caller-supplied (host-built) BVH"""
    scene = pkg.scenes.heightfield_scene((160, 96), nx=129, nz=65)
    bare = scene.build_scene()
    assert bare.bvh is None          # build_scene leaves the BVH to the library, as the reference's does to bvh_from_mesh
    flat = copy.copy(bare)
    flat.bvh, _ = pkg.bvh_from_mesh(list(scene.mesh_map_.values())[0])

    def render(f, param):
        with pkg.PathTracer(device=0, max_bounces=6) as pt:
            if param is not None:
                pt.set_param("bvh_build_on_device", param)
            pt.create_buffers((160, 96), f)
            for _ in range(3):
                pt.path_trace(scene.camera)
            return pt.download("color"), pt.stats(), pt.upload_times()

    want, want_stats, t0 = render(flat, None)
    got, got_stats, t1 = render(bare, on_device)
    assert t0["bvh_on_device"] == 0 and t0["bvh_build_ms"] == 0.0
    assert t1["bvh_on_device"] == on_device and t1["bvh_build_ms"] > 0.0
    assert np.array_equal(got, want) and got_stats["rays_total"] == want_stats["rays_total"]
    assert got_stats["bvh_node_count"] == want_stats["bvh_node_count"] and got_stats["bvh_max_depth"] == want_stats["bvh_max_depth"]


def _layouts(pkg, flat, on_device, size=(64, 64)):
    with pkg.PathTracer() as pt:
        pt.set_param("layout_on_device", on_device)
        pt.set_param("bvh_build_on_device", on_device)
        pt.create_buffers(size, flat)
        t = pt.upload_times()
        assert t["layout_on_device"] == (on_device if len(flat.indices) else 0)
        return {k: pt.download_layout(k) for k in pt.LAYOUTS}, pt.stats()


def test_device_layouts_equal_host_layouts(pkg):
    """The traversal data derived on the GPU -- four-wide quantised nodes in depth-first preorder, parent boxes and
    triangle records in depth-first leaf order, two-child records -- are byte for byte what pt_scene_host.cpp
    builds (one source for the decisions: pt_layout_rules.hpp), for meshes of several shapes and instance counts."""
    glm = pkg.glmlite
    scenes = []
    for name, mesh in _meshes(pkg):
        sc = pkg.SceneDescription()
        sc.add_material("m", pkg.DiffuseMateral((0.5, 0.5, 0.5)))
        sc.add_mesh(name, mesh)
        sc.add_object(mesh, glm.compose([glm.translate((0.5, -1.0, 2.0))]), "m")
        if name in ("sphere24x48", "soup5"):   # a second, rotated and non-uniformly scaled instance
            sc.add_object(mesh, glm.compose([glm.rotate(np.float32(0.6), (0.3, 1.0, 0.2)), glm.scale((0.7, 0.4, 0.9))]), "m")
        scenes.append((name, sc))
    scenes.append(("cornell_bunny", pkg.scenes.cornell_bunny((64, 64))))
    scenes.append(("spheres_only", pkg.scenes.cornell_spheres((64, 64))))
    for name, sc in scenes:
        flat = sc.build_scene()
        dev, dev_stats = _layouts(pkg, flat, 1)
        host, host_stats = _layouts(pkg, flat, 0)
        for k in dev:
            assert dev[k].shape == host[k].shape and np.array_equal(dev[k], host[k]), (name, k)
        assert dev_stats["bvh_node_count"] == host_stats["bvh_node_count"], name
        assert dev_stats["stack_capacity"] == host_stats["stack_capacity"], name


def test_device_layouts_benchmark_mesh(pkg):
    flat = pkg.scenes.heightfield_scene((64, 64)).build_scene()
    dev, _ = _layouts(pkg, flat, 1)
    host, _ = _layouts(pkg, flat, 0)
    assert len(dev["bvh"]) == 1_999_999 * 32 and len(dev["tris"]) == 1_000_001 * 64
    for k in dev:
        assert np.array_equal(dev[k], host[k]), k


def test_caller_bvh_in_another_order_goes_to_the_host(pkg):
    """a caller's BVH numbered children-after-parents but not depth by depth: the layouts come from the host code"""
    scene = pkg.scenes.heightfield_scene((96, 64), nx=33, nz=17)
    mesh = list(scene.mesh_map_.values())[0]
    nodes, _ = pkg.bvh_from_mesh(mesh)
    # renumber depth-first: children still follow their parent and stay adjacent; same topology, same image
    res = [nodes[0].copy()]
    work = [(0, 0)]
    while work:
        src, dst = work.pop()
        if nodes[src]["primitive_count"] == 0:
            f = nodes[src]["first_child_or_primitive"]
            at = len(res)
            res.append(nodes[f].copy()); res.append(nodes[f + 1].copy())
            res[dst]["first_child_or_primitive"] = at
            work.append((f + 1, at + 1)); work.append((f, at))
    shuffled = np.array(res, dtype=nodes.dtype)
    flat = scene.build_scene()
    a = copy.copy(flat); a.bvh = nodes
    b = copy.copy(flat); b.bvh = shuffled

    def render(f):
        with pkg.PathTracer(device=0, max_bounces=5) as pt:
            pt.create_buffers((96, 64), f)
            t = pt.upload_times()
            for _ in range(2):
                pt.path_trace(scene.camera)
            return pt.download("color"), t

    img_a, t_a = render(a)
    img_b, t_b = render(b)
    assert t_a["layout_on_device"] == 1 and t_b["layout_on_device"] == 0
    assert np.array_equal(img_a, img_b)
