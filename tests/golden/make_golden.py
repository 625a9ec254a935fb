"""Regenerates the golden fixtures under tests/golden/ from the CPU oracle (oracle/liboracle.so).

    python tests/golden/make_golden.py

The reference cannot run in this pipeline (CUDA), and it ships no result-pinning tests or images for the
hot path, so these vectors pin the ORACLE (and through it the HIP path), not the CUDA renderer:
"parity unpinned" with respect to the reference, see DESIGN.md.  Fixtures are data only: inputs are
rebuilt by the deterministic scene generators in cuda-path-tracer_amd/scenes.py."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
orc = graft.load_oracle()


def golden_scenes():
    """name -> (SceneDescription, width, height); shared with the tests."""
    return {
        "spheres": (pkg.scenes.cornell_spheres((32, 32)), 32, 32),
        "mesh": (pkg.scenes.cornell_bunny((48, 32), n_lat=8, n_lon=16), 48, 32),
        "heightfield": (pkg.scenes.heightfield_scene((48, 32), nx=33, nz=17), 48, 32),
    }


def multimesh_scenes():
    """Scenes with SEVERAL distinct meshes (ptc_mesh_range / OScene::meshes; the reference keeps one mesh per scene,
    scene_description.cpp:42,95, so these pin the extension, not the reference): name -> (SceneDescription, w, h, max_bounces)."""
    glm = pkg.glmlite
    two = pkg.scenes.cornell_spheres((48, 32))            # walls and three small spheres
    a = pkg.scenes.displaced_sphere_mesh(16, 32)
    b = pkg.scenes.heightfield_mesh(33, 17, 2.0, 1.0, seed=4)
    two.add_mesh("a", a)
    two.add_mesh("b", b)
    two.add_material("ma", pkg.DiffuseMateral((0.8, 0.3, 0.2)))
    two.add_material("mb", pkg.MetalMaterial((0.7, 0.7, 0.9), 0.1))
    two.add_object(a, glm.compose([glm.scale(0.5), glm.translate((-0.7, 0.2, 0.4))]), "ma")
    two.add_object(b, glm.compose([glm.rotate(np.float32(0.4), (0.0, 1.0, 0.0)), glm.translate((0.2, -0.9, 0.0))]), "mb")
    two.add_object(a, glm.compose([glm.rotate(np.float32(0.6), (0.3, 1.0, 0.2)), glm.scale((0.4, 0.25, 0.5)),
                                   glm.translate((0.8, 0.5, -0.3))]), "mb")
    # three meshes, two of them placed exactly on top of each other (every hit there is a t == t_max tie, won by the
    # later object), a glass mesh, a sphere between the mesh objects
    ties = pkg.SceneDescription()
    ties.resolution = (40, 40)
    ties.camera = pkg.Camera(position=(0.0, 0.3, 3.2), rotation=(1.0, 0.0, 0.0, 0.0), vfov=float(np.radians(45)))
    c = pkg.scenes.displaced_sphere_mesh(10, 20)
    d = pkg.scenes.heightfield_mesh(17, 9, 3.0, 2.0, seed=9)
    e = pkg.scenes.displaced_sphere_mesh(6, 12)
    for name, mat in (("red", pkg.DiffuseMateral((0.8, 0.2, 0.2))), ("grey", pkg.DiffuseMateral((0.6, 0.6, 0.6))),
                      ("steel", pkg.MetalMaterial((0.8, 0.8, 0.85), 0.05)), ("glass", pkg.DielectricMaterial(1.5))):
        ties.add_material(name, mat)
    for name, mesh in (("c", c), ("d", d), ("e", e)):
        ties.add_mesh(name, mesh)
    place = glm.compose([glm.scale(0.6), glm.translate((-0.5, 0.3, 0.0))])
    ties.add_object(c, place, "red")
    ties.add_object(d, glm.translate((0.0, -0.6, 0.0)), "grey")
    ties.add_object(pkg.Sphere((0, 0, 0), 0.3), glm.translate((0.6, 0.0, 0.6)), "steel")
    ties.add_object(c, place, "steel")                    # coincident with the first object: wins every tie
    ties.add_object(e, glm.compose([glm.scale(0.5), glm.translate((0.5, 0.5, -0.2))]), "glass")
    return {"two_meshes": (two, 48, 32, 6), "three_meshes_ties": (ties, 40, 40, 8)}


def main_multimesh():
    out = {}
    rng = np.random.default_rng(21)
    for name, (scene, w, h, mb) in multimesh_scenes().items():
        flat = scene.build_scene(distinct_meshes=True)
        r = orc.render_streaming(flat, scene.camera, w, h, 0, 3, mb)
        for k in ("color", "normal", "depth", "live"):
            out[f"{name}_{k}"] = r[k]
        out[f"{name}_rays"] = np.array([r["rays"]], dtype=np.uint64)
        n = 1500
        origin = rng.uniform(-3, 3, size=(n, 3)).astype(np.float32)
        target = rng.uniform(-1.0, 1.0, size=(n, 3)).astype(np.float32)
        dirs = target - origin
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        rays = np.zeros((n, 8), dtype=np.float32)
        rays[:, 0:3], rays[:, 3], rays[:, 4:7], rays[:, 7] = origin, 1e-4, dirs, np.finfo(np.float32).max
        recs, hit = orc.intersect_rays(flat, rays)
        out[f"{name}_probe_rays"] = rays
        out[f"{name}_probe_hit"] = hit
        out[f"{name}_probe_t"] = recs["t"]
        out[f"{name}_probe_normal"] = recs["normal"]
        out[f"{name}_probe_material"] = recs["material_id"].astype(np.uint32)
        out[f"{name}_probe_side"] = recs["side"]
    np.savez_compressed(os.path.join(HERE, "multimesh.npz"), **out)
    print("wrote", os.path.join(HERE, "multimesh.npz"))


INTERLEAVED = dict(w=48, h=72, block_rows=8, max_bounces=8, iterations=3, worlds=(2, 3, 8))


def interleaved_scene():
    """The small heightfield at 48 x 72: nine blocks of 8 rows, so that with 8 ranks rank 0 owns two blocks and the
    others one, with 3 ranks three each, with 2 ranks five and four."""
    c = INTERLEAVED
    return pkg.scenes.heightfield_scene((c["w"], c["h"]), nx=33, nz=17)


def main_interleaved():
    """tests/golden/interleaved.npz: the multi-GPU split that bench.py --gpus N times (interleaved row blocks, paths
    numbered per rank, "slot_offset" = rank * W * H), every rank of worlds 2 / 3 / 8 rendered by the oracle
    (orc_render_streaming_interleaved) and assembled into frame order; per-rank live counts and ray totals."""
    c = INTERLEAVED
    scene = interleaved_scene()
    flat = scene.build_scene()
    w, h = c["w"], c["h"]
    out = {}
    for world in c["worlds"]:
        parts = [orc.render_interleaved(flat, scene.camera, w, h, r, world, c["block_rows"], r * w * h, 0, c["iterations"],
                                        c["max_bounces"]) for r in range(world)]
        for k in ("color", "normal", "depth"):
            out[f"w{world}_{k}"] = pkg.bands.assemble_interleaved([p[k] for p in parts], h, world, c["block_rows"])
        out[f"w{world}_live"] = np.stack([p["live"] for p in parts])
        out[f"w{world}_rays"] = np.array([p["rays"] for p in parts], dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "interleaved.npz"), **out)
    print("wrote", os.path.join(HERE, "interleaved.npz"))


def main():
    out = {}
    for name, (scene, w, h) in golden_scenes().items():
        flat = scene.build_scene()
        for mb in (4, 8, 50):
            r = orc.render_streaming(flat, scene.camera, w, h, 0, 4, mb)
            out[f"{name}_mb{mb}_color"] = r["color"]
            out[f"{name}_mb{mb}_normal"] = r["normal"]
            out[f"{name}_mb{mb}_depth"] = r["depth"]
            out[f"{name}_mb{mb}_live"] = r["live"]
            out[f"{name}_mb{mb}_rays"] = np.array([r["rays"]], dtype=np.uint64)
            # first iteration alone (iteration index 0) as well
            r0 = orc.render_streaming(flat, scene.camera, w, h, 0, 1, mb)
            out[f"{name}_mb{mb}_color_it0"] = r0["color"]
        m = orc.render_megakernel(flat, scene.camera, w, h, 0, 2, 8)
        out[f"{name}_mega_color"] = m["color"]
        out[f"{name}_mega_rays"] = np.array([m["rays"]], dtype=np.uint64)
    # denoiser on the heightfield G-buffer (64x64, 2 iterations accumulated)
    scene = pkg.scenes.heightfield_scene((64, 64), nx=33, nz=17)
    flat = scene.build_scene()
    r = orc.render_streaming(flat, scene.camera, 64, 64, 0, 2, 8)
    den, touched = orc.denoise(scene.camera, 64, 64, r["color"], r["normal"], r["depth"])
    out["denoise_in_color"] = r["color"]
    out["denoise_in_normal"] = r["normal"]
    out["denoise_in_depth"] = r["depth"]
    out["denoise_out"] = den
    out["denoise_touched_oob"] = touched
    out["preview_rgba"] = orc.preview(r["color"], 64, 64, 0)
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **out)

    # scalar KATs: generate_ray on a 4x3 image, intersection edge cases, BVH of a 2x2-quad grid
    kat = {}
    cam = orc.OCamera()
    cam.position[:] = [0, 0, 0]
    cam.rotation_wxyz[:] = [1, 0, 0, 0]
    cam.vfov = float(np.float32(np.radians(60.0)))
    g = orc.OGPUCamera()
    orc.lib().orc_to_gpu_camera(C.byref(cam), 4, 3, C.byref(g))
    rays = np.zeros((3, 4, 8), dtype=np.float32)
    for y in range(3):
        for x in range(4):
            ray = orc.ORay()
            orc.lib().orc_generate_ray(C.byref(g), x + 0.5, y + 0.5, C.byref(ray))
            rays[y, x] = np.frombuffer(bytes(ray), dtype=np.float32)
    kat["generate_ray_4x3"] = rays
    mesh = pkg.scenes.heightfield_mesh(3, 3, 2.0, 2.0, seed=3)
    nodes, depth = orc.build_bvh(mesh.positions, mesh.indices)
    kat["bvh_grid_positions"] = mesh.positions
    kat["bvh_grid_indices"] = mesh.indices
    kat["bvh_grid_nodes"] = nodes.view(np.uint8).reshape(-1, 32)
    kat["bvh_grid_depth"] = np.array([depth], dtype=np.uint32)
    np.savez_compressed(os.path.join(HERE, "kat.npz"), **kat)
    print("wrote", os.path.join(HERE, "frames.npz"), os.path.join(HERE, "kat.npz"))


if __name__ == "__main__":
    if sys.argv[1:] == ["multimesh"]:     # only the multi-mesh fixtures (the others stay as committed)
        main_multimesh()
    elif sys.argv[1:] == ["interleaved"]:
        main_interleaved()
    else:
        main()
        main_multimesh()
        main_interleaved()
