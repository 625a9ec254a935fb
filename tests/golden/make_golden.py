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
    main()
