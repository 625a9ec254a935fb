"""Object lists drawn at random against the oracle.  The launch plan of a bounce depends on the ORDER of the scene's objects
(ray_scene_intersection_test walks them in list order, path_tracer.cu:110-128): a sphere run in front of the first mesh is walked
by the kernel that ends the bounce before ("prefold") or by k_spheres, consecutive instances of one mesh share a launch
(k_traverse4m), a single instance takes k_traverse4 (with entry points when it opens the list), spheres between two meshes go
through k_spheres with the closest hit carried in the hit record, the run that ends the list belongs to k_shade_fused -- and
which of those forms a scene gets was so far decided by the handful of hand-built scenes of the other test files.  Here the
list is random: 1-7 objects, each a sphere (translated; or its centre kept in the Sphere; or scaled) or an instance of one of
two small meshes under a random similarity / affine transform, three material kinds, a random camera looking at the heap.
Every scene is rendered by the oracle (orc_render_streaming, the restatement of path_tracer.cu:389-477) and by the library with
the default schedule and with one schedule knob flipped; frames, ray totals and live counts must be the oracle's bits.
Fixed seeds: a failure names its seed."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, ITERS, MB = 96, 64, 3, 6
KNOBS = [(), (("prefold", 0),), (("merge_instances", 0),), (("frames_in_flight", 1),), (("fused_shade", 0),), (("filter_rays", 0), ("beam", 0)),
         (("sphere_lanes", 0), ("sphere_fold", 0)), (("frames_in_flight", 4), ("batch_frames", 2), ("traverse_waves", 64))]


def _random_scene(pkg, seed):
    rng = np.random.default_rng(1000 + seed)
    glm = pkg.glmlite
    s = pkg.SceneDescription()
    s.resolution = (W, H)
    mats = [("white", pkg.DiffuseMateral((0.8, 0.8, 0.8))), ("red", pkg.DiffuseMateral((0.8, 0.2, 0.2))),
            ("steel", pkg.MetalMaterial((0.8, 0.8, 0.9), 0.2)), ("mirror", pkg.MetalMaterial((0.9, 0.9, 0.9), 0.0)),
            ("glass", pkg.DielectricMaterial(1.5))]
    for name, m in mats:
        s.add_material(name, m)
    meshes = [pkg.scenes.displaced_sphere_mesh(10, 20), pkg.scenes.heightfield_mesh(17, 9, 2.0, 1.0, seed=3 + seed)]
    for k, m in enumerate(meshes):
        s.add_mesh(f"m{k}", m)

    def vec(lo, hi):
        return tuple(float(v) for v in rng.uniform(lo, hi, 3).astype(np.float32))

    n = int(rng.integers(1, 8))
    kinds = []
    last_mesh = None
    for _ in range(n):
        mat = mats[int(rng.integers(0, len(mats)))][0]
        roll = rng.random()
        if roll < 0.45:                                   # a translated sphere (the reference's scenes: centre in the transform)
            r = float(np.float32(rng.choice([0.15, 0.4, 0.8, 30.0])))
            c = vec(-1.2, 1.2)
            if r > 10.0:                                  # a wall: pushed out so that its surface passes near the heap
                axis = int(rng.integers(0, 3))
                c = tuple((-(r + 1.5) if rng.random() < 0.5 else (r + 1.5)) if a == axis else 0.0 for a in range(3))
            if rng.random() < 0.25:
                s.add_object(pkg.Sphere(c, r), glm.translate((0.0, 0.0, 0.0)), mat)       # centre kept in the Sphere
            else:
                s.add_object(pkg.Sphere((0.0, 0.0, 0.0), r), glm.translate(c), mat)
            kinds.append("s")
        elif roll < 0.55:                                 # a scaled sphere: no fold, no per-lane candidates for its run
            s.add_object(pkg.Sphere((0.0, 0.0, 0.0), 1.0), glm.compose([glm.scale(float(rng.uniform(0.2, 0.6))), glm.translate(vec(-1.0, 1.0))]), mat)
            kinds.append("S")
        else:                                             # an instance; often of the mesh before it (a run for k_traverse4m)
            k = last_mesh if (last_mesh is not None and rng.random() < 0.5) else int(rng.integers(0, 2))
            last_mesh = k
            parts = []
            if rng.random() < 0.5:
                parts.append(glm.rotate(np.float32(rng.uniform(0.0, 3.0)), vec(-1.0, 1.0)))
            parts.append(glm.scale(vec(0.3, 0.8)) if rng.random() < 0.3 else glm.scale(float(rng.uniform(0.3, 0.9))))
            parts.append(glm.translate(vec(-1.0, 1.0)))
            s.add_object(meshes[k], glm.compose(parts), mat)
            kinds.append("ab"[k])
    eye = vec(-0.6, 0.6)
    s.camera = pkg.Camera(position=(eye[0], eye[1] + 0.3, 3.2), rotation=(1.0, 0.0, 0.0, 0.0), vfov=float(np.radians(rng.uniform(40, 65))))
    return s, "".join(kinds)


# (PT_RANDOM_SEEDS=N: a longer hunt with the same generator -- 600 seeds passed on the round's final code)
@pytest.mark.parametrize("seed", range(int(os.environ.get("PT_RANDOM_SEEDS", "40"))))
def test_random_object_lists_against_the_oracle(pkg, orc, seed):
    scene, kinds = _random_scene(pkg, seed)
    flat = scene.build_scene(distinct_meshes=True)
    ref = orc.render_streaming(flat, scene.camera, W, H, 0, ITERS, MB)
    for params in ((), KNOBS[1 + seed % (len(KNOBS) - 1)]):
        with pkg.PathTracer(device=0, max_bounces=MB) as pt:
            for k, v in params:
                pt.set_param(k, v)
            pt.create_buffers((W, H), flat)
            pt.max_iterations = ITERS
            for _ in range(ITERS):
                pt.path_trace(scene.camera)
            for k in ("color", "normal", "depth"):
                got = pt.download(k)
                assert np.array_equal(got, ref[k]), (seed, kinds, params, k, int(np.sum(got != ref[k])))
            st = pt.stats()
            assert st["rays_total"] == ref["rays"], (seed, kinds, params)
            assert st["last_live"] == [int(x) for x in ref["live"][-1]], (seed, kinds, params)
