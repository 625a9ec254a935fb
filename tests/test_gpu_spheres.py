"""The sphere run that ends the object list in k_shade_fused: "select approximately, verify exactly" (round 4,
sphere_run_lanes in pt_kernels.hip) -- every lane collects its candidate spheres from approximate bounds on the root the
reference would accept, then runs the reference's own sequence (path_tracer.cu:84-96, intersections.cuh:7-41) for each
candidate with its object's data.  The bounds may only ever rule out what the reference rules out, so the frames must stay
the oracle's bits: scenes built to sit on the bounds' edges -- a camera inside nested and overlapping spheres, radii from
1e-3 to 1e3 (the subtraction in the discriminant loses the last digits there), coincident spheres (every hit of the pair
is a tie the LATER object wins, intersections.cuh:30), spheres touching, glass inside glass -- and the runs the per-lane
form must hand back to the object-by-object form: a scaled sphere (another t unit: transform.hpp:51-58 copies the ray's
range unscaled), more than eight spheres, spheres in front of a mesh."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(pkg, scene, flat, w, h, iters, mb, params=()):
    with pkg.PathTracer(device=0, max_bounces=mb) as pt:
        for k, v in params:
            pt.set_param(k, v)
        pt.create_buffers((w, h), flat)
        pt.max_iterations = iters
        for _ in range(iters):
            pt.path_trace(scene.camera)
        out = {k: pt.download(k) for k in ("color", "normal", "depth")}
        out["rays"] = pt.stats()["rays_total"]
        out["live"] = pt.stats()["last_live"]
    return out


def _soup(pkg, kind):
    glm = pkg.glmlite
    s = pkg.SceneDescription()
    s.resolution = (96, 64)
    s.camera = pkg.Camera(position=(0.0, 0.4, 3.0), rotation=(1.0, 0.0, 0.0, 0.0), vfov=float(np.radians(50)))
    for name, mat in (("white", pkg.DiffuseMateral((0.8, 0.8, 0.8))), ("red", pkg.DiffuseMateral((0.8, 0.2, 0.2))),
                      ("steel", pkg.MetalMaterial((0.8, 0.8, 0.9), 0.15)), ("mirror", pkg.MetalMaterial((0.9, 0.9, 0.9), 0.0)),
                      ("glass", pkg.DielectricMaterial(1.5)), ("water", pkg.DielectricMaterial(1.33))):
        s.add_material(name, mat)

    def ball(center, radius, material, transform=None):
        # the reference's scenes put the centre into the transform (json_parser.cpp:133-159); some here keep it in the sphere
        s.add_object(pkg.Sphere((0.0, 0.0, 0.0), radius), glm.translate(center) if transform is None else transform, material)

    if kind == "nested":      # camera inside a glass shell inside a huge sphere; a floor of radius 1000 just below
        ball((0.0, -1000.5, 0.0), 1000.0, "white")
        ball((0.0, 0.4, 3.0), 0.6, "glass")           # around the camera
        ball((0.0, 0.4, 3.0), 0.3, "water")           # and another inside it
        ball((0.0, 0.0, 0.0), 0.5, "steel")
        ball((0.0, 0.0, 0.0), 0.5, "red")             # coincident with the one before: wins every tie
        ball((1.0, 0.0, 0.0), 0.5, "glass")           # touches its neighbour in one point
        ball((-1.2, 0.2, 0.3), 0.7, "mirror")
        ball((0.2, 0.9, -0.4), 1e-3, "red")           # a speck
    elif kind == "overlap":   # eight overlapping spheres, centres kept in the Sphere struct for some
        for k in range(8):
            c = (0.45 * k - 1.6, 0.25 * ((k * 5) % 3) - 0.2, -0.3 * (k % 2))
            if k % 2:
                s.add_object(pkg.Sphere(c, 0.45), glm.translate((0.0, 0.0, 0.0)), ("glass", "steel", "white", "mirror")[k % 4])
            else:
                ball(c, 0.45, ("glass", "steel", "white", "mirror")[k % 4])
    elif kind == "scaled":    # one sphere under a scale: the whole run goes object by object
        ball((0.0, -1000.5, 0.0), 1000.0, "white")
        ball((0.0, 0.0, 0.0), 1.0, "glass", glm.compose([glm.scale(0.5), glm.translate((0.6, 0.0, 0.0))]))
        ball((-0.6, 0.0, 0.0), 0.5, "steel")
    elif kind == "many":      # nine spheres: more than a lane's candidate byte holds
        ball((0.0, -1000.5, 0.0), 1000.0, "white")
        for k in range(8):
            ball((0.5 * k - 1.75, 0.1 * (k % 3), -0.2 * k), 0.3, ("red", "steel", "glass", "mirror")[k % 4])
    elif kind == "behind_mesh":   # a mesh first, the spheres behind it in the list (config 3's shape), one sphere in front too
        ball((0.0, 0.3, 1.2), 0.25, "glass")
        mesh = pkg.scenes.heightfield_mesh(33, 17, 4.0, 2.0, seed=5)
        s.add_mesh("ground", mesh)
        s.add_object(mesh, glm.translate((0.0, -0.5, 0.0)), "white")
        ball((0.0, 0.0, 0.0), 0.5, "mirror")
        ball((0.9, 0.1, 0.2), 0.45, "glass")
        ball((0.9, 0.1, 0.2), 0.45, "red")            # coincident again
        ball((-0.9, 0.0, 0.0), 0.5, "steel")
    return s


@pytest.mark.parametrize("kind", ["nested", "overlap", "scaled", "many", "behind_mesh"])
def test_sphere_runs_against_the_oracle(pkg, orc, kind):
    scene = _soup(pkg, kind)
    flat = scene.build_scene()
    w, h, iters, mb = 96, 64, 3, 12
    ref = orc.render_streaming(flat, scene.camera, w, h, 0, iters, mb)
    got = _frames(pkg, scene, flat, w, h, iters, mb)
    off = _frames(pkg, scene, flat, w, h, iters, mb, params=(("sphere_lanes", 0),))
    serial = _frames(pkg, scene, flat, w, h, iters, mb, params=(("frames_in_flight", 1),))
    for k in ("color", "normal", "depth"):
        assert np.array_equal(got[k], ref[k]), (kind, k, int(np.sum(got[k] != ref[k])))
        assert np.array_equal(off[k], ref[k]), (kind, k, "object by object")
        assert np.array_equal(serial[k], ref[k]), (kind, k, "one frame in flight")
    assert got["rays"] == ref["rays"] == off["rays"]
    assert got["live"] == [int(x) for x in ref["live"][-1]]


def test_rays_from_inside_and_along_the_surface_of_a_huge_sphere(pkg, orc):
    """The Cornell box's walls are spheres of radius 1000: rays start a hair above such a surface and run nearly along
    it, which is where the approximate discriminant is worth least.  Camera on the floor sphere looking along it."""
    glm = pkg.glmlite
    s = pkg.SceneDescription()
    s.resolution = (128, 48)
    s.camera = pkg.Camera(position=(0.0, 1e-3, 2.0), rotation=(1.0, 0.0, 0.0, 0.0), vfov=float(np.radians(70)))
    s.add_material("white", pkg.DiffuseMateral((0.8, 0.8, 0.8)))
    s.add_material("mirror", pkg.MetalMaterial((0.95, 0.95, 0.95), 0.0))
    s.add_material("glass", pkg.DielectricMaterial(1.5))
    s.add_object(pkg.Sphere((0, 0, 0), 1000.0), glm.translate((0.0, -1000.0, 0.0)), "mirror")     # the floor, a mirror: grazing reflections
    s.add_object(pkg.Sphere((0, 0, 0), 1000.0), glm.translate((0.0, 0.0, -1003.0)), "white")      # a wall
    s.add_object(pkg.Sphere((0, 0, 0), 1000.0), glm.translate((1003.0, 0.0, 0.0)), "white")
    s.add_object(pkg.Sphere((0, 0, 0), 0.5), glm.translate((0.0, 0.5, 0.0)), "glass")             # rests on the floor: touches it
    s.add_object(pkg.Sphere((0, 0, 0), 0.25), glm.translate((0.9, 0.25, 0.5)), "white")
    flat = s.build_scene()
    w, h, iters, mb = 128, 48, 4, 16
    ref = orc.render_streaming(flat, s.camera, w, h, 0, iters, mb)
    got = _frames(pkg, s, flat, w, h, iters, mb)
    for k in ("color", "normal", "depth"):
        assert np.array_equal(got[k], ref[k]), (k, int(np.sum(got[k] != ref[k])))
    assert got["rays"] == ref["rays"]
