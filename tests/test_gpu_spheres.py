"""The sphere run that ends the object list in k_shade_fused: "select approximately, verify exactly" (round 4,
sphere_run_lanes in pt_kernels.hip) -- every lane collects its candidate spheres from approximate bounds on the root the
reference would accept, then runs the reference's own sequence (path_tracer.cu:84-96, intersections.cuh:7-41) for each
candidate with its object's data.  The bounds may only ever rule out what the reference rules out, so the frames must stay
the oracle's bits: scenes built to sit on the bounds' edges -- a camera inside nested and overlapping spheres, radii from
1e-3 to 1e3 (the subtraction in the discriminant loses the last digits there), coincident spheres (every hit of the pair
is a tie the LATER object wins, intersections.cuh:30), spheres touching, glass inside glass -- and the runs the per-lane
form must hand back to the object-by-object form: a scaled sphere (another t unit: transform.hpp:51-58 copies the ray's
range unscaled), more than eight spheres, spheres in front of a mesh.

The run IN FRONT of a mesh (k_spheres) has a form of its own for translated spheres, sphere_fold: the reference's sequence
object by object, with the matrix products that are sums with zeros written as the sums they are and the hit record's
normal finished once behind the loop.  Its exceptions (a -0.0 coordinate against a zero translation, a direction with a
zero component, anything not finite) go back to the plain form wavefront by wavefront: the rays of the last test sit on
exactly those."""
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
    elif kind in ("room_mesh", "room_mesh_scaled"):   # a room of wall spheres and balls IN FRONT of a mesh (config 2's shape): sphere_fold
        big = 1000.0
        ball((0.0, -big - 1.0, 0.0), big, "white")
        ball((0.0, 0.0, -big - 2.0), big, "white")
        ball((-big - 2.0, 0.0, 0.0), big, "red")
        ball((big + 2.0, 0.0, 0.0), big, "steel")
        ball((0.0, big + 2.5, 0.0), big, "white")
        ball((0.0, 0.0, big + 4.0), big, "white")       # behind the camera
        ball((0.9, -0.5, 0.3), 0.5, "glass")
        ball((0.9, -0.5, 0.3), 0.25, "water")           # inside the glass ball
        s.add_object(pkg.Sphere((-1.0, -0.6, 0.2), 0.4), glm.translate((0.0, 0.0, 0.0)), "mirror")   # centre in the sphere, zero translation
        s.add_object(pkg.Sphere((-1.0, -0.6, 0.2), 0.4), glm.translate((0.0, 0.0, 0.0)), "red")      # coincident: wins every tie
        if kind == "room_mesh_scaled":                  # one scaled sphere in the run: the whole run takes the plain form
            ball((0.0, 0.0, 0.0), 1.0, "steel", glm.compose([glm.scale(0.3), glm.translate((0.0, 0.6, -0.5))]))
        mesh = pkg.scenes.heightfield_mesh(33, 17, 2.0, 1.0, seed=5)
        s.add_mesh("ground", mesh)
        s.add_object(mesh, glm.translate((0.0, -0.8, -0.5)), "white")
        ball((0.0, -0.3, -0.5), 0.3, "glass")           # and a run that ends the list
    return s


@pytest.mark.parametrize("kind", ["nested", "overlap", "scaled", "many", "behind_mesh", "room_mesh", "room_mesh_scaled"])
def test_sphere_runs_against_the_oracle(pkg, orc, kind):
    scene = _soup(pkg, kind)
    flat = scene.build_scene()
    w, h, iters, mb = 96, 64, 3, 12
    ref = orc.render_streaming(flat, scene.camera, w, h, 0, iters, mb)
    got = _frames(pkg, scene, flat, w, h, iters, mb)
    off = _frames(pkg, scene, flat, w, h, iters, mb, params=(("sphere_lanes", 0), ("sphere_fold", 0)))
    serial = _frames(pkg, scene, flat, w, h, iters, mb, params=(("frames_in_flight", 1),))
    # "prefold" (round 5, default on): a run in front of a mesh is walked by the kernel that ends the bounce before (room_mesh:
    # sphere_fold there, room_mesh_scaled: the plain form); off, every bounce starts with k_spheres again
    apart = _frames(pkg, scene, flat, w, h, iters, mb, params=(("prefold", 0),))
    for k in ("color", "normal", "depth"):
        assert np.array_equal(got[k], ref[k]), (kind, k, int(np.sum(got[k] != ref[k])))
        assert np.array_equal(off[k], ref[k]), (kind, k, "object by object")
        assert np.array_equal(serial[k], ref[k]), (kind, k, "one frame in flight")
        assert np.array_equal(apart[k], ref[k]), (kind, k, "k_spheres as a pass of its own")
    assert got["rays"] == ref["rays"] == off["rays"] == apart["rays"]
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


def test_rays_on_the_exceptions_of_the_fold(pkg, orc):
    """ptc_intersect_rays on the room in front of a mesh with rays made for sphere_fold's exceptions: origins with -0.0 and
    +0.0 coordinates (the walls' translations have two zero components each, the camera of the frame tests sits at x = +0),
    directions with zero and negative-zero components, origins on a sphere's centre plane (a hit normal with a zero
    component), rays that start exactly on a wall, huge and tiny t_max."""
    scene = _soup(pkg, "room_mesh")
    flat = scene.build_scene()
    rng = np.random.default_rng(7)
    n = 16384
    rays = np.zeros((n, 8), dtype=np.float32)
    rays[:, 0:3] = rng.uniform(-1.8, 1.8, size=(n, 3)).astype(np.float32)
    rays[:, 1] = rng.uniform(-0.9, 2.3, size=n)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 3] = np.where(rng.uniform(size=n) < 0.5, 1e-4, 1e-5)
    rays[:, 7] = np.where(rng.uniform(size=n) < 0.7, np.finfo(np.float32).max, rng.uniform(0.05, 6.0, size=n))
    k = np.arange(n)
    for axis in range(3):
        rays[k % 7 == axis, axis] = 0.0                   # +0 origin coordinate
        rays[k % 7 == axis + 3, axis] = -0.0              # -0 origin coordinate
        rays[k % 11 == axis, 4 + axis] = 0.0              # zero direction component
        rays[k % 11 == axis + 3, 4 + axis] = -0.0
    rays[k % 13 == 0, 0:3] = np.float32([-1.0, -0.6, 0.2]) + np.float32([0.0, 0.0, 1.5])     # on the x and y planes of a sphere's centre
    rays[k % 13 == 0, 4:7] = np.float32([0.0, 0.0, -1.0])
    rays[k % 17 == 0, 1] = -1.0                           # exactly on the floor sphere's lowest tangent plane... and
    rays[k % 19 == 0, 0:3] = np.float32([0.9, -0.5, 0.3])  # at the centre of the nested glass balls
    recs, hit = orc.intersect_rays(flat, rays)
    for fold in (1, 0):
        with pkg.PathTracer() as pt:
            pt.set_param("sphere_fold", fold)
            pt.create_buffers((64, 64), flat)
            t, nrm, mat, side = pt.intersect_rays(rays)
        m = hit.astype(bool)
        assert 0.5 < m.mean() <= 1.0
        assert np.array_equal(t >= 0, m), fold
        assert np.array_equal(t[m].view(np.uint32), recs["t"][m].view(np.uint32)), fold
        assert np.array_equal(nrm[m].view(np.uint32), recs["normal"][m].view(np.uint32)), fold    # signs of zeros included
        assert np.array_equal(mat[m], recs["material_id"][m].astype(np.uint32)), fold
        assert np.array_equal(side[m], recs["side"][m]), fold


def test_scaled_instances_behind_closer_spheres(pkg, orc):
    """The work list the sphere kernel builds for the mesh launch behind it drops a ray whose instance boxes all START beyond
    the closest hit so far.  That is a comparison of world distances, and it is right for a mesh of any scale because the
    reference tests a mesh's triangles in WORLD space (path_tracer.cu:57-62: transform_point on the three vertices, the world
    ray) -- only a sphere's root is compared in object space (transform.hpp:51-58).  Pinned here: small spheres in front of
    a three-fold and a 0.4-fold instance, with the filter and without it, against the oracle."""
    glm = pkg.glmlite
    s = pkg.SceneDescription()
    s.resolution = (128, 64)
    s.camera = pkg.Camera(position=(0.0, 0.5, 4.0), rotation=(1.0, 0.0, 0.0, 0.0), vfov=float(np.radians(45)))
    s.add_material("white", pkg.DiffuseMateral((0.8, 0.8, 0.8)))
    s.add_material("red", pkg.DiffuseMateral((0.8, 0.2, 0.2)))
    s.add_material("steel", pkg.MetalMaterial((0.8, 0.8, 0.9), 0.1))
    s.add_object(pkg.Sphere((0, 0, 0), 0.35), glm.translate((-0.9, 0.45, 1.5)), "red")     # in front of the big instance
    s.add_object(pkg.Sphere((0, 0, 0), 0.35), glm.translate((0.9, 0.45, 1.5)), "steel")    # in front of the small one
    s.add_object(pkg.Sphere((0, 0, 0), 0.2), glm.translate((0.0, 0.2, 2.5)), "red")
    mesh = pkg.scenes.heightfield_mesh(33, 17, 1.0, 0.5, seed=9)
    s.add_mesh("ground", mesh)
    s.add_object(mesh, glm.compose([glm.scale(3.0), glm.translate((-1.2, -0.2, -1.5))]), "white")
    s.add_object(mesh, glm.compose([glm.scale(0.4), glm.translate((1.0, 0.0, 0.6))]), "white")
    s.add_object(pkg.Sphere((0, 0, 0), 0.25), glm.translate((0.0, 0.9, 0.0)), "steel")     # a run that ends the list
    flat = s.build_scene()
    w, h, iters, mb = 128, 64, 3, 8
    ref = orc.render_streaming(flat, s.camera, w, h, 0, iters, mb)
    got = _frames(pkg, s, flat, w, h, iters, mb)
    plain = _frames(pkg, s, flat, w, h, iters, mb, params=(("filter_rays", 0),))
    for k in ("color", "normal", "depth"):
        assert np.array_equal(plain[k], ref[k]), (k, "filter_rays 0", int(np.sum(plain[k] != ref[k])))
        assert np.array_equal(got[k], ref[k]), (k, int(np.sum(got[k] != ref[k])))
    assert got["rays"] == ref["rays"]


@pytest.mark.parametrize("shift", [(1000.0, 800.0, -1200.0), (9000.0, -7000.0, 4000.0), (0.0, 0.0, 0.0)])
@pytest.mark.parametrize("with_mesh", [False, True])
def test_small_spheres_far_from_the_origin(pkg, orc, shift, with_mesh):
    """Round 4's advisor finding: the candidate bounds of sphere_candidates took the INFLATED ball (radius + the float
    rounding of its centre) also for "surely hit" and for the upper bounds, which is the wrong way round -- a ray that
    grazes just outside a small sphere far from the origin was declared a sure hit, capped the list, and the wall behind it
    was dropped.  Every other case in this file has a zero centre or a zero translation (no rounding of the centre at all).
    Here: spheres with their centre in the Sphere struct AND a translation of 1e3 .. 1e4 (the world centre is not a float),
    a wall behind them, the camera about a unit away so that a good share of the rays graze; without a mesh the run ends the
    object list (k_shade_fused, caps of the earlier objects only), with a mesh at the end it stands in front of one
    (k_spheres); per-lane candidates on and off."""
    glm = pkg.glmlite
    s = pkg.SceneDescription()
    s.resolution = (160, 96)
    sx, sy, sz = shift
    s.camera = pkg.Camera(position=(sx + 0.3, sy + 0.25, sz + 1.6), rotation=(1.0, 0.0, 0.0, 0.0), vfov=float(np.radians(55)))
    s.add_material("white", pkg.DiffuseMateral((0.8, 0.8, 0.8)))
    s.add_material("red", pkg.DiffuseMateral((0.8, 0.2, 0.2)))
    s.add_material("mirror", pkg.MetalMaterial((0.9, 0.9, 0.9), 0.0))
    s.add_material("glass", pkg.DielectricMaterial(1.5))
    s.add_object(pkg.Sphere((0.3, 0.2, 0.1), 0.5), glm.translate(shift), "red")
    s.add_object(pkg.Sphere((-0.55, 0.25, 0.2), 0.35), glm.translate(shift), "glass")          # touches the first one's silhouette
    s.add_object(pkg.Sphere((0.31, 0.2, 0.1), 0.5), glm.translate(shift), "mirror")            # nearly coincident with the first
    s.add_object(pkg.Sphere((0.0, 0.0, -1003.0), 1000.0), glm.translate(shift), "white")       # the wall behind them
    s.add_object(pkg.Sphere((0.0, -1000.4, 0.0), 1000.0), glm.translate(shift), "white")       # a floor
    if with_mesh:
        mesh = pkg.scenes.heightfield_mesh(17, 9, 1.0, 0.5, seed=3)
        s.add_mesh("ground", mesh)
        s.add_object(mesh, glm.translate((sx, sy - 0.3, sz + 0.4)), "white")
    flat = s.build_scene()
    w, h, iters, mb = 160, 96, 2, 6
    ref = orc.render_streaming(flat, s.camera, w, h, 0, iters, mb)
    for params in ((), (("sphere_lanes", 0), ("sphere_fold", 0)), (("fused_shade", 0),)):
        got = _frames(pkg, s, flat, w, h, iters, mb, params=params)
        for k in ("color", "normal", "depth"):
            assert np.array_equal(got[k], ref[k]), (shift, with_mesh, params, k, int(np.sum(got[k] != ref[k])))
        assert got["rays"] == ref["rays"]
