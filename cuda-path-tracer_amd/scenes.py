"""Synthetic scenes for the configurations of BASELINE.md section 3.  The reference's own meshes are
git-LFS pointers (assets/models/*.obj) and its sphere scene does not parse (three_balls.json uses an
unknown transform key), so every scene here is generated; all generators are deterministic integer-hash
noise, no RNG state."""
import math

import numpy as np

from . import glmlite as glm
from .scene_description import (Camera, DielectricMaterial, DiffuseMateral, Mesh, MetalMaterial, SceneDescription,
                                Sphere)


def _hash_u32(a):
    a = np.asarray(a, dtype=np.uint64)
    m = np.uint64(0xFFFFFFFF)
    a = ((a + np.uint64(0x7ED55D16)) + (a << np.uint64(12))) & m
    a = ((a ^ np.uint64(0xC761C23C)) ^ (a >> np.uint64(19))) & m
    a = ((a + np.uint64(0x165667B1)) + (a << np.uint64(5))) & m
    a = ((a + np.uint64(0xD3A2646C)) ^ (a << np.uint64(9))) & m
    a = ((a + np.uint64(0xFD7046C5)) + (a << np.uint64(3))) & m
    a = ((a ^ np.uint64(0xB55A4F09)) ^ (a >> np.uint64(16))) & m
    return a


def hash_noise(i, j, seed):
    """[-1, 1) noise of two integer lattice coordinates."""
    h = _hash_u32(_hash_u32(np.asarray(i, dtype=np.uint64) * np.uint64(73856093) % np.uint64(1 << 32))
                  ^ _hash_u32((np.asarray(j, dtype=np.uint64) + np.uint64(seed) * np.uint64(19349663)) % np.uint64(1 << 32)))
    return (h.astype(np.float64) / 2147483648.0 - 1.0).astype(np.float32)


def heightfield_mesh(nx=1001, nz=501, size_x=8.0, size_z=4.0, seed=7):
    """(nx-1)*(nz-1)*2 triangles over [-size_x/2, size_x/2] x [-size_z/2, size_z/2]:
    y = 0.15 sin(3x) cos(5z) + 0.02 noise.  1001 x 501 vertices -> exactly 1,000,000 triangles."""
    ix, iz = np.meshgrid(np.arange(nx), np.arange(nz), indexing="xy")  # [nz, nx]
    x = (ix.astype(np.float64) / (nx - 1) - 0.5) * size_x
    z = (iz.astype(np.float64) / (nz - 1) - 0.5) * size_z
    y = 0.15 * np.sin(3.0 * x) * np.cos(5.0 * z) + 0.02 * hash_noise(ix, iz, seed).astype(np.float64)
    positions = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float32)
    v = (iz[:-1, :-1] * nx + ix[:-1, :-1]).astype(np.uint32)
    # two triangles per quad, counter-clockwise seen from +y
    tri0 = np.stack([v, v + nx, v + 1], axis=-1)
    tri1 = np.stack([v + 1, v + nx, v + nx + 1], axis=-1)
    indices = np.stack([tri0, tri1], axis=2).reshape(-1).astype(np.uint32)
    return Mesh(positions, indices)


def displaced_sphere_mesh(n_lat=108, n_lon=324, radius=0.5, seed=1):
    """UV sphere, n_lat x n_lon quads -> 2*n_lat*n_lon triangles (69,984 at the defaults; the stand-in for
    the reference's bunny), radially displaced by 3 octaves of lattice noise.  Pole rows are displaced per
    vertex too, so no two triangle centroids coincide (the builder rejects coincident centroids)."""
    lat = np.arange(n_lat + 1)
    lon = np.arange(n_lon)
    jj, ii = np.meshgrid(lon, lat, indexing="xy")  # [n_lat+1, n_lon]
    theta = (ii.astype(np.float64) + 0.5 * (ii == 0) * 1e-3 - 0.5 * (ii == n_lat) * 1e-3) / n_lat * math.pi
    phi = jj.astype(np.float64) / n_lon * 2.0 * math.pi
    disp = np.zeros_like(theta)
    for octave in range(3):
        s = 1 << octave
        disp += (0.06 / s) * hash_noise(ii // max(1, 8 // s), jj // max(1, 8 // s), seed + octave).astype(np.float64)
        disp += (0.004 / s) * hash_noise(ii, jj, seed + 10 + octave).astype(np.float64)
    r = radius * (1.0 + disp)
    x = r * np.sin(theta) * np.cos(phi)
    y = r * np.cos(theta)
    z = r * np.sin(theta) * np.sin(phi)
    positions = np.stack([x, y, z], axis=-1).reshape(-1, 3).astype(np.float32)
    a = (ii[:-1] * n_lon + jj[:-1]).astype(np.uint32)
    b = (ii[:-1] * n_lon + (jj[:-1] + 1) % n_lon).astype(np.uint32)
    c = a + n_lon
    d = b + n_lon
    tri0 = np.stack([a, b, c], axis=-1)
    tri1 = np.stack([b, d, c], axis=-1)
    indices = np.stack([tri0, tri1], axis=2).reshape(-1).astype(np.uint32)
    return Mesh(positions, indices)


def _camera_from_look_at(frm, at, up=(0.0, 1.0, 0.0), vfov_deg=45.0):
    m = glm.look_at(frm, at, up)
    return Camera(position=tuple(float(v) for v in m[3, 0:3]), rotation=tuple(float(v) for v in glm.quat_from_matrix(m)),
                  vfov=float(np.float32(math.radians(vfov_deg))))


def _add_box_and_balls(scene, with_balls=True):
    """Open-top, open-front box made of five radius-1000 spheres plus three unit-test spheres
    (diffuse / metal fuzz 0.2 / glass 1.5), sky-lit (the reference has no emitters)."""
    scene.add_material("floor", DiffuseMateral((0.73, 0.73, 0.73)))
    scene.add_material("back", DiffuseMateral((0.73, 0.73, 0.73)))
    scene.add_material("left", DiffuseMateral((0.65, 0.05, 0.05)))
    scene.add_material("right", DiffuseMateral((0.12, 0.45, 0.15)))
    scene.add_material("ball_diffuse", DiffuseMateral((0.1, 0.2, 0.5)))
    scene.add_material("ball_metal", MetalMaterial((0.8, 0.6, 0.2), 0.2))
    scene.add_material("ball_glass", DielectricMaterial(1.5))
    big = 1000.0
    scene.add_object(Sphere((0, 0, 0), big), glm.translate((0.0, -big - 1.0, 0.0)), "floor")
    scene.add_object(Sphere((0, 0, 0), big), glm.translate((0.0, 0.0, -big - 2.0)), "back")
    scene.add_object(Sphere((0, 0, 0), big), glm.translate((-big - 2.0, 0.0, 0.0)), "left")
    scene.add_object(Sphere((0, 0, 0), big), glm.translate((big + 2.0, 0.0, 0.0)), "right")
    if with_balls:
        scene.add_object(Sphere((0, 0, 0), 0.5), glm.translate((0.0, -0.5, -0.6)), "ball_diffuse")
        scene.add_object(Sphere((0, 0, 0), 0.5), glm.translate((1.1, -0.5, 0.1)), "ball_metal")
        scene.add_object(Sphere((0, 0, 0), 0.5), glm.translate((-1.1, -0.5, 0.2)), "ball_glass")


def cornell_spheres(resolution=(256, 256)):
    """Config 1: spheres only."""
    s = SceneDescription()
    _add_box_and_balls(s)
    s.camera = _camera_from_look_at((0.0, 0.0, 4.0), (0.0, -0.3, 0.0), vfov_deg=45.0)
    s.resolution = tuple(resolution)
    s.spp = 16
    return s


def cornell_bunny(resolution=(1280, 720), n_lat=108, n_lon=324):
    """Config 2: the box + two instances of one displaced-sphere mesh (69,984 triangles), placed like
    the reference's bunny.json:37-57 (one unit instance, one scaled by 0.5)."""
    s = SceneDescription()
    _add_box_and_balls(s, with_balls=False)
    s.add_material("bunny", DiffuseMateral((0.8, 0.8, 0.5)))
    s.add_material("bunny2", MetalMaterial((0.6, 0.4, 0.8), 0.1))
    s.add_material("glass", DielectricMaterial(1.5))
    mesh = s.add_mesh("models/displaced_sphere.obj", displaced_sphere_mesh(n_lat, n_lon))
    s.add_object(mesh, glm.translate((0.9, -0.5, -0.4)), "bunny")
    s.add_object(mesh, glm.compose([glm.scale(0.5), glm.translate((-0.9, -0.75, 0.3))]), "bunny2")
    s.add_object(Sphere((0, 0, 0), 0.3), glm.translate((0.0, -0.7, 0.9)), "glass")
    s.camera = _camera_from_look_at((0.0, 0.2, 4.0), (0.0, -0.4, 0.0), vfov_deg=45.0)
    s.resolution = tuple(resolution)
    s.spp = 1
    return s


def heightfield_scene(resolution=(1920, 1080), nx=1001, nz=501):
    """Configs 3-5: 1,000,000-triangle heightfield (diffuse 0.7) + three spheres above it."""
    s = SceneDescription()
    s.add_material("ground", DiffuseMateral((0.7, 0.7, 0.7)))
    s.add_material("ball_diffuse", DiffuseMateral((0.1, 0.2, 0.5)))
    s.add_material("ball_metal", MetalMaterial((0.8, 0.6, 0.2), 0.2))
    s.add_material("ball_glass", DielectricMaterial(1.5))
    mesh = s.add_mesh("models/heightfield.obj", heightfield_mesh(nx, nz))
    s.add_object(mesh, glm.identity(), "ground")
    s.add_object(Sphere((0, 0, 0), 0.5), glm.translate((0.0, 0.7, 0.0)), "ball_diffuse")
    s.add_object(Sphere((0, 0, 0), 0.5), glm.translate((1.3, 0.7, 0.3)), "ball_metal")
    s.add_object(Sphere((0, 0, 0), 0.5), glm.translate((-1.3, 0.7, 0.3)), "ball_glass")
    s.camera = _camera_from_look_at((0.0, 2.5, 5.0), (0.0, 0.0, 0.0), vfov_deg=50.0)
    s.resolution = tuple(resolution)
    s.spp = 1
    return s
