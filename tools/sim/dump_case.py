"""Dump the benchmark scene's reference BVH, triangles and a set of representative rays (primary rays of the camera and
diffuse bounce rays from their hit points) for tools/sim/walk_sim.cpp.  CPU only (uses the oracle)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/sim_case.bin"
full = len(sys.argv) > 2 and sys.argv[2] == "full"     # packet_sim.cpp: every primary ray of the 1920 x 1080 frame, no bounce rays
W, H = (1920, 1080) if full else (480, 270)
scene = pkg.scenes.heightfield_scene((W, H)); flat = scene.build_scene()
flat.bvh, depth = pkg.bvh_from_mesh(list(scene.mesh_map_.values())[0])
# mesh only: drop the spheres (objects[0] is the mesh with an identity transform)
import copy
mesh_only = copy.copy(flat)
mesh_only.objects = flat.objects[:1].copy(); mesh_only.object_material_indices = flat.object_material_indices[:1].copy()
mesh_only.spheres = flat.spheres[:0].copy()
cam = scene.camera
# primary rays: through pixel centres, like generate_ray (ray_gen.cu:34-61)
glm = pkg.glmlite
import math
q = np.array(cam.rotation, dtype=np.float64); pos = np.array(cam.position, dtype=np.float64)
w, x, y, z = q
R = np.array([[1-2*(y*y+z*z), 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), 1-2*(x*x+z*z), 2*(y*z-w*x)], [2*(x*z-w*y), 2*(y*z+w*x), 1-2*(x*x+y*y)]])
vh = 2*math.tan(cam.vfov/2); vw = W/H*vh
px, py = np.meshgrid(np.arange(W)+0.5, np.arange(H)+0.5)
u = px/(W-1); v = (H-py)/(H-1)
d = np.stack([-vw/2+u*vw, -vh/2+v*vh, -np.ones_like(u)], axis=-1).reshape(-1, 3) @ R.T
d /= np.linalg.norm(d, axis=1, keepdims=True)
n = len(d)
rays = np.zeros((n, 8), dtype=np.float32)
rays[:, 0:3] = pos; rays[:, 3] = 1e-4; rays[:, 4:7] = d; rays[:, 7] = np.finfo(np.float32).max
all_rays = [rays]
rng = np.random.default_rng(1)
cur = rays
for bounce in range(0 if full else 3):
    recs, hit = orc.intersect_rays(mesh_only, cur)
    m = hit.astype(bool)
    p = cur[m, 0:3] + cur[m, 4:7] * recs["t"][m][:, None]
    nrm = recs["normal"][m]
    r = rng.normal(size=nrm.shape); r /= np.linalg.norm(r, axis=1, keepdims=True)
    nd = nrm + r; nd /= np.maximum(np.linalg.norm(nd, axis=1, keepdims=True), 1e-9)
    nxt = np.zeros((len(p), 8), dtype=np.float32)
    nxt[:, 0:3] = p + nrm * 1e-4; nxt[:, 3] = 1e-4; nxt[:, 4:7] = nd; nxt[:, 7] = np.finfo(np.float32).max
    all_rays.append(nxt); cur = nxt
rays = np.concatenate(all_rays)
print("rays per bounce", [len(a) for a in all_rays], "bvh nodes", len(flat.bvh), "depth", depth)
with open(out, "wb") as f:
    np.array([len(flat.bvh), len(flat.indices)//3, len(flat.positions), len(rays)], dtype=np.uint32).tofile(f)
    flat.bvh.tofile(f); flat.indices.astype(np.uint32).tofile(f); flat.positions.astype(np.float32).tofile(f); rays.tofile(f)
    np.array([len(a) for a in all_rays] + [0]*(8-len(all_rays)), dtype=np.uint32).tofile(f)
