"""Device BVH builder against the host builder on 4M and 8M triangles (time, byte equality).  GPU box: python3 tools/big_bvh_check.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
for nx, nz in ((2001, 1001), (4001, 1001)):
    mesh = pkg.scenes.heightfield_mesh(nx, nz, 16.0, 8.0, seed=2)
    t = time.time(); want, wd = pkg.bvh_from_mesh(mesh); th = time.time() - t
    with pkg.PathTracer() as pt:
        t = time.time(); got, gd = pt.build_bvh(mesh); td = time.time() - t
    print(mesh.triangle_count(), "triangles: host", round(th * 1e3), "ms, device (incl. copies)", round(td * 1e3), "ms, equal:",
          bool(np.array_equal(got.view(np.uint8), want.view(np.uint8))), "depth", wd, gd)
