#!/usr/bin/env python3
"""Start-up cost of the benchmark scene on the GPU box: reference BVH build (host), scene upload by stage
(ptc_get_upload_times).  usage: tools/time_upload.py [grid=1001x501] [device_bvh=0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
grid = sys.argv[1] if len(sys.argv) > 1 else "1001x501"
nx, nz = (int(v) for v in grid.split("x"))
scene = pkg.scenes.heightfield_scene((1920, 1080), nx=nx, nz=nz)
mesh = list(scene.mesh_map_.values())[0]
t = time.time(); bvh, depth = pkg.bvh_from_mesh(mesh); host_ms = (time.time() - t) * 1e3
print(f"host ptc_build_bvh: {host_ms:.1f} ms, {len(bvh)} nodes, depth {depth}")
flat = scene.build_scene()
import copy
for mode in ("caller", "device", "host"):
    f = copy.copy(flat)
    f.bvh = bvh if mode == "caller" else None
    with pkg.PathTracer(device=0) as pt:
        pt.set_param("bvh_build_on_device", 0 if mode == "host" else 1)
        for rep in range(2):
            t = time.time(); pt.create_buffers((1920, 1080), f); wall = (time.time() - t) * 1e3
            print({"caller": "caller's BVH", "device": "BVH built on the GPU", "host": "BVH built on the host"}[mode],
                  f"create_buffers {wall:.1f} ms", pt.upload_times())
