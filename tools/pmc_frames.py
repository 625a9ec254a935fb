"""Workload for rocprofv3 --pmc passes: N streaming frames of config 3 (1080p, 1M triangles, 8 bounces)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = pkg.scenes.heightfield_scene((1920, 1080)); flat = sc.build_scene()
mesh = list(sc.mesh_map_.values())[0]; flat.bvh, _ = pkg.bvh_from_mesh(mesh)
with pkg.PathTracer(max_bounces=8) as pt:
    pt.create_buffers((1920, 1080), flat); pt.max_iterations = 1 << 30
    for _ in range(n):
        pt.path_trace(sc.camera)
    pt.synchronize()
