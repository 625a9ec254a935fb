"""N streaming frames of a named scene with a given trace variant (workload for rocprofv3 runs).
usage: run_frames.py <heightfield|bunny|spheres> <variant> <frames>"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
which = sys.argv[1]; variant = int(sys.argv[2]); n = int(sys.argv[3])
if which == 'heightfield':
    W, H = 1920, 1080; sc = pkg.scenes.heightfield_scene((W, H))
elif which == 'bunny':
    W, H = 1280, 720; sc = pkg.scenes.cornell_bunny((W, H))
else:
    W, H = 1280, 720; sc = pkg.scenes.cornell_spheres((W, H))
flat = sc.build_scene()
if sc.mesh_map_:
    flat.bvh, _ = pkg.bvh_from_mesh(list(sc.mesh_map_.values())[0])
with pkg.PathTracer(max_bounces=8) as pt:
    pt.create_buffers((W, H), flat); pt.max_iterations = 1 << 30
    pt.set_trace_variant(variant)
    for _ in range(n):
        pt.path_trace(sc.camera)
    pt.synchronize()
