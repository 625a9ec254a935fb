"""N streaming frames of a named scene (workload for rocprofv3 runs); prints one JSON line with the ray count.
usage: run_frames.py <heightfield|bunny|spheres> <frames> [batch_frames] [streams] [trace_variant] [name=value ...]
(name=value: ptc_set_param before the scene upload, e.g. fused_shade=0)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
which = sys.argv[1]; n = int(sys.argv[2])
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 32
streams = int(sys.argv[4]) if len(sys.argv) > 4 else 2
variant = int(sys.argv[5]) if len(sys.argv) > 5 and "=" not in sys.argv[5] else -1
params = [a.split("=") for a in sys.argv[3:] if "=" in a]
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
    pt.set_param("frames_in_flight", batch * streams)
    pt.set_param("batch_frames", batch)
    if variant >= 0:
        pt.set_trace_variant(variant)
    for name, value in params:
        pt.set_param(name, int(value))
    pt.create_buffers((W, H), flat); pt.max_iterations = 1 << 30
    for _ in range(n):
        pt.path_trace(sc.camera)
    pt.synchronize()
    st = pt.stats()
print(json.dumps({"scene": which, "frames": n, "frames_per_launch": batch, "streams": streams, "rays_total": st["rays_total"],
                  "resolution": [W, H]}))
