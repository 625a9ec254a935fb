"""One traced 1080p frame of config 3's scene, then N denoise calls (workload for rocprofv3 runs of the A-Trous kernel).
usage: run_denoise.py <calls> [variant]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]); variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W, H = 1920, 1080
sc = pkg.scenes.heightfield_scene((W, H)); flat = sc.build_scene()
flat.bvh, _ = pkg.bvh_from_mesh(list(sc.mesh_map_.values())[0])
with pkg.PathTracer(max_bounces=8) as pt:
    pt.set_param("frames_in_flight", 1)
    pt.set_param("denoise_variant", variant)
    pt.create_buffers((W, H), flat); pt.max_iterations = 1 << 30
    pt.path_trace(sc.camera)
    for _ in range(n):
        pt.denoise()
    pt.synchronize()
print(json.dumps({"scene": "denoise", "frames": n, "frames_per_launch": 1, "rays_total": W * H * n * 4, "resolution": [W, H]}))
