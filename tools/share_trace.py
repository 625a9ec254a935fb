"""Workload for a kernel trace of a 1/SHARE interleaved share of config 3: args F B waves frames"""
import os, sys; sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
pkg = g.load_package()
F, B, waves, n = (int(a) for a in sys.argv[1:5])
share = int(os.environ.get('SHARE', '8'))
W, H = 1920, 1080
sc = pkg.scenes.heightfield_scene((W, H)); flat = sc.build_scene()
flat.bvh, _ = pkg.bvh_from_mesh(list(sc.mesh_map_.values())[0])
with pkg.PathTracer(max_bounces=8) as pt:
    pt.set_param('frames_in_flight', F); pt.set_param('batch_frames', B); pt.set_param('traverse_waves', waves)
    pt.create_buffers((W, H), flat); pt.max_iterations = 1 << 30
    if share > 1: pt.set_interleave(0, share, 8)
    for _ in range(n):
        pt.path_trace(sc.camera)
    pt.synchronize()
