"""Config 2: Cornell box + two instances of the 69,984-triangle displaced sphere, 1280x720, 8 bounces."""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
W, H = 1280, 720
sc = pkg.scenes.cornell_bunny((W, H)); flat = sc.build_scene()
flat.bvh, _ = pkg.bvh_from_mesh(list(sc.mesh_map_.values())[0])
for params in ((('frames_in_flight', 1),), (('frames_in_flight', 16), ('batch_frames', 1)), ()):
    with pkg.PathTracer(max_bounces=8) as pt:
        for k, v in params: pt.set_param(k, v)
        pt.create_buffers((W, H), flat); pt.max_iterations = 1 << 30
        for _ in range(64): pt.path_trace(sc.camera)
        pt.synchronize(); r0 = pt.stats()['rays_total']
        K = 256; t = time.time()
        for _ in range(K): pt.path_trace(sc.camera)
        pt.synchronize(); dt = time.time() - t
        rays = pt.stats()['rays_total'] - r0
        print(f'config 2 {dict(params) or "defaults"}: {dt/K*1e3:.3f} ms/frame {rays/dt/1e6:.1f} Mrays/s  rays/frame {rays/K:.0f}', flush=True)
