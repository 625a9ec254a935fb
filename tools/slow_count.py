"""How many rays take the reference-order fallback (k_slow_rays) per bounce, config 3, default schedule."""
import sys; sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
pkg = g.load_package()
W, H = 1920, 1080
sc = pkg.scenes.heightfield_scene((W, H)); flat = sc.build_scene()
flat.bvh, _ = pkg.bvh_from_mesh(list(sc.mesh_map_.values())[0])
with pkg.PathTracer(max_bounces=8) as pt:
    pt.create_buffers((W, H), flat); pt.max_iterations = 1 << 30
    pt.reset_profile()
    for _ in range(32): pt.path_trace(sc.camera)
    pr = pt.profile()
    print('paths    ', pr['paths'])
    print('slow rays', pr['slow_rays'])
