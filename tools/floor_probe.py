import sys, time; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
pkg=g.load_package()
W,H=1920,1080
sc=pkg.scenes.heightfield_scene((W,H)); flat=sc.build_scene()
mesh=list(sc.mesh_map_.values())[0]; flat.bvh,depth=pkg.bvh_from_mesh(mesh)
# camera far above looking up: every ray misses every object box
cam=pkg.Camera(position=(0.0,50.0,0.0), rotation=(0.70710678,0.70710678,0.0,0.0), vfov=float(np.radians(50)))
for v,wv in ((1,4096),(3,1024),(3,2048),(3,4096),(3,8192),(3,16384)):
  with pkg.PathTracer(max_bounces=8) as pt:
    pt.set_param('traverse_waves', wv)
    pt.create_buffers((W,H), flat); pt.max_iterations=1<<30; pt.set_trace_variant(v)
    for i in range(2): pt.path_trace(cam)
    pt.set_profiling(True, False); pt.reset_profile()
    t=time.time()
    for i in range(8): pt.path_trace(cam)
    pt.synchronize(); dt=(time.time()-t)/8
    pr=pt.profile(); st=pt.stats()
    print('variant',v,'waves',wv,'frame ms %.3f'%(dt*1e3),'live',st['last_live'][:3],'trace us per bounce', [round(x/8*1e3,1) for x in pr['trace_ms']], flush=True)
