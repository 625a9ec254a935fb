import sys, time, os; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
pkg=g.load_package()
W,H=1920,1080
sc=pkg.scenes.heightfield_scene((W,H)); flat=sc.build_scene()
mesh=list(sc.mesh_map_.values())[0]; flat.bvh,depth=pkg.bvh_from_mesh(mesh)
print('GPU_MAX_HW_QUEUES', os.environ.get('GPU_MAX_HW_QUEUES'), flush=True)
cfgs=[(8,16,1024),(8,24,1024),(8,32,1024),(8,32,512),(8,48,1024),(4,16,2048),(4,24,2048),(2,12,4096),(2,16,4096),(1,8,6144),(1,12,6144),(1,16,6144)]
for nr,F,waves in cfgs:
  with pkg.PathTracer(max_bounces=8) as pt:
    pt.set_param('frames_in_flight', F); pt.set_param('traverse_waves', waves)
    pt.create_buffers((W,H), flat); pt.max_iterations=1<<30
    if nr>1: pt.set_interleave(0, nr, 8)
    for i in range(F): pt.path_trace(sc.camera)
    pt.synchronize()
    K=96
    t=time.time()
    for i in range(K): pt.path_trace(sc.camera)
    t_enq=time.time()-t
    pt.synchronize(); dt=time.time()-t
    print(f'ranks={nr} F={F} waves={waves}: enqueue {t_enq/K*1e6:.1f} us/frame, total {dt/K*1e3:.3f} ms/frame', flush=True)
