import sys, time; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
pkg=g.load_package()
W,H=1920,1080
sc=pkg.scenes.heightfield_scene((W,H)); flat=sc.build_scene()
mesh=list(sc.mesh_map_.values())[0]; flat.bvh,depth=pkg.bvh_from_mesh(mesh)
for F,waves,refill,lb,se in ((16,6144,20,1,7),(16,6144,20,1,4),(16,6144,20,1,0),(16,6144,20,1,2),(16,6144,20,1,7),(1,6144,20,1,7),(1,6144,20,1,4),(1,6144,20,1,0)):
  with pkg.PathTracer(max_bounces=8) as pt:
    pt.set_param('frames_in_flight', F); pt.set_param('traverse_waves', waves); pt.set_param('refill_lanes', refill); pt.set_param('static_eighths', se)
    pt.create_buffers((W,H), flat); pt.max_iterations=1<<30
    for i in range(8): pt.path_trace(sc.camera)
    pt.synchronize(); r0=pt.stats()['rays_total']
    K=48
    t=time.time()
    for i in range(K): pt.path_trace(sc.camera)
    pt.synchronize(); dt=(time.time()-t)
    rays=pt.stats()['rays_total']-r0
    print(f'F={F} waves={waves} refill={refill} leaf_batch={lb} static={se}/8: {dt/K*1e3:.3f} ms/frame  {rays/dt/1e6:.1f} Mrays/s', flush=True)
