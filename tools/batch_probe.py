import sys, time; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
pkg=g.load_package()
W,H=1920,1080
sc=pkg.scenes.heightfield_scene((W,H)); flat=sc.build_scene()
mesh=list(sc.mesh_map_.values())[0]; flat.bvh,depth=pkg.bvh_from_mesh(mesh)
rows=int(sys.argv[1]) if len(sys.argv)>1 else H
ref=None
for F,B,waves in ((16,1,6144),(16,2,6144),(16,4,6144),(16,8,6144),(8,4,6144),(8,8,6144),(32,8,6144),(16,4,12288),(16,1,6144)):
  with pkg.PathTracer(max_bounces=8) as pt:
    pt.set_param('frames_in_flight', F); pt.set_param('batch_frames', B); pt.set_param('traverse_waves', waves)
    pt.create_buffers((W,H), flat); pt.max_iterations=1<<30
    if rows!=H: pt.set_rows(0, rows)
    for i in range(16): pt.path_trace(sc.camera)
    pt.synchronize(); r0=pt.stats()['rays_total']
    K=64
    t=time.time()
    for i in range(K): pt.path_trace(sc.camera)
    pt.synchronize(); dt=(time.time()-t)
    rays=pt.stats()['rays_total']-r0
    c=pt.download('color')
    if ref is None: ref=c
    print(f'rows={rows} F={F} B={B} waves={waves}: {dt/K*1e3:.3f} ms/frame  {rays/dt/1e6:.1f} Mrays/s  same={np.array_equal(c,ref)}', flush=True)
