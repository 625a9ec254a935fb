import os, sys, time; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
pkg=g.load_package()
W,H=1920,1080
sc=pkg.scenes.heightfield_scene((W,H)); flat=sc.build_scene()
mesh=list(sc.mesh_map_.values())[0]; flat.bvh,depth=pkg.bvh_from_mesh(mesh)
tag=sys.argv[1]
for F,B,waves in [tuple(int(x) for x in a.split(',')) if ',' in a else (16,2,int(a)) for a in sys.argv[2:]]:
  with pkg.PathTracer(max_bounces=8) as pt:
    pt.set_param('frames_in_flight', F); pt.set_param('batch_frames', B); pt.set_param('traverse_waves', waves)
    for kv in filter(None, os.environ.get('EXTRA','').split(',')):
      k,v=kv.split('='); pt.set_param(k,int(v))
    pt.create_buffers((W,H), flat); pt.max_iterations=1<<30
    if os.environ.get('SHARE'): pt.set_interleave(0, int(os.environ['SHARE']), 8)
    for i in range(64): pt.path_trace(sc.camera)
    pt.synchronize(); r0=pt.stats()['rays_total']
    K=int(os.environ.get('K','256'))
    t=time.time()
    for i in range(K): pt.path_trace(sc.camera)
    pt.synchronize(); dt=(time.time()-t)
    rays=pt.stats()['rays_total']-r0
    c=pt.download('color')
    print(f'{tag} F={F} B={B} waves={waves}: {dt/K*1e3:.3f} ms/frame  {rays/dt/1e6:.1f} Mrays/s  sum={float(c.astype(np.float64).sum()):.6f}', flush=True)
