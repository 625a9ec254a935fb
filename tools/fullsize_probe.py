import sys, time; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
pkg=g.load_package()
W,H=1920,1080
t=time.time(); sc=pkg.scenes.heightfield_scene((W,H)); print('scene gen', time.time()-t)
t=time.time(); flat=sc.build_scene(); print('flatten', time.time()-t)
mesh=list(sc.mesh_map_.values())[0]
t=time.time(); nodes,depth=pkg.bvh_from_mesh(mesh); print('bvh build', time.time()-t, len(nodes), 'depth', depth)
flat.bvh=nodes
with pkg.PathTracer(max_bounces=8) as pt:
    t=time.time(); pt.create_buffers((W,H), flat); print('upload', time.time()-t)
    pt.max_iterations=1000
    for i in range(3): pt.path_trace(sc.camera)
    pt.synchronize()
    st0=pt.stats()
    t=time.time()
    K=10
    for i in range(K): pt.path_trace(sc.camera)
    pt.synchronize(); dt=time.time()-t
    st=pt.stats()
    rays=st['rays_total']-st0['rays_total']
    print('frames',K,'time',dt,'ms/frame',dt/K*1e3,'Mrays/s',rays/dt/1e6,'live',st['last_live'])
