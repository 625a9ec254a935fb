import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
import __graft_entry__ as g
pkg=g.load_package()
W,H=1920,1080
sc=pkg.scenes.heightfield_scene((W,H)); flat=sc.build_scene()
mesh=list(sc.mesh_map_.values())[0]; flat.bvh,depth=pkg.bvh_from_mesh(mesh)
import os
with pkg.PathTracer(max_bounces=8) as pt:
    for kv in filter(None, os.environ.get('EXTRA','').split(',')):
        k,v=kv.split('='); pt.set_param(k,int(v))
    pt.create_buffers((W,H), flat); pt.max_iterations=1<<30
    pt.path_trace(sc.camera); pt.denoise(); pt.synchronize()
    K=40
    t=time.time()
    for i in range(K): pt.denoise()
    pt.synchronize(); dt=(time.time()-t)/K
    print('denoise only: %.3f ms per call (4 passes)'%(dt*1e3))
    t=time.time()
    for i in range(K):
        pt.path_trace(sc.camera); pt.denoise()
    pt.synchronize(); dt=(time.time()-t)/K
    print('1 spp + denoise per frame (config 5): %.3f ms/frame'%(dt*1e3))
    out=pt.download('final'); print('finite', np.isfinite(out).all(), out.mean())
