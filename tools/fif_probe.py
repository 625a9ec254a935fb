import sys, time; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
pkg=g.load_package()
W,H=1920,1080
sc=pkg.scenes.heightfield_scene((W,H)); flat=sc.build_scene()
mesh=list(sc.mesh_map_.values())[0]; flat.bvh,depth=pkg.bvh_from_mesh(mesh)
ref=None
for variant,F in ((4,8),(3,8),(4,1),(3,1)):
  with pkg.PathTracer(max_bounces=8) as pt:
    pt.set_param('frames_in_flight', F)
    pt.create_buffers((W,H), flat); pt.max_iterations=1<<30; pt.set_trace_variant(variant)
    for i in range(8): pt.path_trace(sc.camera)
    pt.synchronize(); r0=pt.stats()['rays_total']
    K=32
    t=time.time()
    for i in range(K): pt.path_trace(sc.camera)
    pt.synchronize(); dt=(time.time()-t)
    rays=pt.stats()['rays_total']-r0
    col=pt.download('color')
    if ref is None: ref=col
    print(f'variant {variant} F={F}: {dt/K*1e3:.3f} ms/frame  {rays/dt/1e6:.1f} Mrays/s  identical_to_first={np.array_equal(col,ref)}', flush=True)
