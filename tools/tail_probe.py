import sys, time; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
pkg=g.load_package()
W,H=1920,1080
variants=[int(v) for v in (sys.argv[1] if len(sys.argv)>1 else '3').split(',')]
sc=pkg.scenes.heightfield_scene((W,H)); flat=sc.build_scene()
mesh=list(sc.mesh_map_.values())[0]; flat.bvh,depth=pkg.bvh_from_mesh(mesh)
for v in variants:
  with pkg.PathTracer(max_bounces=8) as pt:
    pt.set_param('frames_in_flight', 1)
    pt.create_buffers((W,H), flat); pt.max_iterations=1<<30; pt.set_trace_variant(v)
    for i in range(2): pt.path_trace(sc.camera)
    pt.set_profiling(False, True); pt.reset_profile(); pt.path_trace(sc.camera); pr=pt.profile()
    print('variant',v,'box/ray', [round(pr['box_tests'][b]/pr['paths'][b],1) for b in range(8)], 'tri/ray',[round(pr['tri_tests'][b]/pr['paths'][b],2) for b in range(8)], 'max_box', pr['max_box_tests'][:8], 'max_ray_cyc', pr['max_ray_cycles'][:8], 'max_wave_cyc', pr['max_wave_cycles'][:8])
    pt.set_profiling(True, False); pt.reset_profile()
    t=time.time()
    for i in range(8): pt.path_trace(sc.camera)
    pt.synchronize(); dt=(time.time()-t)/8
    pr=pt.profile(); print('variant',v,'frame ms %.2f'%(dt*1e3),'trace us per bounce', [round(x/8*1e3,1) for x in pr['trace_ms']], flush=True)
