import sys, time; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
pkg=g.load_package()
W,H=1920,1080
sc=pkg.scenes.heightfield_scene((W,H)); flat=sc.build_scene()
mesh=list(sc.mesh_map_.values())[0]; flat.bvh,depth=pkg.bvh_from_mesh(mesh)
with pkg.PathTracer(max_bounces=8) as pt:
    pt.create_buffers((W,H), flat); pt.max_iterations=1<<30
    for i in range(2): pt.path_trace(sc.camera)
    pt.set_profiling(False, True); pt.reset_profile(); pt.path_trace(sc.camera); pr=pt.profile()
    for b in range(8):
        n=pr['paths'][b]
        print(b, n, 'box/ray %.1f tri/ray %.2f max_box %d'%(pr['box_tests'][b]/n, pr['tri_tests'][b]/n, pr['max_box_tests'][b]))
    pt.set_profiling(True, False); pt.reset_profile()
    for i in range(8): pt.path_trace(sc.camera)
    pr=pt.profile(); print('trace us per bounce', [round(x/8*1e3,1) for x in pr['trace_ms']])
