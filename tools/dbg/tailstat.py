import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
W, H = 1920, 1080
sc = pkg.scenes.heightfield_scene((W, H)); flat = sc.build_scene()
flat.bvh, _ = pkg.bvh_from_mesh(list(sc.mesh_map_.values())[0])
for split, mw, rpw in ((8,1024,0),(8,512,0),(8,2048,0),(0,1024,0)):
    with pkg.PathTracer(max_bounces=8) as pt:
        pt.set_param("frames_in_flight", 1)
        pt.set_param("min_waves", mw)
        pt.create_buffers((W, H), flat); pt.max_iterations = 1 << 30
        pt.set_param("split_idle", split)
        for _ in range(4): pt.path_trace(sc.camera)
        pt.synchronize(); pt.reset_profile(); pt.set_profiling(True, False)
        t = time.time()
        for _ in range(16): pt.path_trace(sc.camera)
        pt.synchronize(); dt = time.time() - t
        pr = pt.profile()
        print('split', split, 'min_waves', mw, 'rays/wave', rpw, 'frame ms %.3f' % (dt / 16 * 1e3), 'trace us per bounce', [round(x / 16 * 1e3, 1) for x in pr['trace_ms']])
        pt.set_profiling(False, True); pt.reset_profile()
        pt.path_trace(sc.camera); pt.synchronize()
        pr = pt.profile()
        print('   paths', pr['paths'], 'max_box', pr['max_box_tests'][:8], 'nodes/ray', [round(pr['node_visits'][b] / max(pr['paths'][b], 1), 1) for b in range(8)])
