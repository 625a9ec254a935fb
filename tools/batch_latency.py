"""Per-launch duration of the traversal kernel by bounce (HIP events), one stream, for batch sizes 1,2,4,8:
how does a launch's duration grow with the number of frames it carries?  SHARE env = interleaved share."""
import os, sys; sys.path.insert(0, '.')
import __graft_entry__ as g
pkg = g.load_package()
W, H = 1920, 1080
share = int(os.environ.get('SHARE', '8'))
sc = pkg.scenes.heightfield_scene((W, H)); flat = sc.build_scene()
flat.bvh, _ = pkg.bvh_from_mesh(list(sc.mesh_map_.values())[0])
for B, waves in ((1, 1024), (2, 1024), (4, 1024), (8, 1024), (4, 2048), (8, 4096), (1, 256), (1, 4096)):
    with pkg.PathTracer(max_bounces=8) as pt:
        pt.set_param('frames_in_flight', max(B, 2)); pt.set_param('batch_frames', B); pt.set_param('traverse_waves', waves)
        pt.create_buffers((W, H), flat); pt.max_iterations = 1 << 30
        if share > 1: pt.set_interleave(0, share, 8)
        for _ in range(2 * B): pt.path_trace(sc.camera)
        pt.synchronize()
        pt.set_profiling(True, False); pt.reset_profile()
        n = 8 * B
        for _ in range(n): pt.path_trace(sc.camera)
        pt.synchronize()
        pr = pt.profile()
        per = [pr['trace_ms'][b] / max(1, pr['trace_launches'][b]) * 1e3 for b in range(8)]
        print(f"share 1/{share} B={B} waves={waves}: launches/bounce {pr['trace_launches'][0]}  us/launch by bounce "
              + ' '.join(f'{x:.0f}' for x in per) + f"  sum {sum(per):.0f} us  per frame {sum(per)/B:.0f} us", flush=True)
