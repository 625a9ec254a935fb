"""Where the wall time of the driver's 20 timed steps goes on the host side: the ptc_trace calls, the flush of the batch (on the
20th call), the present, the synchronisation -- with and without per-launch HIP events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
import torch
pkg = g.load_package()
W, H, MB = 1920, 1080, 8
scene = pkg.scenes.heightfield_scene((W, H))
flat = scene.build_scene()
out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
with pkg.PathTracer(device=0, max_bounces=MB) as pt:
    pt.set_param("frames_in_flight", 20)
    pt.set_param("batch_frames", 20)
    pt.create_buffers((W, H), flat)
    pt.set_stream(torch.cuda.current_stream().cuda_stream)
    pt.max_iterations = 1 << 30
    for events in (False, True, True, False):
        for _ in range(5):
            pt.path_trace(scene.camera)
        pt.download_to_device("color", out.data_ptr())
        torch.cuda.synchronize()
        pt.reset_profile()
        pt.set_profiling(time_trace_kernel=events, count_tests=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(19):
            pt.path_trace(scene.camera)
        t1 = time.perf_counter()
        pt.path_trace(scene.camera)      # the batch is full: enqueued here
        t2 = time.perf_counter()
        pt.download_to_device("color", out.data_ptr())
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print("events", events, "19 calls %.3f ms, 20th call (flush) %.3f ms, present %.3f ms, synchronize %.3f ms, total %.3f ms"
              % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3), flush=True)
        pt.set_profiling(False, False)
