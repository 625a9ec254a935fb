import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
scene = pkg.scenes.heightfield_scene((160, 96), nx=129, nz=65)
flat = scene.build_scene()
for share in (0, 1):
    t0 = time.time()
    with pkg.PathTracer(device=0, max_bounces=6) as pt:
        pt.set_param("share_rays", share)
        pt.create_buffers((160, 96), flat)
        pt.max_iterations = 3
        for _ in range(3):
            pt.path_trace(scene.camera)
        try:
            c = pt.download("color"); st = pt.stats()
        except Exception as e:
            print("share", share, "ERROR", e, time.time() - t0)
            import ctypes as C
            lib = pkg.lib()
            if hasattr(lib, "ptc_debug_guard"):
                buf = np.zeros(4096, dtype=np.uint32)
                lib.ptc_debug_guard.argtypes = [C.c_void_p, C.c_size_t]
                lib.ptc_debug_guard(buf.ctypes.data, buf.nbytes)
                print("guard records", buf[0])
                for i in range(min(int(buf[0]), 12)):
                    print([int(x) for x in buf[16 + 16 * i: 16 + 16 * i + 16]])
                for i in range(min(int(buf[1]), 12)):
                    print("lane", [int(x) for x in buf[2048 + 8 * i: 2048 + 8 * i + 8]])
            continue
    ref = orc.render_streaming(flat, scene.camera, 160, 96, 0, 3, 6)
    print("share", share, "equal", np.array_equal(c, ref["color"]), st["rays_total"], ref["rays"], round(time.time() - t0, 2), flush=True)
