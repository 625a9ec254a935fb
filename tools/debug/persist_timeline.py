"""Per frame and bounce, when the phases of one persistent launch opened and completed (a -DPT_PERSIST_DEBUG build:
make -C cuda-path-tracer_amd/csrc variant TAG=pdbg EXTRA=-DPT_PERSIST_DEBUG; PTCORE_LIB=.../libptcore_w_pdbg.so).
usage: persist_timeline.py [frames=20] [name=value ...]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
from cuda_path_tracer_amd import _capi
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
params = [a.split("=") for a in sys.argv[2:]]
W, H, MB = 1920, 1080, 8
scene = pkg.scenes.heightfield_scene((W, H))
flat = scene.build_scene()
with pkg.PathTracer(device=0, max_bounces=MB) as pt:
    pt.set_param("persist", 1)
    pt.set_param("frames_in_flight", frames)
    pt.set_param("batch_frames", frames)
    for k, v in params:
        pt.set_param(k, int(v))
    pt.create_buffers((W, H), flat)
    pt.max_iterations = 1 << 30
    for rep in range(3):
        for _ in range(frames):
            pt.path_trace(scene.camera)
        pt.synchronize()
    size = 256 + 128 + 32 * 11 * 128 + 32 * 16 * 8 * 4
    buf = (C.c_uint8 * size)()
    assert _capi.lib().ptc_debug_persist(pt._ctx, 0, buf, size) == 0
    raw = np.frombuffer(buf, dtype=np.uint32)
    dbg = raw[-32 * 16 * 8:].reshape(32, 16, 8).astype(np.int64)
    t0 = int(dbg[0][15][0])
    us = lambda v: ((int(v) - t0) & 0xffffffff) / 100.0
    print("frame | per bounce: [S(b) done / T(b+1) opens at us, T(b+1) rays] then T(b) complete at us")
    for f in range(frames):
        row = []
        for b in range(MB):
            s_done, t_done = us(dbg[f][b][4]), (us(dbg[f][b][2]) if b >= 1 else 0.0)
            row.append("b%d T<%7.0f S<%7.0f n%7d" % (b, t_done, s_done, dbg[f][b][0] if b >= 1 else 0))
        print("%2d  " % f + " | ".join(row))
