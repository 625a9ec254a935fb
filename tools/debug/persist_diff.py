"""persist on / off on a small heightfield scene: where do the frames differ?  usage: persist_diff.py [W H iters mb]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
W, H, iters, mb = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (160, 96, 12, 6)))
scene = pkg.scenes.heightfield_scene((W, H), nx=65, nz=33)
if os.environ.get("NOSPHERES"):
    glm = pkg.glmlite
    s2 = pkg.SceneDescription()
    s2.resolution, s2.camera = (W, H), scene.camera
    s2.add_material("ground", pkg.DiffuseMateral((0.7, 0.7, 0.7)))
    mesh = list(scene.mesh_map_.values())[0]
    s2.add_mesh("ground", mesh)
    s2.add_object(mesh, glm.translate((0.0, 0.0, 0.0)), "ground")
    scene = s2
flat = scene.build_scene()

def render(params, batch):
    with pkg.PathTracer(device=0, max_bounces=mb) as pt:
        pt.set_param("persist", 1)
        for k, v in params:
            pt.set_param(k, v)
        pt.set_param("frames_in_flight", batch)
        pt.set_param("batch_frames", batch)
        pt.create_buffers((W, H), flat)
        pt.max_iterations = iters
        for _ in range(iters):
            pt.path_trace(scene.camera)
        out = {k: pt.download(k) for k in ("color", "normal", "depth")}
        st = pt.stats()
        out["rays"], out["live"] = st["rays_total"], st["last_live"][:mb]
    return out

ref = render((("persist", 0),), iters)
def render2(params, fif, batch):
    global iters
    with pkg.PathTracer(device=0, max_bounces=mb) as pt:
        pt.set_param("persist", 1)
        for k, v in params:
            pt.set_param(k, v)
        pt.set_param("frames_in_flight", fif)
        pt.set_param("batch_frames", batch)
        pt.create_buffers((W, H), flat)
        pt.max_iterations = iters
        pt.reset_profile()
        for _ in range(iters):
            pt.path_trace(scene.camera)
        out = {k: pt.download(k) for k in ("color", "normal", "depth")}
        st = pt.stats()
        out["rays"], out["live"] = st["rays_total"], st["last_live"][:mb]
        print("   persist launches", pt.profile()["persist_launches"])
        import ctypes as C
        from cuda_path_tracer_amd import _capi
        size = 256 + 128 + 32 * 11 * 128 + 32 * 16 * 8 * 4
        for slot in (0, 1):
            buf = (C.c_uint8 * size)()
            rc = _capi.lib().ptc_debug_persist(pt._ctx, slot, buf, size)
            if rc: continue
            raw = np.frombuffer(buf, dtype=np.uint32)
            dbg = raw[-32 * 16 * 8:].reshape(32, 16, 8)
            print("   slot", slot, "started/frames_done/lock/error", raw[64:68].tolist(), "state0", hex(int(raw[0]) | (int(raw[1]) << 32)), "lib", _capi.LIB_PATH[-24:], "frame 0 per bounce {n, fetched, finalized, tiles, n_all, live_out}:")
            for b in range(mb):
                print("      b", b, dbg[0][b][:6].tolist())
            print("      over-count events", dbg[0][15][6], "finalized outside T", dbg[0][15][7])
    return out
print("ref live", ref["live"])
for params, fif, batch in (((("persist_min_frames", 1), ("traverse_waves", 64)), 2, 1),):
    got = render2(params, fif, batch)
    bad = np.any(got["color"] != ref["color"], axis=-1)
    badn = np.any(got["normal"] != ref["normal"], axis=-1) | (got["depth"] != ref["depth"])
    ys, xs = np.nonzero(bad)
    print(params, "fif", fif, "batch", batch, "color px", int(bad.sum()), "gbuffer px", int(badn.sum()), "rays", got["rays"], ref["rays"], "live", got["live"],
          "rows", sorted(set(ys.tolist()))[:12], "max |d|", float(np.max(np.abs(got["color"] - ref["color"]))))
