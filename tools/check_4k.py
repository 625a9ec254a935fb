"""3840x2160 check: the automatic frames-in-flight cap (24 GiB of path state) and bit-identity of the default schedule
with the reference-order kernel at a size where one slot holds fewer frames than batch_frames."""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
W, H = 3840, 2160
sc = pkg.scenes.heightfield_scene((W, H), nx=501, nz=251); flat = sc.build_scene()
flat.bvh, _ = pkg.bvh_from_mesh(list(sc.mesh_map_.values())[0])
res = {}
for name, variant, fif in (("reference order", 0, 1), ("default", None, None)):
    with pkg.PathTracer(max_bounces=8) as pt:
        if fif: pt.set_param("frames_in_flight", fif)
        pt.create_buffers((W, H), flat); pt.max_iterations = 1 << 30
        if variant is not None: pt.set_trace_variant(variant)
        t = time.time()
        for _ in range(20): pt.path_trace(sc.camera)
        res[name] = {k: pt.download(k) for k in ("color", "normal", "depth")}
        st = pt.stats(); dt = time.time() - t
        print(f"{name}: 20 iterations in {dt:.2f} s, {st['rays_total']/dt/1e6:.0f} Mrays/s", flush=True)
print("identical:", all(np.array_equal(res["default"][k], res["reference order"][k]) for k in ("color", "normal", "depth")))
