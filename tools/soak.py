"""Soak test: N iterations of config 3 under several schedules must give the same accumulated image, bit for bit
(any single differing hit anywhere changes every later random number of that frame)."""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
W, H = 1920, 1080
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sc = pkg.scenes.heightfield_scene((W, H)); flat = sc.build_scene()
flat.bvh, _ = pkg.bvh_from_mesh(list(sc.mesh_map_.values())[0])
ref = None
for name, params, variant in (("defaults", (), None),
                              ("8 x 1-frame launches, 1024 wavefronts", (("frames_in_flight", 8), ("batch_frames", 1), ("traverse_waves", 1024)), None),
                              ("4 x 8-frame launches, 2048 wavefronts, static 7/8", (("frames_in_flight", 32), ("batch_frames", 8), ("traverse_waves", 2048), ("static_eighths", 7)), None),
                              ("12 x 1-frame launches, 6 stack entries in LDS", (("frames_in_flight", 12), ("batch_frames", 1), ("debug_lds_entries", 6)), None),
                              ("no ray filter, three-kernel end of a bounce, 3 x 20-frame launches", (("filter_rays", 0), ("fused_shade", 0), ("frames_in_flight", 60), ("batch_frames", 20)), None),
                              ("ray filter without the fused shade kernel, 5 x 7-frame launches", (("fused_shade", 0), ("frames_in_flight", 35), ("batch_frames", 7)), None),
                              ("round 3's feed (refill at 20 idle lanes, 3/8 static), no entry points, spheres object by object, 2 x 13-frame launches",
                               (("refill_lanes", 20), ("static_eighths", 3), ("beam", 0), ("sphere_lanes", 0), ("sphere_fold", 0), ("frames_in_flight", 26), ("batch_frames", 13)), None),
                              ("refill at 48 idle lanes, all of a region dynamic, 3072 wavefronts, 3 x 9-frame launches",
                               (("refill_lanes", 48), ("static_eighths", 0), ("traverse_waves", 3072), ("frames_in_flight", 27), ("batch_frames", 9)), None)):
    with pkg.PathTracer(max_bounces=8) as pt:
        for k, v in params: pt.set_param(k, v)
        pt.create_buffers((W, H), flat); pt.max_iterations = 1 << 30
        if variant is not None: pt.set_trace_variant(variant)
        t = time.time()
        for _ in range(N): pt.path_trace(sc.camera)
        out = {k: pt.download(k) for k in ("color", "normal", "depth")}
        st = pt.stats(); dt = time.time() - t
    if ref is None: ref = (out, st)
    same = all(np.array_equal(out[k], ref[0][k]) for k in out) and st["rays_total"] == ref[1]["rays_total"]
    print(f"{name}: {N} iterations in {dt:.2f} s, rays {st['rays_total']}, identical to the first: {same}", flush=True)
    assert same
print("soak ok")
