"""Soak test: N iterations of config 3 (or, `soak.py N bunny`, config 2) under several schedules must give the same accumulated
image, bit for bit (any single differing hit anywhere changes every later random number of that frame)."""
import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
BUNNY = len(sys.argv) > 2 and sys.argv[2] == "bunny"
if BUNNY:
    W, H = 1280, 720
    sc = pkg.scenes.cornell_bunny((W, H))
else:
    W, H = 1920, 1080
    sc = pkg.scenes.heightfield_scene((W, H))
flat = sc.build_scene()
flat.bvh, _ = pkg.bvh_from_mesh(list(sc.mesh_map_.values())[0])
ref = None
CONFIG2 = (("defaults (prefold, the run's launch on 3072 wavefronts)", (), None),
           ("k_spheres as a pass of its own", (("prefold", 0),), None),
           ("one launch per instance, 5120 wavefronts", (("merge_instances", 0), ("run_waves", 5120)), None),
           ("three-kernel end of a bounce, 3 x 10-frame launches", (("fused_shade", 0), ("frames_in_flight", 30), ("batch_frames", 10)), None),
           ("no ray filter, spheres object by object, 8 x 1-frame launches",
            (("filter_rays", 0), ("sphere_lanes", 0), ("sphere_fold", 0), ("frames_in_flight", 8), ("batch_frames", 1)), None),
           ("one frame in flight", (("frames_in_flight", 1),), None))
CONFIG3_MORE = (("the bounce-spanning persistent launch, 2 x 16 frames", (("persist", 1), ("frames_in_flight", 32), ("batch_frames", 16)), None),
                ("paired batches taking turns, 2 x 10 frames", (("pair_batches", 1), ("frames_in_flight", 20), ("batch_frames", 10)), None))
CONFIG3 = (("defaults", (), None),
                              ("8 x 1-frame launches, 1024 wavefronts", (("frames_in_flight", 8), ("batch_frames", 1), ("traverse_waves", 1024)), None),
                              ("4 x 8-frame launches, 2048 wavefronts, static 7/8", (("frames_in_flight", 32), ("batch_frames", 8), ("traverse_waves", 2048), ("static_eighths", 7)), None),
                              ("12 x 1-frame launches, 6 stack entries in LDS", (("frames_in_flight", 12), ("batch_frames", 1), ("debug_lds_entries", 6)), None),
                              ("no ray filter, three-kernel end of a bounce, 3 x 20-frame launches", (("filter_rays", 0), ("fused_shade", 0), ("frames_in_flight", 60), ("batch_frames", 20)), None),
                              ("ray filter without the fused shade kernel, 5 x 7-frame launches", (("fused_shade", 0), ("frames_in_flight", 35), ("batch_frames", 7)), None),
                              ("round 3's feed (refill at 20 idle lanes, 3/8 static), no entry points, spheres object by object, 2 x 13-frame launches",
                               (("refill_lanes", 20), ("static_eighths", 3), ("beam", 0), ("sphere_lanes", 0), ("sphere_fold", 0), ("frames_in_flight", 26), ("batch_frames", 13)), None),
                              ("refill at 48 idle lanes, all of a region dynamic, 3072 wavefronts, 3 x 9-frame launches",
                               (("refill_lanes", 48), ("static_eighths", 0), ("traverse_waves", 3072), ("frames_in_flight", 27), ("batch_frames", 9)), None))
for name, params, variant in CONFIG2 if BUNNY else CONFIG3 + CONFIG3_MORE:
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
