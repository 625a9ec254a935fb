"""Interleaved A/B of parameter sets on config 3 (1080p, 1M triangles, 8 bounces): every set is timed ROUNDS
times in rotation (box clocks drift by ~10 % over a run) and the median is reported.
usage: tuner.py [rounds] 'F,B,waves[,k=v...]' ...     env SHARE=n: interleaved 1/n share of the rows"""
import os, sys, time; sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
W, H = 1920, 1080
sc = pkg.scenes.heightfield_scene((W, H)); flat = sc.build_scene()
flat.bvh, _ = pkg.bvh_from_mesh(list(sc.mesh_map_.values())[0])
args = sys.argv[1:]
rounds = 3
if args and args[0].isdigit():
    rounds = int(args.pop(0))
K = int(os.environ.get('K', '192'))
share = int(os.environ.get('SHARE', '1'))
res = {a: [] for a in args}
sums = set()
for r in range(rounds):
    for a in args:
        parts = a.split(',')
        F, B, waves = (int(x) for x in parts[:3])
        with pkg.PathTracer(max_bounces=8) as pt:
            pt.set_param('frames_in_flight', F); pt.set_param('batch_frames', B); pt.set_param('traverse_waves', waves)
            for kv in parts[3:]:
                k, v = kv.split('='); pt.set_param(k, int(v))
            pt.create_buffers((W, H), flat); pt.max_iterations = 1 << 30
            if share > 1: pt.set_interleave(0, share, 8)
            for i in range(max(F, 32)): pt.path_trace(sc.camera)
            pt.synchronize(); r0 = pt.stats()['rays_total']
            t = time.time()
            for i in range(K): pt.path_trace(sc.camera)
            pt.synchronize(); dt = time.time() - t
            rays = pt.stats()['rays_total'] - r0
            res[a].append(rays / dt / 1e6)
            pt.restart()
            for i in range(2): pt.path_trace(sc.camera)
            sums.add(float(pt.download('color').astype(np.float64).sum()))
for a in args:
    v = sorted(res[a])
    print(f"{a:<48} median {v[len(v)//2]:8.1f} Mrays/s   all {' '.join(f'{x:.0f}' for x in res[a])}", flush=True)
print('distinct images (2 iterations):', len(sums))
