"""Full-size GPU-vs-GPU check: the fast traversal (variant 1) against the reference-order traversal
(variant 0) on the 1080p / 1M-triangle configuration, bit for bit, plus timing of both."""
import sys, time; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
pkg=g.load_package()
W,H=1920,1080
K=int(sys.argv[1]) if len(sys.argv)>1 else 6
which=sys.argv[2] if len(sys.argv)>2 else 'heightfield'
if which=='heightfield':
    sc=pkg.scenes.heightfield_scene((W,H))
else:
    W,H=1280,720
    sc=pkg.scenes.cornell_bunny((W,H))
flat=sc.build_scene()
mesh=list(sc.mesh_map_.values())[0]
flat.bvh,depth=pkg.bvh_from_mesh(mesh)
res={}
VARIANTS=tuple(int(v) for v in (sys.argv[3] if len(sys.argv)>3 else '0,1,2').split(','))
for variant in VARIANTS:
    with pkg.PathTracer(max_bounces=8) as pt:
        pt.create_buffers((W,H), flat); pt.max_iterations=1<<30
        pt.set_trace_variant(variant)
        pt.path_trace(sc.camera); pt.synchronize()
        pt.restart()
        t=time.time()
        for i in range(K): pt.path_trace(sc.camera)
        pt.synchronize(); dt=time.time()-t
        st=pt.stats()
        res[variant]={k:pt.download(k) for k in ('color','normal','depth')}
        res[variant]['live']=st['last_live']
        pt.set_profiling(False, True); pt.reset_profile(); pt.set_iteration(0); pt.path_trace(sc.camera); pr=pt.profile()
        rays=sum(pr['paths'])
        print(f"variant {variant}: {dt/K*1e3:.2f} ms/frame, live {st['last_live']}, box/ray {sum(pr['box_tests'])/rays:.1f} tri/ray {sum(pr['tri_tests'])/rays:.2f}", flush=True)
base=VARIANTS[0]
for v in VARIANTS[1:]:
    for k in ('color','normal','depth'):
        a,b=res[base][k],res[v][k]
        print(f'variant {v} vs {base}:',k,'identical',np.array_equal(a,b),'ndiff',int(np.sum(a!=b)))
    print(f'variant {v} vs {base}: live identical', res[base]['live']==res[v]['live'])
