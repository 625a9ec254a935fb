import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as g
pkg=g.load_package(); orc=g.load_oracle()
import importlib.util
spec=importlib.util.spec_from_file_location('ts','/root/repo/tests/test_gpu_schedules.py'); ts=importlib.util.module_from_spec(spec); spec.loader.exec_module(ts)
scene=pkg.scenes.heightfield_scene((64,64)); flat=scene.build_scene()
flat.bvh,depth=pkg.bvh_from_mesh(list(scene.mesh_map_.values())[0])
rays=ts._adversarial_rays(flat, flat.bvh, np.random.default_rng(7))
recs,hit=orc.intersect_rays(flat,rays); m=hit.astype(bool)
print('rays', rays.shape, 'hit frac', m.mean())
with pkg.PathTracer() as pt:
    pt.create_buffers((64,64), flat)
    for variant in (0,1,3):
        pt.set_trace_variant(variant)
        for rep in range(2):
            t,nrm,mat,side=pt.intersect_rays(rays)
            bad=np.nonzero(((t>=0)!=m) | (m & (t!=recs['t'])))[0]
            kinds=np.bincount(bad%15, minlength=15)
            print('variant',variant,'rep',rep,'bad',len(bad),'by kind',kinds.tolist())
            for i in bad[:6]:
                print('   ray',i,'kind',i%15,rays[i].tolist(),'gpu t',t[i],'mat',mat[i],'oracle hit',hit[i],'t',recs['t'][i],'mat',recs['material_id'][i])
        if variant==3: print('slow rays', sum(pt.profile()['slow_rays']))
