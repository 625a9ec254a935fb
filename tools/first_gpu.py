import sys, time; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
pkg=g.load_package(); orc=g.load_oracle()
t=time.time(); g.smoke(); print('smoke time', time.time()-t)
# math selftest
rng=np.random.default_rng(0)
a=np.concatenate([rng.uniform(0,6.2831855,100000), rng.uniform(0,1e6,1000), [0,1e-30,1e-40,3.4e38]]).astype(np.float32)
b=np.concatenate([rng.normal(size=100000), rng.uniform(-1e-3,1e-3,1000),[1e-40,3e38,1,2e-39]]).astype(np.float32)
with pkg.PathTracer() as pt:
    d,s,si,co=pt.selftest_math(a,b)
with np.errstate(all='ignore'):
    print('div exact', np.array_equal(d.view(np.uint32),(a/b).view(np.uint32)), 'sqrt exact', np.array_equal(s.view(np.uint32), np.sqrt(a).view(np.uint32)))
import ctypes as C
L=orc.lib()
so=np.empty_like(a); cc=np.empty_like(a)
sv=C.c_float(); cv=C.c_float()
for i in range(0,len(a)):
    L.orc_sincos(float(a[i]), C.byref(sv), C.byref(cv)); so[i]=sv.value; cc[i]=cv.value
print('sin exact', np.array_equal(si.view(np.uint32), so.view(np.uint32)), 'cos exact', np.array_equal(co.view(np.uint32), cc.view(np.uint32)))
m=a<7
print('sin err vs np (ulp-ish)', np.max(np.abs(so[m]-np.sin(a[m].astype(np.float64)))), np.max(np.abs(cc[m]-np.cos(a[m].astype(np.float64)))))
# bigger parity: config1 small, config2 small
for name,sc,W,H,it,mb in [('c1',pkg.scenes.cornell_spheres((128,128)),128,128,4,8),('c2',pkg.scenes.cornell_bunny((160,90),n_lat=24,n_lon=48),160,90,3,8),('c3',pkg.scenes.heightfield_scene((160,90),nx=101,nz=51),160,90,3,8)]:
    flat=sc.build_scene()
    with pkg.PathTracer(max_bounces=mb) as pt:
        pt.create_buffers((W,H), flat); pt.max_iterations=it
        lives=[]
        for i in range(it):
            pt.path_trace(sc.camera); lives.append(pt.stats()['last_live'])
        col=pt.download('color'); nrm=pt.download('normal'); dep=pt.download('depth'); st=pt.stats()
    ref=orc.render_streaming(flat, sc.camera, W,H,0,it,mb)
    print(name, 'rays', st['rays_total'], ref['rays'], 'color exact', np.array_equal(col,ref['color']), 'normal exact', np.array_equal(nrm, ref['normal']), 'depth exact', np.array_equal(dep, ref['depth']), 'live eq', np.array_equal(np.array(lives,dtype=np.uint32), ref['live']), 'mse', float(np.mean(np.sum((col-ref['color'])**2,-1))), 'depth', st['bvh_max_depth'])
