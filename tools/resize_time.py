import sys, time; sys.path.insert(0,'/root/repo')
import __graft_entry__ as g
pkg=g.load_package()
sc=pkg.scenes.heightfield_scene((1920,1080), nx=129, nz=65); flat=sc.build_scene()
for i in range(3):
    t=time.time()
    pt=pkg.PathTracer(max_bounces=8); pt.create_buffers((1920,1080), flat)
    t1=time.time()-t
    t=time.time(); pt.path_trace(sc.camera); pt.synchronize(); t2=time.time()-t
    t=time.time(); pt.close(); t3=time.time()-t
    print(f'create_buffers {t1*1e3:.1f} ms, first frame {t2*1e3:.1f} ms, close {t3*1e3:.1f} ms')
