import sys; sys.path.insert(0,'.')
import numpy as np
import __graft_entry__ as g
pkg=g.load_package()
W,H=480,270
sc=pkg.scenes.heightfield_scene((W,H), nx=201, nz=101); flat=sc.build_scene()
out={}
for v in (1,4):
    with pkg.PathTracer(max_bounces=1) as pt:
        pt.set_param('frames_in_flight',1)
        pt.create_buffers((W,H), flat); pt.set_trace_variant(v); pt.path_trace(sc.camera)
        out[v]=(pt.download('depth'), pt.download('normal'), pt.stats()['last_live'])
d1,n1,l1=out[1]; d4,n4,l4=out[4]
print('live',l1,l4)
bad=np.argwhere(d1!=d4)
print('depth diffs',len(bad))
for y,x in bad[:10]:
    print(y,x,d1[y,x],d4[y,x],n1[y,x],n4[y,x])
