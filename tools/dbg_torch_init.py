import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, ctypes
import __graft_entry__ as g
pkg=g.load_package()
import importlib
ts=importlib.import_module('test_gpu_schedules')
which=sys.argv[1]
scene=pkg.scenes.cornell_spheres((96,96)); flat=scene.build_scene()
def hipcount():
    hip=ctypes.CDLL('libamdhip64.so.7'); n=ctypes.c_int(-1); rc=hip.hipGetDeviceCount(ctypes.byref(n)); return rc,n.value
print('start', hipcount(), flush=True)
if which=='batched':
    for fif,batch,waves in ((3,3,6144),(8,4,6144),(8,8,256),(6,2,24),(4,4,8)):
        ts.frames(pkg, scene, flat, 96,96, 7, 8, fif=fif, params=(("batch_frames",batch),("traverse_waves",waves)))
        print(fif,batch,waves, hipcount(), flush=True)
elif which=='plain':
    for fif in (1,3,8,16):
        ts.frames(pkg, scene, flat, 96,96, 7, 8, fif=fif)
        print(fif, hipcount(), flush=True)
import torch
print('torch import', hipcount(), torch.cuda.device_count(), flush=True)
x=torch.zeros(2,device='cuda'); print('ok', x)
