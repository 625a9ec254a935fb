"""Largest deviation of the GPU denoiser from the CPU oracle (tolerance stage), several sizes."""
import sys; sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
for (w, h, nx, nz, it) in ((96, 64, 65, 33, 2), (320, 180, 129, 65, 1), (320, 180, 129, 65, 8)):
    sc = pkg.scenes.heightfield_scene((w, h), nx=nx, nz=nz); flat = sc.build_scene()
    with pkg.PathTracer(max_bounces=8) as pt:
        pt.create_buffers((w, h), flat); pt.max_iterations = it
        for _ in range(it): pt.path_trace(sc.camera)
        c, n, d = (pt.download(k) for k in ('color', 'normal', 'depth'))
        pt.denoise(); out = pt.download('final')
    den, touched = orc.denoise(sc.camera, w, h, c, n, d)
    ok = ~touched
    err = np.abs(out[ok] - den[ok])
    print(f'{w}x{h} iters {it}: max abs err {err.max():.3e}  mean {err.mean():.3e}  max value {den[ok].max():.3f}', flush=True)
