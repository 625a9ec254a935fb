"""What the three libm substitutions change in the image (CPU only; INTEGRATION.md section 5 quotes these figures).

Both the oracle and the kernels replace three libm calls of the reference by fixed IEEE sequences so that the two sides
agree bit for bit: sinf / cosf in random_in_unit_sphere (distributions.cuh:13-17), pow(1 - cos, 5) in reflectance
(path_tracer.cu:135) and tan(vfov / 2) in generate_ray (ray_gen.cu:40).  oracle/liboracle_libm.so (make -C oracle libm)
is the same oracle with the platform's sinf / cosf / powf and a correctly rounded tangent instead.  This module renders
the golden scenes with both and reports
  at 1 spp:   the fraction of pixels whose radiance differs at all, the largest per-pixel difference, and the per-bounce
              live-path counts of both builds (the streaming loop's RNG is keyed on the compacted slot index, so ONE
              path that bounces differently renumbers every later path of the frame: this is the amplifier);
  at 64 spp:  the MSE between the two builds next to the MSE between two independent 64-spp estimates of the SAME
              build (iterations 0-63 against 64-127): the noise floor the difference has to be read against.

    python tests/libm_sensitivity.py        # prints the table as JSON
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

LIBM_SO = os.path.join(ROOT, "oracle", "liboracle_libm.so")


def libm_oracle():
    if not os.path.exists(LIBM_SO) or os.path.getmtime(LIBM_SO) < os.path.getmtime(os.path.join(ROOT, "oracle", "oracle.c")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libm"], check=True, stdout=subprocess.DEVNULL)
    h = C.CDLL(LIBM_SO)
    h.orc_render_streaming.restype = C.c_uint64
    h.orc_render_streaming.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return h


def render(handle, orc, sh, camera, w, h, first, count, mb):
    cam = orc.camera_c(camera)
    live = np.zeros((count, mb), dtype=np.uint32)
    # Every iteration is rendered on its own and the samples are averaged here: the oracle keeps the reference's running
    # mean over the ABSOLUTE iteration index, so a run that starts at `first` > 0 cannot be folded by it from zero.
    acc = np.zeros((h, w, 3), dtype=np.float64)
    rays = 0
    for k in range(count):
        s1 = np.zeros((h, w, 3), dtype=np.float32)
        n1 = np.zeros((h, w, 3), dtype=np.float32)
        d1 = np.zeros((h, w), dtype=np.float32)
        l1 = np.zeros((1, mb), dtype=np.uint32)
        rays += int(handle.orc_render_streaming(C.byref(sh.c), C.byref(cam), w, h, first + k, 1, mb, s1.ctypes.data,
                                                n1.ctypes.data, d1.ctypes.data, l1.ctypes.data, 0))
        # iteration index i > 0 leaves (0 * i + sample) / (i + 1) in a zeroed framebuffer: undo the division
        acc += s1.astype(np.float64) * (1.0 if first + k == 0 else float(first + k + 1))
        live[k] = l1[0]
    color = (acc / count).astype(np.float32)
    return {"color": color, "live": live, "rays": rays}


def measure(scene, w, h, mb, spp=64):
    orc = graft.load_oracle()
    flat = scene.build_scene()
    sh = orc.SceneHandle(flat)
    det, libm = orc.lib(), libm_oracle()
    out = {}
    a = render(det, orc, sh, scene.camera, w, h, 0, 1, mb)
    b = render(libm, orc, sh, scene.camera, w, h, 0, 1, mb)
    diff = np.abs(a["color"].astype(np.float64) - b["color"].astype(np.float64))
    out["one_spp"] = {"pixels": w * h, "differing_pixel_fraction": float(np.mean(np.any(diff > 0, axis=-1))),
                      "pixels_off_by_more_than_1e-3": float(np.mean(np.sqrt(np.sum(diff * diff, axis=-1)) > 1e-3)),
                      "max_abs_difference": float(diff.max()),
                      "live_fixed": [int(v) for v in a["live"][0]], "live_libm": [int(v) for v in b["live"][0]],
                      "max_live_delta_relative": float(max(abs(int(x) - int(y)) / max(int(x), 1) for x, y in zip(a["live"][0], b["live"][0]))),
                      "rays_fixed": a["rays"], "rays_libm": b["rays"]}
    a64 = render(det, orc, sh, scene.camera, w, h, 0, spp, mb)
    b64 = render(libm, orc, sh, scene.camera, w, h, 0, spp, mb)
    a64b = render(det, orc, sh, scene.camera, w, h, spp, spp, mb)
    mse = lambda x, y: float(np.mean(np.sum((x["color"].astype(np.float64) - y["color"].astype(np.float64)) ** 2, axis=-1)))
    out["many_spp"] = {"spp": spp, "mse_fixed_vs_libm": mse(a64, b64), "mse_fixed_vs_fixed_other_iterations": mse(a64, a64b),
                       "mean_radiance_fixed": float(a64["color"].mean()), "mean_radiance_libm": float(b64["color"].mean()),
                       "rays_relative_delta": abs(a64["rays"] - b64["rays"]) / a64["rays"]}
    return out


def scenes():
    pkg = graft.load_package()
    return {"cornell_spheres (diffuse, metal, glass)": (pkg.scenes.cornell_spheres((96, 96)), 96, 96, 8),
            "cornell_mesh (two mesh instances)": (pkg.scenes.cornell_bunny((96, 64), n_lat=10, n_lon=20), 96, 64, 8),
            "heightfield + 3 spheres": (pkg.scenes.heightfield_scene((96, 64), nx=65, nz=33), 96, 64, 8)}


if __name__ == "__main__":
    print(json.dumps({name: measure(*args) for name, args in scenes().items()}, indent=1))
