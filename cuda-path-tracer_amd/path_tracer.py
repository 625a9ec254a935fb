"""Python mirror of the reference's `class PathTracer` (src/lib/path_tracer.hpp:60-99) over the C ABI.

Same public surface: fields ``max_iterations``, ``current_gpu_method``, ``atrous_denoiser``; methods
``restart``, ``iteration``, ``resize_image``, ``create_buffers``, ``path_trace``, ``denoise``,
``send_to_preview``.  Everything runs in libptcore.so on the GPU; this file only marshals arguments."""
import ctypes as C
from dataclasses import dataclass
from enum import IntEnum

import numpy as np

from . import _capi
from .scene_description import Camera, FlatScene, SceneDescription


class GPUMethod(IntEnum):  # path_tracer.hpp:57
    megakernel = 0
    streaming = 1


class DisplayBufferType(IntEnum):  # path_tracer.hpp:19
    final = 0
    color = 1
    normal = 2
    depth = 3


@dataclass
class EdgeAvoidingATrousDenoiser:  # denoising/edge_avoiding_a_trous_denoiser.hpp:7-12
    filter_size: int = 10
    color_weight: float = 0.45
    normal_weight: float = 0.30
    position_weight: float = 0.25


class PathTracer:
    def __init__(self, device=0, max_bounces=50):
        self._lib = _capi.lib()
        self.max_iterations = 1
        self.current_gpu_method = GPUMethod.streaming
        self.atrous_denoiser = EdgeAvoidingATrousDenoiser()
        self.max_bounces = max_bounces  # reference: compile-time 50 (path_tracer.cu:27)
        cfg = _capi.ptc_config(device=device, max_bounces=max_bounces, method=int(self.current_gpu_method), reserved=0)
        handle = C.c_void_p()
        _capi.check(self._lib.ptc_create(C.byref(cfg), C.byref(handle)))
        self._ctx = handle
        self._resolution = None
        self._rows = None

    # -- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.ptc_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        return _capi.check(rc, self._ctx)

    def _push_fields(self):
        self._check(self._lib.ptc_set_max_iterations(self._ctx, int(self.max_iterations)))
        self._check(self._lib.ptc_set_method(self._ctx, int(self.current_gpu_method)))
        self._check(self._lib.ptc_set_max_bounces(self._ctx, int(self.max_bounces)))
        d = self.atrous_denoiser
        p = _capi.ptc_denoiser_params(int(d.filter_size), float(d.color_weight), float(d.normal_weight),
                                      float(d.position_weight))
        self._check(self._lib.ptc_set_denoiser_params(self._ctx, C.byref(p)))

    # -- reference API --------------------------------------------------------------------------
    def restart(self):
        self._check(self._lib.ptc_restart(self._ctx))

    def iteration(self):
        return self._lib.ptc_iteration(self._ctx)

    def resize_image(self, resolution):
        w, h = resolution
        self._check(self._lib.ptc_resize(self._ctx, int(w), int(h)))
        self._resolution = (int(w), int(h))
        self._rows = (0, int(h))

    def create_buffers(self, resolution, scene):
        """scene: a SceneDescription (flattened with build_scene(), like the reference) or a FlatScene."""
        flat = scene.build_scene() if isinstance(scene, SceneDescription) else scene
        assert isinstance(flat, FlatScene)
        desc = flat.to_c()
        self._check(self._lib.ptc_upload_scene(self._ctx, C.byref(desc)))
        self.resize_image(resolution)

    def path_trace(self, camera, resolution=None):
        if resolution is not None and tuple(resolution) != self._resolution:
            raise ValueError("resolution differs from the allocated buffers (call resize_image first)")
        self._push_fields()
        cam = camera.to_c() if isinstance(camera, Camera) else camera
        self._check(self._lib.ptc_trace(self._ctx, C.byref(cam)))

    def denoise(self, resolution=None):
        self._push_fields()
        self._check(self._lib.ptc_denoise(self._ctx))

    def send_to_preview(self, dev_pbo=None, resolution=None, display_type=DisplayBufferType.final):
        """With dev_pbo=None returns an [h, w, 4] uint8 array; otherwise writes RGBA8 to the device pointer."""
        if dev_pbo is not None:
            self._check(self._lib.ptc_present_rgba8(self._ctx, C.c_void_p(int(dev_pbo)), 1, int(display_type)))
            return None
        out = np.empty((self.pixel_count(), 4), dtype=np.uint8)
        self._check(self._lib.ptc_present_rgba8(self._ctx, out.ctypes.data_as(C.c_void_p), 0, int(display_type)))
        return out.reshape(self._rows[1] - self._rows[0], self._resolution[0], 4)

    # -- additions (no reference equivalent) ----------------------------------------------------------
    def set_rows(self, row_begin, row_end):
        self._check(self._lib.ptc_set_rows(self._ctx, int(row_begin), int(row_end)))
        self._rows = (int(row_begin), int(row_end))

    def set_interleave(self, rank, nranks, block_rows):
        """Rows dealt in blocks of block_rows round-robin over nranks contexts (load-balanced multi-GPU split)."""
        self._check(self._lib.ptc_set_interleave(self._ctx, int(rank), int(nranks), int(block_rows)))
        h = self._resolution[1]
        blocks = (h + block_rows - 1) // block_rows
        rows = sum(min(block_rows, h - gb * block_rows) for gb in range(rank, blocks, nranks))
        self._rows = (0, rows)

    def pixel_count(self):
        return (self._rows[1] - self._rows[0]) * self._resolution[0]

    def set_iteration(self, it):
        self._check(self._lib.ptc_set_iteration(self._ctx, int(it)))

    def set_stream(self, hip_stream):
        self._check(self._lib.ptc_set_stream(self._ctx, C.c_void_p(int(hip_stream)) if hip_stream else None))

    def trace_begin(self, camera):
        self._push_fields()
        cam = camera.to_c() if isinstance(camera, Camera) else camera
        self._check(self._lib.ptc_trace_begin(self._ctx, C.byref(cam)))

    def trace_bounce(self, bounce, slot_base_dev=None):
        self._check(self._lib.ptc_trace_bounce(self._ctx, int(bounce), C.c_void_p(int(slot_base_dev)) if slot_base_dev else None))

    def trace_end(self):
        self._check(self._lib.ptc_trace_end(self._ctx))

    def live_count_dev(self, bounce):
        p = C.c_void_p()
        self._check(self._lib.ptc_live_count_dev(self._ctx, int(bounce), C.byref(p)))
        return p.value

    def copy_live_count(self, bounce, dst_dev):
        self._check(self._lib.ptc_copy_live_count(self._ctx, int(bounce), C.c_void_p(int(dst_dev))))

    def read_live_count(self, bounce):
        """Host value of the live-path count entering `bounce` of the frame being built (synchronises its stream)."""
        v = C.c_uint32(0)
        self._check(self._lib.ptc_read_live_count(self._ctx, int(bounce), C.byref(v)))
        return int(v.value)

    # -- several GPUs: bands over HIP inter-process memory (include/ptcore.h) -----------------------------------
    def band_export(self):
        """bytes of this rank's ptc_band_handle (to be sent to the root over the application's own channel)"""
        h = _capi.ptc_band_handle()
        self._check(self._lib.ptc_band_export(self._ctx, C.byref(h)))
        return bytes(h)

    def band_import(self, rank, handle_bytes):
        h = _capi.ptc_band_handle.from_buffer_copy(handle_bytes)
        self._check(self._lib.ptc_band_import(self._ctx, int(rank), C.byref(h)))

    def band_publish(self, which="color"):
        sel = {"color": _capi.BUF_COLOR, "normal": _capi.BUF_NORMAL, "depth": _capi.BUF_DEPTH, "final": _capi.BUF_FINAL}[which]
        self._check(self._lib.ptc_band_publish(self._ctx, sel))

    def gather_frame(self, which="color", dev_ptr=None):
        """root: the whole frame [h, w, 3] (or [h, w] for depth) from its own rows and the published rows of the
        imported ranks; into a device pointer when given"""
        sel = {"color": _capi.BUF_COLOR, "normal": _capi.BUF_NORMAL, "depth": _capi.BUF_DEPTH, "final": _capi.BUF_FINAL}[which]
        w, h = self._resolution
        if dev_ptr is not None:
            self._check(self._lib.ptc_gather_frame(self._ctx, sel, C.c_void_p(int(dev_ptr)), 1))
            return None
        ch = 1 if which == "depth" else 3
        out = np.empty(w * h * ch, dtype=np.float32)
        self._check(self._lib.ptc_gather_frame(self._ctx, sel, out.ctypes.data_as(C.c_void_p), 0))
        return out.reshape(h, w) if ch == 1 else out.reshape(h, w, 3)

    def gather_last_us(self):
        """root: device time of the most recent gather launch, in microseconds"""
        us = C.c_float(0.0)
        self._check(self._lib.ptc_gather_last_us(self._ctx, C.byref(us)))
        return float(us.value)

    def gather_present(self, display_type=DisplayBufferType.final):
        w, h = self._resolution
        out = np.empty((h * w, 4), dtype=np.uint8)
        self._check(self._lib.ptc_gather_present_rgba8(self._ctx, out.ctypes.data_as(C.c_void_p), 0, int(display_type)))
        return out.reshape(h, w, 4)

    def synchronize(self):
        self._check(self._lib.ptc_synchronize(self._ctx))

    def download(self, which):
        """which: 'color' | 'normal' | 'depth' | 'final' -> float32 array [rows, w, 3] (or [rows, w] for depth)."""
        sel = {"color": _capi.BUF_COLOR, "normal": _capi.BUF_NORMAL, "depth": _capi.BUF_DEPTH, "final": _capi.BUF_FINAL}[which]
        n = self.pixel_count()
        ch = 1 if which == "depth" else 3
        out = np.empty(n * ch, dtype=np.float32)
        self._check(self._lib.ptc_download(self._ctx, sel, out.ctypes.data_as(C.c_void_p), 0))
        rows, w = self._rows[1] - self._rows[0], self._resolution[0]
        return out.reshape(rows, w) if ch == 1 else out.reshape(rows, w, 3)

    def download_to_device(self, which, dev_ptr):
        sel = {"color": _capi.BUF_COLOR, "normal": _capi.BUF_NORMAL, "depth": _capi.BUF_DEPTH, "final": _capi.BUF_FINAL}[which]
        self._check(self._lib.ptc_download(self._ctx, sel, C.c_void_p(int(dev_ptr)), 1))

    def stats(self):
        s = _capi.ptc_stats()
        self._check(self._lib.ptc_get_stats(self._ctx, C.byref(s)))
        return {"rays_total": int(s.rays_total), "frames": int(s.frames),
                "last_live": [int(x) for x in s.last_live[: self.max_bounces]],
                "bvh_node_count": int(s.bvh_node_count), "bvh_max_depth": int(s.bvh_max_depth),
                "triangle_count": int(s.triangle_count), "stack_capacity": int(s.stack_capacity)}

    def build_bvh(self, mesh):
        """bvh_from_mesh (accelerators/bvh.cpp:211-253) on this context's GPU (ptc_build_bvh_device): the nodes the
        host builder (scene_description.bvh_from_mesh) returns.  -> (nodes structured array, max_depth)"""
        from .scene_description import BVH_NODE_DTYPE
        t = mesh.triangle_count()
        nodes = np.zeros(max(2 * t - 1, 1), dtype=BVH_NODE_DTYPE)
        depth = C.c_uint32(0)
        rc = self._lib.ptc_build_bvh_device(self._ctx, mesh.positions.ctypes.data_as(C.POINTER(C.c_float)), len(mesh.positions),
                                            mesh.indices.ctypes.data_as(C.POINTER(C.c_uint32)), len(mesh.indices),
                                            nodes.ctypes.data_as(C.POINTER(_capi.ptc_bvh_node)), C.byref(depth))
        if rc < 0:
            self._check(rc)
        return nodes[:rc], depth.value

    def upload_times(self):
        """milliseconds of the last scene upload by stage (ptc_upload_times)"""
        t = _capi.ptc_upload_times()
        self._check(self._lib.ptc_get_upload_times(self._ctx, C.byref(t)))
        return {k: (int(getattr(t, k)) if k.endswith("on_device") else round(float(getattr(t, k)), 2)) for k, _ in t._fields_}

    LAYOUTS = {"bvh4q": 0, "leaf_parent": 1, "tris": 2, "wide": 3, "bvh": 4}

    def download_layout(self, which):
        """bytes of one device-resident traversal array of the uploaded scene (ptc_download_layout)"""
        n = C.c_uint64(0)
        self._check(self._lib.ptc_download_layout(self._ctx, self.LAYOUTS[which], None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        self._check(self._lib.ptc_download_layout(self._ctx, self.LAYOUTS[which], out.ctypes.data, n.value, None))
        return out

    def set_trace_variant(self, variant):
        self._check(self._lib.ptc_set_trace_variant(self._ctx, int(variant)))

    def set_param(self, name, value):
        self._check(self._lib.ptc_set_param(self._ctx, name.encode(), int(value)))

    def set_profiling(self, time_trace_kernel=False, count_tests=False):
        self._check(self._lib.ptc_set_profiling(self._ctx, int(time_trace_kernel), int(count_tests)))

    def reset_profile(self):
        self._check(self._lib.ptc_reset_profile(self._ctx))

    def profile(self):
        p = _capi.ptc_profile()
        self._check(self._lib.ptc_get_profile(self._ctx, C.byref(p)))
        n = self.max_bounces
        return {"paths": [int(x) for x in p.paths[:n]], "box_tests": [int(x) for x in p.box_tests[:n]],
                "tri_tests": [int(x) for x in p.tri_tests[:n]], "trace_ms": [float(x) for x in p.trace_ms[:n]],
                "trace_launches": [int(x) for x in p.trace_launches[:n]],
                "max_box_tests": [int(x) for x in p.max_box_tests[:n]],
                "listed_rays": [int(x) for x in p.listed_rays[:n]],
                "slow_rays": [int(x) for x in p.slow_rays[:n]],
                "node_visits": [int(x) for x in p.node_visits[:n]],
                "denoise_ms": float(p.denoise_ms), "denoise_passes": int(p.denoise_passes),
                "persist_launches": int(p.persist_launches)}

    def intersect_rays(self, rays):
        """rays: [n, 8] float32 (origin, t_min, direction, t_max).  Returns t, normal, material, side."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = len(rays)
        t = np.empty(n, dtype=np.float32)
        nrm = np.empty((n, 3), dtype=np.float32)
        mat = np.empty(n, dtype=np.uint32)
        side = np.empty(n, dtype=np.uint8)
        self._check(self._lib.ptc_intersect_rays(
            self._ctx, rays.ctypes.data_as(C.POINTER(C.c_float)), n, t.ctypes.data_as(C.POINTER(C.c_float)),
            nrm.ctypes.data_as(C.POINTER(C.c_float)), mat.ctypes.data_as(C.POINTER(C.c_uint32)),
            side.ctypes.data_as(C.POINTER(C.c_uint8))))
        return t, nrm, mat, side

    def selftest_math(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        outs = [np.empty_like(a) for _ in range(4)]
        fp = C.POINTER(C.c_float)
        self._check(self._lib.ptc_selftest_math(self._ctx, a.ctypes.data_as(fp), b.ctypes.data_as(fp), len(a),
                                                *[o.ctypes.data_as(fp) for o in outs]))
        return outs
