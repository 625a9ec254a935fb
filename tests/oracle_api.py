"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product never imports it.
Takes the same FlatScene arrays the product uploads (identical struct layouts)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.environ.get("ORACLE_LIB") or os.path.join(ORACLE_DIR, "liboracle.so")  # ORACLE_LIB: sanitizer build (oracle/Makefile)


class OScene(C.Structure):
    _fields_ = [("objects", C.c_void_p), ("object_count", C.c_uint32), ("object_material_indices", C.c_void_p),
                ("spheres", C.c_void_p), ("sphere_count", C.c_uint32), ("materials", C.c_void_p),
                ("material_count", C.c_uint32), ("positions", C.c_void_p), ("vertex_count", C.c_uint32),
                ("indices", C.c_void_p), ("index_count", C.c_uint32), ("bvh", C.c_void_p), ("bvh_node_count", C.c_uint32),
                ("meshes", C.c_void_p), ("mesh_count", C.c_uint32)]


class OCamera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("rotation_wxyz", C.c_float * 4), ("vfov", C.c_float)]


class OGPUCamera(C.Structure):
    _fields_ = [("camera_matrix", C.c_float * 16), ("vfov", C.c_float), ("width", C.c_uint32), ("height", C.c_uint32)]


class ORay(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("t_min", C.c_float), ("direction", C.c_float * 3), ("t_max", C.c_float)]


class OIntersection(C.Structure):
    _fields_ = [("t", C.c_float), ("point", C.c_float * 3), ("normal", C.c_float * 3), ("material_id", C.c_size_t),
                ("side", C.c_uint8)]


assert C.sizeof(ORay) == 32 and C.sizeof(OIntersection) == 48

EXCHANGE_FN = C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32)

BVH_NODE_DTYPE = np.dtype([("aabb_min", "<f4", (3,)), ("aabb_max", "<f4", (3,)),
                           ("first_child_or_primitive", "<u4"), ("primitive_count", "<u4")])
INTERSECTION_DTYPE = np.dtype({"names": ["t", "point", "normal", "material_id", "side"],
                               "formats": ["<f4", ("<f4", (3,)), ("<f4", (3,)), "<u8", "u1"],
                               "offsets": [0, 4, 16, 32, 40], "itemsize": 48})

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.run(["make"], cwd=ORACLE_DIR, check=True, stdout=subprocess.DEVNULL)
        h = C.CDLL(LIB_PATH)
        h.orc_hash.restype = C.c_uint32
        h.orc_hash.argtypes = [C.c_uint32]
        h.orc_rng_seed.restype = C.c_uint32
        h.orc_rng_seed.argtypes = [C.c_uint32]
        h.orc_rng_next.restype = C.c_uint32
        h.orc_rng_next.argtypes = [C.POINTER(C.c_uint32)]
        h.orc_rng_discard.restype = None
        h.orc_rng_discard.argtypes = [C.POINTER(C.c_uint32), C.c_uint64]
        h.orc_rng_uniform.restype = C.c_float
        h.orc_rng_uniform.argtypes = [C.POINTER(C.c_uint32)]
        h.orc_path_seed.restype = C.c_uint32
        h.orc_path_seed.argtypes = [C.c_uint32, C.c_uint64]
        h.orc_sincos.restype = None
        h.orc_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        h.orc_to_gpu_camera.restype = None
        h.orc_to_gpu_camera.argtypes = [C.POINTER(OCamera), C.c_uint32, C.c_uint32, C.POINTER(OGPUCamera)]
        h.orc_generate_ray.restype = None
        h.orc_generate_ray.argtypes = [C.POINTER(OGPUCamera), C.c_float, C.c_float, C.POINTER(ORay)]
        h.orc_ray_sphere.restype = C.c_int
        h.orc_ray_sphere.argtypes = [C.POINTER(ORay), C.c_void_p, C.POINTER(OIntersection)]
        h.orc_ray_triangle.restype = C.c_int
        h.orc_ray_triangle.argtypes = [C.POINTER(ORay), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(OIntersection)]
        h.orc_ray_aabb.restype = C.c_int
        h.orc_ray_aabb.argtypes = [C.POINTER(ORay), C.c_void_p]
        h.orc_inverse_transform_ray.restype = None
        h.orc_inverse_transform_ray.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ORay), C.POINTER(ORay)]
        h.orc_transform_aabb.restype = None
        h.orc_transform_aabb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        h.orc_mat4_inverse.restype = None
        h.orc_mat4_inverse.argtypes = [C.c_void_p, C.c_void_p]
        h.orc_make_object.restype = None
        h.orc_make_object.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        h.orc_aabb_surface_area.restype = C.c_float
        h.orc_aabb_surface_area.argtypes = [C.c_void_p]
        h.orc_aabb_extent.restype = None
        h.orc_aabb_extent.argtypes = [C.c_void_p, C.c_void_p]
        h.orc_aabb_max_extent.restype = C.c_int
        h.orc_aabb_max_extent.argtypes = [C.c_void_p]
        h.orc_aabb_offset.restype = None
        h.orc_aabb_offset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        h.orc_bvh_build.restype = C.c_int
        h.orc_bvh_build.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
        h.orc_scene_intersect.restype = C.c_int
        h.orc_scene_intersect.argtypes = [C.POINTER(OScene), C.POINTER(ORay), C.POINTER(OIntersection)]
        h.orc_intersect_rays.restype = None
        h.orc_intersect_rays.argtypes = [C.POINTER(OScene), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        h.orc_render_streaming.restype = C.c_uint64
        h.orc_render_streaming.argtypes = [C.POINTER(OScene), C.POINTER(OCamera), C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        h.orc_render_streaming_band.restype = C.c_uint64
        h.orc_render_streaming_band.argtypes = [C.POINTER(OScene), C.POINTER(OCamera), C.c_uint32, C.c_uint32, C.c_uint32,
                                                C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, EXCHANGE_FN, C.c_void_p, C.c_int]
        h.orc_interleaved_rows.restype = C.c_uint32
        h.orc_interleaved_rows.argtypes = [C.c_uint32] * 4
        h.orc_render_streaming_interleaved.restype = C.c_uint64
        h.orc_render_streaming_interleaved.argtypes = [C.POINTER(OScene), C.POINTER(OCamera)] + [C.c_uint32] * 9 + \
                                                       [C.c_void_p] * 4 + [C.c_int]
        h.orc_render_megakernel.restype = C.c_uint64
        h.orc_render_megakernel.argtypes = [C.POINTER(OScene), C.POINTER(OCamera), C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        h.orc_denoise.restype = C.c_int
        h.orc_denoise.argtypes = [C.POINTER(OCamera), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int]
        h.orc_preview.restype = None
        h.orc_preview.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        h.orc_hardware_threads.restype = C.c_int
        h.orc_hardware_threads.argtypes = []
        _lib = h
    return _lib


def _p(a):
    return a.ctypes.data if a is not None and a.size else None


def build_bvh(positions, indices):
    """orc_bvh_build -> (nodes, max_depth); raises ValueError with the oracle's error code."""
    positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
    indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
    t = len(indices) // 3
    nodes = np.zeros(max(2 * t - 1, 1), dtype=BVH_NODE_DTYPE)
    depth = C.c_uint32(0)
    rc = lib().orc_bvh_build(_p(positions), len(positions), _p(indices), len(indices), nodes.ctypes.data, C.byref(depth))
    if rc < 0:
        raise ValueError(rc)
    return nodes[:rc], depth.value


class SceneHandle:
    """Keeps the numpy arrays of a FlatScene alive next to the OScene that points into them."""

    def __init__(self, flat):
        self.flat = flat
        bvh = flat.bvh
        self.mesh_ranges = None
        ranges = getattr(flat, "mesh_ranges", None)
        if ranges is not None:
            # a scene with a mesh table (extension, oracle.h): every mesh gets its own reference BVH (orc_bvh_build on
            # its slice, as bvh_from_mesh sees it), concatenated; the table says where each one starts
            ranges = np.array(ranges, dtype=np.uint32).reshape(-1, 6)
            positions = np.ascontiguousarray(flat.positions, dtype=np.float32).reshape(-1, 3)
            trees, depth, first = [], 0, 0
            for r in ranges:
                if r[3] == 0:
                    r[4], r[5] = first, 0
                    continue
                nodes, d = build_bvh(positions[r[0]:r[0] + r[1]], flat.indices[r[2]:r[2] + r[3]])
                r[4], r[5] = first, len(nodes)
                first += len(nodes)
                depth = max(depth, d)
                trees.append(nodes)
            bvh = np.concatenate(trees) if trees else None
            self.depth = depth
            self.mesh_ranges = np.ascontiguousarray(ranges)
        elif bvh is None and len(flat.indices):
            bvh, self.depth = build_bvh(flat.positions, flat.indices)
        self.bvh = bvh
        s = OScene()
        s.objects = _p(flat.objects)
        s.object_count = len(flat.objects)
        s.object_material_indices = _p(flat.object_material_indices)
        s.spheres = _p(flat.spheres)
        s.sphere_count = len(flat.spheres)
        s.materials = _p(flat.materials)
        s.material_count = len(flat.materials)
        s.positions = _p(flat.positions)
        s.vertex_count = len(flat.positions)
        s.indices = _p(flat.indices)
        s.index_count = len(flat.indices)
        s.bvh = _p(bvh) if bvh is not None else None
        s.bvh_node_count = len(bvh) if bvh is not None else 0
        s.meshes = _p(self.mesh_ranges) if self.mesh_ranges is not None else None
        s.mesh_count = len(self.mesh_ranges) if self.mesh_ranges is not None else 0
        self.c = s


def camera_c(camera):
    c = OCamera()
    c.position[:] = [float(x) for x in camera.position]
    c.rotation_wxyz[:] = [float(x) for x in camera.rotation]
    c.vfov = float(camera.vfov)
    return c


def render_streaming(flat, camera, w, h, iter_begin, iter_count, max_bounces, nthreads=0, prev=None, scene_handle=None):
    sh = scene_handle or SceneHandle(flat)
    cam = camera_c(camera)
    if prev is None:
        color = np.zeros((h, w, 3), dtype=np.float32)
        normal = np.zeros((h, w, 3), dtype=np.float32)
        depth = np.zeros((h, w), dtype=np.float32)
    else:
        color, normal, depth = (np.array(prev[k], dtype=np.float32, copy=True) for k in ("color", "normal", "depth"))
    live = np.zeros((iter_count, max_bounces), dtype=np.uint32)
    rays = lib().orc_render_streaming(C.byref(sh.c), C.byref(cam), w, h, iter_begin, iter_count, max_bounces,
                                      color.ctypes.data, normal.ctypes.data, depth.ctypes.data, live.ctypes.data, nthreads)
    return {"color": color, "normal": normal, "depth": depth, "live": live, "rays": int(rays)}


def render_band(flat, camera, w, h, rows, iteration, max_bounces, exchange=None, prev=None, nthreads=0, scene_handle=None):
    """One iteration of the row band rows=(row0,row1).  exchange(bounce, my_live) -> slot base, or None."""
    sh = scene_handle or SceneHandle(flat)
    cam = camera_c(camera)
    row0, row1 = rows
    if prev is None:
        color = np.zeros((row1 - row0, w, 3), dtype=np.float32)
        normal = np.zeros((row1 - row0, w, 3), dtype=np.float32)
        depth = np.zeros((row1 - row0, w), dtype=np.float32)
    else:
        color, normal, depth = (np.array(prev[k], dtype=np.float32, copy=True) for k in ("color", "normal", "depth"))
    live = np.zeros(max_bounces, dtype=np.uint32)
    cb = EXCHANGE_FN((lambda user, bounce, mine: int(exchange(int(bounce), int(mine)))) if exchange else 0)
    rays = lib().orc_render_streaming_band(C.byref(sh.c), C.byref(cam), w, h, row0, row1, iteration, max_bounces,
                                           color.ctypes.data, normal.ctypes.data, depth.ctypes.data, live.ctypes.data,
                                           cb, None, nthreads)
    return {"color": color, "normal": normal, "depth": depth, "live": live, "rays": int(rays)}


def render_interleaved(flat, camera, w, h, rank, nranks, block_rows, slot_offset, iter_begin, iter_count, max_bounces,
                       nthreads=0, prev=None, scene_handle=None):
    """The rows of rank `rank` under interleaved row blocks (ptc_set_interleave + "slot_offset"), packed in frame order:
    running means over iterations [iter_begin, iter_begin + iter_count)."""
    sh = scene_handle or SceneHandle(flat)
    cam = camera_c(camera)
    rows = int(lib().orc_interleaved_rows(h, rank, nranks, block_rows))
    if prev is None:
        color = np.zeros((rows, w, 3), dtype=np.float32)
        normal = np.zeros((rows, w, 3), dtype=np.float32)
        depth = np.zeros((rows, w), dtype=np.float32)
    else:
        color, normal, depth = (np.array(prev[k], dtype=np.float32, copy=True) for k in ("color", "normal", "depth"))
    live = np.zeros((iter_count, max_bounces), dtype=np.uint32)
    rays = lib().orc_render_streaming_interleaved(C.byref(sh.c), C.byref(cam), w, h, rank, nranks, block_rows, slot_offset,
                                                  iter_begin, iter_count, max_bounces, color.ctypes.data, normal.ctypes.data,
                                                  depth.ctypes.data, live.ctypes.data, nthreads)
    return {"color": color, "normal": normal, "depth": depth, "live": live, "rays": int(rays)}


def render_megakernel(flat, camera, w, h, iter_begin, iter_count, max_bounces, nthreads=0, prev=None, scene_handle=None):
    sh = scene_handle or SceneHandle(flat)
    cam = camera_c(camera)
    if prev is None:
        color = np.zeros((h, w, 3), dtype=np.float32)
        normal = np.zeros((h, w, 3), dtype=np.float32)
        depth = np.zeros((h, w), dtype=np.float32)
    else:
        color, normal, depth = (np.array(prev[k], dtype=np.float32, copy=True) for k in ("color", "normal", "depth"))
    rays = lib().orc_render_megakernel(C.byref(sh.c), C.byref(cam), w, h, iter_begin, iter_count, max_bounces,
                                       color.ctypes.data, normal.ctypes.data, depth.ctypes.data, nthreads)
    return {"color": color, "normal": normal, "depth": depth, "rays": int(rays)}


def intersect_rays(flat, rays, scene_handle=None):
    sh = scene_handle or SceneHandle(flat)
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
    n = len(rays)
    recs = np.zeros(n, dtype=INTERSECTION_DTYPE)
    hit = np.zeros(n, dtype=np.uint8)
    lib().orc_intersect_rays(C.byref(sh.c), rays.ctypes.data, n, recs.ctypes.data, hit.ctypes.data)
    return recs, hit


def denoise(camera, w, h, color, normal, depth, filter_size=10, c_phi=0.45, n_phi=0.30, p_phi=0.25, nthreads=0):
    cam = camera_c(camera)
    color = np.ascontiguousarray(color, dtype=np.float32)
    normal = np.ascontiguousarray(normal, dtype=np.float32)
    depth = np.ascontiguousarray(depth, dtype=np.float32)
    a = np.zeros((h, w, 3), dtype=np.float32)
    b = np.zeros((h, w, 3), dtype=np.float32)
    touched = np.zeros((h, w), dtype=np.uint8)
    which = lib().orc_denoise(C.byref(cam), w, h, color.ctypes.data, normal.ctypes.data, depth.ctypes.data,
                              a.ctypes.data, b.ctypes.data, filter_size, c_phi, n_phi, p_phi, touched.ctypes.data, nthreads)
    return (a if which == 0 else b), touched.astype(bool)


def preview(buffer, w, h, mode):
    buffer = np.ascontiguousarray(buffer, dtype=np.float32)
    out = np.zeros((h, w, 4), dtype=np.uint8)
    lib().orc_preview(buffer.ctypes.data, w, h, mode, out.ctypes.data)
    return out
