"""Host-side scene container mirroring the reference's SceneDescription
(src/lib/scene_description.hpp:27-49, scene_description.cpp:12-154): same method names, same
flattening rules (materials indexed in name-sorted order; only the first mesh in name order is
uploaded and every mesh object uses it; per-object world AABBs).  build_scene() returns the flat arrays
that ptc_upload_scene takes -- the six cudaMemcpy uploads of the reference."""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _capi
from . import glmlite as glm


@dataclass
class DiffuseMateral:  # (sic) material.hpp:6-8
    albedo: tuple


@dataclass
class MetalMaterial:  # material.hpp:10-13
    albedo: tuple
    fuzz: float = 0.0


@dataclass
class DielectricMaterial:  # material.hpp:15-17
    refraction_index: float = 1.0


@dataclass
class Sphere:  # sphere.hpp:8-11
    center: tuple = (0.0, 0.0, 0.0)
    radius: float = 0.0


@dataclass
class Mesh:  # mesh.hpp:9-18
    positions: np.ndarray  # [V,3] float32
    indices: np.ndarray    # [3T] uint32
    aabb: tuple = None     # (min xyz, max xyz); the OBJ loader takes assimp's bounding box

    def __post_init__(self):
        self.positions = np.ascontiguousarray(self.positions, dtype=np.float32).reshape(-1, 3)
        self.indices = np.ascontiguousarray(self.indices, dtype=np.uint32).reshape(-1)
        if self.aabb is None and len(self.positions):
            self.aabb = (self.positions.min(axis=0), self.positions.max(axis=0))

    def triangle_count(self):
        return len(self.indices) // 3


@dataclass
class Camera:  # camera.hpp:17-23
    position: tuple = (0.0, 0.0, 0.0)
    rotation: tuple = (1.0, 0.0, 0.0, 0.0)  # w x y z
    vfov: float = float(np.pi / 2)

    def to_c(self):
        c = _capi.ptc_camera()
        c.position[:] = [float(x) for x in self.position]
        c.rotation_wxyz[:] = [float(x) for x in self.rotation]
        c.vfov = float(self.vfov)
        return c


@dataclass
class FlatScene:
    """What build_scene() uploads: numpy views laid out exactly like include/ptcore.h's structs."""
    objects: np.ndarray            # [N] structured, 160 B each
    object_material_indices: np.ndarray
    spheres: np.ndarray            # [S,4] float32
    materials: np.ndarray          # [M] structured, 20 B each
    positions: np.ndarray          # [V,3] float32
    indices: np.ndarray            # [3T] uint32
    bvh: np.ndarray = None         # optional [2T-1] structured, 32 B each
    mesh_ranges: np.ndarray = None  # optional [M,6] uint32 (ptc_mesh_range): several distinct meshes, objects[i].index = mesh
    keepalive: list = field(default_factory=list)

    def to_c(self):
        d = _capi.ptc_scene_desc()

        def ptr(a, ty):
            return a.ctypes.data_as(C.POINTER(ty)) if a is not None and a.size else None

        d.objects = ptr(self.objects, _capi.ptc_object)
        d.object_count = len(self.objects)
        d.object_material_indices = ptr(self.object_material_indices, C.c_uint32)
        d.spheres = ptr(self.spheres, _capi.ptc_sphere)
        d.sphere_count = len(self.spheres)
        d.materials = ptr(self.materials, _capi.ptc_material)
        d.material_count = len(self.materials)
        d.positions = ptr(self.positions, C.c_float)
        d.vertex_count = len(self.positions)
        d.indices = ptr(self.indices, C.c_uint32)
        d.index_count = len(self.indices)
        d.bvh = ptr(self.bvh, _capi.ptc_bvh_node) if self.bvh is not None else None
        d.bvh_node_count = len(self.bvh) if self.bvh is not None else 0
        if self.mesh_ranges is not None:
            ranges = np.ascontiguousarray(self.mesh_ranges, dtype=np.uint32).reshape(-1, 6)
            self.keepalive.append(ranges)
            d.meshes = ranges.ctypes.data_as(C.POINTER(_capi.ptc_mesh_range))
            d.mesh_count = len(ranges)
        return d


OBJECT_DTYPE = np.dtype([("type", "<u4"), ("index", "<u4"), ("m", "<f4", (16,)), ("inv_m", "<f4", (16,)),
                         ("aabb_min", "<f4", (3,)), ("aabb_max", "<f4", (3,))])
MATERIAL_DTYPE = np.dtype([("type", "<i4"), ("p", "<f4", (4,))])
BVH_NODE_DTYPE = np.dtype([("aabb_min", "<f4", (3,)), ("aabb_max", "<f4", (3,)),
                           ("first_child_or_primitive", "<u4"), ("primitive_count", "<u4")])
assert OBJECT_DTYPE.itemsize == 160 and MATERIAL_DTYPE.itemsize == 20 and BVH_NODE_DTYPE.itemsize == 32


def material_record(material):
    rec = np.zeros((), dtype=MATERIAL_DTYPE)
    if isinstance(material, DiffuseMateral):
        rec["type"] = 0
        rec["p"][:3] = material.albedo
    elif isinstance(material, MetalMaterial):
        rec["type"] = 1
        rec["p"][:3] = material.albedo
        rec["p"][3] = material.fuzz
    elif isinstance(material, DielectricMaterial):
        rec["type"] = 2
        rec["p"][0] = material.refraction_index
    else:
        raise TypeError(f"unsupported material {material!r}")
    return rec


class SceneDescription:
    def __init__(self):
        self.objects_ = []                    # (shape, transform 4x4)
        self.material_map_ = {}               # name -> material (iterated in sorted-name order, like std::map)
        self.objects_material_mapping_ = []
        self.mesh_map_ = {}                   # name -> Mesh
        self.filename = ""
        self.camera = Camera()
        self.resolution = (0, 0)
        self.spp = 1

    def add_material(self, name, material):   # scene_description.cpp:150-153 (try_emplace: first one wins)
        self.material_map_.setdefault(name, material)

    def add_object(self, shape, transform, material_name):  # scene_description.cpp:119-129
        if material_name not in self.material_map_:
            raise KeyError(f"Cannot find material {material_name}")
        self.objects_.append((shape, np.asarray(transform, dtype=np.float32).reshape(4, 4)))
        self.objects_material_mapping_.append(material_name)

    def get_mesh(self, name):                  # scene_description.cpp:131-139
        return self.mesh_map_.get(name)

    def add_mesh(self, name, mesh):            # scene_description.cpp:141-148
        if name in self.mesh_map_:
            raise ValueError("Cannot add the same mesh twice!")
        self.mesh_map_[name] = mesh
        return mesh

    def build_scene(self, prebuilt_bvh=None, distinct_meshes=False):  # scene_description.cpp:12-117
        """distinct_meshes=False is the reference: ONE mesh per scene (the first of the mesh map by name), whatever
        shape a mesh object was added with (scene_description.cpp:42,95).  True: every mesh object instantiates its
        own mesh (ptc_mesh_range)."""
        lib = _capi.lib()
        names = sorted(self.material_map_, key=lambda s: s.encode())
        index_of = {n: i for i, n in enumerate(names)}
        materials = np.array([material_record(self.material_map_[n]) for n in names], dtype=MATERIAL_DTYPE)

        mesh = None
        if self.mesh_map_:
            mesh = self.mesh_map_[sorted(self.mesh_map_, key=lambda s: s.encode())[0]]
        distinct = []   # the meshes of the scene in the order the objects first use them
        if distinct_meshes:
            for shape, _ in self.objects_:
                if not isinstance(shape, Sphere) and not any(shape is m for m in distinct):
                    distinct.append(shape)

        objects = np.zeros(len(self.objects_), dtype=OBJECT_DTYPE)
        spheres = []
        for i, (shape, transform) in enumerate(self.objects_):
            m16 = np.ascontiguousarray(transform, dtype=np.float32).reshape(16)
            out = _capi.ptc_object()
            if isinstance(shape, Sphere):
                sp = _capi.ptc_sphere()
                sp.center[:] = [float(x) for x in shape.center]
                sp.radius = float(shape.radius)
                _capi.check(lib.ptc_make_object(0, len(spheres), m16.ctypes.data_as(C.POINTER(C.c_float)), C.byref(sp),
                                                None, C.byref(out)))
                spheres.append([*shape.center, shape.radius])
            else:
                box = np.concatenate([np.asarray(shape.aabb[0], dtype=np.float32),
                                      np.asarray(shape.aabb[1], dtype=np.float32)])
                mesh_index = next(k for k, m in enumerate(distinct) if m is shape) if distinct_meshes else 0
                _capi.check(lib.ptc_make_object(1, mesh_index, m16.ctypes.data_as(C.POINTER(C.c_float)), None,
                                                box.ctypes.data_as(C.POINTER(C.c_float)), C.byref(out)))
            objects[i] = np.frombuffer(bytes(out), dtype=OBJECT_DTYPE)[0]

        if distinct_meshes and distinct:
            if prebuilt_bvh is not None:
                raise ValueError("prebuilt_bvh describes one mesh; with distinct_meshes the library builds the trees")
            ranges, v0, i0 = [], 0, 0
            for m in distinct:
                ranges.append([v0, len(m.positions), i0, len(m.indices), 0, 0])
                v0 += len(m.positions)
                i0 += len(m.indices)
            return FlatScene(
                objects=objects,
                object_material_indices=np.array([index_of[n] for n in self.objects_material_mapping_], dtype=np.uint32),
                spheres=np.array(spheres, dtype=np.float32).reshape(-1, 4),
                materials=materials,
                positions=np.ascontiguousarray(np.concatenate([m.positions for m in distinct]), dtype=np.float32),
                indices=np.ascontiguousarray(np.concatenate([m.indices for m in distinct]), dtype=np.uint32),
                mesh_ranges=np.array(ranges, dtype=np.uint32))
        return FlatScene(
            objects=objects,
            object_material_indices=np.array([index_of[n] for n in self.objects_material_mapping_], dtype=np.uint32),
            spheres=np.array(spheres, dtype=np.float32).reshape(-1, 4),
            materials=materials,
            positions=mesh.positions if mesh is not None else np.zeros((0, 3), dtype=np.float32),
            indices=mesh.indices if mesh is not None else np.zeros((0,), dtype=np.uint32),
            bvh=prebuilt_bvh)


def bvh_from_mesh(mesh):
    """bvh_from_mesh (accelerators/bvh.cpp:211-253) through the library's host builder.
    Returns (nodes structured array, max_depth)."""
    lib = _capi.lib()
    t = mesh.triangle_count()
    nodes = np.zeros(max(2 * t - 1, 1), dtype=BVH_NODE_DTYPE)
    depth = C.c_uint32(0)
    rc = lib.ptc_build_bvh(mesh.positions.ctypes.data_as(C.POINTER(C.c_float)), len(mesh.positions),
                           mesh.indices.ctypes.data_as(C.POINTER(C.c_uint32)), len(mesh.indices),
                           nodes.ctypes.data_as(C.POINTER(_capi.ptc_bvh_node)), C.byref(depth))
    if rc < 0:
        raise _capi.PtcError(rc, "BVH build failed")
    return nodes[:rc], depth.value
