"""MI355X (gfx950) path-tracing core behind the render API of LesleyLai/cuda-path-tracer.

The directory name contains a hyphen, so it is imported through ``__graft_entry__.load_package()``
under the module name ``cuda_path_tracer_amd``.  All rendering happens in ``libptcore.so``
(``csrc/``, C ABI in ``include/ptcore.h``); the Python files are the host-side mirror of the reference's
``PathTracer`` / ``SceneDescription`` interface used by the tests and by bench.py."""
from . import _capi, bands, camera_controller, glmlite, json_parser, scenes, viewer
from ._capi import LIB_PATH, PtcError, lib
from .path_tracer import DisplayBufferType, EdgeAvoidingATrousDenoiser, GPUMethod, PathTracer
from .scene_description import (Camera, DielectricMaterial, DiffuseMateral, FlatScene, Mesh, MetalMaterial,
                                SceneDescription, Sphere, bvh_from_mesh)

__all__ = ["PathTracer", "GPUMethod", "DisplayBufferType", "EdgeAvoidingATrousDenoiser", "SceneDescription", "Camera",
           "Sphere", "Mesh", "DiffuseMateral", "MetalMaterial", "DielectricMaterial", "FlatScene", "bvh_from_mesh",
           "scenes", "bands", "glmlite", "json_parser", "lib", "LIB_PATH", "PtcError"]
