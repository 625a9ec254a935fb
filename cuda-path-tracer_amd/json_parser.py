"""Python mirror of the reference's scene front-end: scene_from_json (assets/json_parser.cpp:174-224) and
load_obj (assets/model_loader.cpp:11-44, there through assimp; here a plain OBJ reader: `v` and `f` records,
fan triangulation, bounding box from the vertices).  Same grammar as host/scene_description.cpp."""
import json
import math
import os

import numpy as np

from . import glmlite as glm
from .scene_description import (Camera, DielectricMaterial, DiffuseMateral, Mesh, MetalMaterial, SceneDescription,
                                Sphere)


def load_obj(filename):
    """model_loader.cpp:11-44 keeps assimp's mMeshes[0] only.  assimp's OBJ importer opens a new mesh at every `o` / `g` that
    names another object or group and at every `usemtl` that names another material, and drops meshes without faces: the
    first chunk that holds a face is kept, with the vertices its faces use (file order); fan triangulation.  The same rule
    as host/scene_description.cpp (tests/test_cli_frontend.py compares the two)."""
    positions, faces = [], []
    cur = {"o": "", "g": "", "usemtl": ""}
    closed = False
    with open(filename) as f:
        for line in f:
            if line.startswith("v "):
                positions.append([float(x) for x in line.split()[1:4]])
                continue
            key = line.split(None, 1)[0] if line.strip() else ""
            if key in cur and line[len(key):len(key) + 1] in (" ", "\t"):
                name = line[len(key):].strip()
                if name != cur[key] and faces:
                    closed = True
                cur[key] = name
            elif line.startswith("f ") and not closed:
                face = []
                for tok in line.split()[1:]:
                    idx = int(tok.split("/")[0])
                    face.append(len(positions) + idx if idx < 0 else idx - 1)
                for k in range(1, len(face) - 1):
                    faces += [face[0], face[k], face[k + 1]]
    if not positions or not faces:
        raise ValueError(f"Unable to load {filename}")
    positions = np.array(positions, dtype=np.float32)
    faces = np.array(faces, dtype=np.int64)
    if faces.min() < 0 or faces.max() >= len(positions):
        raise ValueError(f"Unable to load {filename}: face index out of range")
    used = np.zeros(len(positions), dtype=bool)
    used[faces] = True
    remap = np.cumsum(used) - 1
    return Mesh(positions[used], remap[faces].astype(np.uint32))


def _command(j):
    """One transform command, json_parser.cpp:40-75."""
    if "translate" in j:
        return glm.translate(j["translate"])
    if "scale" in j:
        s = j["scale"]
        return glm.scale(s if isinstance(s, (int, float)) else s)
    if "rotate" in j:
        return glm.rotate(np.float32(np.float32(j["rotate"]) * np.float32(0.01745329251994329576923690768489)), j["axis"])
    if "from" in j and "at" in j and "up" in j:
        return glm.look_at(j["from"], j["at"], j["up"])
    raise ValueError(f"Json parser: Unrecognized transform command: {j}")


def _transform(j):
    """json_parser.cpp:78-95: an object is one command, an array applies its commands left to right."""
    if isinstance(j, dict):
        return _command(j)
    if isinstance(j, list):
        return glm.compose([_command(e) for e in j])
    raise ValueError("Json Parser: Transform must be either an object or an array!")


def scene_from_json(filename):
    with open(filename) as f:
        root = json.load(f)
    file_dir = os.path.dirname(os.path.abspath(filename))
    scene = SceneDescription()
    scene.filename = filename
    for m in root["materials"]:
        t = m["type"]
        if t == "lambertian":
            scene.add_material(m["name"], DiffuseMateral(tuple(m["albedo"])))
        elif t == "dielectric":
            scene.add_material(m["name"], DielectricMaterial(m["refraction_index"]))
        elif t == "metal":
            scene.add_material(m["name"], MetalMaterial(tuple(m["albedo"]), m["fuzz"]))
        else:
            raise ValueError(f"Json Parser: Unsupported material type {t}")
    for s in root["surfaces"]:
        transform = _transform(s["transform"])
        if s["type"] == "sphere":
            scene.add_object(Sphere((0.0, 0.0, 0.0), s["radius"]), transform, s["material"])
        elif s["type"] == "mesh":
            path = os.path.normpath(os.path.join(file_dir, s["filename"]))
            mesh = scene.get_mesh(path) or scene.add_mesh(path, load_obj(path))
            scene.add_object(mesh, transform, s["material"])
        else:
            raise ValueError(f"Json Parser: Not supported surface type {s['type']}")
    cam = root["camera"]
    camera = Camera()
    if "transform" in cam:
        m = _transform(cam["transform"])
        camera.position = tuple(float(v) for v in m[3, 0:3])
        camera.rotation = tuple(float(v) for v in glm.quat_from_matrix(m))
    camera.vfov = float(np.float32(np.float32(cam["vfov"]) * np.float32(0.01745329251994329576923690768489)))
    scene.camera = camera
    if "resolution" in cam:
        scene.resolution = (int(cam["resolution"][0]), int(cam["resolution"][1]))
    scene.spp = int(root["sampler"]["samples"]) if "sampler" in root else 1
    return scene
