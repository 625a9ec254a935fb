"""The C++ host front-end (cuda-path-tracer_amd/host: hip_pt = the reference's cuda_pt command line,
SceneDescription, JSON/OBJ readers) against the Python mirror: same scene files, same flattened arrays; and on
the GPU the same image through both."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_PT = os.path.join(ROOT, "cuda-path-tracer_amd", "host", "hip_pt")
SCENES = ["cornell_spheres.json", "cornell_mesh.json"]


def _ensure_cli():
    if not os.path.exists(HIP_PT):
        subprocess.run(["make"], cwd=os.path.dirname(HIP_PT), check=True, stdout=subprocess.DEVNULL)


def _parse_dump(path, pkg):
    sd = pkg.scene_description
    data = open(path, "rb").read()
    off = [0]

    def vec(dtype):
        n = struct.unpack_from("<Q", data, off[0])[0]
        off[0] += 8
        arr = np.frombuffer(data, dtype=dtype, count=n, offset=off[0])
        off[0] += n * np.dtype(dtype).itemsize
        return arr

    sphere_dt = np.dtype([("c", "<f4", (3,)), ("r", "<f4")])
    out = {"objects": vec(sd.OBJECT_DTYPE), "object_material_indices": vec("<u4"), "spheres": vec(sphere_dt),
           "materials": vec(sd.MATERIAL_DTYPE), "positions": vec("<f4"), "indices": vec("<u4")}
    out["camera"] = np.frombuffer(data, dtype="<f4", count=8, offset=off[0])
    off[0] += 32
    out["tail"] = np.frombuffer(data, dtype="<i4", count=3, offset=off[0])
    return out


@pytest.mark.parametrize("scene_file", SCENES)
def test_cpp_front_end_flattens_like_python(pkg, tmp_path, scene_file):
    _ensure_cli()
    path = os.path.join(ROOT, "assets", "scenes", scene_file)
    dump = tmp_path / "scene.bin"
    # run from a subdirectory: the asset directory is located like the reference does (walk up to "assets/")
    subprocess.run([HIP_PT, "--dump-scene", str(dump), "scenes/" + scene_file], cwd=os.path.join(ROOT, "tests"), check=True)
    cpp = _parse_dump(str(dump), pkg)
    py = pkg.json_parser.scene_from_json(path)
    flat = py.build_scene()
    assert np.array_equal(cpp["object_material_indices"], flat.object_material_indices)
    assert np.array_equal(cpp["indices"], flat.indices)
    assert np.array_equal(cpp["positions"], flat.positions.reshape(-1))
    assert np.array_equal(cpp["materials"]["type"], flat.materials["type"]) and np.array_equal(cpp["materials"]["p"], flat.materials["p"])
    assert np.array_equal(cpp["spheres"]["r"], flat.spheres[:, 3])
    for field in ("type", "index"):
        assert np.array_equal(cpp["objects"][field], flat.objects[field])
    for field in ("m", "inv_m", "aabb_min", "aabb_max"):
        assert np.allclose(cpp["objects"][field], flat.objects[field], rtol=1e-6, atol=1e-6), field
    cam = np.array([*py.camera.position, *py.camera.rotation, py.camera.vfov], dtype=np.float32)
    assert np.allclose(cpp["camera"], cam, rtol=1e-6, atol=1e-7)
    assert tuple(cpp["tail"][:2]) == tuple(py.resolution) and cpp["tail"][2] == py.spp


def test_obj_keeps_the_first_mesh_only(pkg, tmp_path):
    """model_loader.cpp:22 takes assimp's mMeshes[0]: of a file with several objects / groups / materials only the first chunk
    that holds faces, with the vertices those faces use.  Both readers, on a file whose first `o` is empty, whose second
    holds a quad and a triangle (negative and a/b/c indices), and whose later chunks (another `o`, another `usemtl`) must
    not be read; an unused vertex in front must not end up in the mesh (assimp's mesh has its own vertices, and its box
    bounds those)."""
    _ensure_cli()
    root = tmp_path / "assets"
    (root / "models").mkdir(parents=True)
    (root / "scenes").mkdir()
    (root / "models" / "two.obj").write_text(
        "# two objects\n"
        "o empty\n"
        "v 9 9 9\n"                      # used by nobody in the first mesh
        "o first\n"
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 0.5 1\n"
        "usemtl a\n"
        "f 2/1/1 3/2/1 4/3/1 5/4/1\n"      # a quad: two triangles
        "f -1 -5 -4\n"                    # relative indices: vertex 6, 2, 3
        "usemtl a\n"                      # the same material again: no new mesh
        "f 2 4 6\n"
        "usemtl b\n"                      # another material: assimp's second mesh
        "f 1 2 3\n"
        "o second\n"
        "v 5 5 5\nv 6 5 5\nv 5 6 5\n"
        "f 7 8 9\n")
    (root / "scenes" / "two.json").write_text(
        '{"camera": {"vfov": 45, "transform": {"from": [0, 0, 4], "at": [0, 0, 0], "up": [0, 1, 0]}, "resolution": [32, 32]},'
        ' "materials": [{"name": "m", "type": "lambertian", "albedo": [0.5, 0.5, 0.5]}],'
        ' "surfaces": [{"type": "mesh", "filename": "../models/two.obj", "material": "m", "transform": {"translate": [0, 0, 0]}}]}')
    mesh = pkg.json_parser.load_obj(str(root / "models" / "two.obj"))
    assert mesh.positions.shape == (5, 3) and np.array_equal(mesh.positions[0], [0, 0, 0]) and np.array_equal(mesh.positions[4], [0.5, 0.5, 1])
    assert [int(i) for i in mesh.indices] == [0, 1, 2, 0, 2, 3, 4, 0, 1, 0, 2, 4]
    dump = tmp_path / "scene.bin"
    subprocess.run([HIP_PT, "--dump-scene", str(dump), "scenes/two.json"], cwd=str(root / "scenes"), check=True)
    cpp = _parse_dump(str(dump), pkg)
    assert np.array_equal(cpp["positions"], mesh.positions.reshape(-1))
    assert np.array_equal(cpp["indices"], mesh.indices)
    flat = pkg.json_parser.scene_from_json(str(root / "scenes" / "two.json")).build_scene()
    assert np.allclose(cpp["objects"]["aabb_min"], flat.objects["aabb_min"]) and np.allclose(flat.objects["aabb_max"][0], [1, 1, 1])


def test_cli_errors_and_flags(tmp_path):
    _ensure_cli()
    r = subprocess.run([HIP_PT], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr
    r = subprocess.run([HIP_PT, "--help"], capture_output=True, text=True)
    assert r.returncode == 0
    r = subprocess.run([HIP_PT, "scenes/cornell_spheres.json"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 1 and "headless" in r.stderr         # the reference would open its GLFW viewer here
    bad = tmp_path / "bad.json"
    bad.write_text('{"camera": {"vfov": 45, "transform": {"o": [0, 0, 4]}}, "materials": [], "surfaces": []}')
    r = subprocess.run([HIP_PT, "--dump-scene", str(tmp_path / "x"), str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "Unrecognized transform command" in r.stderr   # like three_balls.json in the reference


@pytest.mark.gpu
@pytest.mark.parametrize("scene_file", SCENES)
def test_cli_image_equals_python_path(pkg, tmp_path, scene_file):
    from PIL import Image
    _ensure_cli()
    out = tmp_path / "out.png"
    r = subprocess.run([HIP_PT, "scenes/" + scene_file, "-o", str(out), "--spp", "3", "--max-bounces", "6"], cwd=ROOT,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Path Tracing:" in r.stdout and "Total:" in r.stdout      # the reference's Stopwatch report
    got = np.array(Image.open(out))
    scene = pkg.json_parser.scene_from_json(os.path.join(ROOT, "assets", "scenes", scene_file))
    with pkg.PathTracer(max_bounces=6) as pt:
        pt.create_buffers(scene.resolution, scene)
        pt.max_iterations = 3
        for _ in range(3):
            pt.path_trace(scene.camera)
        want = pt.send_to_preview()
    assert got.shape == want.shape and np.array_equal(got, want)


def _read_raw(path):
    data = open(path, "rb").read()
    assert data[:4] == b"PTRF"
    w, h, ch = struct.unpack_from("<III", data, 4)
    assert ch == 7 and len(data) == 16 + 4 * 7 * w * h
    f = np.frombuffer(data, dtype="<f4", offset=16)
    P = w * h
    return {"color": f[:3 * P].reshape(h, w, 3), "normal": f[3 * P:6 * P].reshape(h, w, 3), "depth": f[6 * P:].reshape(h, w)}


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [1, 2])
def test_cli_display_views_and_raw_dump(pkg, tmp_path, gpus):
    """hip_pt --display normal|depth|color writes send_to_preview's views (path_tracer.cu:487-520: n*0.5+0.5; 1/depth
    with alpha 1) and --dump-raw the float framebuffers behind them; one process and --gpus 2 (rows dealt in blocks of
    8, the ranks publish the buffer the image shows)."""
    from PIL import Image
    _ensure_cli()
    scene_file, spp, mb = "cornell_mesh.json", 2, 5
    scene = pkg.json_parser.scene_from_json(os.path.join(ROOT, "assets", "scenes", scene_file))
    w, h = scene.resolution
    views, raws = {}, {}
    if gpus == 1:
        with pkg.PathTracer(max_bounces=mb) as pt:
            pt.create_buffers(scene.resolution, scene)
            pt.max_iterations = spp
            for _ in range(spp):
                pt.path_trace(scene.camera)
            for name in ("color", "normal", "depth"):
                views[name] = pt.send_to_preview(display_type=getattr(pkg.DisplayBufferType, name))
                raws[name] = pt.download(name)
    else:
        parts = {k: [] for k in ("color", "normal", "depth")}
        rparts = {k: [] for k in ("color", "normal", "depth")}
        for rank in range(gpus):
            with pkg.PathTracer(max_bounces=mb) as pt:
                pt.create_buffers(scene.resolution, scene)
                pt.set_interleave(rank, gpus, 8)
                pt.set_param("slot_offset", rank * w * h)
                pt.max_iterations = spp
                for _ in range(spp):
                    pt.path_trace(scene.camera)
                for name in parts:
                    parts[name].append(pt.send_to_preview(display_type=getattr(pkg.DisplayBufferType, name)))
                    rparts[name].append(pt.download(name))
        for name in parts:
            views[name] = pkg.bands.assemble_interleaved(parts[name], h, gpus, 8)
            raws[name] = pkg.bands.assemble_interleaved(rparts[name], h, gpus, 8)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for name in ("color", "normal", "depth"):
        out, raw = tmp_path / f"{name}.png", tmp_path / f"{name}.raw"
        cmd = [HIP_PT, "scenes/" + scene_file, "-o", str(out), "--spp", str(spp), "--max-bounces", str(mb), "--display", name,
               "--dump-raw", str(raw)] + (["--gpus", str(gpus)] if gpus > 1 else [])
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
        assert r.returncode == 0, r.stdout + r.stderr
        got = np.array(Image.open(out))
        assert got.shape == views[name].shape and np.array_equal(got, views[name]), name
        dumped = _read_raw(raw)
        for k in ("color", "normal", "depth"):
            assert np.array_equal(dumped[k], raws[k].reshape(dumped[k].shape)), (name, k)
    assert np.all(views["depth"][..., 3] == 1) and np.all(views["normal"][..., 3] == 255)   # preview_depth_kernel's alpha
    r = subprocess.run([HIP_PT, "scenes/" + scene_file, "-o", str(tmp_path / "x.png"), "--display", "albedo"], cwd=ROOT,
                       capture_output=True, text=True)
    assert r.returncode == 1 and "unknown buffer" in r.stderr
