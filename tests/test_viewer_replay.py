"""the interactive front-end, headless.  The viewer's camera controller
(interactive-app/first_person_camera_controller.cpp:12-100) restated in C++ (host/first_person_camera_controller.hpp)
and Python (camera_controller.py), and its frame loop (app.cpp:141-170) replayed from a script of viewer events by
`hip_pt --replay` and `viewer.replay`.  CPU: the controller's arithmetic against hand-computed values and the two
implementations against each other; GPU: the PNG sequence of the C++ replay equals the frames of the Python replay."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_PT = os.path.join(ROOT, "cuda-path-tracer_amd", "host", "hip_pt")

SCRIPT = {"window": [64, 48], "iterations_per_frame": 2, "events": [
    {"frames": 2}, {"key": "W", "count": 3}, {"frames": 1}, {"mouse": [90, 0]}, {"key": "W"}, {"mouse": [0, 200]},
    {"key": "R", "count": 2}, {"speed": 0.5}, {"key": "A"}, {"mouse": [-400, -30]}, {"key": "S", "count": 2}, {"frames": 2},
    {"position": [1, 2, 3]}, {"key": "D"}, {"reset": True}, {"key": "F"}, {"key": "X"},
    {"display": "normal"}, {"frames": 1}, {"display": "final"}, {"denoise": True}, {"max_iterations": 3}, {"frames": 3},
    {"space": True}, {"method": "megakernel"}, {"frames": 1}, {"resize": [40, 40]}, {"method": "streaming"}, {"denoise": False},
    {"frames": 2}]}


def _ensure_cli():
    if not os.path.exists(HIP_PT):
        subprocess.run(["make"], cwd=os.path.dirname(HIP_PT), check=True, stdout=subprocess.DEVNULL)


def test_controller_arithmetic(pkg):
    cc = pkg.camera_controller
    cam = pkg.Camera(position=(0.0, 0.0, 0.0), rotation=(1.0, 0.0, 0.0, 0.0), vfov=1.0)
    c = cc.FirstPersonCameraController(cam)
    assert c.yaw == 0 and c.pitch == 0 and np.isclose(c.speed, 0.1)
    # identity: W = -z, S = +z, R = +y, F = -y, A = +x, D = -x (the reference's own signs, cpp:58-76)
    for key, want in (("W", (0, 0, -0.1)), ("S", (0, 0, 0)), ("R", (0, 0.1, 0)), ("F", (0, 0, 0)), ("A", (0.1, 0, 0)), ("D", (0, 0, 0))):
        assert c.on_key_press(key)
        assert np.allclose(cam.position, want, atol=1e-7), key
    assert not c.on_key_press("X") and not c.on_key_press(" ")
    # a quarter turn of yaw: forward is now -x; glm::yawPitchRoll = Ry(yaw) * Rx(pitch)
    c.on_mouse_move(np.pi / 2, 0.0)
    assert np.isclose(c.yaw, np.pi / 2, atol=1e-6)
    c.on_key_press("W")
    assert np.allclose(cam.position, (-0.1, 0, 0), atol=1e-6)
    assert np.allclose(cam.rotation, (np.sqrt(0.5), 0, np.sqrt(0.5), 0), atol=1e-6)      # quat_cast(Ry(pi/2))
    # pitch is clamped to +-pi/2 (cpp:40-43), yaw wraps into [-pi, pi) (cpp:45-52)
    c.on_mouse_move(0.0, 10.0)
    assert np.isclose(c.pitch, np.pi / 2)
    c.on_mouse_move(0.0, -20.0)
    assert np.isclose(c.pitch, -np.pi / 2)
    c.set_pitch(0.0)
    c.set_yaw(3.5)
    assert np.isclose(c.yaw, 3.5 - 2 * np.pi, atol=1e-6)
    c.set_yaw(-3.5)
    assert np.isclose(c.yaw, 2 * np.pi - 3.5, atol=1e-6)
    c.set_yaw(np.pi)
    assert np.isclose(c.yaw, -np.pi, atol=1e-6)
    # reset() reads pitch and yaw back from the camera's quaternion (glm::eulerAngles, cpp:22-27): a round trip
    for yaw, pitch in ((0.3, -0.7), (-2.0, 0.4), (1.2, 1.0)):
        c.set_yaw(yaw)
        c.set_pitch(pitch)
        c.update_camera()
        q = cam.rotation
        # the quaternion is that of Ry(yaw) * Rx(pitch)
        want = (np.cos(yaw / 2) * np.cos(pitch / 2), np.cos(yaw / 2) * np.sin(pitch / 2), np.sin(yaw / 2) * np.cos(pitch / 2),
                -np.sin(yaw / 2) * np.sin(pitch / 2))
        assert np.allclose(q, want, atol=1e-6) or np.allclose(q, [-v for v in want], atol=1e-6)
        c.speed = 7.0
        c.reset()
        assert np.isclose(c.speed, 0.1)
        if abs(yaw) <= np.pi / 2:
            assert np.isclose(c.pitch, pitch, atol=1e-5) and np.isclose(c.yaw, yaw, atol=1e-5)
        else:
            # glm::yaw is an asin and the controller drops the roll glm::eulerAngles reports (cpp:22-27): behind the half
            # turn around the front the reference's Reset comes back with another (yaw, pitch) -- a quirk, kept
            assert abs(c.yaw) <= np.pi / 2 + 1e-6 and np.isclose(abs(c.pitch - pitch), np.pi, atol=1e-5)


def test_cpp_controller_equals_python(pkg, tmp_path):
    """hip_pt --replay SCRIPT --dry-run prints the camera after every controller event (no GPU needed)"""
    _ensure_cli()
    script = tmp_path / "replay.json"
    script.write_text(json.dumps(SCRIPT))
    r = subprocess.run([HIP_PT, "scenes/cornell_mesh.json", "--replay", str(script), "--dry-run"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = [ln.split() for ln in r.stdout.splitlines() if ln.startswith(("start:", "event:"))]
    scene = pkg.json_parser.scene_from_json(os.path.join(ROOT, "assets", "scenes", "cornell_mesh.json"))
    import copy
    cam = copy.copy(scene.camera)
    c = pkg.camera_controller.FirstPersonCameraController(cam)

    def state():
        return np.array([*cam.position, c.yaw, c.pitch, *cam.rotation], dtype=np.float64)

    def parsed(tok):
        return np.array([float(tok[i]) for i in (2, 3, 4, 6, 8, 10, 11, 12, 13)])

    assert np.allclose(parsed(lines[0]), state(), atol=1e-7, rtol=1e-7)
    k, restarts = 1, 0
    for ev in SCRIPT["events"]:
        moved = None
        if "key" in ev:
            moved = any([c.on_key_press(ev["key"]) for _ in range(ev.get("count", 1))])
        elif "mouse" in ev:
            moved = c.on_mouse_move(np.float32(np.radians(1)) * np.float32(ev["mouse"][0]), np.float32(np.radians(1)) * np.float32(ev["mouse"][1]))
        elif "speed" in ev:
            c.speed, moved = np.float32(ev["speed"]), False
        elif "position" in ev:
            c.set_position(ev["position"])
            c.update_camera()
            moved = True
        elif "reset" in ev:
            c.reset()
            moved = True
        if moved is None:
            continue
        restarts += 1 if moved else 0
        assert np.allclose(parsed(lines[k]), state(), atol=1e-7, rtol=1e-7), (ev, lines[k])
        assert int(lines[k][-1]) == restarts, ev
        k += 1
    assert k == len(lines) and restarts == 12


@pytest.mark.gpu
def test_replay_frames_cpp_equals_python(pkg, tmp_path):
    """the viewer loop on the GPU: C++ (`hip_pt --replay`, PNG sequence) and Python (`viewer.replay`) show the same
    frames -- camera moves and Space restart the accumulation, max_iterations stops it (path_trace is a no-op past it,
    path_tracer.cu:391), the denoised buffer is what `final` shows after a denoise, a resize reallocates"""
    from PIL import Image
    _ensure_cli()
    script = tmp_path / "replay.json"
    script.write_text(json.dumps(SCRIPT))
    prefix = tmp_path / "view"
    r = subprocess.run([HIP_PT, "scenes/cornell_mesh.json", "--replay", str(script), "-o", str(prefix), "--max-bounces", "6"], cwd=ROOT,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    scene = pkg.json_parser.scene_from_json(os.path.join(ROOT, "assets", "scenes", "cornell_mesh.json"))
    frames, controller = pkg.viewer.replay(scene, SCRIPT, max_bounces=6)
    shown = sum(ev.get("frames", 0) for ev in SCRIPT["events"])
    assert len(frames) == shown == 12 and f"{shown} frames shown" in r.stdout
    for i, want in enumerate(frames):
        got = np.array(Image.open(f"{prefix}_{i:04d}.png"))
        assert got.shape == want.shape and np.array_equal(got, want), i
    assert frames[0].shape == (48, 64, 4) and frames[-1].shape == (40, 40, 4)
    # frames 0 and 1 accumulate (2 + 2 iterations of one image); a camera move restarts: frame 2 is another image
    assert not np.array_equal(frames[0], frames[1]) and not np.array_equal(frames[1], frames[2])
    # max_iterations = 3: frame 5 (normal view) had used iterations 0 and 1; the first of the next three turns traces
    # iteration 2 and then path_trace is a no-op -- the three turns show the same finished, denoised image
    assert np.array_equal(frames[6], frames[7]) and np.array_equal(frames[7], frames[8])
    # Space + megakernel: another image (another RNG stream per pixel, path_tracer.cu:239-243)
    assert not np.array_equal(frames[9], frames[8])
    assert np.all(frames[5][..., 3] == 255)   # the normal view is opaque
