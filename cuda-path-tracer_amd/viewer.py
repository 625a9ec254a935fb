"""The interactive front-end without a window: App::main_loop / run_cuda (interactive-app/app.cpp:141-170) driven by a
script of the events the reference takes from GLFW and ImGui.  Python mirror of `hip_pt --replay` (host/main.cpp,
run_replay: the script format is described there); returns the displayed frames instead of writing PNGs."""
import copy
import time

import numpy as np

from .camera_controller import FirstPersonCameraController
from .path_tracer import DisplayBufferType, GPUMethod, PathTracer

RADIANS = np.float32(0.01745329251994329576923690768489)


def replay(scene, script, device=0, max_bounces=50):
    """scene: SceneDescription; script: dict (see host/main.cpp).  Returns (frames, controller): every displayed frame as
    an [h, w, 4] uint8 array, and the controller in its final state."""
    window = tuple(int(v) for v in script.get("window", (800, 800)))          # app.cpp:36
    per_frame = int(script.get("iterations_per_frame", 1))
    budget_ms = float(script.get("budget_ms", 0.0))
    camera = copy.copy(scene.camera)
    controller = FirstPersonCameraController(camera)                             # app.cpp:18
    frames = []
    with PathTracer(device=device, max_bounces=max_bounces) as pt:
        pt.max_iterations = scene.spp                                            # app.cpp:130
        pt.create_buffers(window, scene)
        state = {"denoise": False, "display": DisplayBufferType.final, "resolution": window}

        def turn():                                                              # run_cuda, app.cpp:141-160
            start, done = time.perf_counter(), 0
            while True:
                pt.path_trace(camera)
                if state["denoise"]:
                    pt.denoise()
                pt.synchronize()
                done += 1
                if (budget_ms > 0 and (time.perf_counter() - start) * 1e3 >= budget_ms) or (budget_ms <= 0 and done >= per_frame):
                    break
            frames.append(pt.send_to_preview(display_type=state["display"]))

        for ev in script["events"]:
            if "frames" in ev:
                for _ in range(int(ev["frames"])):
                    turn()
            elif "key" in ev:
                for _ in range(int(ev.get("count", 1))):
                    if controller.on_key_press(ev["key"][:1]):                  # app.cpp:64-67
                        pt.restart()
            elif "mouse" in ev:
                if controller.on_mouse_move(RADIANS * np.float32(ev["mouse"][0]), RADIANS * np.float32(ev["mouse"][1])):
                    pt.restart()                                                 # app.cpp:108-113
            elif "space" in ev:
                pt.restart()                                                     # app.cpp:56
            elif "resize" in ev:
                state["resolution"] = tuple(int(v) for v in ev["resize"])
                pt.resize_image(state["resolution"])                             # app.cpp:45
            elif "denoise" in ev:
                state["denoise"] = bool(ev["denoise"])
            elif "display" in ev:
                state["display"] = getattr(DisplayBufferType, ev["display"], DisplayBufferType.final)
            elif "method" in ev:
                pt.current_gpu_method = GPUMethod.megakernel if ev["method"] == "megakernel" else GPUMethod.streaming
            elif "max_iterations" in ev:
                pt.max_iterations = max(1, int(ev["max_iterations"]))           # gui.cpp:103-104
            elif "filter_size" in ev:
                pt.atrous_denoiser.filter_size = int(ev["filter_size"])
            elif "speed" in ev:
                controller.speed = np.float32(ev["speed"])
            elif "position" in ev:
                controller.set_position(ev["position"])
                controller.update_camera()
                pt.restart()
            elif "reset" in ev:
                controller.reset()
                pt.restart()
            else:
                raise ValueError(f"replay: unknown event {ev}")
    return frames, controller
