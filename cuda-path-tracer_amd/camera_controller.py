"""Python mirror of the viewer's FirstPersonCameraController
(/root/reference/src/interactive-app/first_person_camera_controller.{hpp,cpp}; the C++ mirror is
host/first_person_camera_controller.hpp): position + pitch + yaw, W/A/S/D/R/F move along the camera's axes, a
right-drag turns it.  float32 in glm's operation order; host side, not on the parity path."""
import math

import numpy as np

F = np.float32
PI, TWO_PI, HALF_PI = F(3.14159265358979323846), F(6.28318530717958647692), F(1.57079632679489661923)


def yaw_pitch(yaw, pitch):
    """glm::yawPitchRoll(yaw, pitch, 0), upper 3x3 as M[col, row]"""
    ch, sh, cp, sp = (F(f(float(a))) for f, a in ((math.cos, yaw), (math.sin, yaw), (math.cos, pitch), (math.sin, pitch)))
    return np.array([[ch, F(0), -sh], [sh * sp, cp, ch * sp], [sh * cp, -sp, ch * cp]], dtype=F)


def quat_cast(m):
    """glm::quat_cast(mat3) -> (w, x, y, z)"""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = (F(v) for v in np.asarray(m, dtype=F).reshape(9))
    four = [m00 + m11 + m22, m00 - m11 - m22, m11 - m00 - m22, m22 - m00 - m11]   # w, x, y, z
    biggest = 0
    for i in (1, 2, 3):
        if four[i] > four[biggest]:
            biggest = i
    val = F(math.sqrt(float(four[biggest] + F(1)))) * F(0.5)
    mult = F(0.25) / val
    if biggest == 0:
        return (val, (m12 - m21) * mult, (m20 - m02) * mult, (m01 - m10) * mult)
    if biggest == 1:
        return ((m12 - m21) * mult, val, (m01 + m10) * mult, (m20 + m02) * mult)
    if biggest == 2:
        return ((m20 - m02) * mult, (m01 + m10) * mult, val, (m12 + m21) * mult)
    return ((m01 - m10) * mult, (m20 + m02) * mult, (m12 + m21) * mult, val)


class FirstPersonCameraController:
    default_speed = F(0.1)     # first_person_camera_controller.hpp:18

    def __init__(self, camera):
        self.camera = camera
        self.speed = self.default_speed
        self.reset()

    def reset(self):           # cpp:18-31
        self.position = np.array(self.camera.position, dtype=F)
        w, x, y, z = (F(v) for v in self.camera.rotation)
        py, px = F(2) * (y * z + w * x), w * w - x * x - y * y + z * z
        eps = F(1.1920929e-7)
        self.pitch = (F(2) * F(math.atan2(float(x), float(w)))) if (abs(px) <= eps and abs(py) <= eps) else F(math.atan2(float(py), float(px)))
        self.yaw = F(math.asin(float(min(F(1), max(F(-1), F(-2) * (x * z - w * y))))))
        self.speed = self.default_speed
        self.update_camera()

    def set_position(self, position):   # cpp:33-38
        self.position = np.array(position, dtype=F)

    def set_pitch(self, pitch):         # cpp:40-43
        self.pitch = min(HALF_PI, max(-HALF_PI, F(pitch)))

    def set_yaw(self, yaw):             # cpp:45-52
        yaw = F(math.fmod(float(F(yaw) + PI), float(TWO_PI)))
        if yaw < 0:
            yaw = yaw + TWO_PI
        self.yaw = yaw - PI

    def on_key_press(self, key):        # cpp:54-90
        direction = {"R": (0, 1, 0), "F": (0, -1, 0), "A": (1, 0, 0), "D": (-1, 0, 0), "W": (0, 0, -1), "S": (0, 0, 1)}.get(key)
        if direction is None:
            return False
        m = yaw_pitch(self.yaw, self.pitch)
        t = np.array(direction, dtype=F) * self.speed
        self.position = self.position + ((m[0] * t[0] + m[1] * t[1]) + m[2] * t[2])
        self.update_camera()
        return True

    def on_mouse_move(self, x_offset, y_offset):   # cpp:92-100 (radians)
        self.set_yaw(self.yaw + F(x_offset))
        self.set_pitch(self.pitch + F(y_offset))
        self.update_camera()
        return True

    def update_camera(self):            # cpp:12-16
        self.camera.position = tuple(float(v) for v in self.position)
        self.camera.rotation = tuple(float(v) for v in quat_cast(yaw_pitch(self.yaw, self.pitch)))
