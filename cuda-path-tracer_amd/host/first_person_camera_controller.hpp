// first_person_camera_controller.hpp -- the viewer's camera controller
// (/root/reference/src/interactive-app/first_person_camera_controller.{hpp,cpp}), restated without glm / GLFW / ImGui:
// position + pitch + yaw, w/a/s/d/r/f move along the camera's axes, a right-drag turns it.  Keys are the characters
// the reference binds ('W' forward ... GLFW_KEY_W == 'W').  Every function cites the lines it follows; float32 math in
// glm's operation order (host side, not on the parity path).  Transcendentals are evaluated in double precision and
// rounded to float -- the Python mirror (camera_controller.py) does the same through math.*, so the two front-ends hand
// the render core the same camera bits and their replays show the same frames.
#pragma once

#include <cmath>

#include "scene_description.hpp"

namespace hip_pt {

class FirstPersonCameraController {
public:
  static constexpr float default_speed = 0.1f;  // first_person_camera_controller.hpp:18
  float speed = default_speed;

  explicit FirstPersonCameraController(Camera& camera) : camera_(camera) { reset(); }  // hpp:21-25

  [[nodiscard]] const float* position() const noexcept { return position_; }
  [[nodiscard]] float pitch() const noexcept { return pitch_; }
  [[nodiscard]] float yaw() const noexcept { return yaw_; }
  void set_position(float x, float y, float z) noexcept  // cpp:33-38 (the camera follows at the next update)
  {
    position_[0] = x;
    position_[1] = y;
    position_[2] = z;
  }
  void set_pitch(float pitch) noexcept  // cpp:40-43: clamp to [-pi/2, pi/2]
  {
    const float h = 1.57079632679489661923f;
    pitch_ = pitch < -h ? -h : (pitch > h ? h : pitch);
  }
  void set_yaw(float yaw) noexcept  // cpp:45-52: wrapped into [-pi, pi)
  {
    const float pi = 3.14159265358979323846f, two_pi = 6.28318530717958647692f;
    yaw = (float)std::fmod((double)(yaw + pi), (double)two_pi);
    if (yaw < 0.0f) yaw += two_pi;
    yaw_ = yaw - pi;
  }

  // cpp:18-31: take position, pitch and yaw from the camera (glm::eulerAngles: x = pitch, y = yaw), default speed
  void reset()
  {
    for (int i = 0; i < 3; ++i) position_[i] = camera_.position[i];
    const float w = camera_.rotation_wxyz[0], x = camera_.rotation_wxyz[1], y = camera_.rotation_wxyz[2], z = camera_.rotation_wxyz[3];
    // glm::pitch (gtc/quaternion.inl): atan(2 (y z + w x), w w - x x - y y + z z), 2 atan(x, w) when both vanish
    const float py = 2.0f * (y * z + w * x), px = w * w - x * x - y * y + z * z;
    const float eps = 1.1920929e-7f;
    pitch_ = (std::fabs(px) <= eps && std::fabs(py) <= eps) ? 2.0f * (float)std::atan2((double)x, (double)w)
                                                             : (float)std::atan2((double)py, (double)px);
    // glm::yaw: asin(clamp(-2 (x z - w y), -1, 1))
    float sy = -2.0f * (x * z - w * y);
    sy = sy < -1.0f ? -1.0f : (sy > 1.0f ? 1.0f : sy);
    yaw_ = (float)std::asin((double)sy);
    speed = default_speed;
    update_camera();
  }

  // cpp:54-90: true when the key moved the camera (the viewer then restarts the accumulation, app.cpp:64-67)
  bool on_key_press(int key)
  {
    float d[3];
    switch (key) {
    case 'R': d[0] = 0, d[1] = 1, d[2] = 0; break;    // up
    case 'F': d[0] = 0, d[1] = -1, d[2] = 0; break;   // down
    case 'A': d[0] = 1, d[1] = 0, d[2] = 0; break;    // "left" (the reference's own sign: +x)
    case 'D': d[0] = -1, d[1] = 0, d[2] = 0; break;   // "right"
    case 'W': d[0] = 0, d[1] = 0, d[2] = -1; break;   // forward
    case 'S': d[0] = 0, d[1] = 0, d[2] = 1; break;    // backward
    default: return false;
    }
    float m[9];
    yaw_pitch(yaw_, pitch_, m);
    const float t[3] = {speed * d[0], speed * d[1], speed * d[2]};
    // mat4 * vec4(t, 0): column 0 * t.x + column 1 * t.y + column 2 * t.z
    for (int r = 0; r < 3; ++r) position_[r] += (m[r] * t[0] + m[3 + r] * t[1]) + m[6 + r] * t[2];
    update_camera();
    return true;
  }
  // cpp:92-100 (offsets in radians: the viewer converts pixels with glm::radians, app.cpp:110-111)
  bool on_mouse_move(float x_offset, float y_offset)
  {
    set_yaw(yaw_ + x_offset);
    set_pitch(pitch_ + y_offset);
    update_camera();
    return true;
  }
  // cpp:12-16
  void update_camera()
  {
    for (int i = 0; i < 3; ++i) camera_.position[i] = position_[i];
    float m[9];
    yaw_pitch(yaw_, pitch_, m);
    quat_cast(m, camera_.rotation_wxyz);
  }

  // glm::yawPitchRoll(yaw, pitch, 0) (gtx/euler_angles.inl), upper 3x3, column-major m[3 * col + row]
  static void yaw_pitch(float yaw, float pitch, float m[9])
  {
    const float ch = (float)std::cos((double)yaw), sh = (float)std::sin((double)yaw);
    const float cp = (float)std::cos((double)pitch), sp = (float)std::sin((double)pitch);
    m[0] = ch;       m[1] = 0.0f;  m[2] = -sh;
    m[3] = sh * sp;  m[4] = cp;    m[5] = ch * sp;
    m[6] = sh * cp;  m[7] = -sp;   m[8] = ch * cp;
  }
  // glm::quat_cast(mat3) (gtc/quaternion.inl): largest of the four squared components first
  static void quat_cast(const float m[9], float wxyz[4])
  {
    const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
    const float fx = m00 - m11 - m22, fy = m11 - m00 - m22, fz = m22 - m00 - m11, fw = m00 + m11 + m22;
    int biggest = 0;
    float big = fw;
    if (fx > big) { big = fx; biggest = 1; }
    if (fy > big) { big = fy; biggest = 2; }
    if (fz > big) { big = fz; biggest = 3; }
    const float val = (float)std::sqrt((double)(big + 1.0f)) * 0.5f, mult = 0.25f / val;
    switch (biggest) {
    case 0: wxyz[0] = val; wxyz[1] = (m12 - m21) * mult; wxyz[2] = (m20 - m02) * mult; wxyz[3] = (m01 - m10) * mult; break;
    case 1: wxyz[0] = (m12 - m21) * mult; wxyz[1] = val; wxyz[2] = (m01 + m10) * mult; wxyz[3] = (m20 + m02) * mult; break;
    case 2: wxyz[0] = (m20 - m02) * mult; wxyz[1] = (m01 + m10) * mult; wxyz[2] = val; wxyz[3] = (m12 + m21) * mult; break;
    default: wxyz[0] = (m01 - m10) * mult; wxyz[1] = (m20 + m02) * mult; wxyz[2] = (m12 + m21) * mult; wxyz[3] = val; break;
    }
  }

private:
  Camera& camera_;
  float position_[3] = {0, 0, 0};
  float pitch_ = 0.0f, yaw_ = 0.0f;
};

}  // namespace hip_pt
