// pt_math.hpp -- float vector / matrix arithmetic of the render core.
//
// Every function is a fixed sequence of IEEE binary32 operations; the library is compiled with
// -ffp-contract=off and correctly rounded divide/sqrt, so host and gfx950 produce the same bits.
// The operation ORDER follows glm 0.9.9.8 (the reference's math library, conanfile.txt:6), because
// the reference's results are defined by it:  dot = (x*x' + y*y') + z*z',  normalize = v * (1/sqrt(dot)),
// mat4*vec4 = (c0*x + c1*y) + (c2*z + c3*w),  min(a,b) = b<a ? b : a,  max(a,b) = a<b ? b : a.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_HD __host__ __device__ __forceinline__

namespace pt {

struct f3 {
  float x, y, z;
};

PT_HD f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
PT_HD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_HD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_HD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_HD f3 operator/(f3 a, f3 b) { return f3{a.x / b.x, a.y / b.y, a.z / b.z}; }
PT_HD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
PT_HD f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
PT_HD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }

PT_HD float sel_min(float a, float b) { return (b < a) ? b : a; }
PT_HD float sel_max(float a, float b) { return (a < b) ? b : a; }
PT_HD f3 min3(f3 a, f3 b) { return f3{sel_min(a.x, b.x), sel_min(a.y, b.y), sel_min(a.z, b.z)}; }
PT_HD f3 max3(f3 a, f3 b) { return f3{sel_max(a.x, b.x), sel_max(a.y, b.y), sel_max(a.z, b.z)}; }

PT_HD float dot(f3 a, f3 b)
{
  const float px = a.x * b.x, py = a.y * b.y, pz = a.z * b.z;
  return (px + py) + pz;
}
PT_HD f3 cross(f3 a, f3 b)
{
  return f3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
PT_HD float ieee_sqrt(float x)
{
  return __builtin_sqrtf(x);  // correctly rounded: built with -fhip-fp32-correctly-rounded-divide-sqrt
}
PT_HD float length(f3 a) { return ieee_sqrt(dot(a, a)); }
PT_HD f3 normalize(f3 a) { return a * (1.0f / ieee_sqrt(dot(a, a))); }
PT_HD float sign_of(float x) { return (float)((int)(0.0f < x) - (int)(x < 0.0f)); }

// column-major 4x4: c[col][row]
struct m4 {
  float c[4][4];
};

struct f4 {
  float x, y, z, w;
};

PT_HD f4 mul(const m4& m, float x, float y, float z, float w)
{
  f4 r;
  r.x = (m.c[0][0] * x + m.c[1][0] * y) + (m.c[2][0] * z + m.c[3][0] * w);
  r.y = (m.c[0][1] * x + m.c[1][1] * y) + (m.c[2][1] * z + m.c[3][1] * w);
  r.z = (m.c[0][2] * x + m.c[1][2] * y) + (m.c[2][2] * z + m.c[3][2] * w);
  r.w = (m.c[0][3] * x + m.c[1][3] * y) + (m.c[2][3] * z + m.c[3][3] * w);
  return r;
}
// transpose(m) * v
PT_HD f4 mul_transposed(const m4& m, float x, float y, float z, float w)
{
  f4 r;
  r.x = (m.c[0][0] * x + m.c[0][1] * y) + (m.c[0][2] * z + m.c[0][3] * w);
  r.y = (m.c[1][0] * x + m.c[1][1] * y) + (m.c[1][2] * z + m.c[1][3] * w);
  r.z = (m.c[2][0] * x + m.c[2][1] * y) + (m.c[2][2] * z + m.c[2][3] * w);
  r.w = (m.c[3][0] * x + m.c[3][1] * y) + (m.c[3][2] * z + m.c[3][3] * w);
  return r;
}
// transform_point: (M * (p,1)).xyz / w         (reference transform.hpp:37-42)
PT_HD f3 xform_point(const m4& m, f3 p)
{
  const f4 v = mul(m, p.x, p.y, p.z, 1.0f);
  return mk3(v.x, v.y, v.z) / v.w;
}
// transform_vector: (M * (v,0)).xyz            (transform.hpp:44-49)
PT_HD f3 xform_vector(const m4& m, f3 p)
{
  const f4 v = mul(m, p.x, p.y, p.z, 0.0f);
  return mk3(v.x, v.y, v.z);
}
// transform_normal: (transpose(M^-1) * (n,0)).xyz, not renormalised  (transform.hpp:60-66)
PT_HD f3 xform_normal(const m4& inv_m, f3 n)
{
  const f4 v = mul_transposed(inv_m, n.x, n.y, n.z, 0.0f);
  return mk3(v.x, v.y, v.z);
}

// Deterministic sin/cos for |x| < 8192: three-constant Cody-Waite reduction to [-pi/4, pi/4] and
// the classic single-precision minimax polynomials, spelled out operation by operation so that the
// host and the GPU agree bit for bit (libm / OCML sinf differ from each other in the last ulp).
PT_HD void det_sincos(float x, float& s_out, float& c_out)
{
  float sign_s = 1.0f, sign_c = 1.0f;
  float ax = x;
  if (x < 0.0f) {
    sign_s = -1.0f;
    ax = -x;
  }
  uint32_t j = (uint32_t)(ax * 1.27323954473516f);
  float y = (float)j;
  if (j & 1u) {
    j += 1u;
    y += 1.0f;
  }
  j &= 7u;
  if (j > 3u) {
    sign_s = -sign_s;
    sign_c = -sign_c;
    j -= 4u;
  }
  if (j > 1u) sign_c = -sign_c;
  const float r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
  const float z = r * r;
  const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
  pc = pc - 0.5f * z;
  pc = pc + 1.0f;
  const bool swap = (j == 1u) || (j == 2u);
  s_out = sign_s * (swap ? pc : ps);
  c_out = sign_c * (swap ? ps : pc);
}

}  // namespace pt
