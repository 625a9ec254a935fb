/*
 * oracle.c -- CPU restatement of the hot path of LesleyLai/cuda-path-tracer (see oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP path, and bench.py's cpu_baseline ("port").
 *
 * How it follows the reference.  The reference path is CUDA + Thrust + glm and cannot be compiled
 * here (no nvcc; glm/Thrust-CUDA absent), so every function below restates the reference code it
 * cites, keeping the reference's data layout (32-B Ray, 48-B Intersection, 6-array Paths SoA,
 * 32-B BVH nodes), its control flow (per-bounce intersect -> material -> stable partition with
 * dead paths kept behind, final gather over all W*H slots) and all quirks of SURVEY.md Appendix A.
 *
 * Third-party arithmetic restated (sources absent from /root/reference):
 *   - Thrust 12.1 minstd_rand / uniform_real_distribution / discard: restated from the published
 *     algorithm; checked against rocThrust 7.2 run on the host (tests/golden/rng_kat.json).
 *   - glm 0.9.9.8 (conanfile.txt:6): dot/cross/normalize/reflect/refract/mix/inverse/mat*vec
 *     restated from glm's published formulas WITH glm's operation order (noted at each function).
 *
 * Deliberate, documented deviations from "call libm like the CUDA code calls libdevice":
 *   - sinf/cosf in random_in_unit_sphere (distributions.cuh:9-18) use orc_sincos below, a fixed
 *     sequence of IEEE binary32 adds/multiplies (Cephes-style; <= ~1 ulp from libm on [0, 2 pi]).
 *     Reason: the streaming mode re-seeds its RNG from the COMPACTED slot index, so one path whose
 *     hit/miss decision differs in one ulp shifts every later path's random numbers; a checker that
 *     can only be matched statistically cannot pin anything.  With a fully specified sin/cos the
 *     HIP path can be (and is) compared with this oracle bit for bit.
 *   - pow(1-cos, 5) in reflectance (path_tracer.cu:130-136) is x2=x*x; x4=x2*x2; x5=x4*x.
 *   - tan(vfov/2) in generate_ray (ray_gen.cu:40) is evaluated once on the host with libm tanf.
 *   These three are the only places where this file is not a transliteration.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math (see Makefile).
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* ------------------------------------------------------------------------------------------------
 * vec / mat helpers with glm's operation order
 * ---------------------------------------------------------------------------------------------- */
static inline ovec3 v3(float x, float y, float z) { ovec3 r = {x, y, z}; return r; }
static inline ovec3 vadd(ovec3 a, ovec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline ovec3 vsub(ovec3 a, ovec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline ovec3 vmul(ovec3 a, ovec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline ovec3 vdiv(ovec3 a, ovec3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline ovec3 vscale(ovec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline ovec3 vdivs(ovec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline ovec3 vneg(ovec3 a) { return v3(-a.x, -a.y, -a.z); }
/* glm::dot(vec3): tmp = a*b; return tmp.x + tmp.y + tmp.z */
static inline float vdot(ovec3 a, ovec3 b) { ovec3 t = vmul(a, b); return t.x + t.y + t.z; }
/* glm::cross */
static inline ovec3 vcross(ovec3 x, ovec3 y)
{
  return v3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* glm::length = sqrt(dot(v,v)) */
static inline float vlength(ovec3 a) { return sqrtf(vdot(a, a)); }
/* glm::normalize = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x) */
static inline ovec3 vnormalize(ovec3 a) { return vscale(a, 1.0f / sqrtf(vdot(a, a))); }
/* glm::min(x,y) = (y < x) ? y : x ; glm::max(x,y) = (x < y) ? y : x   (std::min/max are the same) */
static inline float fmin_sel(float x, float y) { return (y < x) ? y : x; }
static inline float fmax_sel(float x, float y) { return (x < y) ? y : x; }
static inline ovec3 vmin(ovec3 a, ovec3 b) { return v3(fmin_sel(a.x, b.x), fmin_sel(a.y, b.y), fmin_sel(a.z, b.z)); }
static inline ovec3 vmax(ovec3 a, ovec3 b) { return v3(fmax_sel(a.x, b.x), fmax_sel(a.y, b.y), fmax_sel(a.z, b.z)); }
static inline ovec3 vload(const float* p) { return v3(p[0], p[1], p[2]); }

typedef struct { float x, y, z, w; } ovec4;
/* glm operator*(mat4, vec4): (m[0]*v.x + m[1]*v.y) + (m[2]*v.z + m[3]*v.w) */
static inline ovec4 mat_mul_vec4(const omat4* m, float x, float y, float z, float w)
{
  ovec4 r;
  r.x = (m->c[0][0] * x + m->c[1][0] * y) + (m->c[2][0] * z + m->c[3][0] * w);
  r.y = (m->c[0][1] * x + m->c[1][1] * y) + (m->c[2][1] * z + m->c[3][1] * w);
  r.z = (m->c[0][2] * x + m->c[1][2] * y) + (m->c[2][2] * z + m->c[3][2] * w);
  r.w = (m->c[0][3] * x + m->c[1][3] * y) + (m->c[2][3] * z + m->c[3][3] * w);
  return r;
}
/* transpose(m) * vec4 without materialising the transpose */
static inline ovec4 matT_mul_vec4(const omat4* m, float x, float y, float z, float w)
{
  ovec4 r;
  r.x = (m->c[0][0] * x + m->c[0][1] * y) + (m->c[0][2] * z + m->c[0][3] * w);
  r.y = (m->c[1][0] * x + m->c[1][1] * y) + (m->c[1][2] * z + m->c[1][3] * w);
  r.z = (m->c[2][0] * x + m->c[2][1] * y) + (m->c[2][2] * z + m->c[2][3] * w);
  r.w = (m->c[3][0] * x + m->c[3][1] * y) + (m->c[3][2] * z + m->c[3][3] * w);
  return r;
}
/* glm operator*(mat4, mat4): Result[j] = ((A0*b0 + A1*b1) + A2*b2) + A3*b3 */
static void mat_mul(const omat4* a, const omat4* b, omat4* out)
{
  omat4 r;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i)
      r.c[j][i] = ((a->c[0][i] * b->c[j][0] + a->c[1][i] * b->c[j][1]) + a->c[2][i] * b->c[j][2]) +
                  a->c[3][i] * b->c[j][3];
  *out = r;
}

/* glm::inverse(mat4) (func_matrix.inl, compute_inverse<4,4>) */
void orc_mat4_inverse(const omat4* mm, omat4* out)
{
#define M(c_, r_) (mm->c[c_][r_])
  float Coef00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
  float Coef02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
  float Coef03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
  float Coef04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
  float Coef06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
  float Coef07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
  float Coef08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
  float Coef10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
  float Coef11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
  float Coef12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
  float Coef14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
  float Coef15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
  float Coef16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
  float Coef18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
  float Coef19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
  float Coef20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
  float Coef22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
  float Coef23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);

  float Fac0[4] = {Coef00, Coef00, Coef02, Coef03};
  float Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
  float Fac2[4] = {Coef08, Coef08, Coef10, Coef11};
  float Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
  float Fac4[4] = {Coef16, Coef16, Coef18, Coef19};
  float Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
  float Vec0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)};
  float Vec1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)};
  float Vec2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)};
  float Vec3[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};
  static const float SignA[4] = {+1, -1, +1, -1};
  static const float SignB[4] = {-1, +1, -1, +1};
  omat4 inv;
  for (int i = 0; i < 4; ++i) {
    float Inv0 = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i];
    float Inv1 = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i];
    float Inv2 = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i];
    float Inv3 = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i];
    inv.c[0][i] = Inv0 * SignA[i];
    inv.c[1][i] = Inv1 * SignB[i];
    inv.c[2][i] = Inv2 * SignA[i];
    inv.c[3][i] = Inv3 * SignB[i];
  }
  float d0 = M(0, 0) * inv.c[0][0], d1 = M(0, 1) * inv.c[1][0], d2 = M(0, 2) * inv.c[2][0],
        d3 = M(0, 3) * inv.c[3][0];
  float Dot1 = (d0 + d1) + (d2 + d3);
  float OneOverDeterminant = 1.0f / Dot1;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i) out->c[j][i] = inv.c[j][i] * OneOverDeterminant;
#undef M
}

/* ------------------------------------------------------------------------------------------------
 * hash + RNG   (hash.cuh:4-14; Thrust minstd_rand + uniform_real_distribution<float>)
 * ---------------------------------------------------------------------------------------------- */
uint32_t orc_hash(uint32_t a)
{
  a = (a + 0x7ed55d16u) + (a << 12);
  a = (a ^ 0xc761c23cu) ^ (a >> 19);
  a = (a + 0x165667b1u) + (a << 5);
  a = (a + 0xd3a2646cu) ^ (a << 9);
  a = (a + 0xfd7046c5u) + (a << 3);
  a = (a ^ 0xb55a4f09u) ^ (a >> 16);
  return a;
}

#define LCG_A 48271u
#define LCG_M 2147483647u

/* linear_congruential_engine::seed: c == 0, so s % m == 0 maps to 1 */
uint32_t orc_rng_seed(uint32_t s)
{
  uint32_t x = s % LCG_M;
  return x == 0 ? 1u : x;
}
uint32_t orc_rng_next(uint32_t* state)
{
  *state = (uint32_t)(((uint64_t)*state * LCG_A) % LCG_M);
  return *state;
}
/* linear_congruential_engine_discard (uint32, c == 0): state *= a^z mod m */
void orc_rng_discard(uint32_t* state, uint64_t z)
{
  uint64_t mult = LCG_A, mult_to_z = 1;
  while (z > 0) {
    if (z & 1) mult_to_z = (mult_to_z * mult) % LCG_M;
    z >>= 1;
    mult = (mult * mult) % LCG_M;
  }
  *state = (uint32_t)((mult_to_z * (uint64_t)*state) % LCG_M);
}
/* uniform_real_distribution<float>(0,1): float(x - min) / (1.0f + float(max - min)), min=1, max=m-1.
 * float(2147483645) + 1.0f == 2^31, so this can return exactly 1.0f. */
float orc_rng_uniform(uint32_t* state)
{
  uint32_t x = orc_rng_next(state);
  float result = (float)(x - 1u);
  result /= (1.0f + (float)(LCG_M - 1u - 1u));
  return (result * (1.0f - 0.0f)) + 0.0f;
}
/* hash(hash(index) ^ iteration): the XOR is size_t-wide then truncated by hash's parameter
 * (ray_gen.cu:18, path_tracer.cu:239,300) */
uint32_t orc_path_seed(uint32_t index, uint64_t iteration)
{
  return orc_hash((uint32_t)((uint64_t)orc_hash(index) ^ iteration));
}

/* Deterministic sin/cos: Cephes sinf/cosf argument reduction and polynomials, every operation an
 * IEEE binary32 add or multiply in the order written (no FMA).  Valid for |x| < 8192. */
void orc_sincos(float x, float* s_out, float* c_out)
{
  float sign_s = 1.0f, sign_c = 1.0f;
  float ax = x;
  if (x < 0.0f) { sign_s = -1.0f; ax = -x; }
  uint32_t j = (uint32_t)(ax * 1.27323954473516f); /* 4/pi */
  float y = (float)j;
  if (j & 1u) { j += 1u; y += 1.0f; }
  j &= 7u;
  if (j > 3u) { sign_s = -sign_s; sign_c = -sign_c; j -= 4u; }
  if (j > 1u) sign_c = -sign_c;
  float r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
  float z = r * r;
  float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
  pc = pc - 0.5f * z;
  pc = pc + 1.0f;
  int swap = (j == 1u) || (j == 2u);
  *s_out = sign_s * (swap ? pc : ps);
  *c_out = sign_c * (swap ? ps : pc);
}

/* ------------------------------------------------------------------------------------------------
 * camera + raygen  (camera.cpp:5-13, ray_gen.cu:34-61)
 * ---------------------------------------------------------------------------------------------- */
void orc_to_gpu_camera(const OCamera* cam, uint32_t w, uint32_t h, OGPUCamera* out)
{
  /* glm::mat4_cast(quat) */
  const float qw = cam->rotation_wxyz[0], qx = cam->rotation_wxyz[1], qy = cam->rotation_wxyz[2],
              qz = cam->rotation_wxyz[3];
  float qxx = qx * qx, qyy = qy * qy, qzz = qz * qz, qxz = qx * qz, qxy = qx * qy, qyz = qy * qz,
        qwx = qw * qx, qwy = qw * qy, qwz = qw * qz;
  omat4 r;
  memset(&r, 0, sizeof r);
  r.c[0][0] = 1.0f - 2.0f * (qyy + qzz);
  r.c[0][1] = 2.0f * (qxy + qwz);
  r.c[0][2] = 2.0f * (qxz - qwy);
  r.c[1][0] = 2.0f * (qxy - qwz);
  r.c[1][1] = 1.0f - 2.0f * (qxx + qzz);
  r.c[1][2] = 2.0f * (qyz + qwx);
  r.c[2][0] = 2.0f * (qxz + qwy);
  r.c[2][1] = 2.0f * (qyz - qwx);
  r.c[2][2] = 1.0f - 2.0f * (qxx + qyy);
  r.c[3][3] = 1.0f;
  /* glm::translate(identity, position): Result[3] = m[0]*v.x + m[1]*v.y + m[2]*v.z + m[3] */
  omat4 t;
  memset(&t, 0, sizeof t);
  t.c[0][0] = t.c[1][1] = t.c[2][2] = t.c[3][3] = 1.0f;
  for (int i = 0; i < 4; ++i)
    t.c[3][i] = ((t.c[0][i] * cam->position[0] + t.c[1][i] * cam->position[1]) +
                 t.c[2][i] * cam->position[2]) + t.c[3][i];
  mat_mul(&t, &r, &out->camera_matrix);
  out->vfov = cam->vfov;
  out->width = w;
  out->height = h;
}

void orc_generate_ray(const OGPUCamera* camera, float x, float y, ORay* out)
{
  const float aspect_ratio = (float)camera->width / (float)camera->height;
#ifdef ORC_LIBM
  /* "another libm": the tangent rounded from double precision (a correctly rounded tanf; glibc's differs from it by
   * at most an ulp, CUDA's by a little more), see the ORC_LIBM note at random_in_unit_sphere */
  const float viewport_height = 2.0f * (float)tan((double)(camera->vfov / 2));
#else
  const float viewport_height = 2.0f * tanf(camera->vfov / 2);
#endif
  const float viewport_width = aspect_ratio * viewport_height;
  const float focal_length = 1.0f;

  const ovec3 origin = v3(0, 0, 0);
  const ovec3 horizontal = v3(viewport_width, 0, 0);
  const ovec3 vertical = v3(0, viewport_height, 0);
  const ovec3 lower_left_corner =
      vsub(vsub(vsub(origin, vdivs(horizontal, 2.f)), vdivs(vertical, 2.f)), v3(0, 0, focal_length));

  const float u = x / (float)(camera->width - 1);
  const float v = ((float)camera->height - y) / (float)(camera->height - 1);
  const ovec3 direction =
      vsub(vadd(vadd(lower_left_corner, vscale(horizontal, u)), vscale(vertical, v)), origin);

  const ovec4 wo = mat_mul_vec4(&camera->camera_matrix, origin.x, origin.y, origin.z, 1.0f);
  const ovec4 wd = mat_mul_vec4(&camera->camera_matrix, direction.x, direction.y, direction.z, 0.0f);
  out->origin = v3(wo.x, wo.y, wo.z);
  out->t_min = 1e-4f;
  out->direction = vnormalize(v3(wd.x, wd.y, wd.z));
  out->t_max = FLT_MAX;
}

/* ------------------------------------------------------------------------------------------------
 * AABB / transform / intersections
 * ---------------------------------------------------------------------------------------------- */
static inline int aabb_is_empty(const OAABB* b)
{
  return b->min.x > b->max.x || b->min.y > b->max.y || b->min.z > b->max.z;
}
static inline OAABB aabb_empty(void)
{
  OAABB b = {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
  return b;
}
static inline OAABB aabb_enclose_pt(OAABB b, ovec3 p) { OAABB r = {vmin(b.min, p), vmax(b.max, p)}; return r; }
static inline OAABB aabb_union(OAABB a, OAABB b) { OAABB r = {vmin(a.min, b.min), vmax(a.max, b.max)}; return r; }
static inline ovec3 aabb_center(const OAABB* b) { return vdivs(vadd(b->min, b->max), 2.0f); }

float orc_aabb_surface_area(const OAABB* b)
{
  const ovec3 d = vsub(b->max, b->min);
  return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
}
/* aabb.hpp:41-44: extent() = max - min */
void orc_aabb_extent(const OAABB* b, float* out)
{
  const ovec3 e = vsub(b->max, b->min);
  out[0] = e.x; out[1] = e.y; out[2] = e.z;
}
int orc_aabb_max_extent(const OAABB* b)
{
  const ovec3 e = vsub(b->max, b->min);
  return (e.x > e.y && e.x > e.z) ? 0 : (e.y > e.z) ? 1 : 2;
}
void orc_aabb_offset(const OAABB* b, const float* p, float* out)
{
  ovec3 o = vsub(vload(p), b->min);
  if (b->max.x > b->min.x) o.x /= b->max.x - b->min.x;
  if (b->max.y > b->min.y) o.y /= b->max.y - b->min.y;
  if (b->max.z > b->min.z) o.z /= b->max.z - b->min.z;
  out[0] = o.x; out[1] = o.y; out[2] = o.z;
}

/* transform.hpp:37-42: (m * vec4(p,1)).xyz / w */
static inline ovec3 transform_point(const omat4* m, ovec3 p)
{
  const ovec4 v = mat_mul_vec4(m, p.x, p.y, p.z, 1.0f);
  return vdivs(v3(v.x, v.y, v.z), v.w);
}
/* transform.hpp:44-49 */
static inline ovec3 transform_vector(const omat4* m, ovec3 p)
{
  const ovec4 v = mat_mul_vec4(m, p.x, p.y, p.z, 0.0f);
  return v3(v.x, v.y, v.z);
}
/* transform.hpp:60-66: transpose(inverse_m) * vec4(n,0), not re-normalised */
static inline ovec3 transform_normal(const omat4* inv_m, ovec3 n)
{
  const ovec4 v = matT_mul_vec4(inv_m, n.x, n.y, n.z, 0.0f);
  return v3(v.x, v.y, v.z);
}
/* transform.hpp:51-58: t_min/t_max copied unscaled, direction re-normalised */
void orc_inverse_transform_ray(const omat4* m, const omat4* inv_m, const ORay* ray, ORay* out)
{
  (void)m;
  const ovec3 origin = transform_point(inv_m, ray->origin);
  const ovec4 d = mat_mul_vec4(inv_m, ray->direction.x, ray->direction.y, ray->direction.z, 0.0f);
  out->origin = origin;
  out->t_min = ray->t_min;
  out->direction = vnormalize(v3(d.x, d.y, d.z));
  out->t_max = ray->t_max;
}
/* transform.hpp:69-88 */
void orc_transform_aabb(const omat4* m, const OAABB* in, OAABB* out)
{
  if (aabb_is_empty(in)) { *out = *in; return; }
  ovec3 pts[8];
  pts[0].x = pts[1].x = pts[2].x = pts[3].x = in->min.x;
  pts[4].x = pts[5].x = pts[6].x = pts[7].x = in->max.x;
  pts[0].y = pts[1].y = pts[4].y = pts[5].y = in->min.y;
  pts[2].y = pts[3].y = pts[6].y = pts[7].y = in->max.y;
  pts[0].z = pts[2].z = pts[4].z = pts[6].z = in->min.z;
  pts[1].z = pts[3].z = pts[5].z = pts[7].z = in->max.z;
  const ovec3 p0 = transform_point(m, pts[0]);
  OAABB nb = {p0, p0};
  for (int i = 1; i < 8; ++i) nb = aabb_enclose_pt(nb, transform_point(m, pts[i]));
  *out = nb;
}

/* scene_description.cpp:17-52 */
void orc_make_object(uint32_t type, uint32_t index, const omat4* m, const OSphere* sphere,
                     const OAABB* mesh_aabb, OObject* out)
{
  memset(out, 0, sizeof *out);
  out->type = type;
  out->index = index;
  out->m = *m;
  orc_mat4_inverse(m, &out->inv_m);
  if (type == 0) {
    const ovec3 transformed_origin = transform_point(m, sphere->center);
    const float transformed_radius = vlength(transform_vector(m, v3(1.0f, 0.0f, 0.0f))) * sphere->radius;
    const ovec3 r3 = v3(transformed_radius, transformed_radius, transformed_radius);
    out->aabb.min = vsub(transformed_origin, r3);
    out->aabb.max = vadd(transformed_origin, r3);
  } else {
    out->index = 0; /* scene_description.cpp:42: every mesh object references mesh 0 */
    orc_transform_aabb(m, mesh_aabb, &out->aabb);
  }
}

static inline ovec3 ray_at(const ORay* r, float t) { return vadd(r->origin, vscale(r->direction, t)); }

/* intersections.cuh:7-41 */
int orc_ray_sphere(const ORay* ray, const OSphere* sphere, OIntersection* record)
{
  const ovec3 center = sphere->center;
  const float radius = sphere->radius;
  const ovec3 oc = vsub(ray->origin, center);
  const float a = vdot(ray->direction, ray->direction);
  const float b = 2 * vdot(ray->direction, oc);
  const float c = vdot(oc, oc) - radius * radius;
  const float discrimination = b * b - 4 * a * c;
  if (discrimination < 0) return 0;
  const float sqrt_delta = sqrtf(discrimination);
  const float t1 = (-b - sqrt_delta) / (2 * a);
  const float t2 = (-b + sqrt_delta) / (2 * a);
  float t;
  if (t1 >= ray->t_min && t1 <= ray->t_max) t = t1;
  else if (t2 >= ray->t_min && t2 <= ray->t_max) t = t2;
  else return 0;
  record->t = t;
  record->point = ray_at(ray, t);
  const ovec3 outward_normal = vdivs(vsub(record->point, center), radius);
  record->side = vdot(ray->direction, outward_normal) < 0 ? 0 : 1;
  record->normal = record->side == 0 ? outward_normal : vneg(outward_normal);
  return 1;
}

/* intersections.cuh:43-85 (Moeller-Trumbore; t == t_max accepted; material_id = 1) */
static int ray_triangle(const ORay* ray, ovec3 pt0, ovec3 pt1, ovec3 pt2, OIntersection* record)
{
  const float EPSILON = 0.0000001f;
  const ovec3 edge1 = vsub(pt1, pt0);
  const ovec3 edge2 = vsub(pt2, pt0);
  const ovec3 h = vcross(ray->direction, edge2);
  const float a = vdot(edge1, h);
  if (a > -EPSILON && a < EPSILON) return 0;
  const float f = 1.0f / a;
  const ovec3 s = vsub(ray->origin, pt0);
  const float u = f * vdot(s, h);
  if (u < 0.0 || u > 1.0) return 0;
  const ovec3 q = vcross(s, edge1);
  const float v = f * vdot(ray->direction, q);
  if (v < 0.0 || u + v > 1.0) return 0;
  const float t = f * vdot(edge2, q);
  if ((t < ray->t_min) || (t > ray->t_max)) return 0;
  record->t = t;
  record->point = ray_at(ray, t);
  const ovec3 outward_normal = vnormalize(vcross(vsub(pt1, pt0), vsub(pt2, pt0)));
  record->side = vdot(ray->direction, outward_normal) < 0 ? 0 : 1;
  record->normal = record->side == 0 ? outward_normal : vneg(outward_normal);
  record->material_id = 1;
  return 1;
}
int orc_ray_triangle(const ORay* ray, const float* p0, const float* p1, const float* p2, OIntersection* rec)
{
  return ray_triangle(ray, vload(p0), vload(p1), vload(p2), rec);
}

/* intersections.cuh:87-103: ignores t_min/t_max, accepts boxes behind the origin */
int orc_ray_aabb(const ORay* ray, const OAABB* aabb)
{
  if (aabb_is_empty(aabb)) return 0;
  const ovec3 t_min = vdiv(vsub(aabb->min, ray->origin), ray->direction);
  const ovec3 t_max = vdiv(vsub(aabb->max, ray->origin), ray->direction);
  const ovec3 real_min = vmin(t_min, t_max);
  const ovec3 real_max = vmax(t_min, t_max);
  const float minmax = fmin_sel(fmin_sel(real_max.x, real_max.y), real_max.z);
  const float maxmin = fmax_sel(fmax_sel(real_min.x, real_min.y), real_min.z);
  return minmax >= maxmin;
}

/* ------------------------------------------------------------------------------------------------
 * BVH builder  (accelerators/bvh.cpp)
 * ---------------------------------------------------------------------------------------------- */
typedef struct BNode {
  OAABB aabb;
  int32_t left, right;   /* -1 for leaves */
  uint32_t tri_begin;    /* leaves: offset into the index array */
} BNode;

typedef struct {
  BNode* nodes;
  uint32_t node_count;
  OAABB* leaf_aabb;      /* per triangle */
  ovec3* leaf_center;    /* per triangle */
  int error;
} BBuild;

static uint32_t bb_new_leaf(BBuild* bb, uint32_t tri)
{
  BNode* n = &bb->nodes[bb->node_count];
  n->aabb = bb->leaf_aabb[tri];
  n->left = n->right = -1;
  n->tri_begin = tri * 3u;
  return bb->node_count++;
}
static uint32_t bb_new_inner(BBuild* bb, uint32_t l, uint32_t r)
{
  BNode* n = &bb->nodes[bb->node_count];
  n->aabb = aabb_union(bb->nodes[l].aabb, bb->nodes[r].aabb); /* bvh.cpp:51 */
  n->left = (int32_t)l;
  n->right = (int32_t)r;
  n->tri_begin = 0;
  return bb->node_count++;
}
static inline float vcomp(ovec3 v, int axis) { return axis == 0 ? v.x : axis == 1 ? v.y : v.z; }

/* order used wherever the reference leaves element order implementation-defined
 * (nth_element ties): centroid along axis, then triangle index */
typedef struct { float key; uint32_t tri; } KeyTri;
static int keytri_cmp(const void* a, const void* b)
{
  const KeyTri* x = (const KeyTri*)a;
  const KeyTri* y = (const KeyTri*)b;
  if (x->key < y->key) return -1;
  if (x->key > y->key) return 1;
  return (x->tri > y->tri) - (x->tri < y->tri);
}

static int32_t bb_build(BBuild* bb, uint32_t* tris, uint32_t n); /* fwd */

/* bvh.cpp:112-182 */
static int32_t bb_split_sah(BBuild* bb, uint32_t* tris, uint32_t n, const OAABB* centroid_bound, int axis)
{
  enum { buckets_count = 12 };
  int count[buckets_count];
  OAABB bounds[buckets_count];
  for (int i = 0; i < buckets_count; ++i) { count[i] = 0; bounds[i] = aabb_empty(); }
  OAABB bound = aabb_empty();

#define FIND_BUCKET(tri_, out_)                                                     \
  do {                                                                              \
    float off3[3];                                                                  \
    orc_aabb_offset(centroid_bound, &bb->leaf_center[tri_].x, off3);                \
    size_t b_ = (size_t)(int)((float)buckets_count * off3[axis]);                   \
    if (b_ == buckets_count) b_ = buckets_count - 1;                                \
    (out_) = b_;                                                                    \
  } while (0)

  for (uint32_t i = 0; i < n; ++i) {
    size_t b;
    FIND_BUCKET(tris[i], b);
    if (b >= buckets_count) { bb->error = -3; return -1; } /* reference: out-of-bounds write (UB) */
    count[b]++;
    bounds[b] = aabb_union(bounds[b], bb->leaf_aabb[tris[i]]);
    bound = aabb_union(bound, bb->leaf_aabb[tris[i]]);
  }

  float cost[buckets_count - 1];
  for (int i = 0; i < buckets_count - 1; ++i) {
    OAABB b0 = aabb_empty(), b1 = aabb_empty();
    int count0 = 0, count1 = 0;
    for (int j = 0; j <= i; ++j) { b0 = aabb_union(b0, bounds[j]); count0 += count[j]; }
    for (int j = i + 1; j < buckets_count; ++j) { b1 = aabb_union(b1, bounds[j]); count1 += count[j]; }
    cost[i] = .125f + ((float)count0 * orc_aabb_surface_area(&b0) + (float)count1 * orc_aabb_surface_area(&b1)) /
                          orc_aabb_surface_area(&bound);
  }
  float min_cost = cost[0];
  size_t min_cost_split_bucket = 0;
  for (size_t i = 1; i < buckets_count - 1; ++i) {
    if (cost[i] < min_cost) { min_cost = cost[i]; min_cost_split_bucket = i; }
  }

  /* std::ranges::partition by bucket <= split (order inside each side is irrelevant: everything
   * downstream depends on the SET only) -- done here as a stable partition. */
  uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * n);
  uint32_t nl = 0, nr = 0;
  for (uint32_t i = 0; i < n; ++i) {
    size_t b;
    FIND_BUCKET(tris[i], b);
    if (b <= min_cost_split_bucket) tris[nl++] = tris[i];
    else tmp[nr++] = tris[i];
  }
  memcpy(tris + nl, tmp, sizeof(uint32_t) * nr);
  free(tmp);
#undef FIND_BUCKET
  if (nl == 0 || nr == 0) { bb->error = -2; return -1; } /* panic("Shouldn't happen!") bvh.cpp:84 */
  int32_t l = bb_build(bb, tris, nl);
  if (l < 0) return -1;
  int32_t r = bb_build(bb, tris + nl, nr);
  if (r < 0) return -1;
  return (int32_t)bb_new_inner(bb, (uint32_t)l, (uint32_t)r);
}

/* bvh.cpp:71-110 */
static int32_t bb_build(BBuild* bb, uint32_t* tris, uint32_t n)
{
  OAABB centroid_bound = aabb_empty();
  for (uint32_t i = 0; i < n; ++i) centroid_bound = aabb_enclose_pt(centroid_bound, bb->leaf_center[tris[i]]);
  const int axis = orc_aabb_max_extent(&centroid_bound);

  if (n == 0) { bb->error = -2; return -1; }
  if (n == 1) return (int32_t)bb_new_leaf(bb, tris[0]);
  if (n == 2) {
    uint32_t l = tris[0], r = tris[1];
    const float kl = vcomp(bb->leaf_center[l], axis), kr = vcomp(bb->leaf_center[r], axis);
    /* bvh.cpp:88-90 swaps when left > right; on a tie the reference keeps whatever order
     * std::ranges::partition left behind (implementation-defined) -- here the lower triangle index goes left */
    if (kl > kr || (kl == kr && l > r)) { uint32_t t = l; l = r; r = t; }
    uint32_t ln = bb_new_leaf(bb, l);
    uint32_t rn = bb_new_leaf(bb, r);
    return (int32_t)bb_new_inner(bb, ln, rn);
  }
  if (n <= 4) {
    /* nth_element at n/2 on the centroid: as sets, the n/2 smallest go left */
    KeyTri kt[4];
    for (uint32_t i = 0; i < n; ++i) { kt[i].key = vcomp(bb->leaf_center[tris[i]], axis); kt[i].tri = tris[i]; }
    qsort(kt, n, sizeof(KeyTri), keytri_cmp);
    for (uint32_t i = 0; i < n; ++i) tris[i] = kt[i].tri;
    const uint32_t mid = n / 2;
    int32_t l = bb_build(bb, tris, mid);
    if (l < 0) return -1;
    int32_t r = bb_build(bb, tris + mid, n - mid);
    if (r < 0) return -1;
    return (int32_t)bb_new_inner(bb, (uint32_t)l, (uint32_t)r);
  }
  return bb_split_sah(bb, tris, n, &centroid_bound, axis);
}

int orc_bvh_build(const float* positions, uint32_t vertex_count, const uint32_t* indices,
                  uint32_t index_count, OBVHNode* out, uint32_t* max_depth)
{
  (void)vertex_count;
  const uint32_t T = index_count / 3u;
  if (T == 0) return -1; /* panic("Cannot create BVH for empty mesh") bvh.cpp:200 */
  BBuild bb;
  bb.nodes = (BNode*)malloc(sizeof(BNode) * (2u * (size_t)T));
  bb.node_count = 0;
  bb.leaf_aabb = (OAABB*)malloc(sizeof(OAABB) * T);
  bb.leaf_center = (ovec3*)malloc(sizeof(ovec3) * T);
  bb.error = 0;
  uint32_t* tris = (uint32_t*)malloc(sizeof(uint32_t) * T);
  for (uint32_t t = 0; t < T; ++t) {
    const ovec3 p1 = vload(positions + 3u * (size_t)indices[3u * t]);
    const ovec3 p2 = vload(positions + 3u * (size_t)indices[3u * t + 1]);
    const ovec3 p3 = vload(positions + 3u * (size_t)indices[3u * t + 2]);
    bb.leaf_aabb[t] = aabb_enclose_pt(aabb_enclose_pt(aabb_enclose_pt(aabb_empty(), p1), p2), p3);
    bb.leaf_center[t] = aabb_center(&bb.leaf_aabb[t]);
    tris[t] = t;
  }
  int32_t root = (T == 1) ? (int32_t)bb_new_leaf(&bb, 0) : bb_build(&bb, tris, T);
  int result;
  if (root < 0) {
    result = bb.error ? bb.error : -2;
  } else {
    /* breadth-first flatten, bvh.cpp:211-253 */
    const uint32_t size = 2u * T - 1u;
    uint32_t* queue_node = (uint32_t*)malloc(sizeof(uint32_t) * size);
    uint32_t* queue_depth = (uint32_t*)malloc(sizeof(uint32_t) * size);
    uint32_t head = 0, tail = 0, count = 0, deepest = 0;
    queue_node[tail] = (uint32_t)root; queue_depth[tail] = 0; ++tail;
    {
      const BNode* n = &bb.nodes[root];
      out[count].aabb = n->aabb;
      out[count].first_child_or_primitive = n->left < 0 ? n->tri_begin : 0;
      out[count].primitive_count = n->left < 0 ? 1u : 0u;
      ++count;
    }
    while (head < tail) {
      const uint32_t index = head; /* node k of the queue has linear index k */
      const BNode* cur = &bb.nodes[queue_node[head]];
      const uint32_t depth = queue_depth[head];
      if (depth > deepest) deepest = depth;
      if (cur->left >= 0) {
        out[index].first_child_or_primitive = count;
        const int32_t kids[2] = {cur->left, cur->right};
        for (int k = 0; k < 2; ++k) {
          const BNode* n = &bb.nodes[kids[k]];
          out[count].aabb = n->aabb;
          out[count].first_child_or_primitive = n->left < 0 ? n->tri_begin : 0;
          out[count].primitive_count = n->left < 0 ? 1u : 0u;
          queue_node[tail] = (uint32_t)kids[k]; queue_depth[tail] = depth + 1; ++tail;
          ++count;
        }
      }
      ++head;
    }
    free(queue_node);
    free(queue_depth);
    if (max_depth) *max_depth = deepest;
    result = (int)count;
  }
  free(tris);
  free(bb.leaf_center);
  free(bb.leaf_aabb);
  free(bb.nodes);
  return result;
}

/* ------------------------------------------------------------------------------------------------
 * scene intersection  (path_tracer.cu:36-128)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t* data; uint32_t size, cap; } OStack; /* StaticStack<u32,24> without the overflow UB */

static void ostack_push(OStack* s, uint32_t v)
{
  if (s->size == s->cap) {
    s->cap = s->cap ? s->cap * 2 : 64;
    s->data = (uint32_t*)realloc(s->data, sizeof(uint32_t) * s->cap);
  }
  s->data[s->size++] = v;
}

/* path_tracer.cu:36-76.  The arrays of "the" mesh: the whole scene arrays in the reference's one-mesh scene, the
 * object's slice of them when the scene has a mesh table (OScene::meshes; extension, see oracle.h). */
static int ray_mesh(ORay ray, const OScene* scene, const OObject* obj, OIntersection* record, OStack* stack)
{
  int hit = 0;
  ORay transformed_ray;
  orc_inverse_transform_ray(&obj->m, &obj->inv_m, &ray, &transformed_ray);

  const float* positions = scene->positions;
  const uint32_t* indices = scene->indices;
  const OBVHNode* bvh = scene->bvh;
  uint32_t bvh_node_count = scene->bvh_node_count;
  if (scene->meshes) {
    if (obj->index >= scene->mesh_count) return 0;
    const struct OMeshRange* r = &scene->meshes[obj->index];
    positions += 3u * (size_t)r->first_vertex;
    indices += r->first_index;
    bvh += r->first_bvh_node;
    bvh_node_count = r->bvh_node_count;
  }

  stack->size = 0;
  if (bvh_node_count == 0) return 0; /* empty mesh: the reference panics at build time */
  ostack_push(stack, 0);
  while (stack->size != 0) {
    const uint32_t node_index = stack->data[--stack->size];
    const OBVHNode node = bvh[node_index];
    if (node.primitive_count != 0) {
      const uint32_t i = node.first_child_or_primitive;
      const uint32_t index0 = indices[i];
      const uint32_t index1 = indices[i + 1];
      const uint32_t index2 = indices[i + 2];
      const ovec3 p0 = transform_point(&obj->m, vload(positions + 3u * (size_t)index0));
      const ovec3 p1 = transform_point(&obj->m, vload(positions + 3u * (size_t)index1));
      const ovec3 p2 = transform_point(&obj->m, vload(positions + 3u * (size_t)index2));
      if (ray_triangle(&ray, p0, p1, p2, record)) {
        hit = 1;
        ray.t_max = record->t;
      }
    } else {
      if (orc_ray_aabb(&transformed_ray, &node.aabb)) {
        const uint32_t left_index = node.first_child_or_primitive;
        const uint32_t right_index = left_index + 1;
        ostack_push(stack, right_index);
        ostack_push(stack, left_index);
      }
    }
  }
  return hit;
}

/* path_tracer.cu:78-108 */
static int ray_object(ORay ray, const OObject* obj, const OScene* scene, OIntersection* record, OStack* stack)
{
  int hit = 0;
  if (!orc_ray_aabb(&ray, &obj->aabb)) return 0;
  switch (obj->type) {
  case 0: {
    ORay transformed_ray;
    orc_inverse_transform_ray(&obj->m, &obj->inv_m, &ray, &transformed_ray);
    const OSphere sphere = scene->spheres[obj->index];
    hit = orc_ray_sphere(&transformed_ray, &sphere, record);
    if (hit) {
      record->point = transform_point(&obj->m, record->point);
      record->t = vlength(vsub(record->point, ray.origin)); /* glm::distance(p0,p1) = length(p1 - p0) */
      record->normal = transform_normal(&obj->inv_m, record->normal);
    }
    break;
  }
  case 1: hit = ray_mesh(ray, scene, obj, record, stack); break;
  default: break;
  }
  return hit;
}

/* path_tracer.cu:110-128 */
static int ray_scene(ORay ray, const OScene* scene, OIntersection* record, OStack* stack)
{
  int hit = 0;
  for (size_t i = 0; i < scene->object_count; ++i) {
    const OObject* obj = &scene->objects[i];
    if (ray_object(ray, obj, scene, record, stack)) {
      hit = 1;
      record->material_id = scene->object_material_indices[i];
      ray.t_max = record->t;
    }
  }
  return hit;
}

int orc_scene_intersect(const OScene* scene, const ORay* ray, OIntersection* rec)
{
  OStack st = {0, 0, 0};
  int h = ray_scene(*ray, scene, rec, &st);
  free(st.data);
  return h;
}

/* ------------------------------------------------------------------------------------------------
 * shading  (path_tracer.cu:29-34, 130-201; distributions.cuh:6-19)
 * ---------------------------------------------------------------------------------------------- */
static ovec3 get_background_color(const ORay* r)
{
  const ovec3 unit_direction = vnormalize(r->direction);
  const float t = 0.5f * (unit_direction.y + 1.0f);
  /* glm::lerp(x, y, a) = mix = x * (1 - a) + y * a */
  const ovec3 x = v3(0.5f, 0.7f, 1.0f), y = v3(1.0f, 1.0f, 1.0f);
  return vadd(vscale(x, 1.0f - t), vscale(y, t));
}

static ovec3 random_in_unit_sphere(uint32_t* rng)
{
  const float phi = 2.f * 3.14159265358979323846264338327950288f * orc_rng_uniform(rng);
  const float cos_theta = 2.f * orc_rng_uniform(rng) - 1.f;
  const float sin_theta = sqrtf(1 - cos_theta * cos_theta);
  float s, c;
#ifdef ORC_LIBM
  /* -DORC_LIBM (liboracle_libm.so, make -C oracle libm): the reference's own calls -- sinf / cosf here
   * (distributions.cuh:13-17), pow(x, 5) in reflectance (path_tracer.cu:135), tan in generate_ray (ray_gen.cu:40) --
   * through the platform's libm instead of the three fixed sequences both the oracle and the kernels use.  Only
   * tests/test_oracle_libm.py loads this build: it measures what the substitutions change in the image. */
  s = sinf(phi);
  c = cosf(phi);
#else
  orc_sincos(phi, &s, &c);
#endif
  return v3(c * sin_theta, s * sin_theta, cos_theta);
}

static float reflectance(float cosine, float ref_idx)
{
  float r0 = (1 - ref_idx) / (1 + ref_idx);
  r0 = r0 * r0;
#ifdef ORC_LIBM
  return r0 + (1 - r0) * powf(1 - cosine, 5);
#else
  const float x = 1 - cosine;
  const float x2 = x * x;
  const float x4 = x2 * x2;
  return r0 + (1 - r0) * (x4 * x);
#endif
}

/* glm::sign */
static inline float fsign(float x) { return (float)((0.0f < x) - (x < 0.0f)); }

static void evaluate_material(ORay* ray, const OIntersection* isect, uint32_t* rng, ovec3* color,
                              const OMaterial* materials)
{
  ray->origin = vsub(isect->point, vscale(isect->normal, 1e-4f * fsign(vdot(ray->direction, isect->normal))));
  const OMaterial* material = &materials[isect->material_id];
  switch (material->type) {
  case 0: {
    const ovec3 albedo = v3(material->p[0], material->p[1], material->p[2]);
    ovec3 scatter_direction = vnormalize(vadd(isect->normal, random_in_unit_sphere(rng)));
    if (fabs((double)scatter_direction.x) < 1e-8 && fabs((double)scatter_direction.y) < 1e-8 &&
        fabs((double)scatter_direction.z) < 1e-8) {
      scatter_direction = isect->normal;
    }
    ray->direction = scatter_direction;
    *color = vmul(*color, albedo);
  } break;
  case 1: {
    const ovec3 albedo = v3(material->p[0], material->p[1], material->p[2]);
    const float fuzz = material->p[3];
    /* glm::reflect(I, N) = I - N * dot(N, I) * 2 */
    const ovec3 reflected =
        vsub(ray->direction, vscale(vscale(isect->normal, vdot(isect->normal, ray->direction)), 2.0f));
    const ovec3 scatter_direction = vadd(reflected, vscale(random_in_unit_sphere(rng), fuzz));
    ray->direction = scatter_direction;
    if (vdot(scatter_direction, isect->normal) > 0) *color = vmul(*color, albedo);
    else *color = v3(0.0f, 0.0f, 0.0f);
  } break;
  case 2: {
    const float ior = material->p[0];
    const float refraction_ratio = isect->side == 0 ? (1.0f / ior) : ior;
    const ovec3 unit_direction = vnormalize(ray->direction);
    const float cos_theta = fmin_sel(vdot(vneg(unit_direction), isect->normal), 1.0f);
    const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    const int cannot_refract = refraction_ratio * sin_theta > 1.0;
    ovec3 direction;
    if (cannot_refract || reflectance(cos_theta, refraction_ratio) > orc_rng_uniform(rng)) {
      direction = vsub(unit_direction,
                       vscale(vscale(isect->normal, vdot(isect->normal, unit_direction)), 2.0f));
    } else {
      /* glm::refract(I, N, eta) */
      const float eta = refraction_ratio;
      const float dot_value = vdot(isect->normal, unit_direction);
      const float k = 1.0f - eta * eta * (1.0f - dot_value * dot_value);
      if (k >= 0.0f)
        direction = vsub(vscale(unit_direction, eta), vscale(isect->normal, eta * dot_value + sqrtf(k)));
      else
        direction = v3(0, 0, 0);
    }
    ray->origin = isect->point;
    ray->t_min = 1e-5f;
    ray->direction = direction;
    ray->t_max = FLT_MAX;
  } break;
  default: break;
  }
}

/* path_tracer.cu:203-219 */
static inline float temporal_accumulate(uint64_t iteration, float old_value, float new_value)
{
  const float sample_count = (float)(iteration + 1);
  return iteration == 0 ? new_value : (old_value * (sample_count - 1) + new_value) / sample_count;
}
static void final_gather(uint64_t iteration, ovec3 new_color, ovec3 new_normal, float new_depth,
                         float* cur_color, float* cur_normal, float* cur_depth)
{
  cur_color[0] = temporal_accumulate(iteration, cur_color[0], new_color.x);
  cur_color[1] = temporal_accumulate(iteration, cur_color[1], new_color.y);
  cur_color[2] = temporal_accumulate(iteration, cur_color[2], new_color.z);
  cur_normal[0] = temporal_accumulate(iteration, cur_normal[0], new_normal.x);
  cur_normal[1] = temporal_accumulate(iteration, cur_normal[1], new_normal.y);
  cur_normal[2] = temporal_accumulate(iteration, cur_normal[2], new_normal.z);
  *cur_depth = temporal_accumulate(iteration, *cur_depth, new_depth);
}

/* ------------------------------------------------------------------------------------------------
 * parallel-for
 * ---------------------------------------------------------------------------------------------- */
int orc_hardware_threads(void)
{
  long n = sysconf(_SC_NPROCESSORS_ONLN);
  return n > 0 ? (int)n : 1;
}

typedef void (*range_fn)(void* ctx, uint32_t begin, uint32_t end, int tid);
typedef struct { range_fn fn; void* ctx; uint32_t begin, end; int tid; } PFJob;
static void* pf_tramp(void* p) { PFJob* j = (PFJob*)p; j->fn(j->ctx, j->begin, j->end, j->tid); return NULL; }

/* contiguous bands, one per thread */
static void parallel_for(uint32_t n, int nthreads, range_fn fn, void* ctx)
{
  if (nthreads <= 0) nthreads = orc_hardware_threads();
  if (nthreads > 256) nthreads = 256;
  if ((uint32_t)nthreads > n) nthreads = n ? (int)n : 1;
  if (nthreads == 1) { fn(ctx, 0, n, 0); return; }
  pthread_t th[256];
  PFJob jobs[256];
  for (int t = 0; t < nthreads; ++t) {
    jobs[t].fn = fn; jobs[t].ctx = ctx; jobs[t].tid = t;
    jobs[t].begin = (uint32_t)((uint64_t)n * (uint64_t)t / (uint64_t)nthreads);
    jobs[t].end = (uint32_t)((uint64_t)n * (uint64_t)(t + 1) / (uint64_t)nthreads);
    pthread_create(&th[t], NULL, pf_tramp, &jobs[t]);
  }
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
}

/* ------------------------------------------------------------------------------------------------
 * streaming mode  (path_tracer.hpp:24-55, ray_gen.cu:11-32, path_tracer.cu:271-330, 413-471)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  ORay* rays;
  int* pixel_indices;
  ovec3* color_buffer;
  ovec3* normal_buffer;
  float* depth_buffer;
  uint8_t* bounces_left_buffer;
} OPaths;

typedef struct {
  const OScene* scene;
  OGPUCamera camera;
  OPaths paths;
  OIntersection* intersections;
  uint64_t iteration;
  uint32_t bounce;
  uint32_t pix_begin;   /* first pixel of the band (0 for a full frame) */
  uint32_t slot_base;   /* added to the local slot index before RNG seeding (0 for a full frame) */
  /* interleaved row blocks (orc_render_streaming_interleaved); il_nranks == 0: a contiguous band */
  uint32_t il_rank, il_nranks, il_block_rows;
  float* fb_color; float* fb_normal; float* fb_depth;
} StreamCtx;

/* Which pixel of the frame the rank's slot `index` holds at ray generation, and back.  Contiguous band: pix_begin + index.
 * Interleaved: the frame is cut into blocks of il_block_rows rows, block gb belongs to rank gb % nranks; a rank's rows
 * are numbered in frame order (its local row lr lies in its (lr / block_rows)-th block). */
static uint32_t band_slot_to_pixel(const StreamCtx* c, uint32_t index)
{
  if (c->il_nranks == 0u) return c->pix_begin + index;
  const uint32_t W = c->camera.width;
  const uint32_t local_row = index / W, x = index % W;
  const uint32_t my_block = local_row / c->il_block_rows, row_in_block = local_row % c->il_block_rows;
  const uint32_t frame_block = my_block * c->il_nranks + c->il_rank;
  return (frame_block * c->il_block_rows + row_in_block) * W + x;
}
static uint32_t band_pixel_to_local(const StreamCtx* c, uint32_t pixel)
{
  if (c->il_nranks == 0u) return pixel - c->pix_begin;
  const uint32_t W = c->camera.width;
  const uint32_t y = pixel / W, x = pixel % W;
  const uint32_t frame_block = y / c->il_block_rows, row_in_block = y % c->il_block_rows;
  return ((frame_block / c->il_nranks) * c->il_block_rows + row_in_block) * W + x;
}

static void raygen_range(void* p, uint32_t begin, uint32_t end, int tid)
{
  (void)tid;
  StreamCtx* c = (StreamCtx*)p;
  const uint32_t W = c->camera.width;
  for (uint32_t index = begin; index < end; ++index) {
    const uint32_t pixel = band_slot_to_pixel(c, index);
    const uint32_t x = pixel % W, y = pixel / W;
    uint32_t rng = orc_rng_seed(orc_path_seed(pixel, c->iteration));
    const float fx = (float)x + orc_rng_uniform(&rng);
    const float fy = (float)y + orc_rng_uniform(&rng);
    ORay ray;
    orc_generate_ray(&c->camera, fx, fy, &ray);
    c->paths.color_buffer[index] = v3(1.0f, 1.0f, 1.0f);
    c->paths.depth_buffer[index] = 1e6f;
    c->paths.normal_buffer[index] = vneg(ray.direction);
    c->paths.bounces_left_buffer[index] = 50;
    c->paths.rays[index] = ray;
    c->paths.pixel_indices[index] = (int)pixel;
  }
}

static void intersection_range(void* p, uint32_t begin, uint32_t end, int tid)
{
  (void)tid;
  StreamCtx* c = (StreamCtx*)p;
  OStack st = {0, 0, 0};
  for (uint32_t index = begin; index < end; ++index) {
    const ORay ray = c->paths.rays[index];
    OIntersection isect;
    memset(&isect, 0, sizeof isect);
    if (ray_scene(ray, c->scene, &isect, &st)) {
      c->intersections[index] = isect;
    } else {
      c->intersections[index].t = -1.0f;
      c->paths.bounces_left_buffer[index] = 0;
    }
  }
  free(st.data);
}

static void material_range(void* p, uint32_t begin, uint32_t end, int tid)
{
  (void)tid;
  StreamCtx* c = (StreamCtx*)p;
  for (uint32_t index = begin; index < end; ++index) {
    uint32_t rng = orc_rng_seed(orc_path_seed(c->slot_base + index, c->iteration));
    orc_rng_discard(&rng, c->bounce);
    const OIntersection isect = c->intersections[index];
    if (isect.t < 0) {
      c->paths.color_buffer[index] = vmul(c->paths.color_buffer[index], get_background_color(&c->paths.rays[index]));
      continue;
    }
    if (c->bounce == 0) {
      c->paths.depth_buffer[index] = isect.t;
      c->paths.normal_buffer[index] = isect.normal;
    }
    evaluate_material(&c->paths.rays[index], &isect, &rng, &c->paths.color_buffer[index], c->scene->materials);
  }
}

static void gather_range(void* p, uint32_t begin, uint32_t end, int tid)
{
  (void)tid;
  StreamCtx* c = (StreamCtx*)p;
  for (uint32_t index = begin; index < end; ++index) {
    const int pixel_index = (int)band_pixel_to_local(c, (uint32_t)c->paths.pixel_indices[index]);
    final_gather(c->iteration, c->paths.color_buffer[index], c->paths.normal_buffer[index],
                 c->paths.depth_buffer[index], c->fb_color + 3 * (size_t)pixel_index,
                 c->fb_normal + 3 * (size_t)pixel_index, c->fb_depth + pixel_index);
  }
}

/* thrust::stable_partition over the 6-array zip, predicate bounces_left > 0
 * (path_tracer.cu:433-437, 454-457): live first, dead behind, both in original order. */
static uint32_t stable_partition_paths(OPaths* p, OPaths* tmp, uint32_t n)
{
  uint32_t live = 0;
  for (uint32_t i = 0; i < n; ++i) live += p->bounces_left_buffer[i] > 0;
  uint32_t a = 0, b = live;
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t d = p->bounces_left_buffer[i] > 0 ? a++ : b++;
    tmp->rays[d] = p->rays[i];
    tmp->pixel_indices[d] = p->pixel_indices[i];
    tmp->color_buffer[d] = p->color_buffer[i];
    tmp->normal_buffer[d] = p->normal_buffer[i];
    tmp->depth_buffer[d] = p->depth_buffer[i];
    tmp->bounces_left_buffer[d] = p->bounces_left_buffer[i];
  }
  memcpy(p->rays, tmp->rays, sizeof(ORay) * n);
  memcpy(p->pixel_indices, tmp->pixel_indices, sizeof(int) * n);
  memcpy(p->color_buffer, tmp->color_buffer, sizeof(ovec3) * n);
  memcpy(p->normal_buffer, tmp->normal_buffer, sizeof(ovec3) * n);
  memcpy(p->depth_buffer, tmp->depth_buffer, sizeof(float) * n);
  memcpy(p->bounces_left_buffer, tmp->bounces_left_buffer, n);
  return live;
}

static void paths_alloc(OPaths* p, size_t n)
{
  p->rays = (ORay*)malloc(sizeof(ORay) * n);
  p->pixel_indices = (int*)malloc(sizeof(int) * n);
  p->color_buffer = (ovec3*)malloc(sizeof(ovec3) * n);
  p->normal_buffer = (ovec3*)malloc(sizeof(ovec3) * n);
  p->depth_buffer = (float*)malloc(sizeof(float) * n);
  p->bounces_left_buffer = (uint8_t*)malloc(n);
}
static void paths_free(OPaths* p)
{
  free(p->rays); free(p->pixel_indices); free(p->color_buffer);
  free(p->normal_buffer); free(p->depth_buffer); free(p->bounces_left_buffer);
}

uint64_t orc_render_streaming(const OScene* scene, const OCamera* cam, uint32_t w, uint32_t h,
                              uint32_t iter_begin, uint32_t iter_count, uint32_t max_bounces,
                              float* fb_color, float* fb_normal, float* fb_depth,
                              uint32_t* live_counts, int nthreads)
{
  const uint32_t pixels_count = w * h;
  StreamCtx c;
  memset(&c, 0, sizeof c);
  c.scene = scene;
  orc_to_gpu_camera(cam, w, h, &c.camera);
  paths_alloc(&c.paths, pixels_count);
  OPaths tmp;
  paths_alloc(&tmp, pixels_count);
  c.intersections = (OIntersection*)malloc(sizeof(OIntersection) * pixels_count);
  c.fb_color = fb_color; c.fb_normal = fb_normal; c.fb_depth = fb_depth;
  uint64_t rays = 0;

  for (uint32_t it = 0; it < iter_count; ++it) {
    c.iteration = (uint64_t)iter_begin + it;
    parallel_for(pixels_count, nthreads, raygen_range, &c);
    uint32_t paths_count = pixels_count;
    if (live_counts) memset(live_counts + (size_t)it * max_bounces, 0, sizeof(uint32_t) * max_bounces);
    for (uint32_t i = 0; i < max_bounces && paths_count > 0; ++i) {
      if (live_counts) live_counts[(size_t)it * max_bounces + i] = paths_count;
      rays += paths_count;
      c.bounce = i;
      parallel_for(paths_count, nthreads, intersection_range, &c);
      parallel_for(paths_count, nthreads, material_range, &c);
      paths_count = stable_partition_paths(&c.paths, &tmp, paths_count);
    }
    parallel_for(pixels_count, nthreads, gather_range, &c);
  }
  free(c.intersections);
  paths_free(&tmp);
  paths_free(&c.paths);
  return rays;
}

uint64_t orc_render_streaming_band(const OScene* scene, const OCamera* cam, uint32_t w, uint32_t h, uint32_t row0,
                                   uint32_t row1, uint32_t iteration, uint32_t max_bounces, float* fb_color,
                                   float* fb_normal, float* fb_depth, uint32_t* live_counts,
                                   orc_exchange_fn exchange, void* user, int nthreads)
{
  const uint32_t pixels_count = (row1 - row0) * w;
  StreamCtx c;
  memset(&c, 0, sizeof c);
  c.scene = scene;
  orc_to_gpu_camera(cam, w, h, &c.camera);
  paths_alloc(&c.paths, pixels_count);
  OPaths tmp;
  paths_alloc(&tmp, pixels_count);
  c.intersections = (OIntersection*)malloc(sizeof(OIntersection) * pixels_count);
  c.fb_color = fb_color; c.fb_normal = fb_normal; c.fb_depth = fb_depth;
  c.iteration = iteration;
  c.pix_begin = row0 * w;
  uint64_t rays = 0;
  parallel_for(pixels_count, nthreads, raygen_range, &c);
  uint32_t paths_count = pixels_count;
  if (live_counts) memset(live_counts, 0, sizeof(uint32_t) * max_bounces);
  for (uint32_t i = 0; i < max_bounces; ++i) {
    /* every band takes part in every exchange, also with zero live paths */
    c.slot_base = exchange ? exchange(user, i, paths_count) : (i == 0 ? c.pix_begin : 0u);
    if (live_counts) live_counts[i] = paths_count;
    rays += paths_count;
    c.bounce = i;
    parallel_for(paths_count, nthreads, intersection_range, &c);
    parallel_for(paths_count, nthreads, material_range, &c);
    paths_count = stable_partition_paths(&c.paths, &tmp, paths_count);
  }
  parallel_for(pixels_count, nthreads, gather_range, &c);
  free(c.intersections);
  paths_free(&tmp);
  paths_free(&c.paths);
  return rays;
}

uint32_t orc_interleaved_rows(uint32_t h, uint32_t rank, uint32_t nranks, uint32_t block_rows)
{
  if (nranks == 0u || block_rows == 0u || rank >= nranks) return 0u;
  uint32_t rows = 0;
  for (uint32_t first = rank * block_rows; first < h; first += nranks * block_rows)
    rows += h - first < block_rows ? h - first : block_rows;
  return rows;
}

/* The multi-GPU split that is benchmarked (no reference equivalent): a rank renders the rows of every nranks-th block
 * of block_rows rows and numbers its paths locally; the material RNG, which the reference keys on the compacted slot
 * index (path_tracer.cu:297-301), is keyed on slot_offset + the rank's own compacted slot at EVERY bounce (bounce 0
 * included, where a full frame would use the pixel index); ray generation stays keyed on the frame's pixel index
 * (ray_gen.cu:17-22).  Everything else is orc_render_streaming on the rank's pixel set. */
uint64_t orc_render_streaming_interleaved(const OScene* scene, const OCamera* cam, uint32_t w, uint32_t h, uint32_t rank,
                                          uint32_t nranks, uint32_t block_rows, uint32_t slot_offset, uint32_t iter_begin,
                                          uint32_t iter_count, uint32_t max_bounces, float* fb_color, float* fb_normal,
                                          float* fb_depth, uint32_t* live_counts, int nthreads)
{
  const uint32_t pixels_count = orc_interleaved_rows(h, rank, nranks, block_rows) * w;
  if (pixels_count == 0u) return 0;
  StreamCtx c;
  memset(&c, 0, sizeof c);
  c.scene = scene;
  orc_to_gpu_camera(cam, w, h, &c.camera);
  paths_alloc(&c.paths, pixels_count);
  OPaths tmp;
  paths_alloc(&tmp, pixels_count);
  c.intersections = (OIntersection*)malloc(sizeof(OIntersection) * pixels_count);
  c.fb_color = fb_color; c.fb_normal = fb_normal; c.fb_depth = fb_depth;
  c.il_rank = rank; c.il_nranks = nranks; c.il_block_rows = block_rows;
  c.slot_base = slot_offset;
  uint64_t rays = 0;
  for (uint32_t it = 0; it < iter_count; ++it) {
    c.iteration = (uint64_t)iter_begin + it;
    parallel_for(pixels_count, nthreads, raygen_range, &c);
    uint32_t paths_count = pixels_count;
    if (live_counts) memset(live_counts + (size_t)it * max_bounces, 0, sizeof(uint32_t) * max_bounces);
    for (uint32_t i = 0; i < max_bounces && paths_count > 0; ++i) {
      if (live_counts) live_counts[(size_t)it * max_bounces + i] = paths_count;
      rays += paths_count;
      c.bounce = i;
      parallel_for(paths_count, nthreads, intersection_range, &c);
      parallel_for(paths_count, nthreads, material_range, &c);
      paths_count = stable_partition_paths(&c.paths, &tmp, paths_count);
    }
    parallel_for(pixels_count, nthreads, gather_range, &c);
  }
  free(c.intersections);
  paths_free(&tmp);
  paths_free(&c.paths);
  return rays;
}

void orc_intersect_rays(const OScene* scene, const ORay* rays, uint32_t n, OIntersection* recs, uint8_t* hit)
{
  OStack st = {0, 0, 0};
  for (uint32_t i = 0; i < n; ++i) {
    memset(&recs[i], 0, sizeof recs[i]);
    hit[i] = (uint8_t)ray_scene(rays[i], scene, &recs[i], &st);
    if (!hit[i]) recs[i].t = -1.0f;
  }
  free(st.data);
}

/* ------------------------------------------------------------------------------------------------
 * megakernel mode  (path_tracer.cu:227-269)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const OScene* scene;
  OGPUCamera camera;
  uint64_t iteration;
  uint32_t max_bounces;
  float* fb_color; float* fb_normal; float* fb_depth;
  uint64_t* rays_per_thread;
} MegaCtx;

static void mega_range(void* p, uint32_t begin, uint32_t end, int tid)
{
  MegaCtx* c = (MegaCtx*)p;
  OStack st = {0, 0, 0};
  uint64_t rays = 0;
  const uint32_t W = c->camera.width;
  for (uint32_t index = begin; index < end; ++index) {
    const uint32_t x = index % W, y = index / W;
    uint32_t rng = orc_rng_seed(orc_path_seed(index, c->iteration));
    const float fx = (float)x + orc_rng_uniform(&rng);
    const float fy = (float)y + orc_rng_uniform(&rng);
    ORay ray;
    orc_generate_ray(&c->camera, fx, fy, &ray);
    ovec3 color = v3(1.0f, 1.0f, 1.0f);
    ovec3 normal = vneg(ray.direction);
    float depth = 1e6f;
    for (uint32_t i = 0; i < c->max_bounces; ++i) {
      OIntersection isect;
      memset(&isect, 0, sizeof isect);
      ++rays;
      const int hit = ray_scene(ray, c->scene, &isect, &st);
      if (!hit) { color = vmul(color, get_background_color(&ray)); break; }
      if (i == 0) { normal = isect.normal; depth = isect.t; }
      evaluate_material(&ray, &isect, &rng, &color, c->scene->materials);
    }
    final_gather(c->iteration, color, normal, depth, c->fb_color + 3 * (size_t)index,
                 c->fb_normal + 3 * (size_t)index, c->fb_depth + index);
  }
  c->rays_per_thread[tid] += rays;
  free(st.data);
}

uint64_t orc_render_megakernel(const OScene* scene, const OCamera* cam, uint32_t w, uint32_t h,
                               uint32_t iter_begin, uint32_t iter_count, uint32_t max_bounces,
                               float* fb_color, float* fb_normal, float* fb_depth, int nthreads)
{
  MegaCtx c;
  uint64_t per_thread[256];
  memset(per_thread, 0, sizeof per_thread);
  c.scene = scene;
  orc_to_gpu_camera(cam, w, h, &c.camera);
  c.max_bounces = max_bounces;
  c.fb_color = fb_color; c.fb_normal = fb_normal; c.fb_depth = fb_depth;
  c.rays_per_thread = per_thread;
  for (uint32_t it = 0; it < iter_count; ++it) {
    c.iteration = (uint64_t)iter_begin + it;
    parallel_for(w * h, nthreads, mega_range, &c);
  }
  uint64_t rays = 0;
  for (int i = 0; i < 256; ++i) rays += per_thread[i];
  return rays;
}

/* ------------------------------------------------------------------------------------------------
 * Edge-avoiding a-trous denoiser  (denoising/edge_avoiding_a_trous_denoiser.cu:24-115)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  OGPUCamera camera;
  const float* color; const float* normal; const float* depth;
  float* out;
  int step_width;
  float c_phi, n_phi, p_phi;
  const uint8_t* touched_in; uint8_t* touched_out;
} DenoiseCtx;

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void denoise_range(void* p, uint32_t begin, uint32_t end, int tid)
{
  (void)tid;
  DenoiseCtx* c = (DenoiseCtx*)p;
  static const float kernel[] = {3.f / 8.f, 1.f / 4.f, 1.f / 16.f};
  const uint32_t W = c->camera.width, H = c->camera.height;
  const uint32_t P = W * H;
  for (uint32_t index = begin; index < end; ++index) {
    const uint32_t x = index % W, y = index / W;
    const ovec3 cval = vload(c->color + 3 * (size_t)index);
    const ovec3 nval = vload(c->normal + 3 * (size_t)index);
    ORay ray;
    orc_generate_ray(&c->camera, (float)x + 0.5f, (float)y + 0.5f, &ray);
    const ovec3 pval = ray_at(&ray, c->depth[index]);
    ovec3 sum = v3(0, 0, 0);
    float cum_w = 0.0f;
    uint8_t touched = c->touched_in ? c->touched_in[index] : 0;
    for (int dy = -2; dy <= 2; ++dy) {
      for (int dx = -2; dx <= 2; ++dx) {
        /* inclusive clamp to [0,W] x [0,H] -- the reference's off-by-one (cu:39-42) */
        const int u = clampi((int)x + dx * c->step_width, 0, (int)W);
        const int v = clampi((int)y + dy * c->step_width, 0, (int)H);
        uint32_t temp_index = (uint32_t)u + (uint32_t)v * W;
        if (temp_index >= P) { temp_index = P - 1; touched = 1; } /* reference: out-of-bounds read */
        if (c->touched_in && c->touched_in[temp_index]) touched = 1;

        const ovec3 ctemp = vload(c->color + 3 * (size_t)temp_index);
        ovec3 t = vsub(cval, ctemp);
        float dist2 = vdot(t, t);
        const float c_w = fmin_sel(expf(-dist2 / c->c_phi), 1.0f);

        const ovec3 ntemp = vload(c->normal + 3 * (size_t)temp_index);
        t = vsub(nval, ntemp);
        dist2 = fmax_sel(vdot(t, t) / (float)(c->step_width * c->step_width), 0.0f);
        const float n_w = fmin_sel(expf(-dist2 / c->n_phi), 1.0f);

        ORay temp_ray;
        orc_generate_ray(&c->camera, (float)u + 0.5f, (float)v + 0.5f, &temp_ray);
        const ovec3 ptmp = ray_at(&temp_ray, c->depth[temp_index]);
        t = vsub(pval, ptmp);
        dist2 = vdot(t, t);
        const float p_w = fmin_sel(expf(-dist2 / c->p_phi), 1.0f);

        const float weight = c_w * n_w * p_w;
        const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
        const int kernel_index = adx < ady ? adx : ady;
        sum = vadd(sum, vscale(vscale(ctemp, weight), kernel[kernel_index]));
        cum_w += weight * kernel[kernel_index];
      }
    }
    const ovec3 o = vdivs(sum, cum_w);
    c->out[3 * (size_t)index] = o.x;
    c->out[3 * (size_t)index + 1] = o.y;
    c->out[3 * (size_t)index + 2] = o.z;
    if (c->touched_out) c->touched_out[index] = touched;
  }
}

int orc_denoise(const OCamera* cam, uint32_t w, uint32_t h, const float* color, const float* normal,
                const float* depth, float* buf_a, float* buf_b, int filter_size, float c_phi,
                float n_phi, float p_phi, uint8_t* touched_oob, int nthreads)
{
  DenoiseCtx c;
  memset(&c, 0, sizeof c);
  orc_to_gpu_camera(cam, w, h, &c.camera);
  c.normal = normal; c.depth = depth;
  c.c_phi = c_phi; c.n_phi = n_phi; c.p_phi = p_phi;
  const float* color_buffer = color;
  float* back_buffer = buf_a;
  float* front_buffer = buf_b;
  uint8_t* t_a = NULL; uint8_t* t_b = NULL;
  const size_t P = (size_t)w * h;
  if (touched_oob) { t_a = (uint8_t*)calloc(P, 1); t_b = (uint8_t*)calloc(P, 1); }
  const uint8_t* t_in = NULL;
  uint8_t* t_out = t_a;
  int ran = 0;
  for (int step_width = 1; step_width <= filter_size; step_width *= 2) {
    c.color = color_buffer;
    c.out = back_buffer;
    c.step_width = step_width;
    c.touched_in = t_in; c.touched_out = touched_oob ? t_out : NULL;
    parallel_for(w * h, nthreads, denoise_range, &c);
    /* std::tie(color, back, front) = (back, front, back)  (cu:105-107) */
    const float* new_color = back_buffer;
    float* new_back = front_buffer;
    float* new_front = back_buffer;
    color_buffer = new_color; back_buffer = new_back; front_buffer = new_front;
    if (touched_oob) { t_in = t_out; t_out = (t_out == t_a) ? t_b : t_a; }
    ran = 1;
  }
  if (touched_oob) {
    if (t_in) memcpy(touched_oob, t_in, P); else memset(touched_oob, 0, P);
    free(t_a); free(t_b);
  }
  if (!ran) return -1; /* the reference returns front_buffer (buf_b) unwritten */
  return front_buffer == buf_a ? 0 : 1;
}

/* ------------------------------------------------------------------------------------------------
 * preview tonemap  (path_tracer.cu:221-225, 334-385)
 * ---------------------------------------------------------------------------------------------- */
static inline uint8_t color_float_to_255(float v)
{
  const float cl = fmin_sel(fmax_sel(v, 0.f), 1.f); /* glm::clamp = min(max(x, lo), hi) */
  return (uint8_t)(cl * 255.99f);
}
void orc_preview(const float* buffer, uint32_t w, uint32_t h, int mode, uint8_t* rgba)
{
  const size_t P = (size_t)w * h;
  for (size_t i = 0; i < P; ++i) {
    ovec3 color;
    uint8_t alpha = 255;
    if (mode == 2) {
      const float d = 1 / buffer[i];
      color = v3(d, d, d);
      alpha = 1;
    } else {
      color = vload(buffer + 3 * i);
      if (mode == 1) color = vadd(vscale(color, 0.5f), v3(0.5f, 0.5f, 0.5f));
    }
    const float g = 1.f / 2.2f;
    color = v3(powf(color.x, g), powf(color.y, g), powf(color.z, g));
    rgba[4 * i] = color_float_to_255(color.x);
    rgba[4 * i + 1] = color_float_to_255(color.y);
    rgba[4 * i + 2] = color_float_to_255(color.z);
    rgba[4 * i + 3] = alpha;
  }
}
