// pt_kernels.hip -- gfx950 kernels of the render core (wave64, LDS traversal stacks, ballot compaction).
//
// Arithmetic contract: every float operation on the path (ray generation, intersection, shading) is a
// single IEEE binary32 operation in a fixed order (built with -ffp-contract=off and correctly rounded
// divide/sqrt), because the reference's streaming mode seeds its RNG from the COMPACTED slot index
// (path_tracer.cu:297-301): one hit/miss decision that differs moves every later path to another slot.
// The parity tests therefore compare with the CPU oracle bit for bit.
//
// Frame pipeline (streaming mode, PathTracer::path_trace path_tracer.cu:413-471):
//   raygen                                 generate_rays / raygen_kernel   (ray_gen.cu:11-32)
//   per bounce b:
//     trace   closest hit per live path    intersection_kernel             (path_tracer.cu:271-290)
//             + per-wavefront live count (ballot/popcount)
//     scan    exclusive scan of the per-wavefront counts -> compaction offsets, live[b+1]
//     shade   material + sky + G-buffer    material_kernel                 (path_tracer.cu:292-315)
//             fused with the stable compaction scatter   thrust::stable_partition (path_tracer.cu:454-457)
//             and with the final gather of paths that end here  final_gathering_kernel (path_tracer.cu:317-330)
// A path that ends (miss, or the bounce cap) is accumulated into the framebuffer at once; only live
// paths are kept, in their original order, so slot indices equal the reference's.

#include "pt_device.hpp"
#include "pt_rng.hpp"
#include "pt_beam_rules.hpp"
#include "pt_feed_rules.hpp"
static_assert(pt::beam_rules::kLeaf == pt::kLeafBit, "pt_beam_rules.hpp restates the leaf bit of the four-wide node (pt_device.hpp)");
static_assert((uint32_t)pt::beam_rules::kEntries == pt::kBeamEntries, "pt_beam_rules.hpp restates the entries per tile (pt_device.hpp)");
static_assert(pt::feed_rules::kBatch == (uint32_t)pt::kWave, "a feed batch is one wavefront's worth of rays");

#include <float.h>

namespace pt {

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t rank_below(uint64_t mask)
{
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }
// Path state, hit records and framebuffers stream through the chip once per kernel (hundreds of MB per frame), next to
// a BVH working set (75 MB of nodes and triangles at 1M triangles) that every ray re-reads and that should own the
// 4 MB L2 of its XCD: streaming accesses are marked non-temporal.
#ifndef PT_NT
#define PT_NT 1
#endif
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldnt(const float4* p)
{
#if PT_NT
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
__device__ __forceinline__ void stnt(float4* p, const float4 v)
{
#if PT_NT
  v4f w;
  w.x = v.x;
  w.y = v.y;
  w.z = v.z;
  w.w = v.w;
  __builtin_nontemporal_store(w, reinterpret_cast<v4f*>(p));
#else
  *p = v;
#endif
}
__device__ __forceinline__ f3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }

struct Ray {
  f3 o;
  float tmin;
  f3 d;
  float tmax;
};
__device__ __forceinline__ f3 ray_at(const Ray& r, float t) { return r.o + r.d * t; }

// per-lane test counters of the instrumented kernel variant
struct Tally {
  uint32_t boxes = 0u, tris = 0u, nodes = 0u;  // ray/box tests, ray/triangle tests, node records fetched
};

struct Hit {
  float t;
  f3 p;
  f3 n;
  uint32_t mat;
  uint32_t side;  // 0 front, 1 back
};

// wavefront totals of the instrumented kernel variants -> the counter block (one atomic each)
__device__ __forceinline__ void flush_tally(const Tally& tally, DeviceCounters* counters, int bounce, bool per_ray_max)
{
  uint32_t b = tally.boxes, t = tally.tris, nd = tally.nodes, mx = tally.boxes;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    b += __shfl_down(b, off, 64);
    t += __shfl_down(t, off, 64);
    nd += __shfl_down(nd, off, 64);
    mx = max(mx, (uint32_t)__shfl_down(mx, off, 64));
  }
  if ((threadIdx.x & 63u) == 0u) {
    atomicAdd(&counters->box_tests[bounce], (unsigned long long)b);
    atomicAdd(&counters->tri_tests[bounce], (unsigned long long)t);
    atomicAdd(&counters->node_visits[bounce], (unsigned long long)nd);
    if (per_ray_max) atomicMax(&counters->max_box_tests[bounce], mx);  // one lane = one ray in those kernels
  }
}

// ------------------------------------------------------------------------------------------------
// intersection tests (reference intersections.cuh)
// ------------------------------------------------------------------------------------------------
// ray_aabb_intersection_test, intersections.cuh:87-103: no t-range, boxes behind the origin pass
__device__ __forceinline__ bool ray_aabb(const f3 o, const f3 d, const f3 bmin, const f3 bmax)
{
  if (bmin.x > bmax.x || bmin.y > bmax.y || bmin.z > bmax.z) return false;
  const f3 t0 = (bmin - o) / d;
  const f3 t1 = (bmax - o) / d;
  const f3 rmin = min3(t0, t1);
  const f3 rmax = max3(t0, t1);
  const float minmax = sel_min(sel_min(rmax.x, rmax.y), rmax.z);
  const float maxmin = sel_max(sel_max(rmin.x, rmin.y), rmin.z);
  return minmax >= maxmin;
}

// ray_sphere_intersection_test, intersections.cuh:7-41
// (a = dot(d, d) comes from the caller: sphere_segment can test many spheres with the same transformed direction)
__device__ __forceinline__ bool ray_sphere_a(const Ray& ray, const float a, const f3 center, const float radius, Hit& rec);
__device__ __forceinline__ bool ray_sphere(const Ray& ray, const f3 center, const float radius, Hit& rec)
{
  return ray_sphere_a(ray, dot(ray.d, ray.d), center, radius, rec);
}
__device__ __forceinline__ bool ray_sphere_a(const Ray& ray, const float a, const f3 center, const float radius, Hit& rec)
{
  const f3 oc = ray.o - center;
  const float b = 2.0f * dot(ray.d, oc);
  const float c = dot(oc, oc) - radius * radius;
  const float disc = b * b - 4.0f * a * c;
  if (disc < 0.0f) return false;
  const float sq = ieee_sqrt(disc);
  const float t1 = (-b - sq) / (2.0f * a);
  const float t2 = (-b + sq) / (2.0f * a);
  float t;
  if (t1 >= ray.tmin && t1 <= ray.tmax) t = t1;
  else if (t2 >= ray.tmin && t2 <= ray.tmax) t = t2;
  else return false;
  rec.t = t;
  rec.p = ray_at(ray, t);
  const f3 outward = (rec.p - center) / radius;
  rec.side = dot(ray.d, outward) < 0.0f ? 0u : 1u;
  rec.n = rec.side == 0u ? outward : -outward;
  return true;
}

// ray_triangle_intersection_test, intersections.cuh:49-85 (t == t_max accepted)
__device__ __forceinline__ bool ray_triangle(const Ray& ray, const f3 p0, const f3 p1, const f3 p2, Hit& rec)
{
  const float EPS = 0.0000001f;
  const f3 e1 = p1 - p0;
  const f3 e2 = p2 - p0;
  const f3 h = cross(ray.d, e2);
  const float a = dot(e1, h);
  if (a > -EPS && a < EPS) return false;
  const float f = 1.0f / a;
  const f3 s = ray.o - p0;
  const float u = f * dot(s, h);
  if (u < 0.0f || u > 1.0f) return false;
  const f3 q = cross(s, e1);
  const float v = f * dot(ray.d, q);
  if (v < 0.0f || u + v > 1.0f) return false;
  const float t = f * dot(e2, q);
  if (t < ray.tmin || t > ray.tmax) return false;
  rec.t = t;
  rec.p = ray_at(ray, t);
  const f3 outward = normalize(cross(e1, e2));
  rec.side = dot(ray.d, outward) < 0.0f ? 0u : 1u;
  rec.n = rec.side == 0u ? outward : -outward;
  return true;
}

// inverse_transform_ray, transform.hpp:51-58: direction re-normalised, t range copied unscaled
__device__ __forceinline__ void inverse_transform_ray(const m4& inv_m, const Ray& ray, f3& o, f3& d)
{
  o = xform_point(inv_m, ray.o);
  d = normalize(xform_vector(inv_m, ray.d));
}

// ray_mesh_intersection_test, path_tracer.cu:36-76.  Depth-first, left child first, every inner box
// the line crosses is entered (the reference has no t culling).  The stack lives in LDS, laid out
// [depth][lane] so that a push or pop of the whole wavefront touches 64 consecutive banks.
template <bool kCount>
__device__ __forceinline__ bool ray_mesh(Ray ray, const DMeshView& mv, const DObject* obj, Hit& rec, uint32_t* stack,
                                         uint32_t& flags, Tally& tally, const int stack_cap = kStackDepth)
{
  bool hit = false;
  f3 oo, od;
  inverse_transform_ray(obj->inv_m, ray, oo, od);
  if (mv.bvh_node_count == 0u) return false;

  int sp = 0;
  stack[0] = 0u;
  sp = 1;
  while (sp > 0) {
    --sp;
    const uint32_t node = stack[sp * kWave];
    const float4 n0 = mv.bvh[2u * node];
    const float4 n1 = mv.bvh[2u * node + 1u];
    const uint32_t first = __float_as_uint(n0.w);
    const uint32_t count = __float_as_uint(n1.w);
    if (count != 0u) {
      const uint32_t i0 = mv.indices[first], i1 = mv.indices[first + 1u], i2 = mv.indices[first + 2u];
      const f3 p0 = xform_point(obj->m, ld3(mv.positions + 3u * (size_t)i0));
      const f3 p1 = xform_point(obj->m, ld3(mv.positions + 3u * (size_t)i1));
      const f3 p2 = xform_point(obj->m, ld3(mv.positions + 3u * (size_t)i2));
      if (kCount) ++tally.tris;
      if (ray_triangle(ray, p0, p1, p2, rec)) {
        hit = true;
        ray.tmax = rec.t;
      }
    } else if ((kCount ? (void)(++tally.boxes, ++tally.nodes) : (void)0), ray_aabb(oo, od, xyz(n0), xyz(n1))) {
      if (sp + 2 > stack_cap) {
        flags |= kFlagStackOverflow;
      } else {
        stack[sp * kWave] = first + 1u;
        stack[(sp + 1) * kWave] = first;
        sp += 2;
      }
    }
  }
  return hit;
}

// ray_scene_intersection_test + ray_object_intersection_test, path_tracer.cu:78-128
template <bool kCount>
__device__ __forceinline__ bool ray_scene(Ray ray, const DScene& sc, Hit& rec, uint32_t* stack, uint32_t& flags,
                                          Tally& tally)
{
  bool hit = false;
  for (uint32_t i = 0; i < sc.object_count; ++i) {
    const DObject* obj = sc.objects + i;
    if (!ray_aabb(ray.o, ray.d, ld3(obj->bmin), ld3(obj->bmax))) continue;
    bool h = false;
    if (obj->type == 0u) {
      Ray tr;
      inverse_transform_ray(obj->inv_m, ray, tr.o, tr.d);
      tr.tmin = ray.tmin;
      tr.tmax = ray.tmax;
      const float4 sp = sc.spheres[obj->index];
      h = ray_sphere(tr, xyz(sp), sp.w, rec);
      if (h) {
        rec.p = xform_point(obj->m, rec.p);
        rec.t = length(rec.p - ray.o);
        rec.n = xform_normal(obj->inv_m, rec.n);
      }
    } else {
      h = ray_mesh<kCount>(ray, sc.mesh_views[sc.object_mesh[i]], obj, rec, stack, flags, tally);
    }
    if (h) {
      hit = true;
      rec.mat = sc.object_material[i];
      ray.tmax = rec.t;
    }
  }
  return hit;
}

// ------------------------------------------------------------------------------------------------
// fast closest hit: same result as ray_mesh / ray_scene above, different schedule
// ------------------------------------------------------------------------------------------------
// What the reference computes for one mesh object (path_tracer.cu:36-76) is, independent of visiting order:
// among the triangles whose inner ancestors ALL pass ray_aabb (exact divisions, object-space ray), the one
// with the smallest accepted t; equal t -> the one the depth-first, left-first order reaches last.
// This traversal keeps exactly that result and changes only the schedule:
//   * box decisions are first evaluated with one reciprocal per axis; when the two slab extremes are closer
//     than the rounding error of that shortcut (or anything is non-finite) the exact, division-based test
//     of the reference decides, so every inner-box decision equals the reference's;
//   * triangles are stored in depth-first leaf order, so "reached last" = larger index (tie rule);
//   * subtrees are skipped only when their box lies, with a safety margin, beyond the closest hit found
//     so far or behind the ray origin (such triangles cannot be accepted), and leaves whose own box is
//     missed with a margin are skipped (the reference tests no leaf boxes; those tests would fail);
//   * children are visited nearest first; one LDS stack slot per level.
constexpr int kWideStack = kStackDepth;  // one entry per level of the two-wide tree (ptc_upload_scene checks the depth)

__device__ __forceinline__ bool finite_f(float x) { return fabsf(x) < __builtin_inff(); }

__device__ __forceinline__ void slab_fast(const f3 bmin, const f3 bmax, const f3 oo, const f3 inv, float& t_near,
                                          float& t_far)
{
  const float ax = (bmin.x - oo.x) * inv.x, bx = (bmax.x - oo.x) * inv.x;
  const float ay = (bmin.y - oo.y) * inv.y, by = (bmax.y - oo.y) * inv.y;
  const float az = (bmin.z - oo.z) * inv.z, bz = (bmax.z - oo.z) * inv.z;
  t_near = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
  t_far = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
}

// Slab interval of a box FOR CULLING ONLY (never for a decision the reference takes), valid also when a direction
// component is zero or so small that its reciprocal is not finite.  On such an axis the ray keeps its coordinate:
// the interval is everything when the origin lies within the box's extent (boundary included, plus a margin far
// beyond rounding: a triangle test can accept a ray that runs along the box face) and nothing otherwise.  The
// products of slab_fast are 0 * inf = NaN exactly in the case that matters -- origin on a box plane -- and fminf /
// fmaxf then pick the other operand, which moved such a box to t = +inf: rays through mesh vertices, along
// triangle edges or down a box face lost their hit (found by the adversarial rays of
// tests/test_gpu_schedules.py::test_benchmark_size_rays_against_oracle).
__device__ __forceinline__ void slab_cull(const f3 bmin, const f3 bmax, const f3 oo, const f3 inv, float& t_near, float& t_far)
{
  const float kInf = __builtin_inff();
  auto axis = [&](float lo_b, float hi_b, float o, float r, float& lo, float& hi) {
    if (finite_f(r)) {
      const float x = (lo_b - o) * r, y = (hi_b - o) * r;
      lo = fminf(x, y);
      hi = fmaxf(x, y);
    } else {
      const float m = 1e-4f * (fabsf(lo_b) + fabsf(hi_b) + fabsf(o)) + 1e-30f;
      const bool inside = o >= lo_b - m && o <= hi_b + m;
      lo = inside ? -kInf : kInf;
      hi = inside ? kInf : -kInf;
    }
  };
  float lx, hx, ly, hy, lz, hz;
  axis(bmin.x, bmax.x, oo.x, inv.x, lx, hx);
  axis(bmin.y, bmax.y, oo.y, inv.y, ly, hy);
  axis(bmin.z, bmax.z, oo.z, inv.z, lz, hz);
  t_near = fmaxf(fmaxf(lx, ly), lz);
  t_far = fminf(fminf(hx, hy), hz);
}

// the reference's test (ray_aabb) that also hands back its two extremes
__device__ __forceinline__ bool slab_exact(const f3 bmin, const f3 bmax, const f3 o, const f3 d, float& t_near,
                                           float& t_far)
{
  const f3 t0 = (bmin - o) / d;
  const f3 t1 = (bmax - o) / d;
  const f3 rmin = min3(t0, t1);
  const f3 rmax = max3(t0, t1);
  t_far = sel_min(sel_min(rmax.x, rmax.y), rmax.z);
  t_near = sel_max(sel_max(rmin.x, rmin.y), rmin.z);
  return t_far >= t_near;
}

// Decision of an INNER node's box, equal to the reference's.  The shortcut's extremes differ from the exact
// ones by at most ~3 ulp each (one rounding of 1/d, one of the product, against one of the quotient).
__device__ __forceinline__ bool box_pass_inner(const f3 bmin, const f3 bmax, const f3 oo, const f3 od, const f3 inv,
                                               const bool exact_only, float& t_near, float& t_far)
{
  if (!exact_only) {
    slab_fast(bmin, bmax, oo, inv, t_near, t_far);
    const float gap = t_far - t_near;
    const float tol = 4e-7f * (fabsf(t_far) + fabsf(t_near)) + 1e-30f;
    if (gap > tol) return true;
    if (gap < -tol) return false;
  }
  return slab_exact(bmin, bmax, oo, od, t_near, t_far);
}

// May this child be skipped although its box test passed?  limit = |M^-1 d| * (closest t so far): the same
// distance measured along the object-space ray.  NaN compares false -> never skipped.
__device__ __forceinline__ bool box_culled(const float t_near, const float t_far, const float limit)
{
  return (t_near > limit * 1.001f + 1e-3f * (t_far - t_near)) || (t_far < -1e-3f * fabsf(t_near));
}

template <bool kCount>
__device__ __forceinline__ void mesh_closest_wide(const Ray& ray, const DScene& sc, const DMeshView& mv, const DObject* obj,
                                                  const uint32_t tri_base, float& best_t, int& best_k, uint32_t* stack,
                                                  uint32_t& flags, Tally& tally)
{
  if (mv.bvh_node_count == 0u) return;
  // inverse_transform_ray (transform.hpp:51-58); scale = length before the re-normalisation
  const f3 v = xform_vector(obj->inv_m, ray.d);
  const float scale = ieee_sqrt(dot(v, v));
  const f3 od = v * (1.0f / scale);
  const f3 oo = xform_point(obj->inv_m, ray.o);
  const f3 inv = mk3(1.0f / od.x, 1.0f / od.y, 1.0f / od.z);
  const bool exact_only = !(finite_f(inv.x) && finite_f(inv.y) && finite_f(inv.z));
  float limit = scale * best_t;

  uint32_t cur = mv.root_ref;
  if (!(cur & kLeafBit)) {
    float tn, tf;
    if (kCount) ++tally.boxes;
    if (!box_pass_inner(ld3(mv.root_min), ld3(mv.root_max), oo, od, inv, exact_only, tn, tf)) return;
    if (exact_only) slab_cull(ld3(mv.root_min), ld3(mv.root_max), oo, inv, tn, tf);
    if (box_culled(tn, tf, limit)) return;
  }
  const float4* tris = sc.tris + kTriVec4 * (size_t)tri_base;
  int sp = 0;
  for (;;) {
    if (cur & kLeafBit) {
      // ray_triangle_intersection_test (intersections.cuh:49-85) on the precomputed world-space edges
      const uint32_t k = cur & ~kLeafBit;
      const float4 ta = tris[kTriVec4 * k], tb = tris[kTriVec4 * k + 1u], tc = tris[kTriVec4 * k + 2u];
      if (kCount) ++tally.tris;
      const f3 p0 = mk3(ta.x, ta.y, ta.z), e1 = mk3(ta.w, tb.x, tb.y), e2 = mk3(tb.z, tb.w, tc.x);
      const f3 h = cross(ray.d, e2);
      const float a = dot(e1, h);
      if (!(a > -0.0000001f && a < 0.0000001f)) {
        const float f = 1.0f / a;
        const f3 sv = ray.o - p0;
        const float u = f * dot(sv, h);
        if (!(u < 0.0f || u > 1.0f)) {
          const f3 q = cross(sv, e1);
          const float w = f * dot(ray.d, q);
          if (!(w < 0.0f || u + w > 1.0f)) {
            const float t = f * dot(e2, q);
            if (!(t < ray.tmin) && (t < best_t || (t == best_t && (int)k > best_k))) {
              best_t = t;
              best_k = (int)k;
              limit = scale * t;
            }
          }
        }
      }
      if (sp == 0) break;
      --sp;
      cur = stack[sp * kWave];
      continue;
    }
    const float4 w0 = mv.wide[4u * (size_t)cur], w1 = mv.wide[4u * (size_t)cur + 1u];
    const float4 w2 = mv.wide[4u * (size_t)cur + 2u], w3 = mv.wide[4u * (size_t)cur + 3u];
    const uint32_t lref = __float_as_uint(w3.x), rref = __float_as_uint(w3.y);
    const f3 lmin = mk3(w0.x, w0.y, w0.z), lmax = mk3(w0.w, w1.x, w1.y);
    const f3 rmin = mk3(w1.z, w1.w, w2.x), rmax = mk3(w2.y, w2.z, w2.w);
    if (kCount) { tally.boxes += 2u; ++tally.nodes; }
    // Both children through the same code.  An inner child's decision must equal the reference's: the
    // shortcut decides unless the slab extremes are closer than its rounding error (then: exact test).
    // A leaf child's box is not part of the reference's decision: its triangle is skipped only when the ray misses
    // the box by a margin far beyond rounding IN SPACE -- the box is grown by 1e-5 of its coordinates (~100 ulp)
    // before the test.  (A margin relative to t does not do: 1/d magnifies one ulp of distance from the box to
    // any t when a direction component is tiny, while the triangle test's own tolerance is relative to the
    // coordinates; found with direction components of 1e-30.)
    const bool l_leaf = (lref & kLeafBit) != 0u, r_leaf = (rref & kLeafBit) != 0u;
    auto grow = [&](const f3 lo, const f3 hi, bool leaf) -> f3 {
      const float k = leaf ? 1e-5f : 0.0f;
      return mk3(k * (fabsf(lo.x) + fabsf(hi.x) + fabsf(oo.x)) + (leaf ? 1e-30f : 0.0f),
                 k * (fabsf(lo.y) + fabsf(hi.y) + fabsf(oo.y)) + (leaf ? 1e-30f : 0.0f),
                 k * (fabsf(lo.z) + fabsf(hi.z) + fabsf(oo.z)) + (leaf ? 1e-30f : 0.0f));
    };
    const f3 lm = grow(lmin, lmax, l_leaf), rm = grow(rmin, rmax, r_leaf);
    const f3 lmin_c = lmin - lm, lmax_c = lmax + lm, rmin_c = rmin - rm, rmax_c = rmax + rm;  // inner: unchanged
    float ln, lf, rn, rf;
    slab_fast(lmin_c, lmax_c, oo, inv, ln, lf);
    slab_fast(rmin_c, rmax_c, oo, inv, rn, rf);
    const float lgap = lf - ln, rgap = rf - rn;
    const float ltol = 4e-7f * (fabsf(lf) + fabsf(ln)) + 1e-30f;
    const float rtol = 4e-7f * (fabsf(rf) + fabsf(rn)) + 1e-30f;
    bool go_l = l_leaf ? !(lgap < -ltol) : (lgap > ltol);
    bool go_r = r_leaf ? !(rgap < -rtol) : (rgap > rtol);
    const bool l_unsure = !l_leaf && (exact_only || !(lgap > ltol || lgap < -ltol));
    const bool r_unsure = !r_leaf && (exact_only || !(rgap > rtol || rgap < -rtol));
    if (__builtin_expect(l_unsure || r_unsure || exact_only, 0)) {
      if (l_unsure) go_l = slab_exact(lmin, lmax, oo, od, ln, lf);
      if (r_unsure) go_r = slab_exact(rmin, rmax, oo, od, rn, rf);
      if (exact_only) {
        // a direction component without a finite reciprocal: decisions as above (inner: the reference's own test;
        // leaf: visit), culling bounds from the form that is safe for such rays
        go_l = go_l || l_leaf;
        go_r = go_r || r_leaf;
        slab_cull(lmin_c, lmax_c, oo, inv, ln, lf);
        slab_cull(rmin_c, rmax_c, oo, inv, rn, rf);
      }
    }
    go_l = go_l && !box_culled(ln, lf, limit);
    go_r = go_r && !box_culled(rn, rf, limit);
    if (go_l && go_r) {
      const bool left_first = !(rn < ln);
      const uint32_t first = left_first ? lref : rref, second = left_first ? rref : lref;
      if (sp >= kWideStack) {
        flags |= kFlagStackOverflow;
      } else {
        stack[sp * kWave] = second;
        ++sp;
      }
      cur = first;
    } else if (go_l || go_r) {
      cur = go_l ? lref : rref;
    } else {
      if (sp == 0) break;
      --sp;
      cur = stack[sp * kWave];
    }
  }
}

// ray_scene_intersection_test with the fast mesh traversal.  Spheres and the per-object world-box test
// are the reference's code; a mesh winner's normal / side / point are filled in once at the end.
template <bool kCount>
__device__ __forceinline__ bool ray_scene_wide(Ray ray, const DScene& sc, Hit& rec, uint32_t* stack, uint32_t& flags,
                                               Tally& tally)
{
  bool hit = false;
  int win_tri = -1;
  for (uint32_t i = 0; i < sc.object_count; ++i) {
    const DObject* obj = sc.objects + i;
    if (!ray_aabb(ray.o, ray.d, ld3(obj->bmin), ld3(obj->bmax))) continue;
    if (obj->type == 0u) {
      Ray tr;
      inverse_transform_ray(obj->inv_m, ray, tr.o, tr.d);
      tr.tmin = ray.tmin;
      tr.tmax = ray.tmax;
      const float4 sp = sc.spheres[obj->index];
      if (ray_sphere(tr, xyz(sp), sp.w, rec)) {
        rec.p = xform_point(obj->m, rec.p);
        rec.t = length(rec.p - ray.o);
        rec.n = xform_normal(obj->inv_m, rec.n);
        rec.mat = sc.object_material[i];
        ray.tmax = rec.t;
        hit = true;
        win_tri = -1;
      }
    } else {
      float t = ray.tmax;
      int k = -1;
      const uint32_t base = sc.object_tri_base[i];
      mesh_closest_wide<kCount>(ray, sc, sc.mesh_views[sc.object_mesh[i]], obj, base, t, k, stack, flags, tally);
      if (k >= 0) {
        ray.tmax = t;
        rec.t = t;
        rec.mat = sc.object_material[i];
        hit = true;
        win_tri = (int)(base + (uint32_t)k);
      }
    }
  }
  if (win_tri >= 0) {
    const float4 tc = sc.tris[kTriVec4 * (size_t)win_tri + 2u];
    const f3 outward = mk3(tc.y, tc.z, tc.w);
    rec.p = ray_at(ray, rec.t);
    rec.side = dot(ray.d, outward) < 0.0f ? 0u : 1u;
    rec.n = rec.side == 0u ? outward : -outward;
  }
  return hit;
}

// ------------------------------------------------------------------------------------------------
// shading (path_tracer.cu:29-34, 130-201; distributions.cuh:6-19)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f3 background(const f3 dir)
{
  const f3 u = normalize(dir);
  const float t = 0.5f * (u.y + 1.0f);
  return mk3(0.5f, 0.7f, 1.0f) * (1.0f - t) + mk3(1.0f, 1.0f, 1.0f) * t;  // glm::lerp = x*(1-a) + y*a
}

__device__ __forceinline__ f3 random_on_unit_sphere(Minstd& rng)
{
  const float phi = 2.f * 3.14159265358979323846264338327950288f * rng.uniform();
  const float cos_theta = 2.f * rng.uniform() - 1.f;
  const float sin_theta = ieee_sqrt(1.0f - cos_theta * cos_theta);
  float s, c;
  det_sincos(phi, s, c);
  return mk3(c * sin_theta, s * sin_theta, cos_theta);
}

__device__ __forceinline__ float schlick(float cosine, float ref_idx)
{
  float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
  r0 = r0 * r0;
  const float x = 1.0f - cosine;
  const float x2 = x * x;
  const float x4 = x2 * x2;
  return r0 + (1.0f - r0) * (x4 * x);
}

// evaluate_material, path_tracer.cu:138-201.  tmin_flag: ray.t_min is 1e-5 from now on (dielectric).
__device__ __forceinline__ void evaluate_material(f3& ro, f3& rd, bool& tmin_flag, const f3 hp, const f3 hn,
                                                  const uint32_t side, const DMaterial m, Minstd& rng, f3& color)
{
  ro = hp - hn * (1e-4f * sign_of(dot(rd, hn)));
  if (m.type == 0) {
    f3 dir = normalize(hn + random_on_unit_sphere(rng));
    if (fabs((double)dir.x) < 1e-8 && fabs((double)dir.y) < 1e-8 && fabs((double)dir.z) < 1e-8) dir = hn;
    rd = dir;
    color = color * mk3(m.p[0], m.p[1], m.p[2]);
  } else if (m.type == 1) {
    const f3 reflected = rd - (hn * dot(hn, rd)) * 2.0f;
    const f3 dir = reflected + random_on_unit_sphere(rng) * m.p[3];
    rd = dir;
    if (dot(dir, hn) > 0.0f) color = color * mk3(m.p[0], m.p[1], m.p[2]);
    else color = mk3(0.0f, 0.0f, 0.0f);
  } else {
    const float ior = m.p[0];
    const float ratio = side == 0u ? (1.0f / ior) : ior;
    const f3 unit = normalize(rd);
    const float cos_theta = sel_min(dot(-unit, hn), 1.0f);
    const float sin_theta = ieee_sqrt(1.0f - cos_theta * cos_theta);
    const bool cannot_refract = ratio * sin_theta > 1.0f;
    f3 dir;
    if (cannot_refract || schlick(cos_theta, ratio) > rng.uniform()) {
      dir = unit - (hn * dot(hn, unit)) * 2.0f;
    } else {
      const float dv = dot(hn, unit);
      const float k = 1.0f - ratio * ratio * (1.0f - dv * dv);
      dir = (k >= 0.0f) ? (unit * ratio - hn * (ratio * dv + ieee_sqrt(k))) : mk3(0.0f, 0.0f, 0.0f);
    }
    ro = hp;
    rd = dir;
    tmin_flag = true;
  }
}

// evaluate_material + the sky of a miss, for a wavefront whose lanes hold DIFFERENT kinds of work (round 4).  From the
// second bounce on the 64 paths of a wavefront have scattered: in the Cornell-box scenes most wavefronts hold diffuse,
// metal and glass hits and a path that left the box, and a `switch` over the kinds executes its four bodies one after
// the other -- 530 VALU instructions where a diffuse-only wavefront needs 205 (profiles/r04_config2_counters.txt).
// The bodies are made of the same few expensive pieces -- two draws, normalize, sqrt(1 - c^2), the sine / cosine -- on
// different operands.  This form runs every piece ONCE, each lane with the operands of its own kind, in an order that
// every kind can follow:
//     draws (all hits; the second one diffuse and metal)
//     A  unit = normalize(rd)                                   glass, miss
//     B  sin = sqrt(1 - cos^2), cos = 2 u2 - 1 | min(-unit.n, 1)  diffuse, metal | glass
//     .  sine / cosine of phi, the point on the unit sphere     diffuse, metal
//     C  sqrt(|n + r|^2) | sqrt(k)                              diffuse (then 1 / it) | glass that refracts
//     the kinds' own few operations
// A lane executes exactly the operations of its kind's body in evaluate_material / background, on the same operands:
// the same bits (the frames tests compare this kernel with k_shade's plain form and with the oracle).
// kind: 0 diffuse, 1 metal, 2 dielectric (Material::type), 3 miss, anything else: nothing to do.
__device__ __forceinline__ void shade_kinds(const uint32_t kind, f3& ro, f3& rd, bool& tmin_flag, const f3 hp, const f3 hn,
                                            const uint32_t side, const DMaterial m, const uint32_t slot, const uint32_t iteration,
                                            const uint32_t bounce, f3& color)
{
  float u1 = 0.0f, u2 = 0.0f;
  if (kind <= 2u) {
    Minstd rng;
    rng.seed(path_seed(slot, iteration));  // re-seeded from the global slot index, then discard(bounce) (path_tracer.cu:300-301)
    rng.discard(bounce);
    u1 = rng.uniform();
    if (kind <= 1u) u2 = rng.uniform();
  }
  if (kind <= 1u) ro = hp - hn * (1e-4f * sign_of(dot(rd, hn)));
  f3 unit = rd;
  if (kind == 2u || kind == 3u) unit = normalize(rd);
  float cos_theta = 0.0f, sin_theta = 0.0f;
  if (kind <= 2u) {
    cos_theta = kind <= 1u ? 2.f * u2 - 1.f : sel_min(dot(-unit, hn), 1.0f);
    sin_theta = ieee_sqrt(1.0f - cos_theta * cos_theta);
  }
  f3 r = mk3(0.f, 0.f, 0.f);  // random_on_unit_sphere
  if (kind <= 1u) {
    const float phi = 2.f * 3.14159265358979323846264338327950288f * u1;
    float sn, cs;
    det_sincos(phi, sn, cs);
    r = mk3(cs * sin_theta, sn * sin_theta, cos_theta);
  }
  const f3 v = hn + r;  // diffuse
  float root_of = dot(v, v);
  float ratio = 1.0f, dv = 0.0f, k = 0.0f;
  bool reflects = false;
  if (kind == 2u) {
    const float ior = m.p[0];
    ratio = side == 0u ? (1.0f / ior) : ior;
    const bool cannot_refract = ratio * sin_theta > 1.0f;
    reflects = cannot_refract || schlick(cos_theta, ratio) > u1;  // (the draw is the path's first either way)
    dv = dot(hn, unit);
    k = 1.0f - ratio * ratio * (1.0f - dv * dv);
    root_of = k;
  }
  float root = 0.0f;
  if (kind == 0u || (kind == 2u && !reflects)) root = ieee_sqrt(root_of);
  if (kind == 0u) {
    f3 dir = v * (1.0f / root);
    if (fabs((double)dir.x) < 1e-8 && fabs((double)dir.y) < 1e-8 && fabs((double)dir.z) < 1e-8) dir = hn;
    rd = dir;
    color = color * mk3(m.p[0], m.p[1], m.p[2]);
  } else if (kind == 1u) {
    const f3 reflected = rd - (hn * dot(hn, rd)) * 2.0f;
    const f3 dir = reflected + r * m.p[3];
    rd = dir;
    if (dot(dir, hn) > 0.0f) color = color * mk3(m.p[0], m.p[1], m.p[2]);
    else color = mk3(0.0f, 0.0f, 0.0f);
  } else if (kind == 2u) {
    f3 dir;
    if (reflects) dir = unit - (hn * dot(hn, unit)) * 2.0f;
    else dir = (k >= 0.0f) ? (unit * ratio - hn * (ratio * dv + root)) : mk3(0.0f, 0.0f, 0.0f);
    ro = hp;
    rd = dir;
    tmin_flag = true;
  } else if (kind == 3u) {
    const float t = 0.5f * (unit.y + 1.0f);  // background(), path_tracer.cu:29-34
    color = color * (mk3(0.5f, 0.7f, 1.0f) * (1.0f - t) + mk3(1.0f, 1.0f, 1.0f) * t);
  }
}

// final_gather, path_tracer.cu:203-219
__device__ __forceinline__ float running_mean(uint32_t iteration, float old_v, float new_v)
{
  const float sc = (float)(iteration + 1u);
  return iteration == 0u ? new_v : (old_v * (sc - 1.0f) + new_v) / sc;
}
__device__ __forceinline__ void accumulate_color(float4* color4, uint32_t local_pixel, uint32_t iteration, f3 c)
{
  float4 old = iteration == 0u ? make_float4(0.f, 0.f, 0.f, 0.f) : ldnt(&color4[local_pixel]);
  old.x = running_mean(iteration, old.x, c.x);
  old.y = running_mean(iteration, old.y, c.y);
  old.z = running_mean(iteration, old.z, c.z);
  old.w = 0.0f;
  stnt(&color4[local_pixel], old);
}
__device__ __forceinline__ void accumulate_nd(float4* nd4, uint32_t local_pixel, uint32_t iteration, f3 n, float depth)
{
  float4 old = iteration == 0u ? make_float4(0.f, 0.f, 0.f, 0.f) : ldnt(&nd4[local_pixel]);
  old.x = running_mean(iteration, old.x, n.x);
  old.y = running_mean(iteration, old.y, n.y);
  old.z = running_mean(iteration, old.z, n.z);
  old.w = running_mean(iteration, old.w, depth);
  stnt(&nd4[local_pixel], old);
}

// generate_ray, ray_gen.cu:34-61 (frame invariants hoisted into DCamera)
__device__ __forceinline__ void generate_ray(const DCamera& cam, float fx, float fy, f3& o, f3& d)
{
  const float u = fx / (float)(cam.width - 1u);
  const float v = ((float)cam.height - fy) / (float)(cam.height - 1u);
  const float dx = cam.llx + cam.vw * u;
  const float dy = cam.lly + cam.vh * v;
  o = cam.origin;
  d = normalize(xform_vector(cam.cam, mk3(dx, dy, -1.0f)));
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// "filter_rays": may this ray hit one of the mesh objects [filt_begin, filt_end) of the traversal launch that follows?
// No, if it SURELY misses the world boxes of all of them (the same test with the same margin by which that launch skips
// an instance, traverse4m_walk::begin_object) or every box starts beyond the closest hit so far (tmax; a mesh hit needs
// t <= t_max).  A degenerate direction is the traversal launch's business (yes).
__device__ __forceinline__ bool may_hit_boxes(const DObject* objects, uint32_t filt_begin, uint32_t filt_end, const f3 o, const f3 d,
                                              const float tmax)
{
  const f3 winv = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
  bool may_hit = !finite_f(winv.x + winv.y + winv.z);
  for (uint32_t i = filt_begin; i < filt_end && !may_hit; ++i) {
    const DObject* ob = objects + i;
    const f3 a0 = (ld3(ob->bmin) - o) * winv, a1 = (ld3(ob->bmax) - o) * winv;
    const float wn = fmaxf(fmaxf(fminf(a0.x, a1.x), fminf(a0.y, a1.y)), fminf(a0.z, a1.z));
    const float wf = fminf(fminf(fmaxf(a0.x, a1.x), fmaxf(a0.y, a1.y)), fmaxf(a0.z, a1.z));
    const bool surely_missed = finite_f(wn) && finite_f(wf) && (wn - wf) > 4e-6f * (fabsf(wf) + fabsf(wn));
    // (with the margin of the reciprocals, and only for a box in front of the origin)
    const bool beyond = finite_f(wn) && wn > 0.0f && wn * (1.0f - 8e-6f) > tmax;
    may_hit = !(surely_missed || beyond);
  }
  return may_hit;
}
// ... and the rays go on the launch's work list (batch-global slots, DeviceCounters::list_count per frame) IN SLOT ORDER.
// The order does not matter for the results (they are written per slot) but it decides what the traversal launch costs:
// its wavefronts take the list in batches from cursors that move through it, so at any moment an XCD works on
// neighbouring list entries.  With the entries in slot order those are neighbouring pixels and the part of the tree
// they walk stays in the XCD's L2; with workgroups appending in the order they happened to finish (one atomicAdd each,
// tried first) the fabric reads of bounce 0's launch were 13.6 GB per 20 frames instead of 2.1 GB (L2 hit rate 0.54
// instead of 0.89; profiles/r03_worklist_order.txt).
// So a workgroup's place in the list is the exclusive prefix of the counts of the workgroups before it, by the
// decoupled look-back of k_shade_fused on the same descriptors (a launch of its own epoch) -- and for the same reason as
// there a workgroup's tile is its TICKET, not its block index (list_tile).  A workgroup of 256 threads lists up to
// kListPer x 256 slots (thread t: slots tile_first + j * 256 + t, bit j of may_mask); the last tile writes the frame's
// total.  Every thread of the workgroup must call this.
constexpr int kListPer = 4;
struct DTileScan {
  unsigned long long* desc;  // this frame's descriptors
  uint32_t epoch;
};
__device__ __forceinline__ uint32_t tile_lookback(const unsigned long long* desc, uint32_t tile, uint32_t epoch, uint32_t* flags);
constexpr unsigned long long kDescAggregate = 1ull << 32, kDescPrefix = 2ull << 32;

// the tile of this workgroup: tickets of the frame's counter, handed out in the order the workgroups start.  Exactly
// `tiles` workgroups of the frame call this (the others have left, see k_shade_fused).
__device__ __forceinline__ uint32_t list_tile(DeviceCounters* counters, uint32_t tiles)
{
  __shared__ uint32_t s_list_tile;
  if (threadIdx.x == 0u) {
    const uint32_t t = __hip_atomic_fetch_add(&counters->shade_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // every tile of the frame is taken once the last ticket is out: the next launch starts from zero
    if (t + 1u == tiles) __hip_atomic_store(&counters->shade_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_list_tile = t;
  }
  __syncthreads();
  return s_list_tile;
}

__device__ __forceinline__ void list_rays(uint32_t may_mask, uint32_t* worklist, DeviceCounters* counters, size_t frame_base,
                                          uint32_t tile, uint32_t tiles, const DTileScan& scan)
{
  __shared__ uint32_t s_list_cnt[kListPer * 4];
  __shared__ uint32_t s_list_base;
  const uint32_t wave = threadIdx.x >> 6;
  uint32_t rank[kListPer];
#pragma unroll
  for (int j = 0; j < kListPer; ++j) {
    const uint64_t m = __ballot((may_mask >> j & 1u) != 0u);
    rank[j] = rank_below(m);
    if ((threadIdx.x & 63u) == 0u) s_list_cnt[j * 4 + (int)wave] = (uint32_t)__popcll(m);
  }
  __syncthreads();
  if (wave == 0u) {
    uint32_t agg = 0u;
#pragma unroll
    for (int k = 0; k < kListPer * 4; ++k) agg += s_list_cnt[k];
    const unsigned long long tag = (unsigned long long)scan.epoch << 34;
    if (threadIdx.x == 0u)
      __hip_atomic_store(&scan.desc[tile], tag | (tile == 0u ? kDescPrefix : kDescAggregate) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t excl = 0u;
    if (tile != 0u) {
      excl = tile_lookback(scan.desc, tile, scan.epoch, &counters->flags);
      if (threadIdx.x == 0u)
        __hip_atomic_store(&scan.desc[tile], tag | kDescPrefix | (excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0u) {
      s_list_base = excl;
      if (tile + 1u == tiles) counters->list_count = excl + agg;
    }
  }
  __syncthreads();
  uint32_t at = s_list_base;
  const uint32_t tile_first = tile * (256u * (uint32_t)kListPer);
#pragma unroll
  for (int j = 0; j < kListPer; ++j) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t c = s_list_cnt[j * 4 + w];
      if ((uint32_t)w == wave && (may_mask >> j & 1u))
        worklist[frame_base + at + rank[j]] = (uint32_t)frame_base + tile_first + (uint32_t)j * 256u + threadIdx.x;
      at += c;
    }
  }
}

// raygen_kernel, ray_gen.cu:11-32.  Slot s of this context holds pixel band_pixel(band, s).
// kFilter ("filter_rays"): the bounce's first launch is a traversal launch over the mesh objects [filt_begin, filt_end)
// (no sphere run in front of it): the rays that may hit one of their world boxes go on its work list, the others get
// their miss record here (what that launch would have written for them) -- the sky pixels of an outdoor scene never
// reach the traversal kernel.
// kFinish (with kFilter, when that launch walks the scene's WHOLE object list): a ray that is not listed hits nothing at
// all, so its path ends here -- throughput (1, 1, 1) times the sky into the frame, exactly what the shade kernel does
// for a miss at bounce 0 (path_tracer.cu:304-307, ray_gen.cu:26-28) -- and neither its ray nor a miss record is written;
// bounce 0's k_shade_fused then walks the work list instead of all slots.  Per sky pixel and frame: 32 bytes written
// here instead of 48, and 48 bytes the shade kernel no longer reads.
template <bool kFilter, bool kFinish>
__global__ __launch_bounds__(256) void k_raygen(DCameras cams, DBatchInfo bi, DBand band, uint32_t pix_count,
                                                DPaths paths, DeviceCounters* counters, const DObject* objects, uint32_t filt_begin,
                                                uint32_t filt_end, uint32_t* worklist, DHits hits, DTileScan scan, uint32_t tile_stride,
                                                DFrame fb, int staged)
{
  const uint32_t frame = blockIdx.x % bi.count;  // see DBatchInfo; frame-fastest: neighbouring workgroups take their tickets on different lines
  const DCamera& cam = cams.c[frame];
  const uint32_t iteration = bi.iteration[frame];
  paths.o4 += (size_t)frame * bi.stride;
  paths.d4 += (size_t)frame * bi.stride;
  counters += frame;
  scan.desc += (size_t)frame * tile_stride;
  if (kFinish && staged) {
    fb.color4 += (size_t)frame * bi.stride;
    fb.nd4 += (size_t)frame * bi.stride;
  }
  const uint32_t acc_iteration = staged ? 0u : iteration;
  const uint32_t tiles = gridDim.x / bi.count;
  const uint32_t tile = kFilter ? list_tile(counters, tiles) : blockIdx.x / bi.count;
  const uint32_t block_first = tile * (256u * kListPer);  // a workgroup generates kListPer x 256 consecutive slots
  if (tile == 0u) {
    if (threadIdx.x == 0u) counters->live[0] = pix_count;
    // fetch cursors of this frame's persistent traversal launches
    for (uint32_t i = threadIdx.x; i < (uint32_t)kWorkSlots * 8u; i += 256u) (&counters->work[0][0][0])[i * 32u] = 0u;
  }
  uint32_t may_mask = 0u;
#pragma unroll
  for (int j = 0; j < kListPer; ++j) {
    const uint32_t s = block_first + (uint32_t)j * 256u + threadIdx.x;
    if (s >= pix_count) continue;
    const uint32_t pixel = band_pixel(band, s);
    const uint32_t x = pixel % cam.width, y = pixel / cam.width;
    Minstd rng;
    rng.seed(path_seed(pixel, iteration));
    const float fx = (float)x + rng.uniform();
    const float fy = (float)y + rng.uniform();
    f3 o, d;
    generate_ray(cam, fx, fy, o, d);
    bool may_hit = true;
    if (kFilter) {
      may_hit = may_hit_boxes(objects, filt_begin, filt_end, o, d, FLT_MAX);
      may_mask |= may_hit ? 1u << j : 0u;
    }
    if (kFinish && !may_hit) {
      const uint32_t local_pixel = band_local(band, pixel);
      const f3 color = mk3(1.0f, 1.0f, 1.0f) * background(d);
      accumulate_nd(fb.nd4, local_pixel, acc_iteration, -d, 1e6f);
      accumulate_color(fb.color4, local_pixel, acc_iteration, color);
      continue;
    }
    stnt(&paths.o4[s], make_float4(o.x, o.y, o.z, __uint_as_float(pixel)));
    stnt(&paths.d4[s], make_float4(d.x, d.y, d.z, 0.0f));
    // (the throughput of a primary ray is (1, 1, 1), ray_gen.cu:25: the shade kernels know that at bounce 0 and neither
    // is it written here nor read there -- 32 bytes per pixel and frame less)
    if (kFilter && !may_hit) stnt(&hits.tp[(size_t)frame * bi.stride + s], make_float4(-1.0f, 0.f, 0.f, 0.f));
  }
  if (kFilter) list_rays(may_mask, worklist, counters, (size_t)frame * bi.stride, tile, tiles, scan);
}

__device__ __forceinline__ Ray load_ray(const DPaths& paths, uint32_t s)
{
  const float4 o = ldnt(&paths.o4[s]);
  const float4 d = ldnt(&paths.d4[s]);
  Ray r;
  r.o = xyz(o);
  r.d = xyz(d);
  r.tmin = (__float_as_uint(o.w) >> 31) ? 1e-5f : 1e-4f;
  r.tmax = FLT_MAX;
  return r;
}

// intersection_kernel, path_tracer.cu:271-290.  One wavefront per 64-path chunk.
template <bool kCount>
__global__ __launch_bounds__(kWave) void k_trace(DScene sc, DPaths paths, DHits hits, int bounce, DeviceCounters* counters)
{
  __shared__ uint32_t s_stack[kStackDepth * kWave];
  const uint32_t n = counters->live[bounce];
  const uint32_t s = blockIdx.x * kWave + threadIdx.x;
  if (blockIdx.x * kWave >= n) return;
  bool hit = false;
  uint32_t flags = 0u;
  Tally tally;
  if (s < n) {
    const Ray ray = load_ray(paths, s);
    Hit rec;
    rec.t = 0.0f;
    rec.p = rec.n = mk3(0.f, 0.f, 0.f);
    rec.mat = 0u;
    rec.side = 0u;
    hit = ray_scene<kCount>(ray, sc, rec, s_stack + threadIdx.x, flags, tally);
    stnt(&hits.tp[s], make_float4(hit ? rec.t : -1.0f, rec.p.x, rec.p.y, rec.p.z));
    stnt(&hits.nm[s], make_float4(rec.n.x, rec.n.y, rec.n.z, __uint_as_float(rec.mat | (rec.side << 31))));
    if (flags) atomicOr(&counters->flags, flags);
  }
  if (kCount) flush_tally(tally, counters, bounce, true);
}

// The same kernel over the wide layout (default).  Chunks are dealt to workgroups so that workgroups that
// share an XCD (blockIdx % 8, MI355X_MICROARCH.md "Workgroup dispatch") get one contiguous run of
// chunks = one contiguous image region: neighbouring paths walk the same subtrees, which keeps that
// XCD's 4 MiB L2 on one part of the BVH.  Placement only affects speed.
template <bool kCount>
__global__ __launch_bounds__(kWave) void k_trace_wide(DScene sc, DPaths paths, DHits hits, int bounce, DeviceCounters* counters)
{
  __shared__ uint32_t s_stack[kWideStack * kWave];
  const uint32_t n = counters->live[bounce];
  const uint32_t chunks = (n + kChunk - 1u) / kChunk;
  const uint32_t per_xcd = (chunks + 7u) / 8u;
  const uint32_t chunk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per_xcd || chunk >= chunks) return;
  const uint32_t s = chunk * kWave + threadIdx.x;
  bool hit = false;
  uint32_t flags = 0u;
  Tally tally;
  if (s < n) {
    const Ray ray = load_ray(paths, s);
    Hit rec;
    rec.t = 0.0f;
    rec.p = rec.n = mk3(0.f, 0.f, 0.f);
    rec.mat = 0u;
    rec.side = 0u;
    hit = ray_scene_wide<kCount>(ray, sc, rec, s_stack + threadIdx.x, flags, tally);
    stnt(&hits.tp[s], make_float4(hit ? rec.t : -1.0f, rec.p.x, rec.p.y, rec.p.z));
    stnt(&hits.nm[s], make_float4(rec.n.x, rec.n.y, rec.n.z, __uint_as_float(rec.mat | (rec.side << 31))));
    if (flags) atomicOr(&counters->flags, flags);
  }
  if (kCount) flush_tally(tally, counters, bounce, true);
}

// ------------------------------------------------------------------------------------------------
// sphere segments
// ------------------------------------------------------------------------------------------------
#ifndef PT_SPHERE_SKIP
#define PT_SPHERE_SKIP 0
#endif
#ifndef PT_SPHERE_LANES
#define PT_SPHERE_LANES 1
#endif
#ifndef PT_SHADE_KINDS
#define PT_SHADE_KINDS 1
#endif
#ifndef PT_FOLD_TAIL
#define PT_FOLD_TAIL 0
#endif
#ifndef PT_SPHERE_LANES_LEADING
#define PT_SPHERE_LANES_LEADING 0
#endif
// Sphere objects [obj_begin, obj_end) in the reference's order (ray_object_intersection_test, path_tracer.cu:78-100)
// for one ray whose closest hit so far is ray.tmax (FLT_MAX: none).  A sphere that is hit replaces `rec`, shrinks
// ray.tmax and sets `changed`.  Used by k_spheres (a run in front of a mesh, a scene without a mesh) and by
// the kernel that ends the bounce (k_shade_fused / k_tail_count: the run behind the last mesh).
// transform_point (transform.hpp:37-42) divides by w.  For an affine matrix w is exactly 1 whatever the (finite) point --
// (0 x + 0 y) + (0 z + 1) -- and x / 1 is x: when every lane of the wavefront has w == 1 the three IEEE divisions (ten
// instructions each) are skipped; the result has the same bits.  Any lane with another w (a projective matrix, a
// non-finite coordinate) sends the wavefront through the divisions.
__device__ __forceinline__ f3 xform_point_w1(const m4& m, f3 p)
{
  const f4 v = mul(m, p.x, p.y, p.z, 1.0f);
  if (__builtin_expect(__ballot(v.w != 1.0f) == 0ull, 1)) return mk3(v.x, v.y, v.z);
  return mk3(v.x, v.y, v.z) / v.w;
}

// kShareDir (k_spheres only: it has the registers to spare; in k_shade_fused the three extra live values cost more than
// they save, profiles/r03_sphere_math.txt): inverse_transform_ray's direction, normalize(M^-1 (d, 0)), is the same for
// every object whose M^-1 has the identity as its upper 3 x 3 (a translated sphere: every sphere of the Cornell box) --
// (1 dx + 0 dy) + (0 dz + t 0) is dx exactly when the components of d are finite and none is a zero (whose sign the sum
// could change) -- so such a wavefront normalises its direction once for all of them.
// Select approximately, verify exactly (round 4): before the reference's own sequence for a sphere object -- world box
// with six divisions, inverse transform with a normalisation, the quadratic with an IEEE square root and two divisions
// (path_tracer.cu:84-96, intersections.cuh:7-41) -- a dozen approximate operations on the WORLD-space ball around the
// object (DScene::sphere_ball) decide whether that sequence can possibly accept a hit.  true = it surely cannot:
//   * the ray's line passes the ball at more than its radius (the reference's discriminant would be negative), or
//   * the ball lies behind the origin (both roots negative: below t_min), or
//   * the ball starts beyond the closest hit so far: the reference compares the OBJECT-space root with the world-space
//     t_max (transform.hpp:51-58 copies the range unscaled), and an object-space distance is at least the world-space
//     one divided by the matrix's largest stretch.
// Everything is held against the ray with margins far beyond the rounding of either side (relative 1e-4 where the
// arithmetic is good to 1e-6), so a sphere the reference would accept is never skipped; what is not skipped takes the
// reference's sequence unchanged, in list order -- ties, the unscaled-t quirk and the box test included.
__device__ __forceinline__ bool sphere_surely_missed(const float4 ball, const float inv_stretch, const f3 o, const f3 d, const float a,
                                                     const float tmax)
{
  const float R = ball.w;
  const f3 oc = mk3(o.x - ball.x, o.y - ball.y, o.z - ball.z);
  const float oc2 = __builtin_fmaf(oc.x, oc.x, __builtin_fmaf(oc.y, oc.y, oc.z * oc.z));
  const float b = __builtin_fmaf(d.x, oc.x, __builtin_fmaf(d.y, oc.y, d.z * oc.z));  // d . oc (d not normalised)
  const float r2 = R * R;
  const float cc = oc2 - r2;                       // > 0: the origin is outside the ball
  const float scale2 = a * (oc2 + r2);             // the size of the terms the discriminant is made of
  const float disc = __builtin_fmaf(b, b, -(a * cc));
  const float margin = 1e-4f * __builtin_fmaf(b, b, scale2);
  const bool no_ball = !(R >= 0.0f) || !(margin < __builtin_inff());  // no ball for this object, or nothing can be said
  const bool line_misses = disc < -margin;
  const bool outside = cc > 1e-4f * (oc2 + r2);
  const bool behind = outside && b > 0.0f && b * b > 1e-8f * scale2;
  // distance (world units) to the ball along the ray, from below; an object-space root is at least that / stretch
  const float inv_len = __builtin_amdgcn_rsqf(a);
  const float bh = b * inv_len;
  const float dd = __builtin_fmaf(bh, bh, -cc);
  const float entry = -bh - __builtin_amdgcn_sqrtf(fmaxf(dd, 0.0f));  // (v_sqrt_f32, 1 ulp: margins below)
  const float lower = (entry - 1e-4f * (fabsf(bh) + R)) * inv_stretch * 0.9999f;
  const bool beyond = outside && dd > 0.0f && lower > tmax;
  return !no_ball && (line_misses || behind || beyond);
}

template <bool kShareDir = false>
__device__ __forceinline__ void sphere_segment(const DScene& sc, uint32_t obj_begin, uint32_t obj_end, Ray& ray, Hit& rec,
                                               bool& changed)
{
  const float ray_a = __builtin_fmaf(ray.d.x, ray.d.x, __builtin_fmaf(ray.d.y, ray.d.y, ray.d.z * ray.d.z));
  // 1/d by the hardware reciprocal: decides the world-box test of almost every ray without the reference's six
  // divisions per object (below)
  const f3 winv = mk3(__builtin_amdgcn_rcpf(ray.d.x), __builtin_amdgcn_rcpf(ray.d.y), __builtin_amdgcn_rcpf(ray.d.z));
  const bool winv_ok = finite_f(winv.x + winv.y + winv.z);
  const bool d_plain = kShareDir && finite_f(ray.d.x + ray.d.y + ray.d.z) && ray.d.x != 0.0f && ray.d.y != 0.0f && ray.d.z != 0.0f;
  bool have_nd = false;
  f3 nd = mk3(0.f, 0.f, 0.f);
  float nd_a = 0.0f;
  for (uint32_t i = obj_begin; i < obj_end; ++i) {
    const DObject* obj = sc.objects + i;
    if (obj->type != 0u) continue;
    if (PT_SPHERE_SKIP) {
      // (wave-uniform loads; the sequence below runs only when some lane of the wavefront cannot be ruled out)
      const float4 ball = sc.sphere_ball[(size_t)kSphereTab * i];
      const float inv_stretch = sc.sphere_ball[(size_t)kSphereTab * i + 1u].x;
      if (sphere_surely_missed(ball, inv_stretch, ray.o, ray.d, ray_a, ray.tmax)) continue;
    }
    {
      // ray_aabb_intersection_test (intersections.cuh:87-103) decides by the sign of min(far) - max(near).  With
      // reciprocals each slab value is within 3 ulp of the reference's quotient, so a gap beyond 2e-6 of the two
      // extremes has the reference's sign; only a ray that grazes the box within that margin (or has a zero /
      // non-finite direction component) takes the divisions.
      const f3 bmin = ld3(obj->bmin), bmax = ld3(obj->bmax);
      const f3 a0 = (bmin - ray.o) * winv, a1 = (bmax - ray.o) * winv;
      const float wn = fmaxf(fmaxf(fminf(a0.x, a1.x), fminf(a0.y, a1.y)), fminf(a0.z, a1.z));
      const float wf = fminf(fminf(fmaxf(a0.x, a1.x), fmaxf(a0.y, a1.y)), fmaxf(a0.z, a1.z));
      const float gap = wf - wn, margin = 2e-6f * (fabsf(wf) + fabsf(wn)) + 1e-30f;
      const bool box_ok = !(bmin.x > bmax.x || bmin.y > bmax.y || bmin.z > bmax.z);
      bool pass = gap > margin;
      const bool unsure = !box_ok || !winv_ok || !(gap > margin || gap < -margin);
      if (__builtin_expect(unsure, 0)) pass = ray_aabb(ray.o, ray.d, bmin, bmax);
      if (!pass) continue;
    }
    Ray tr;
    tr.o = xform_point_w1(obj->inv_m, ray.o);  // inverse_transform_ray, transform.hpp:51-58
    const m4& im = obj->inv_m;
    const bool identity3 = kShareDir && im.c[0][0] == 1.0f && im.c[1][1] == 1.0f && im.c[2][2] == 1.0f && im.c[0][1] == 0.0f &&
                           im.c[0][2] == 0.0f && im.c[1][0] == 0.0f && im.c[1][2] == 0.0f && im.c[2][0] == 0.0f &&
                           im.c[2][1] == 0.0f;  // (wave-uniform: scalar loads and compares)
    float a;
    if (kShareDir && identity3 && __ballot(!d_plain) == 0ull) {
      if (!have_nd) {
        nd = normalize(ray.d);
        nd_a = dot(nd, nd);
        have_nd = true;
      }
      tr.d = nd;
      a = nd_a;
    } else {
      tr.d = normalize(xform_vector(obj->inv_m, ray.d));
      a = dot(tr.d, tr.d);
    }
    tr.tmin = ray.tmin;
    tr.tmax = ray.tmax;
    const float4 sp = sc.spheres[obj->index];
    if (ray_sphere_a(tr, a, xyz(sp), sp.w, rec)) {
      rec.p = xform_point_w1(obj->m, rec.p);
      rec.t = length(rec.p - ray.o);
      rec.n = xform_normal(obj->inv_m, rec.n);
      rec.mat = sc.object_material[i];
      ray.tmax = rec.t;
      changed = true;
    }
  }
}
// ---- a run of SIMPLE sphere objects, candidates per lane (round 4) -------------------------------------------------
// ---- a run of translated spheres as a fold with the hit record's normal deferred (round 4; k_spheres) ----
// In front of a mesh the spheres are typically a room's walls: every ray is inside every one of them, every wall is hit,
// and which one wins is decided only by distance -- there is nothing to rule out, so sphere_segment pays the whole
// sequence of path_tracer.cu:84-96 + intersections.cuh:7-41 for every object (226 VALU instructions per ray and sphere
// on the Cornell box's walls, behind four dependent vector loads of the object's data: 1.15 ms per 29.5 M rays at 39 % of the VALU
// issue rate and 1.8 TB/s -- bound by neither, by its chains of dependent operations; profiles/r04_config2_counters.txt).  When every
// object of the run is "simple" (DScene::fold_run: both matrices pure translations, 3 x 3 part 1.0f / +-0.0f) the same
// operations on the same operands come much cheaper:
//   * transform_point's row (1 x + e y) + (e' z + t 1), e, e' zeros of either sign, IS x + t: the products with the
//     zeros are zeros, they vanish in the sums, one rounding remains -- unless x is -0.0f (with t a zero the zeros'
//     signs then decide the sign of the result; the fold does not look at t), or a coordinate is not finite (0 * inf).
//     w is (0 x + 0 y) + (0 z + 1) = 1 and nothing is divided.  3 instructions instead of 28, for the ray's origin, and
//     again for the hit point.
//   * transform_normal's row (1 nx + e ny) + (e' nz + e'' 0) IS nx, unless nx is -0.0f.
//   * what the NEXT object needs of an accepted hit is its distance (t_max: length(p_world - o)) -- the normal
//     ((p - centre) / radius: three divisions; the side; the transform) is needed of the LAST accepted hit only: the fold
//     keeps p - centre, the radius and the material of the hit it holds, and finishes the record once, behind the loop.
// A lane that meets one of the exceptions (-0.0f where it matters, a non-finite coordinate, a direction with a zero
// component: inverse_transform_ray's normalised direction is then not the same for every object) sends its wavefront
// through sphere_segment with the ray as it came.  Same bits as sphere_segment in every case (tests/test_gpu_spheres.py).
typedef __attribute__((address_space(4))) const float cfloat;
__device__ __forceinline__ bool neg_zero(const float x) { return __float_as_uint(x) == 0x80000000u; }
__device__ __forceinline__ void sphere_fold(const DScene& sc, const uint32_t obj_begin, const uint32_t obj_end, Ray& ray, Hit& rec, bool& changed)
{
  const bool plain = finite_f(ray.d.x + ray.d.y + ray.d.z) && ray.d.x != 0.0f && ray.d.y != 0.0f && ray.d.z != 0.0f &&
                     finite_f(ray.o.x + ray.o.y + ray.o.z) && !neg_zero(ray.o.x) && !neg_zero(ray.o.y) && !neg_zero(ray.o.z);
  if (__builtin_expect(__ballot(!plain) != 0ull, 0)) {
    sphere_segment<true>(sc, obj_begin, obj_end, ray, rec, changed);
    return;
  }
  const float tmax_in = ray.tmax;
  const f3 winv = mk3(__builtin_amdgcn_rcpf(ray.d.x), __builtin_amdgcn_rcpf(ray.d.y), __builtin_amdgcn_rcpf(ray.d.z));
  const bool winv_ok = finite_f(winv.x + winv.y + winv.z);
  const f3 nd = normalize(ray.d);  // inverse_transform_ray's direction, the same for every object of the run
  const float a = dot(nd, nd);
  bool odd = false;
  f3 pc = mk3(0.f, 0.f, 0.f);      // of the hit the fold holds: p - centre (object space), the radius, the material
  float held_r = 1.0f;
  uint32_t held_mat = 0u;
  const bool tmin_pos = ray.tmin > 0.0f;
  for (uint32_t i = obj_begin; i < obj_end; ++i) {
    // what differs between the objects of such a run, from the table the host packed (DScene::sphere_ball rows 2..6),
    // through the constant address space: scalar loads (the object array itself is read with vector loads -- the
    // compiler cannot know that the kernel's stores do not touch it)
    const cfloat* q = (const cfloat*)reinterpret_cast<const float*>(sc.sphere_ball + (size_t)kSphereTab * i + 2u);
    const f3 bmin = mk3(q[0], q[1], q[2]), bmax = mk3(q[4], q[5], q[6]);
    const f3 itr = mk3(q[3], q[7], q[11]);         // the inverse matrix' translation column
    const f3 center = mk3(q[8], q[9], q[10]);
    const f3 ftr = mk3(q[12], q[13], q[14]);       // the matrix' translation column
    const float radius = q[15];
    const uint32_t material = __float_as_uint(q[16]);
    {  // the object's world box (path_tracer.cu:84).  An origin strictly inside the box: every axis has one slab
       // bound behind the origin and one in front of it, whatever the (non-zero) direction -- near < 0 < far, the
       // reference passes; a room's walls are all of this kind, and a wavefront of such rays skips the slab arithmetic
      const bool inside = ray.o.x > bmin.x && ray.o.x < bmax.x && ray.o.y > bmin.y && ray.o.y < bmax.y && ray.o.z > bmin.z && ray.o.z < bmax.z;
      if (__ballot(!inside) != 0ull) {
        // as in sphere_segment
        const f3 a0 = (bmin - ray.o) * winv, a1 = (bmax - ray.o) * winv;
        const float wn = fmaxf(fmaxf(fminf(a0.x, a1.x), fminf(a0.y, a1.y)), fminf(a0.z, a1.z));
        const float wf = fminf(fminf(fmaxf(a0.x, a1.x), fmaxf(a0.y, a1.y)), fmaxf(a0.z, a1.z));
        const float gap = wf - wn, margin = 2e-6f * (fabsf(wf) + fabsf(wn)) + 1e-30f;
        const bool box_ok = !(bmin.x > bmax.x || bmin.y > bmax.y || bmin.z > bmax.z);
        bool pass = gap > margin;
        const bool unsure = !box_ok || !winv_ok || !(gap > margin || gap < -margin);
        if (__builtin_expect(unsure, 0)) pass = ray_aabb(ray.o, ray.d, bmin, bmax);
        if (!pass && !inside) continue;
      }
    }
    Ray tr;
    tr.o = ray.o + itr;
    tr.d = nd;
    // ray_sphere_intersection_test, intersections.cuh:7-41, up to the accepted root
    const f3 oc = tr.o - center;
    const float b = 2.0f * dot(tr.d, oc);
    const float c = dot(oc, oc) - radius * radius;
    const float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) continue;
    const float sq = ieee_sqrt(disc);
    // t1 = (-b - sq) / (2 a), 2 a > 0: a negative numerator gives a quotient that is negative or -0 and fails
    // t1 >= t_min (> 0) without being divided; a wavefront of rays inside their spheres never divides for t1
    const float n1 = -b - sq;
    const bool t1_out = n1 < 0.0f && tmin_pos && a > 0.0f;
    float t1 = -1.0f;
    if (__ballot(!t1_out) != 0ull) {
      asm volatile("" ::: "memory");  // (keeps the division inside the branch: the compiler would divide and select)
      t1 = n1 / (2.0f * a);
    }
    float t;
    if (!t1_out && t1 >= ray.tmin && t1 <= ray.tmax) {
      t = t1;
    } else {
      const float t2 = (-b + sq) / (2.0f * a);
      if (t2 >= ray.tmin && t2 <= ray.tmax) t = t2;
      else continue;
    }
    const f3 p = ray_at(tr, t);
    pc = p - center;
    held_r = radius;
    held_mat = material;
    rec.p = p + ftr;
    rec.t = length(rec.p - ray.o);
    odd = odd || neg_zero(p.x) || neg_zero(p.y) || neg_zero(p.z) || !finite_f(rec.t);  // (a non-finite p gives a non-finite length)
    ray.tmax = rec.t;
    changed = true;
  }
  if (changed) {
    const f3 outward = pc / held_r;
    rec.side = dot(nd, outward) < 0.0f ? 0u : 1u;
    const f3 n = rec.side == 0u ? outward : -outward;
    odd = odd || !finite_f(n.x + n.y + n.z) || neg_zero(n.x) || neg_zero(n.y) || neg_zero(n.z);
    rec.n = n;
    rec.mat = held_mat;
  }
  if (__builtin_expect(__ballot(odd) != 0ull, 0)) {
    ray.tmax = tmax_in;
    changed = false;
    sphere_segment<true>(sc, obj_begin, obj_end, ray, rec, changed);
  }
}

// sphere_segment walks the run object by object, and the wavefront pays the reference's whole sequence for an object
// whenever ANY lane cannot rule it out -- with 64 lanes that is almost every object: the Cornell box's five wall spheres
// are all "hit" by every ray inside it, and k_spheres ran eight full sequences per ray (645 us a launch, config 2).
// Here every lane first collects ITS candidates, for up to kSlots rays at once (k_shade_fused holds four), with the
// approximate arithmetic of sphere_surely_missed extended to bounds on the root the reference would accept:
//   lo  a lower bound of that root (world distance: a simple object does not stretch),
//   hi  an upper bound of it when the sphere is surely hit beyond t_min (else +inf).
// In list order an object is a candidate unless it is surely missed or lo exceeds `cap`, the smallest hi of the
// surely-hit objects BEFORE it (and the closest hit carried in): when its turn comes the reference's t_max is below its
// root whichever of the earlier objects were accepted, so the reference rejects it too.  (Objects AFTER it never
// matter for it: the reference walks the list in order.)  Then the lanes take their candidates one per iteration, lowest
// list index first and slot by slot, each lane with the data of ITS object (five float4 from DScene::sphere_ball) and
// the reference's operations spelled with the literal matrix entries of a translation -- same operands, same order,
// same bits.  The loop runs as often as the busiest lane has candidates: one to three times where sphere_segment paid
// for eight objects.
// mul(m, x, y, z, w) (pt_math.hpp) with the matrix's translation column taken from `tcol` (per lane), everything else from
// `m` (the run's common entries: wave-uniform)
__device__ __forceinline__ f4 mul_tcol(const m4& m, const f3 tcol, float x, float y, float z, float w)
{
  f4 r;
  r.x = (m.c[0][0] * x + m.c[1][0] * y) + (m.c[2][0] * z + tcol.x * w);
  r.y = (m.c[0][1] * x + m.c[1][1] * y) + (m.c[2][1] * z + tcol.y * w);
  r.z = (m.c[0][2] * x + m.c[1][2] * y) + (m.c[2][2] * z + tcol.z * w);
  r.w = (m.c[0][3] * x + m.c[1][3] * y) + (m.c[2][3] * z + m.c[3][3] * w);
  return r;
}

// The reference's sequence for one sphere object of a simple run (path_tracer.cu:84-96, intersections.cuh:7-41): the same
// operations on the same operands as sphere_segment, with the object's own numbers (q0..q3, see DScene::sphere_ball)
// in vector registers and the entries all objects of the run share in `first` (the run's first object, scalar).
__device__ __forceinline__ bool sphere_exact_simple(const DObject* first, const float4 q0, const float4 q1, const float4 q2, const float4 q3,
                                                    const f3 ro, const f3 rd, const float tmin, const float tmax, Hit& rec)
{
  {
    // the object's world box first (path_tracer.cu:84), as in sphere_segment: reciprocals, the reference's divisions
    // only for a ray that grazes it
    const f3 bmin = xyz(q0), bmax = xyz(q1);
    const f3 winv = mk3(__builtin_amdgcn_rcpf(rd.x), __builtin_amdgcn_rcpf(rd.y), __builtin_amdgcn_rcpf(rd.z));
    const f3 a0 = (bmin - ro) * winv, a1 = (bmax - ro) * winv;
    const float wn = fmaxf(fmaxf(fminf(a0.x, a1.x), fminf(a0.y, a1.y)), fminf(a0.z, a1.z));
    const float wf = fminf(fminf(fmaxf(a0.x, a1.x), fmaxf(a0.y, a1.y)), fmaxf(a0.z, a1.z));
    const float gap = wf - wn, margin = 2e-6f * (fabsf(wf) + fabsf(wn)) + 1e-30f;
    const bool box_ok = !(bmin.x > bmax.x || bmin.y > bmax.y || bmin.z > bmax.z);
    bool pass = gap > margin;
    const bool unsure = !box_ok || !finite_f(winv.x + winv.y + winv.z) || !(gap > margin || gap < -margin);
    if (__builtin_expect(unsure, 0)) pass = ray_aabb(ro, rd, bmin, bmax);
    if (!pass) return false;
  }
  const f3 ti = mk3(q0.w, q1.w, q2.w);
  // inverse_transform_ray (transform.hpp:51-58)
  Ray tr;
  const f4 ov = mul_tcol(first->inv_m, ti, ro.x, ro.y, ro.z, 1.0f);
  tr.o = mk3(ov.x, ov.y, ov.z);
  if (__builtin_expect(ov.w != 1.0f, 0)) tr.o = tr.o / ov.w;  // (x / 1 is x: the division transform_point always makes, skipped)
  const f4 dv = mul_tcol(first->inv_m, ti, rd.x, rd.y, rd.z, 0.0f);
  tr.d = normalize(mk3(dv.x, dv.y, dv.z));
  tr.tmin = tmin;
  tr.tmax = tmax;
  if (!ray_sphere_a(tr, dot(tr.d, tr.d), xyz(q2), q3.w, rec)) return false;
  // path_tracer.cu:92-96: point to world, t = distance, normal by transpose(inv_m)
  const f4 pv = mul_tcol(first->m, xyz(q3), rec.p.x, rec.p.y, rec.p.z, 1.0f);
  rec.p = mk3(pv.x, pv.y, pv.z);
  if (__builtin_expect(pv.w != 1.0f, 0)) rec.p = rec.p / pv.w;
  rec.t = length(rec.p - ro);
  rec.n = xform_normal(first->inv_m, rec.n);  // (no entry of the translation column in it)
  return true;
}

// the candidates of one ray among the objects of the run: bit k = object obj_begin + k.
// Bounds: with oc = origin - centre and u the unit direction, the roots are -u.oc -+ sqrt((u.oc)^2 - (|oc|^2 - r^2)).
// The radicand is a difference of terms up to 1e6 (the Cornell box's walls are spheres of radius 1000): every use of it
// carries m = 1e-5 of those terms (forty times what float arithmetic loses there) on the side that keeps the bound a
// bound -- the reference's own float result lies inside [lo, hi] as well.  The cap is the smallest hi of ALL surely-hit
// objects, not only of the earlier ones: the reference's answer is the closest accepted root (ties: the later object),
// it is at most that cap, and an object whose root lies beyond the cap cannot be the answer nor decide between others
// that could (what it may be accepted for in the reference's walk is overwritten by the object that sets the cap).
// kAllCaps false (k_shade_fused: four rays' worth of state and no registers for eight more bounds): the cap an object is
// held against is the one of the objects before it -- the reference's walk at its plainest, nothing to argue.
template <bool kAllCaps>
__device__ __forceinline__ uint32_t sphere_candidates(const DScene& sc, const uint32_t obj_begin, const uint32_t obj_end, const float4 o4,
                                                      const float4 d4, const float closest)
{
  const f3 o = xyz(o4), d = xyz(d4);
  const float tmin = (__float_as_uint(o4.w) >> 31) ? 1e-5f : 1e-4f;
  const float a = __builtin_fmaf(d.x, d.x, __builtin_fmaf(d.y, d.y, d.z * d.z));
  const float inv_len = __builtin_amdgcn_rsqf(a);
  float cap = closest >= 0.0f ? closest * 1.0001f : FLT_MAX;
  uint32_t cand = 0u;
  float lo_of[kAllCaps ? 8 : 1];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (kAllCaps) lo_of[k] = __builtin_inff();
    const uint32_t i = obj_begin + (uint32_t)k;
    if (i < obj_end) {  // (wave-uniform: the balls come through scalar loads)
      const float4 ball = sc.sphere_ball[(size_t)kSphereTab * i];
      const float R = ball.w;
      const f3 oc = mk3(o.x - ball.x, o.y - ball.y, o.z - ball.z);
      const float oc2 = __builtin_fmaf(oc.x, oc.x, __builtin_fmaf(oc.y, oc.y, oc.z * oc.z));
      const float bh = __builtin_fmaf(d.x, oc.x, __builtin_fmaf(d.y, oc.y, d.z * oc.z)) * inv_len;
      const float r2 = R * R;
      const float dd = __builtin_fmaf(bh, bh, -(oc2 - r2));
      const float m = 1e-5f * (__builtin_fmaf(bh, bh, oc2) + r2);
      // the inner ball (sphere_ball_of): what is hit for sure, and the bounds from the other side
      const float Rin = sc.sphere_ball[(size_t)kSphereTab * i + 1u].z;
      const float dd_in = __builtin_fmaf(bh, bh, -(oc2 - Rin * Rin));
      const float sq_up = __builtin_amdgcn_sqrtf(fmaxf(dd + m, 0.0f)), sq_dn = __builtin_amdgcn_sqrtf(fmaxf(dd_in - m, 0.0f));
      const float e = 1e-5f * (fabsf(bh) + sq_up + R) + 1e-6f;
      const bool near_bad = -bh - sq_dn < tmin - e;   // the nearer root is surely below t_min: only the farther one counts
      const bool near_ok = -bh - sq_up > tmin + e;    // ... surely at or above it: it is the one
      const bool missed = dd < -m || -bh + sq_up < tmin - e;
      const bool sure = dd_in > m && (near_ok || (near_bad && -bh + sq_dn > tmin + e));
      const float lo = near_bad ? -bh + sq_dn - e : -bh - sq_up - e;
      const float hi = near_ok ? -bh - sq_dn + e : -bh + sq_up + e;
      const bool known = m < __builtin_inff();        // (anything non-finite: a candidate, and no bound from it)
      const float lo_k = !known ? -__builtin_inff() : (missed ? __builtin_inff() : lo);
      if (kAllCaps) lo_of[k] = lo_k;
      else cand |= lo_k <= cap ? 1u << k : 0u;
      cap = (known && sure) ? fminf(cap, hi * 1.0001f) : cap;
    }
  }
  if (kAllCaps) {
#pragma unroll
    for (int k = 0; k < 8; ++k) cand |= lo_of[k] <= cap ? 1u << k : 0u;
  }
  return cand;
}

// (component by component: a conditional expression on the float4 STRUCT is compiled as a choice between two addresses
// in scratch memory)
__device__ __forceinline__ float4 sel4(const bool c, const float4 a, const float4 b)
{
  return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}

// o4 / d4: the rays as the path arrays hold them; tp / nm: the hit records (tp.x < 0: none yet), updated in place.
// valid: bit j = slot j holds a ray.  Returns the slots whose record changed.  (Every array index below is a literal:
// a loop over the slots, even a fully unrolled one, left the arrays in scratch memory.)
template <int kSlots>
__device__ __forceinline__ uint32_t sphere_run_lanes(const DScene& sc, const uint32_t obj_begin, const uint32_t obj_end, const float4 (&o4)[kSlots],
                                                     const float4 (&d4)[kSlots], float4 (&tp)[kSlots], float4 (&nm)[kSlots], const uint32_t valid)
{
  static_assert(kSlots >= 1 && kSlots <= 4, "slots");
  uint32_t cand = 0u;
  constexpr bool kAllCaps = kSlots == 1;
  if (valid & 1u) cand |= sphere_candidates<kAllCaps>(sc, obj_begin, obj_end, o4[0], d4[0], tp[0].x);
  if constexpr (kSlots > 1) { if (valid & 2u) cand |= sphere_candidates<kAllCaps>(sc, obj_begin, obj_end, o4[1], d4[1], tp[1].x) << 8; }
  if constexpr (kSlots > 2) { if (valid & 4u) cand |= sphere_candidates<kAllCaps>(sc, obj_begin, obj_end, o4[2], d4[2], tp[2].x) << 16; }
  if constexpr (kSlots > 3) { if (valid & 8u) cand |= sphere_candidates<kAllCaps>(sc, obj_begin, obj_end, o4[3], d4[3], tp[3].x) << 24; }
  uint32_t changed = 0u;
  while (__ballot(cand != 0u) != 0ull) {
    if (cand != 0u) {
      const int bit = __ffs((int)cand) - 1;
      cand &= cand - 1u;
      const int j = bit >> 3;
      const uint32_t i = obj_begin + (uint32_t)(bit & 7);
      const float4* q = sc.sphere_ball + (size_t)kSphereTab * i + 2u;
      const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
      float4 ro4 = o4[0], rd4 = d4[0];
      float tcur = tp[0].x;
      if constexpr (kSlots > 1) { ro4 = sel4(j == 1, o4[1], ro4); rd4 = sel4(j == 1, d4[1], rd4); tcur = j == 1 ? tp[1].x : tcur; }
      if constexpr (kSlots > 2) { ro4 = sel4(j == 2, o4[2], ro4); rd4 = sel4(j == 2, d4[2], rd4); tcur = j == 2 ? tp[2].x : tcur; }
      if constexpr (kSlots > 3) { ro4 = sel4(j == 3, o4[3], ro4); rd4 = sel4(j == 3, d4[3], rd4); tcur = j == 3 ? tp[3].x : tcur; }
      const float tmin = (__float_as_uint(ro4.w) >> 31) ? 1e-5f : 1e-4f;
      Hit rec;
      if (sphere_exact_simple(sc.objects + obj_begin, q0, q1, q2, q3, xyz(ro4), xyz(rd4), tmin, tcur >= 0.0f ? tcur : FLT_MAX, rec)) {
        const uint32_t mat = __float_as_uint(q[4].x);
        const float4 ntp = make_float4(rec.t, rec.p.x, rec.p.y, rec.p.z);
        const float4 nnm = make_float4(rec.n.x, rec.n.y, rec.n.z, __uint_as_float(mat | (rec.side << 31)));
        tp[0] = sel4(j == 0, ntp, tp[0]);
        nm[0] = sel4(j == 0, nnm, nm[0]);
        if constexpr (kSlots > 1) { tp[1] = sel4(j == 1, ntp, tp[1]); nm[1] = sel4(j == 1, nnm, nm[1]); }
        if constexpr (kSlots > 2) { tp[2] = sel4(j == 2, ntp, tp[2]); nm[2] = sel4(j == 2, nnm, nm[2]); }
        if constexpr (kSlots > 3) { tp[3] = sel4(j == 3, ntp, tp[3]); nm[3] = sel4(j == 3, nnm, nm[3]); }
        changed |= 1u << j;
      }
    }
  }
  return changed;
}

__device__ __forceinline__ void store_hit(const DHits& hits, uint32_t slot, const Hit& rec)
{
  stnt(&hits.tp[slot], make_float4(rec.t, rec.p.x, rec.p.y, rec.p.z));
  stnt(&hits.nm[slot], make_float4(rec.n.x, rec.n.y, rec.n.z, __uint_as_float(rec.mat | (rec.side << 31))));
}

// ------------------------------------------------------------------------------------------------
// persistent traversal: ray hand-out
// ------------------------------------------------------------------------------------------------
// Traversal lengths differ by an order of magnitude between neighbouring rays (a ray that grazes the terrain
// tests hundreds of boxes, its neighbour a few dozen), so "one wavefront = 64 fixed rays" leaves most lanes idle
// most of the time (measured: 21 % of lanes active per VALU instruction).  A traversal wavefront therefore lives
// for the whole launch and every lane that finishes its ray takes the next unprocessed one.  Results are written
// per slot, so the order in which rays are processed is irrelevant to the output.  One launch handles ONE mesh
// object (its matrices stay in scalar registers); the closest hit so far travels in the hit record between the
// segments of a bounce.  The live paths [0, n) of a frame are cut into eight image regions (one per XCD:
// blockIdx % 8) and each region into 64-ray batches; a share of the batches is dealt statically (no atomics),
// the rest is taken from one cursor per region (own 128-byte line).
// RayFeed for a batch of frames (DBatchInfo): count x 8 regions, region (f, r) = eighth r of frame f's live
// rays, its cursor on frame f's counters.  A wavefront's home keeps the XCD <-> image-region pairing of
// RayFeed (r = blockIdx & 7) and deals the frames round-robin over the wavefronts of that XCD.  Ranges are
// returned as batch-global slots (f * stride + slot).  When its static share is done a wavefront looks at
// all regions at once -- lane p probes the p-th region in its preference order (same eighth of the other
// frames first: same part of the tree in this XCD's L2) -- and takes from the first that has rays left.
struct BatchFeed {
  DeviceCounters* ctr;
  uint32_t stride, count, bounce, work_slot, static_eighths, dyn_batch;
  bool listed;  // the launch walks a work list (DeviceCounters::list_count entries per frame, read through `order`), not all live rays
  __device__ __forceinline__ uint32_t rays_of(uint32_t f) const { return listed ? ctr[f].list_count : ctr[f].live[bounce]; }
  uint32_t home_f, home_r, home_base, home_rs, stat_next, stat_step, stat_count;
  bool in_static, done;

  // (the regions' geometry: pt_feed_rules.hpp, shared with the host's check, ptc_check_feed)
  static __device__ __forceinline__ uint32_t region_size_of(uint32_t n) { return feed_rules::region_size_of(n); }
  static __device__ __forceinline__ uint32_t region_len_of(uint32_t n, uint32_t rs, uint32_t r) { return feed_rules::region_len_of(n, rs, r); }
  static __device__ __forceinline__ uint32_t pos_of(uint32_t rs, uint32_t r, uint32_t local) { return feed_rules::pos_of(rs, r, local); }
  __device__ __forceinline__ uint32_t static_batches_of(uint32_t len) const
  {
    return feed_rules::static_batches_of(len, static_eighths);
  }
  __device__ __forceinline__ void init(DeviceCounters* ctr_, const DBatchInfo& bi, int bounce_, int work_slot_,
                                       uint32_t static_eighths_, bool listed_ = false)
  {
    ctr = ctr_;
    listed = listed_;
    stride = bi.stride;
    count = bi.count;
    bounce = (uint32_t)bounce_;
    work_slot = (uint32_t)work_slot_;
    // every home needs at least one wavefront for its static share
    static_eighths = gridDim.x >= 8u * count ? static_eighths_ : 0u;
    const uint32_t j = blockIdx.x >> 3;
    home_r = blockIdx.x & 7u;
    home_f = j % count;
    const uint32_t with_r = (gridDim.x - home_r + 7u) / 8u;  // wavefronts of this r
    stat_step = (with_r - home_f + count - 1u) / count;      // ... of which this many share the home
    stat_next = j / count;
    const uint32_t n = rays_of(home_f);
    home_rs = region_size_of(n);
    stat_count = static_batches_of(region_len_of(n, home_rs, home_r));
    home_base = home_f * stride;
    in_static = static_eighths != 0u;
    done = false;
    // dynamic batches: one atomic hands out this many rays (a cursor line sustains ~30 atomics/us)
    // (re-measured on round 3's final build: 64 everywhere -6 %, 256 or other thresholds within noise)
    // (the same for every wavefront of the launch -- frame 0's count decides: the cursors then only ever stand on multiples
    // of it, and a batch of two never straddles two blocks of a region)
    dyn_batch = (uint64_t)rays_of(0u) * count / gridDim.x >= 256u ? 128u : (uint32_t)kWave;
  }
  __device__ __forceinline__ bool exhausted() const { return !in_static && done; }
  // wave-uniform: next batch [begin, end) of batch-global slots, or false
  __device__ __forceinline__ bool acquire(uint32_t& begin, uint32_t& end)
  {
    if (in_static) {
      if (stat_next < stat_count) {
        begin = home_base + pos_of(home_rs, home_r, stat_next * kWave);
        end = begin + kWave;  // static batches are full batches inside the region
        stat_next += stat_step;
        return true;
      }
      in_static = false;
    }
    while (!done) {
      bool any = false;
      for (uint32_t first_p = 0u; first_p < 8u * count; first_p += (uint32_t)kWave) {
        const uint32_t p = first_p + threadIdx.x;
        bool has = false;
        uint32_t len = 0u, first = 0u, gbase = 0u, grs = 0u, gr = 0u;
        uint32_t* cursor = nullptr;
        if (p < 8u * count) {
          const uint32_t df = p % count, dr = p / count;
          uint32_t f = home_f + df;
          if (f >= count) f -= count;
          // (own region first, then the next ones in turn.  Every wavefront walking the regions in the same order -- so that
          // the launch ends on a chosen one -- was measured: the end of a primary-ray launch 300 instead of 570 us after its
          // feed, the launch as a whole 2 % LONGER, the later bounces 5-20 %: twenty cursors for 5120 wavefronts.)
          const uint32_t r = (home_r + dr) & 7u;
          const uint32_t n = rays_of(f);
          const uint32_t rs = region_size_of(n);
          len = region_len_of(n, rs, r);
          first = static_batches_of(len) * kWave;
          gbase = f * stride;
          grs = rs;
          gr = r;
          cursor = &ctr[f].work[work_slot][r][0];
          has = first < len && first + __hip_atomic_load(cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < len;
        }
        const uint64_t mask = __ballot(has);
        if (mask == 0ull) continue;
        any = true;
        const int sel = __ffsll((unsigned long long)mask) - 1;
        uint32_t base = len;
        if ((int)threadIdx.x == sel) base = first + atomicAdd(cursor, dyn_batch);
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)base, sel);
        const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)len, sel);
        const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)gbase, sel);
        if (b < l) {
          const uint32_t rs_sel = (uint32_t)__builtin_amdgcn_readlane((int)grs, sel), r_sel = (uint32_t)__builtin_amdgcn_readlane((int)gr, sel);
          begin = g + pos_of(rs_sel, r_sel, b);
          end = begin + (min(l, b + dyn_batch) - b);
          return true;
        }
        break;  // another wavefront took the rest of that region: look again
      }
      if (!any) done = true;
    }
    return false;
  }
};

// ------------------------------------------------------------------------------------------------
// variant 3 (default): persistent wavefronts over the four-wide collapse, 64-byte quantised nodes
// ------------------------------------------------------------------------------------------------
// Conservative FMA slabs on four children per step, children visited nearest first, optimistic acceptance,
// one exact test of the winner (finalize below), results written in batches just before a refill.
#ifndef PT_T4_WAVES
#define PT_T4_WAVES 5
#endif
// iterations between two rounds of work splitting at the end of a launch (4 until round 3: every iteration is 3.6 % faster
// on a 1/64 share of the frame, 1.2 % on an eighth, +-0 on the whole; every eighth is slower everywhere)
#ifndef PT_SPLIT_EVERY
#define PT_SPLIT_EVERY 1u
#endif
// end of a launch: finished lanes are retired when this many wait, or every so many iterations
#ifndef PT_RETIRE_LANES
#define PT_RETIRE_LANES 4u
#endif
#ifndef PT_RETIRE_EVERY
#define PT_RETIRE_EVERY 3u
#endif
#ifndef PT_FULL_SORT
#define PT_FULL_SORT 1
#endif
#ifndef PT_FLAT_TRI
#define PT_FLAT_TRI 1
#endif


// -DPT_TAILPROF (diagnostic builds only, tools/tailprof.py): per traversal launch and wavefront the times (100 MHz
// wall clock) at which it started, found the feed exhausted and left the loop, and its loop iterations / split rounds
// taken -- where does the end of a launch go?
#ifdef PT_TAILPROF
__device__ unsigned long long g_tailprof[16][8192][8];
extern "C" int ptc_debug_tailprof(void* dst, size_t bytes)
{
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (!dst) {  // clear
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tailprof)) != hipSuccess) return -1;
    return hipMemset(p, 0, sizeof(g_tailprof)) == hipSuccess && hipDeviceSynchronize() == hipSuccess ? 0 : -1;
  }
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tailprof), bytes < sizeof(g_tailprof) ? bytes : sizeof(g_tailprof)) == hipSuccess ? 0 : -1;
}
#endif

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, kWave);
  return v;
}

// End of a persistent traversal launch, run by its LAST wavefront (after the exact redo): the bookkeeping of the next
// launch on this stream starts from zero -- the redo list, the sign-off counter and the fetch cursors of this launch's
// set for every frame of the batch (plain stores: the next launch starts after this one has completed).
__device__ __forceinline__ void launch_epilogue(DeviceCounters* counters, int bounce, int work_slot, uint32_t redone,
                                                const DBatchInfo& bi, bool was_listed)
{
  for (uint32_t i = threadIdx.x; i < bi.count * 8u; i += (uint32_t)kWave) counters[i >> 3].work[work_slot][i & 7u][0] = 0u;
  // what was on the work lists goes into the profile (list_count itself stays: the kernel that builds a list always
  // writes it, and bounce 0's k_shade_fused may walk the same list after this launch)
  uint32_t listed = 0u;
  if (was_listed)
    for (uint32_t f = threadIdx.x; f < bi.count; f += (uint32_t)kWave) listed += counters[f].list_count;
  listed = wave_sum(listed);
  if (threadIdx.x == 0u) {
    if (was_listed) counters->listed_now[bounce] = counters->list_count;
    counters->listed_rays[bounce] += listed;
    counters->slow_rays[bounce] += redone;
    counters->slow_count = 0u;
    counters->waves_done = 0u;
  }
}

// Set a ray aside for the exact redo at the end of the launch (redo_slow_rays).  The entry is written with an
// agent-scope atomic store: the wavefront that drains the list may run on another XCD (its own L2).
__device__ __forceinline__ void set_aside(DeviceCounters* counters, uint32_t* slow_list, uint32_t slot)
{
  const uint32_t at = atomicAdd(&counters->slow_count, 1u);
  __hip_atomic_store(&slow_list[at], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool kCount, bool kFirst, bool kBeam>
__device__ __forceinline__ void traverse4_walk(const DScene& sc, uint32_t obj_index, const DPaths& paths, const DHits& hits,
                                               int bounce, int work_slot, DeviceCounters* counters, uint32_t* slow_list,
                                               const uint32_t* order, const DBatchInfo& bi, const bool listed)
{
  __shared__ uint32_t s_stack[kLds4 * kWave];
  // work splitting at the end of a launch (see `split` below): per lane = per ray group led by that lane
  __shared__ unsigned long long s_grp_best[kWave];  // best candidate of the group so far: t bits << 32 | ~triangle
  __shared__ uint32_t s_grp_count[kWave];           // lanes still walking for the group
  __shared__ uint32_t s_leader[kWave];              // per lane: the lane that leads the ray it is walking for
  __shared__ uint32_t s_pair[kWave];                // scratch: r-th donor of a split round
  // explicitly an LDS pointer: as a generic pointer the pop below compiles to a flat load
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  lds_u32* stack = (lds_u32*)s_stack + threadIdx.x;
  const uint32_t gid = blockIdx.x * kWave + threadIdx.x;
  // `slot` below is batch-global (frame * bi.stride + slot in the frame); flags, the slow-ray list and the
  // test tallies of the whole batch go to frame 0's counters
  uint32_t n_max = 0u;
  for (uint32_t f = 0; f < bi.count; ++f) n_max = max(n_max, listed ? counters[f].list_count : counters[f].live[bounce]);
  if (n_max == 0u) return;
  const DObject* obj = sc.objects + obj_index;
  const uint32_t mat = sc.object_material[obj_index];
  const float4* tris = sc.tris + kTriVec4 * (size_t)sc.object_tri_base[obj_index];
  // more wavefronts than batches (the margin keeps every wavefront that owns a static batch, see BatchFeed)
  if (blockIdx.x >= ((n_max + kWave - 1u) / kWave + 8u) * bi.count) return;
  // the object's world box: the same for every ray of the launch (scalar registers)
  auto uni = [](float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); };
  const f3 obj_bmin = mk3(uni(obj->bmin[0]), uni(obj->bmin[1]), uni(obj->bmin[2]));
  const f3 obj_bmax = mk3(uni(obj->bmax[0]), uni(obj->bmax[1]), uni(obj->bmax[2]));
#ifdef PT_TAILPROF
  const unsigned long long tp_start = wall_clock64();
  unsigned long long tp_exhausted = 0ull;
  uint32_t tp_iters = 0u, tp_splits = 0u, tp_iters_exh = 0u, tp_lanes_exh = 0u;
  unsigned long long tp_c_split = 0ull, tp_c_retire = 0ull, tp_c_step = 0ull, tp_lanes_tail = 0ull;
#endif
  BatchFeed feed;
  feed.init(counters, bi, bounce, work_slot, sc.static_eighths, listed);
  uint32_t priv_next = 0u, priv_end = 0u;

  bool active = false;
  bool pending = false;
  uint32_t slot = 0u, cur = 0u, flags = 0u;
  int sp = 0, sbase = 0, best_k = -1;  // the lane's stack is entries [sbase, sp) of its column
  bool split_mode = false;             // wave-uniform: some ray of this wavefront is walked by several lanes
  uint32_t since_split = 0u, since_retire = 0u;
  f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 0), inv = mk3(0, 0, 0);
  f3 oin = mk3(0, 0, 0), oif = mk3(0, 0, 0);
  bool neg_x = false, neg_y = false, neg_z = false;  // sign of 1/d per axis: which plane of a box is the near one
  float tmin = 0.0f, best_t = 0.0f, scale = 0.0f, limit = 0.0f;
  Tally tally;
  uint32_t ray_boxes = 0u;

  // entries [0, lds_cap) of a lane's stack live in LDS, the rest in the launch's global overflow area (lds_cap is
  // kLds4 except in tests that want the overflow path exercised by small scenes)
  const int lds_cap = min((int)sc.lds_cap, kLds4);  // (the host sets lds_cap <= kLds4: same header)
  auto push = [&](uint32_t ref) {
    if (sp < lds_cap) stack[sp * kWave] = ref;
    else if (sp < lds_cap + (int)sc.spill_cap) sc.spill[(size_t)(sp - lds_cap) * sc.spill_stride + gid].x = ref;
    else {
      flags |= kFlagStackOverflow;
      return;
    }
    ++sp;
  };
  // the conservative slab pair of one box for this lane's ray, tolerance folded INWARDS: if even that interval is
  // non-empty, the reference's exact test of the box passes
  auto surely_inside = [&](const f3 lo, const f3 hi) -> bool {
    const bool nx = neg_x, ny = neg_y, nz = neg_z;
    const float tn = fmaxf(fmaxf(__builtin_fmaf(nx ? hi.x : lo.x, inv.x, oif.x), __builtin_fmaf(ny ? hi.y : lo.y, inv.y, oif.y)),
                           __builtin_fmaf(nz ? hi.z : lo.z, inv.z, oif.z));
    const float tf = fminf(fminf(__builtin_fmaf(nx ? lo.x : hi.x, inv.x, oin.x), __builtin_fmaf(ny ? lo.y : hi.y, inv.y, oin.y)),
                           __builtin_fmaf(nz ? lo.z : hi.z, inv.z, oin.z));
    return tf >= tn;
  };
  // Result of a finished ray.  The winner is the closest of ALL candidates; it is the reference's answer iff the
  // reference reaches it: the object's world box passes (path_tracer.cu:84, tested here instead of before the
  // walk) and the box of the winner's parent passes the reference's own test (nesting).  Both tests have a
  // cheap sufficient form (approximate arithmetic with the error bound held against the ray); the exact
  // divisions run only for rays that graze a box.
  auto finalize = [&]() {
    // everything a winner needs from memory -- its parent's box, its normal -- requested in one go (three
    // dependent round trips otherwise: this code runs every few loop iterations)
    const size_t win = (size_t)max(best_k, 0);
    const float4 pb0 = sc.cur.leaf_parent[2u * win], pb1 = sc.cur.leaf_parent[2u * win + 1u];
    const float4 tc = tris[kTriVec4 * win + 2u];
    if (best_k >= 0) {
      // world box (ray_aabb, intersections.cuh:87-103): quotients by reciprocal, each within 3 ulp of the quotient
      const f3 bmin = obj_bmin, bmax = obj_bmax;
      const f3 winv = mk3(__builtin_amdgcn_rcpf(rd.x), __builtin_amdgcn_rcpf(rd.y), __builtin_amdgcn_rcpf(rd.z));
      const f3 a0 = (bmin - ro) * winv, a1 = (bmax - ro) * winv;
      const float wn = fmaxf(fmaxf(fminf(a0.x, a1.x), fminf(a0.y, a1.y)), fminf(a0.z, a1.z));
      const float wf = fminf(fminf(fmaxf(a0.x, a1.x), fmaxf(a0.y, a1.y)), fmaxf(a0.z, a1.z));
      const bool box_ok = !(bmin.x > bmax.x || bmin.y > bmax.y || bmin.z > bmax.z);
      bool world_sure = box_ok && finite_f(winv.x + winv.y + winv.z) && (wf - wn) > 2e-6f * (fabsf(wf) + fabsf(wn));
      const bool parent_sure = surely_inside(xyz(pb0), xyz(pb1));
      asm volatile("" ::"v"(tc.y), "v"(tc.z), "v"(tc.w));  // keeps the normal's load up there with the box's
      if (__builtin_expect(!(world_sure && parent_sure) || sc.force_slow == 2u, 0)) {
        if (!ray_aabb(ro, rd, bmin, bmax)) {
          best_k = -1;  // the reference skips the object: the carried hit (or the miss) stands
        } else {
          const f3 od = normalize(xform_vector(obj->inv_m, rd));  // inverse_transform_ray, transform.hpp:51-58
          const f3 oo = xform_point(obj->inv_m, ro);
          float en, ef;
          if (!slab_exact(xyz(pb0), xyz(pb1), oo, od, en, ef) || sc.force_slow == 2u) {
            // a ray grazing the parent's box within rounding: redone with exact box decisions by the launch's epilogue (redo_slow_rays)
            set_aside(counters, slow_list, slot);
            best_k = -2;
          }
        }
      }
    }
    if (best_k >= 0) {
      const f3 outward = mk3(tc.y, tc.z, tc.w);
      const f3 p = ro + rd * best_t;
      const uint32_t side = dot(rd, outward) < 0.0f ? 0u : 1u;
      const f3 nn = side == 0u ? outward : -outward;
      stnt(&hits.tp[slot], make_float4(best_t, p.x, p.y, p.z));
      stnt(&hits.nm[slot], make_float4(nn.x, nn.y, nn.z, __uint_as_float(mat | (side << 31))));
    } else if (kFirst && best_k == -1) {
      stnt(&hits.tp[slot], make_float4(-1.0f, 0.f, 0.f, 0.f));
    }
    if (kCount) atomicMax(&counters->max_box_tests[bounce], ray_boxes);
  };

  // ---- work splitting: the tail of a launch -------------------------------------------------------------------
  // When the launch has no rays left to hand out, a wavefront's lanes fall idle one by one while a few long rays
  // (a ray grazing the terrain visits hundreds of nodes) keep the launch -- and the whole bounce behind it -- alive:
  // 100-150 us per launch, whatever its size, which is most of the time of a small launch (a single frame, a rank's
  // share of a multi-GPU frame, the late bounces).  An idle lane then takes the BOTTOM entry of a busy lane's stack
  // (the farthest, usually largest pending subtree) together with a copy of its ray and walks it as a member of
  // that ray's group.  Candidates are merged in LDS with a 64-bit minimum (t, then the reference's tie rule: the
  // later triangle); the last member to finish writes the ray's result like an unsplit lane would.  The closest
  // hit is the same whoever walks which subtree, so results do not change (tests: every schedule bit-identical).
  auto candidate_key = [&]() -> unsigned long long {
    return best_k >= 0 ? ((unsigned long long)__float_as_uint(best_t) << 32) | (unsigned long long)(~(uint32_t)best_k) : ~0ull;
  };
  auto adopt = [&](unsigned long long key) {
    if (key != ~0ull) {
      const float t = __uint_as_float((uint32_t)(key >> 32));
      const int k = (int)~(uint32_t)key;
      if (t < best_t || (t == best_t && k > best_k)) {
        best_t = t;
        best_k = k;
        limit = scale * t * 1.001f;
      }
    }
  };
  // a lane whose walk has ended: alone -> finalize; in a group -> hand in the candidate, and finalize only as the last
  auto retire = [&]() {
    if (split_mode) {
      const uint32_t leader = s_leader[threadIdx.x];
      __hip_atomic_fetch_min(&s_grp_best[leader], candidate_key(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t left = __hip_atomic_fetch_sub(&s_grp_count[leader], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (left != 1u) return;  // another member is still walking: it will finish the ray
      best_k = -1;
      best_t = FLT_MAX;
      adopt(__hip_atomic_load(&s_grp_best[leader], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    }
    finalize();
  };
  auto split = [&]() {
    // share what the members of a group know (tightens every member's culling limit)
    if (split_mode && active) {
      const uint32_t leader = s_leader[threadIdx.x];
      const unsigned long long mine = candidate_key();
      const unsigned long long seen = __hip_atomic_fetch_min(&s_grp_best[leader], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      adopt(seen);
    }
    const bool can_give = active && sp - sbase >= 1 && sbase < lds_cap;
    const uint64_t donors = __ballot(can_give), takers = __ballot(!active && !pending);
    const uint32_t pairs = min((uint32_t)__popcll(donors), (uint32_t)__popcll(takers));
    if (pairs == 0u) return;
    if (!split_mode) {  // first split of this wavefront: every lane leads its own ray
      s_leader[threadIdx.x] = threadIdx.x;
      s_grp_count[threadIdx.x] = active || pending ? 1u : 0u;  // (a finished lane that has not retired yet still owes its ray's result)
      s_grp_best[threadIdx.x] = ~0ull;
      split_mode = true;
    }
    const uint32_t drank = rank_below(donors), trank = rank_below(takers);
    if (can_give && drank < pairs) s_pair[drank] = threadIdx.x;
    const bool take = !active && !pending && trank < pairs;
    const uint32_t d = take ? s_pair[trank] : threadIdx.x;  // my donor (myself: no change)
    // the donor's ray and walk state (every lane reads its partner's registers)
    auto from = [&](float v) { return __shfl(v, (int)d, kWave); };
    const f3 d_ro = mk3(from(ro.x), from(ro.y), from(ro.z)), d_rd = mk3(from(rd.x), from(rd.y), from(rd.z));
    const f3 d_inv = mk3(from(inv.x), from(inv.y), from(inv.z));
    const f3 d_oin = mk3(from(oin.x), from(oin.y), from(oin.z)), d_oif = mk3(from(oif.x), from(oif.y), from(oif.z));
    const float d_tmin = from(tmin), d_best_t = from(best_t), d_scale = from(scale), d_limit = from(limit);
    const int d_best_k = __shfl(best_k, (int)d, kWave), d_sbase = __shfl(sbase, (int)d, kWave);
    const uint32_t d_slot = (uint32_t)__shfl((int)slot, (int)d, kWave);
    if (take) {
      ro = d_ro;
      rd = d_rd;
      inv = d_inv;
      oin = d_oin;
      oif = d_oif;
      neg_x = inv.x < 0.0f;
      neg_y = inv.y < 0.0f;
      neg_z = inv.z < 0.0f;
      tmin = d_tmin;
      best_t = d_best_t;
      best_k = d_best_k;
      scale = d_scale;
      limit = d_limit;
      slot = d_slot;
      cur = ((lds_u32*)s_stack)[d_sbase * kWave + (int)d];  // the bottom of the donor's stack
      sp = sbase = 0;
      ray_boxes = 0u;
      const uint32_t leader = s_leader[d];
      s_leader[threadIdx.x] = leader;
      __hip_atomic_fetch_add(&s_grp_count[leader], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      active = true;
    }
    if (can_give && drank < pairs) ++sbase;
  };

  // One step of every active lane (the body of both loops below)
  auto step = [&]() {
    // One loop iteration = one step per lane, and ONE memory round trip: the lane's current reference is either a
    // node or a leaf, both records are fetched by the SAME four 16-byte loads from a per-lane address (a 64-byte
    // node, or a 48-byte triangle record whose last 16 bytes are simply requested twice), and the stack entry the
    // lane falls back to is read from LDS meanwhile.  Vector-memory instructions and dependent round trips are
    // what this loop is bound by (lesson x): four loads and one wait per iteration, where
    // separate node and triangle phases needed seven loads and two waits.
    if (active) {
      const bool is_leaf = (cur & kLeafBit) != 0u;
      const uint32_t index = cur & ~kLeafBit;
      // The four loads are written as instructions: left to the compiler they are split by use (the leaf branch
      // needs 36 of the 64 bytes), narrowed and partly sunk into the branches -- five to seven loads again.
      // Loads, the LDS read of the stack entry the lane falls back to, and the ONE wait for all of them are a single asm
      // statement (round 4).  Until then the loads and the wait were separate statements around compiler-issued code:
      // the compiler believes a hand-issued load complete the moment it is issued and is free to spill or move a
      // destination register in between (nothing it does shows up in its wait bookkeeping either: a build of round 2
      // waited between the loads, -17 %).  As one statement there is no "in between"; early-clobber outputs keep the
      // addresses alive until the last load is out.
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      u32x4 w0, w1, w2, w3;
      // (the "fourth quarter" of a 48-byte triangle record would be the head of the next record: its last 16 bytes
      // are requested twice instead; a 64-byte record is simply read whole)
      constexpr bool kLeaf48 = kTriVec4 == 3u;
      const char* rec = is_leaf ? reinterpret_cast<const char*>(tris) + (16u * kTriVec4) * (size_t)index
                                : reinterpret_cast<const char*>(sc.cur.bvh4q) + 64u * (size_t)index;
      const char* rec3 = rec + (kLeaf48 && is_leaf ? 32 : 48);
      const int top = sp - 1;  // (peek(): the top entry without removing it)
      const uint32_t below_addr = (uint32_t)(uintptr_t)(stack + min(max(top, 0), lds_cap - 1) * kWave);
      uint32_t below;
      asm volatile(
          "global_load_dwordx4 %0, %5, off\n\t"
          "global_load_dwordx4 %1, %5, off offset:16\n\t"
          "global_load_dwordx4 %2, %5, off offset:32\n\t"
          "global_load_dwordx4 %3, %6, off\n\t"
          "ds_read_b32 %4, %7\n\t"
          "s_waitcnt vmcnt(0) lgkmcnt(0)"
          : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(below)
          : "v"(rec), "v"(rec3), "v"(below_addr)
          : "memory");
      if (__builtin_expect(top >= lds_cap, 0)) below = sc.spill[(size_t)(top - lds_cap) * sc.spill_stride + gid].x;
      below = top >= sbase ? below : kNoChild;
      const uint4 q0 = make_uint4(w0.x, w0.y, w0.z, w0.w), q1 = make_uint4(w1.x, w1.y, w1.z, w1.w);
      const uint4 q2 = make_uint4(w2.x, w2.y, w2.z, w2.w), q3 = make_uint4(w3.x, w3.y, w3.z, w3.w);
      if (!is_leaf) {
        if (kCount) ++tally.nodes;
        // 64-byte node: origin + power-of-two grid steps + 8-bit plane coordinates (Wide4Accel::nodes_q).  A plane
        // is origin + q * step, so its slab term is fma(q, step / d, fma(origin, 1/d, -o/d -+ tol)): two terms per
        // axis and node, one fma per plane.  The quantised boxes contain the exact ones, so the walk stays
        // conservative; the exact tests of the winner use the exact parent box (leaf_parent) as before.
        const f3 org = mk3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
        const f3 ax = mk3(__uint_as_float(q0.w) * inv.x, __uint_as_float(q2.z) * inv.y, __uint_as_float(q2.w) * inv.z);
        const f3 bn = mk3(__builtin_fmaf(org.x, inv.x, oin.x), __builtin_fmaf(org.y, inv.y, oin.y), __builtin_fmaf(org.z, inv.z, oin.z));
        const f3 bf = mk3(__builtin_fmaf(org.x, inv.x, oif.x), __builtin_fmaf(org.y, inv.y, oif.y), __builtin_fmaf(org.z, inv.z, oif.z));
        const uint32_t nqx = neg_x ? q1.w : q1.x, fqx = neg_x ? q1.x : q1.w;
        const uint32_t nqy = neg_y ? q2.x : q1.y, fqy = neg_y ? q1.y : q2.x;
        const uint32_t nqz = neg_z ? q2.y : q1.z, fqz = neg_z ? q1.z : q2.y;
        float key[4];
        uint32_t ref[4] = {q3.x, q3.y, q3.z, q3.w};
        // tn is a lower bound of the true entry distance and tf an upper bound of the true exit distance (the
        // tolerance is inside oin / oif), so the child can be skipped when the interval [max(tn, 0), min(tf,
        // limit)] is empty: box missed, box behind the origin (every t in it < 0 < t_min), or box beyond the
        // closest hit so far (limit carries a 0.1 % margin; ties at equal t start no farther than the hit).
        // (An unused slot carries an inside-out box and needs no test of its own, see Collapse::quantise.)
        // (Measured dead end: the 24 plane FMAs as 12 v_pk_fma_f32 -- clean code, no pair-forming moves -- run 1 %
        // SLOWER; packed f32 does not issue faster than two plain FMAs on gfx950, MI355X_MICROARCH.md constants table.)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float cnx = (float)((nqx >> (8 * c)) & 0xffu), cny = (float)((nqy >> (8 * c)) & 0xffu), cnz = (float)((nqz >> (8 * c)) & 0xffu);
          const float cfx = (float)((fqx >> (8 * c)) & 0xffu), cfy = (float)((fqy >> (8 * c)) & 0xffu), cfz = (float)((fqz >> (8 * c)) & 0xffu);
          const float tn = fmaxf(fmaxf(fmaxf(__builtin_fmaf(cnx, ax.x, bn.x), __builtin_fmaf(cny, ax.y, bn.y)),
                                       __builtin_fmaf(cnz, ax.z, bn.z)), 0.0f);
          const float tf = fminf(fminf(fminf(__builtin_fmaf(cfx, ax.x, bf.x), __builtin_fmaf(cfy, ax.y, bf.y)),
                                       __builtin_fmaf(cfz, ax.z, bf.z)), limit);
          if (kCount && ref[c] != sc.cur.dummy_ref) { ++tally.boxes; ++ray_boxes; }
          key[c] = tn <= tf ? tn : __builtin_inff();
        }
        // children nearest first: sort the four (key, ref) pairs, 5 compare-exchanges
        auto cx = [&](int a, int b) {
          const bool sw = key[b] < key[a];
          const float ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b];
          const uint32_t ra = sw ? ref[b] : ref[a], rb = sw ? ref[a] : ref[b];
          key[a] = ka;
          key[b] = kb;
          ref[a] = ra;
          ref[b] = rb;
        };
        cx(0, 1);
        cx(2, 3);
        cx(0, 2);
        cx(1, 3);
#if PT_FULL_SORT
        cx(1, 2);
#endif
        // the others go on the stack, farthest first
        if (__builtin_expect(sp + 3 <= lds_cap, 1)) {
          // room for all three in LDS: write unconditionally, advance only past the ones that count (a slot that
          // does not count is overwritten by the next write or stays above the top)
          stack[sp * kWave] = ref[3];
          sp += key[3] < __builtin_inff() ? 1 : 0;
          stack[sp * kWave] = ref[2];
          sp += key[2] < __builtin_inff() ? 1 : 0;
          stack[sp * kWave] = ref[1];
          sp += key[1] < __builtin_inff() ? 1 : 0;
        } else {
          if (key[3] < __builtin_inff()) push(ref[3]);
          if (key[2] < __builtin_inff()) push(ref[2]);
          if (key[1] < __builtin_inff()) push(ref[1]);
        }
        if (key[0] < __builtin_inff()) {
          cur = ref[0];
        } else {  // nothing was pushed: `below` is still the top
          cur = below;
          sp = max(sp - 1, sbase);
        }
      } else {
        // ray_triangle_intersection_test (intersections.cuh:49-85) on the precomputed world-space edges
        if (kCount) ++tally.tris;
        const f3 p0 = mk3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
        const f3 e1 = mk3(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
        const f3 e2 = mk3(__uint_as_float(q1.z), __uint_as_float(q1.w), __uint_as_float(q2.x));
#if PT_FLAT_TRI
        // the same operations in the same order as the nested form below, evaluated unconditionally (a lane whose
        // test fails early computes garbage that is never looked at: with a dozen lanes per wavefront in this
        // branch some lane reaches every stage anyway, so the early outs only cost their branches)
        const f3 h = cross(rd, e2);
        const float a = dot(e1, h);
        const f3 sv = ro - p0;
        const float f = 1.0f / a;
        const float u = f * dot(sv, h);
        const f3 qv = cross(sv, e1);
        const float w = f * dot(rd, qv);
        const float t = f * dot(e2, qv);
        const bool hit = !(a > -0.0000001f && a < 0.0000001f) & !(u < 0.0f || u > 1.0f) & !(w < 0.0f || u + w > 1.0f) &
                         !(t < tmin) & (t < best_t || (t == best_t && (int)index > best_k));
        best_t = hit ? t : best_t;
        best_k = hit ? (int)index : best_k;
        limit = hit ? scale * t * 1.001f : limit;
#else
        const f3 h = cross(rd, e2);
        const float a = dot(e1, h);
        const f3 sv = ro - p0;
        if (!(a > -0.0000001f && a < 0.0000001f)) {
          const float f = 1.0f / a;
          const float u = f * dot(sv, h);
          if (!(u < 0.0f || u > 1.0f)) {
            const f3 qv = cross(sv, e1);
            const float w = f * dot(rd, qv);
            if (!(w < 0.0f || u + w > 1.0f)) {
              const float t = f * dot(e2, qv);
              if (!(t < tmin) && (t < best_t || (t == best_t && (int)index > best_k))) {
                best_t = t;
                best_k = (int)index;
                limit = scale * t * 1.001f;
              }
            }
          }
        }
#endif
        cur = below;
        sp = max(sp - 1, sbase);
      }
      if (cur == kNoChild) {
        active = false;
        pending = true;
      }
    }
  };

  for (;;) {
    const uint64_t idle_mask = __ballot(!active);
    const uint32_t idle = (uint32_t)__popcll(idle_mask);
    const bool more = priv_next < priv_end || !feed.exhausted();
#ifdef PT_TAILPROF
    ++tp_iters;
#endif
    if (!more) break;  // nothing left to fetch: the lanes still walking finish in the second loop
    if (more && (idle == (uint32_t)kWave || idle >= sc.refill_lanes)) {
      if (pending) {
        finalize();  // (no ray is shared between lanes before the second loop)
        pending = false;
      }
      if (priv_next >= priv_end && !feed.acquire(priv_next, priv_end)) priv_next = priv_end = 0u;
      const uint32_t mine = priv_next + rank_below(idle_mask);
      const uint32_t range_end = priv_end;
      priv_next = min(priv_end, priv_next + idle);
      if (!active && mine < range_end) {
        slot = order ? order[mine] : mine;  // (k_sort_octant: the same rays, picked up in a more coherent order)
        const float4 o4 = ldnt(&paths.o4[slot]);
        const float4 d4 = ldnt(&paths.d4[slot]);
        // entry points (DBeam): the tile's four boxes, requested together with the ray -- at bounce 0 the slot says which
        // pixel the ray belongs to, so the address does not wait for the ray (one round trip for both; the first version
        // took the pixel from the loaded ray and paid a second one, in front of every lane of the wavefront)
        float4 eb[kBeam ? 8 : 1];
        if (kBeam) {
          const uint32_t frame = slot / bi.stride;
          const uint32_t pixel = band_pixel(sc.beam.band, slot - frame * bi.stride);
          const uint32_t py = pixel / sc.beam.width, px = pixel - py * sc.beam.width;
          const float4* e = sc.beam.entries + ((size_t)sc.beam.beam_of[frame] * sc.beam.tiles + (size_t)(py / kBeamTile) * sc.beam.tiles_x + px / kBeamTile) * (2u * kBeamEntries);
#pragma unroll
          for (int k = 0; k < 8; ++k) eb[k] = e[k];
        }
        ro = xyz(o4);
        rd = xyz(d4);
        tmin = (__float_as_uint(o4.w) >> 31) ? 1e-5f : 1e-4f;
        float t_in = FLT_MAX;
        if (!kFirst) {
          const float carried = ldnt(&hits.tp[slot]).x;
          if (carried >= 0.0f) t_in = carried;
        }
        bool go = sc.cur.bvh_node_count != 0u;
        bool wrote = false;
        if (go) {
          // inverse_transform_ray (transform.hpp:51-58) for the walk only: the walk has to be conservative, not
          // exact, so the normalisation and the reciprocals are the hardware approximations (1 ulp) and the
          // error bound below covers them; everything that decides the result is recomputed exactly in finalize
          const f3 v = xform_vector(obj->inv_m, rd);
          const float len2 = dot(v, v);
          const float rlen = __builtin_amdgcn_rsqf(len2);
          scale = len2 * rlen;
          const f3 od = v * rlen;
          // the origin in object space for the walk: the reference's (M^-1 (o,1)).xyz / w with the division by w
          // (1 for an affine transform) as a reciprocal -- finalize recomputes the exact one where it decides
          const f4 ow = mul(obj->inv_m, ro.x, ro.y, ro.z, 1.0f);
          const f3 oo_walk = mk3(ow.x, ow.y, ow.z) * __builtin_amdgcn_rcpf(ow.w);
          inv = mk3(__builtin_amdgcn_rcpf(od.x), __builtin_amdgcn_rcpf(od.y), __builtin_amdgcn_rcpf(od.z));
          // Slab form t = fma(b, 1/d, -o/d).  Against the reference's (b - o)/d (d normalised with IEEE sqrt and
          // divide) it is off by at most ~1e-6 of |b/d| + |o/d| per axis (rsq, rcp: 1 ulp each, three roundings);
          // four times that bound (|b| <= the root box) is folded into the two origin terms so the near side
          // can only move nearer and the far side farther.
          const f3 oi = mk3(-(oo_walk.x * inv.x), -(oo_walk.y * inv.y), -(oo_walk.z * inv.z));
          const float bx = fmaxf(fabsf(sc.cur.root_min[0]), fabsf(sc.cur.root_max[0]));
          const float by = fmaxf(fabsf(sc.cur.root_min[1]), fabsf(sc.cur.root_max[1]));
          const float bz = fmaxf(fabsf(sc.cur.root_min[2]), fabsf(sc.cur.root_max[2]));
          const f3 tol = mk3(4e-6f * (fabsf(oi.x) + bx * fabsf(inv.x)) + 1e-30f,
                             4e-6f * (fabsf(oi.y) + by * fabsf(inv.y)) + 1e-30f,
                             4e-6f * (fabsf(oi.z) + bz * fabsf(inv.z)) + 1e-30f);
          if (__builtin_expect(!(finite_f(inv.x) && finite_f(inv.y) && finite_f(inv.z) && finite_f(tol.x + tol.y + tol.z)) ||
                               sc.force_slow == 1u, 0)) {
            // degenerate direction (0/0 or overflow in the slab terms voids the error bound): set aside for
            // the launch's epilogue (redo_slow_rays), which takes every box decision with the reference's own test
            set_aside(counters, slow_list, slot);
            wrote = true;
            go = false;
          } else {
            oin = oi - tol;
            oif = oi + tol;
            neg_x = inv.x < 0.0f;
            neg_y = inv.y < 0.0f;
            neg_z = inv.z < 0.0f;
            best_t = t_in;
            best_k = -1;
            limit = scale * best_t * 1.001f;
            cur = sc.cur.bvh4_root;
            sp = sbase = 0;
            ray_boxes = 0u;
            if (kBeam) {
              // entry points (DBeam): the four boxes of the ray's tile, tested like the children of one node -- same
              // conservative slab arithmetic, boxes in the object's space -- and entered nearest first.  Everything the
              // tile's frustum cannot reach was left out by k_beam; what the walk finds is verified exactly as always.
              float key[4];
              uint32_t ref[4];
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const float4 lo = eb[2 * c], hi = eb[2 * c + 1];
                const float tn = fmaxf(fmaxf(fmaxf(__builtin_fmaf(neg_x ? hi.x : lo.x, inv.x, oin.x), __builtin_fmaf(neg_y ? hi.y : lo.y, inv.y, oin.y)),
                                             __builtin_fmaf(neg_z ? hi.z : lo.z, inv.z, oin.z)), 0.0f);
                const float tf = fminf(fminf(fminf(__builtin_fmaf(neg_x ? lo.x : hi.x, inv.x, oif.x), __builtin_fmaf(neg_y ? lo.y : hi.y, inv.y, oif.y)),
                                             __builtin_fmaf(neg_z ? lo.z : hi.z, inv.z, oif.z)), limit);
                key[c] = tn <= tf ? tn : __builtin_inff();
                ref[c] = __float_as_uint(lo.w);
                if (kCount && key[c] < __builtin_inff()) { ++tally.boxes; ++ray_boxes; }
              }
              auto cx = [&](int a, int b) {
                const bool sw = key[b] < key[a];
                const float ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b];
                const uint32_t ra = sw ? ref[b] : ref[a], rb = sw ? ref[a] : ref[b];
                key[a] = ka;
                key[b] = kb;
                ref[a] = ra;
                ref[b] = rb;
              };
              cx(0, 1);
              cx(2, 3);
              cx(0, 2);
              cx(1, 3);
              cx(1, 2);
              if (key[3] < __builtin_inff()) push(ref[3]);
              if (key[2] < __builtin_inff()) push(ref[2]);
              if (key[1] < __builtin_inff()) push(ref[1]);
              cur = ref[0];
              go = key[0] < __builtin_inff();  // (no entry hit: the ray passes this object)
            }
          }
        }
        if (go) active = true;
        else if (kFirst && !wrote) stnt(&hits.tp[slot], make_float4(-1.0f, 0.f, 0.f, 0.f));
      }
    }
    if (__ballot(active) == 0ull) continue;
    step();
  }

  // ---- the end of the launch: no rays left to fetch.  Lanes fall idle one by one; idle lanes take over parts of the
  // busy lanes' walks (split).  A wavefront runs with few others here, so an iteration costs its dependent round trips:
  // a retire is two of its own (the winner's parent box and normal, then the stores the next wait sits out) and a split
  // is a chain of LDS round trips -- measured 1600 + 1600 cycles beside a 2400-cycle step when both ran every iteration
  // (profiles/r04_tailprof_before_*.txt) -- so finished lanes are retired a few at a time and until then are no takers.
#ifdef PT_TAILPROF
  tp_exhausted = wall_clock64();
  tp_iters_exh = tp_iters;
  tp_lanes_exh = (uint32_t)__popcll(__ballot(active));
#endif
  for (;;) {
    const uint32_t idle = (uint32_t)__popcll(__ballot(!active));
#ifdef PT_TAILPROF
    ++tp_iters;
#endif
    if (sc.split_idle != 0u && idle >= sc.split_idle && ++since_split >= PT_SPLIT_EVERY) {
      since_split = 0u;
#ifdef PT_TAILPROF
      const unsigned long long c0 = clock64();
#endif
      const uint32_t waiting = (uint32_t)__popcll(__ballot(pending));
      if (waiting != 0u && (waiting >= PT_RETIRE_LANES || ++since_retire >= PT_RETIRE_EVERY || idle == (uint32_t)kWave)) {
        since_retire = 0u;
        if (pending) {
          retire();
          pending = false;
        }
      }
#ifdef PT_TAILPROF
      const unsigned long long c1 = clock64();
#endif
      split();
#ifdef PT_TAILPROF
      ++tp_splits;
      tp_c_retire += c1 - c0;
      tp_c_split += clock64() - c1;
#endif
    }
    if (__ballot(active) == 0ull) {
      if (pending) {
        retire();
        pending = false;
      }
      break;
    }
#ifdef PT_TAILPROF
    const unsigned long long c2 = clock64();
    tp_lanes_tail += (unsigned long long)__popcll(__ballot(active));
#endif
    step();
#ifdef PT_TAILPROF
    tp_c_step += clock64() - c2;
#endif
  }
#ifdef PT_TAILPROF
  if (threadIdx.x == 0u && blockIdx.x < 8192u && bounce < 16) {
    unsigned long long* tp = g_tailprof[bounce][blockIdx.x];
    tp[0] = tp_start;
    tp[1] = tp_exhausted;
    tp[2] = wall_clock64();
    tp[3] = ((unsigned long long)tp_iters << 40) | ((unsigned long long)tp_iters_exh << 16) | ((unsigned long long)tp_lanes_exh << 8) | min(tp_splits, 255u);
    tp[4] = tp_c_retire;
    tp[5] = tp_c_split;
    tp[6] = tp_c_step;
    tp[7] = tp_lanes_tail;
  }
#endif
  if (flags) atomicOr(&counters->flags, flags);
  if (kCount) flush_tally(tally, counters, bounce, false);
}

// Rays a persistent traversal launch set aside (a direction with a zero / subnormal component, or a winner whose
// parent box the ray only grazes): redone with EXACT box decisions -- the culled near-first walk
// (mesh_closest_wide: every inner box decided like the reference's ray_aabb, also for NaN / infinite slab terms),
// which returns the reference's hit in ~50 box tests instead of the ~125 (worst case thousands) of the reference's
// own order.  Run by the LAST wavefront of the traversal launch to finish (k_traverse4's epilogue): the list is
// almost always empty, and a separate one-wavefront launch for it was a bubble on the stream every bounce (it
// waited for a wavefront slot behind the other stream's persistent wavefronts: 10 % of the kernel time of round 1).
// Its traversal stack lives in global memory (DScene::slow_stack, [depth][lane]): this code is off the fast path.
template <bool kFirst>
__device__ __forceinline__ void redo_slow_rays(const DScene& sc, uint32_t obj_index, const DPaths& paths, const DHits& hits,
                                            const uint32_t* slow_list, uint32_t count, DeviceCounters* counters)
{
  const DObject* obj = sc.objects + obj_index;
  const uint32_t mat = sc.object_material[obj_index];
  const uint32_t tri_base = sc.object_tri_base[obj_index];
  uint32_t flags = 0u;
  for (uint32_t i = threadIdx.x; i < count; i += kWave) {
    const uint32_t slot = __hip_atomic_load(&slow_list[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Ray ray = load_ray(paths, slot);
    if (!kFirst) {
      const float carried = ldnt(&hits.tp[slot]).x;
      if (carried >= 0.0f) ray.tmax = carried;
    }
    float best_t = ray.tmax;
    int best_k = -1;
    Tally unused;
    // the object's world box first (path_tracer.cu:84): the persistent kernel tests it only for its winners
    if (ray_aabb(ray.o, ray.d, ld3(obj->bmin), ld3(obj->bmax)))
      mesh_closest_wide<false>(ray, sc, sc.cur, obj, tri_base, best_t, best_k, sc.slow_stack + threadIdx.x, flags, unused);
    if (best_k >= 0) {
      const float4 tc = sc.tris[kTriVec4 * ((size_t)tri_base + (uint32_t)best_k) + 2u];
      const f3 outward = mk3(tc.y, tc.z, tc.w);
      const f3 p = ray.o + ray.d * best_t;
      const uint32_t side = dot(ray.d, outward) < 0.0f ? 0u : 1u;
      const f3 nn = side == 0u ? outward : -outward;
      stnt(&hits.tp[slot], make_float4(best_t, p.x, p.y, p.z));
      stnt(&hits.nm[slot], make_float4(nn.x, nn.y, nn.z, __uint_as_float(mat | (side << 31))));
    } else if (kFirst) {
      stnt(&hits.tp[slot], make_float4(-1.0f, 0.f, 0.f, 0.f));
    }
  }
  if (flags) atomicOr(&counters->flags, flags);
}

// Entry points for primary rays (DBeam), one thread per tile and camera.  The tile's frustum: four planes through the
// camera, each spanned by two neighbouring corner rays of the tile's pixel rectangle (generate_ray at the rectangle's
// corners, a twentieth of a pixel outside: the jitter keeps a ray of pixel x inside [x, x + 1]); everything in the space
// of the launch's object (the walk's space).  A box is out when its corner farthest along a plane's normal is still
// outside that plane by more than the margin; the pyramid is the forward one only, so what lies behind the camera is out
// -- the reference's line-without-range box test would pass such boxes, but no triangle in them can be hit at t >= t_min.
// Frontier: the root; as long as there is room for them, the largest inner entry is replaced by those of its (quantised)
// children the frustum reaches.  Quantised boxes contain the exact ones and are handed on a millionth larger: the rays
// test them with the walk's own tolerant slabs, and the winner is verified exactly (finalize) as for any other walk.
__global__ __launch_bounds__(64) void k_beam(DScene sc, uint32_t obj_index, DCameras cams, DBeam geo, uint32_t nbeam, uint32_t node_count4, float4* out)
{
  const uint32_t id = blockIdx.x * 64u + threadIdx.x;
  if (id >= nbeam * geo.tiles) return;
  const uint32_t beam = id / geo.tiles, tile = id - beam * geo.tiles;
  const uint32_t ty = tile / geo.tiles_x, tx = tile - ty * geo.tiles_x;
  const DCamera& cam = cams.c[geo.beam_of[beam]];  // (here: the camera OF the beam, filled in by launch_beam)
  const DObject* obj = sc.objects + obj_index;
  const float x0 = (float)(tx * kBeamTile) - 0.05f, x1 = (float)((tx + 1u) * kBeamTile) + 0.05f;
  const float y0 = (float)(ty * kBeamTile) - 0.05f, y1 = (float)((ty + 1u) * kBeamTile) + 0.05f;
  f3 o, d00, d10, d01, d11, dc;
  generate_ray(cam, x0, y0, o, d00);
  generate_ray(cam, x1, y0, o, d10);
  generate_ray(cam, x0, y1, o, d01);
  generate_ray(cam, x1, y1, o, d11);
  generate_ray(cam, 0.5f * (x0 + x1), 0.5f * (y0 + y1), o, dc);
  const beam_rules::Frustum fr = beam_rules::make_frustum(xform_point(obj->inv_m, o), xform_vector(obj->inv_m, d00), xform_vector(obj->inv_m, d10),
                                                          xform_vector(obj->inv_m, d01), xform_vector(obj->inv_m, d11), xform_vector(obj->inv_m, dc));
  f3 lo4[4], hi4[4];
  uint32_t ref4[4];
  int n = 0;
  if (sc.cur.bvh_node_count != 0u)
    n = beam_rules::tile_entries(reinterpret_cast<const uint32_t*>(sc.cur.bvh4q), node_count4, sc.cur.bvh4_root, ld3(sc.cur.root_min), ld3(sc.cur.root_max), fr,
                                 lo4, hi4, ref4);
  float4* e = out + (size_t)id * (2u * kBeamEntries);
#pragma unroll
  for (int k = 0; k < (int)kBeamEntries; ++k) {
    if (k < n) {
      e[2 * k] = make_float4(lo4[k].x, lo4[k].y, lo4[k].z, __uint_as_float(ref4[k]));
      e[2 * k + 1] = make_float4(hi4[k].x, hi4[k].y, hi4[k].z, 0.0f);
    } else {  // nothing: a box no ray is inside of
      e[2 * k] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __uint_as_float(kNoChild));
      e[2 * k + 1] = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), 0.0f);
    }
  }
}

template <bool kCount, bool kFirst, bool kBeam = false>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(PT_T4_WAVES, PT_T4_WAVES)))
void k_traverse4(DScene sc, uint32_t obj_index, DPaths paths, DHits hits, int bounce, int work_slot,
                 DeviceCounters* counters, uint32_t* slow_list, const uint32_t* order, DBatchInfo bi, int listed)
{
  traverse4_walk<kCount, kFirst, kBeam>(sc, obj_index, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed != 0);
  // Epilogue: every wavefront signs off; the last one redoes the rays that were set aside.  The list entries were
  // written with agent-scope atomic stores; waiting for this wavefront's own stores before the sign-off and reading
  // the list with agent-scope loads orders them without a full L2 write-back per wavefront.
  uint32_t prev = 0u;
  if (threadIdx.x == 0u) {
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): this wavefront's list entries have landed
    prev = __hip_atomic_fetch_add(&counters->waves_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  prev = (uint32_t)__builtin_amdgcn_readfirstlane((int)prev);
  if (prev + 1u != gridDim.x) return;
  const uint32_t count = __hip_atomic_load(&counters->slow_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (count != 0u) redo_slow_rays<kFirst>(sc, obj_index, paths, hits, slow_list, count, counters);
  launch_epilogue(counters, bounce, work_slot, count, bi, listed != 0);
}

#include "pt_traverse4m.inc"

// A run of sphere objects that does not end the object list (the spheres in front of a mesh), continuing from /
// handing on the closest hit in the hit record.  (The run that ENDS the list -- or is the whole list -- is part of
// the kernel that ends the bounce.)  (Taking the sphere runs into the traversal kernel instead was tried in round 2: inlined or as a
// call, their temporaries pushed loop-carried state of the walk into scratch, with reloads inside its hot loop.)
// kFilter: the launch is followed by a traversal launch over the mesh objects [filt_begin, filt_end).  A ray that SURELY
// misses the world boxes of all of them (the same test with the same margin by which that launch skips an instance,
// traverse4m_walk::begin_object), or whose boxes all start beyond the closest hit so far, has nothing to do there: only
// the others are put on the work list (batch-global slots, DeviceCounters::list_count per frame; their order is
// irrelevant -- results are written per slot), and the traversal launch fetches its rays through that list.  In the
// Cornell-box scenes most rays of most bounces never come near the meshes.
template <bool kFirst, bool kFilter>
// (at least six wavefronts per SIMD: 80 registers, 3-14 spilled, against 92 and five wavefronts: config 2 +2.7 %; seven: +1.7 %,
// eight (49-62 spilled): +1.1 %; profiles/r04_config2_counters.txt)
#ifndef PT_SPHERES_WAVES
#define PT_SPHERES_WAVES 6
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PT_SPHERES_WAVES, 8))) void k_spheres(DScene sc, uint32_t obj_begin, uint32_t obj_end, DPaths paths, DHits hits,
                                                 int bounce, DeviceCounters* counters, DBatchInfo bi, uint32_t filt_begin,
                                                 uint32_t filt_end, uint32_t* worklist, DTileScan scan, uint32_t tile_stride)
{
  const uint32_t frame = blockIdx.x % bi.count;  // see DBatchInfo; frame-fastest as in k_raygen
  paths.o4 += (size_t)frame * bi.stride;
  paths.d4 += (size_t)frame * bi.stride;
  hits.tp += (size_t)frame * bi.stride;
  hits.nm += (size_t)frame * bi.stride;
  counters += frame;
  scan.desc += (size_t)frame * tile_stride;
  const uint32_t n = counters->live[bounce];
  const uint32_t tiles = (n + 256u * kListPer - 1u) / (256u * kListPer);
  if (blockIdx.x / bi.count >= tiles) {
    if (kFilter && tiles == 0u && blockIdx.x / bi.count == 0u && threadIdx.x == 0u) counters->list_count = 0u;  // nothing alive: an empty list
    return;
  }
  const uint32_t tile = kFilter ? list_tile(counters, tiles) : blockIdx.x / bi.count;
  const uint32_t block_first = tile * (256u * kListPer);  // a workgroup tests kListPer x 256 consecutive slots
  uint32_t may_mask = 0u;
  // (sphere_run_lanes is for the run that ends the list: here, in front of a mesh, the spheres are typically the walls of a
  // room -- every ray hits every one of them, there is little to rule out, and sphere_segment shares the inverse
  // transform's normalised direction among them: measured 907 us against 1114 for the per-lane form, config 2)
  const bool lanes_run = PT_SPHERE_LANES_LEADING && sc.lanes_run != 0u;
  const bool fold_run = sc.fold_run != 0u;
#pragma unroll 1
  for (int j = 0; j < kListPer; ++j) {
    const uint32_t s = block_first + (uint32_t)j * 256u + threadIdx.x;
    if (s >= n) continue;
    Ray ray = load_ray(paths, s);
    if (!kFirst) {
      const float carried = ldnt(&hits.tp[s]).x;
      if (carried >= 0.0f) ray.tmax = carried;
    }
    Hit rec;
    bool changed = false;
    if (lanes_run) {
      const float4 ro4[1] = {ldnt(&paths.o4[s])}, rd4[1] = {ldnt(&paths.d4[s])};
      float4 rtp[1] = {make_float4(ray.tmax < FLT_MAX ? ray.tmax : -1.0f, 0.f, 0.f, 0.f)}, rnm[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
      changed = sphere_run_lanes<1>(sc, obj_begin, obj_end, ro4, rd4, rtp, rnm, 1u) != 0u;
      if (changed) {
        stnt(&hits.tp[s], rtp[0]);
        stnt(&hits.nm[s], rnm[0]);
        ray.tmax = rtp[0].x;
      }
    } else {
      if (fold_run) sphere_fold(sc, obj_begin, obj_end, ray, rec, changed);
      else sphere_segment<true>(sc, obj_begin, obj_end, ray, rec, changed);
      if (changed) store_hit(hits, s, rec);
    }
    if (!changed && kFirst) stnt(&hits.tp[s], make_float4(-1.0f, 0.f, 0.f, 0.f));
    if (kFilter && may_hit_boxes(sc.objects, filt_begin, filt_end, ray.o, ray.d, ray.tmax)) may_mask |= 1u << j;
  }
  if (kFilter) list_rays(may_mask, worklist, counters, (size_t)frame * bi.stride, tile, tiles, scan);
}

// The end of a bounce's closest-hit stage: the sphere run that ends the object list (if any) and the live count of
// every 64-slot chunk (ballot / popcount of "the hit record holds a hit"), for the compaction scan.
// kSpheres: objects [obj_begin, obj_end) are tested; kFirst: nothing has written the hit record in this bounce yet.
// 256-thread workgroups: one wavefront per SIMD fits beside the other stream's persistent traversal wavefronts as soon
// as one of those has left (1024-thread workgroups wait until four per SIMD have: measured 15 % slower end to end,
// together with a scan fused in behind a "last workgroup" sign-off).
template <bool kSpheres, bool kFirst>
__global__ __launch_bounds__(256) void k_tail_count(DScene sc, uint32_t obj_begin, uint32_t obj_end, DPaths paths, DHits hits,
                                                    int bounce, uint32_t* chunk_counts, DeviceCounters* counters, DBatchInfo bi)
{
  const uint32_t frame = blockIdx.y;  // see DBatchInfo
  paths.o4 += (size_t)frame * bi.stride;
  paths.d4 += (size_t)frame * bi.stride;
  hits.tp += (size_t)frame * bi.stride;
  hits.nm += (size_t)frame * bi.stride;
  chunk_counts += (size_t)frame * bi.chunk_stride;
  counters += frame;
  const uint32_t n = counters->live[bounce];
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  // a wavefront beyond the live range owns no chunk: k_scan reads ceil(n / 64) entries, and in a batch the next
  // entries belong to the next frame
  if ((s & ~63u) >= n) return;
  bool hit = false;
  if (s < n) {
    float t_so_far = -1.0f;
    if (!kFirst) t_so_far = ldnt(&hits.tp[s]).x;
    hit = t_so_far >= 0.0f;
    if (kSpheres) {
      Ray ray = load_ray(paths, s);
      if (hit) ray.tmax = t_so_far;
      Hit rec;
      bool changed = false;
      sphere_segment(sc, obj_begin, obj_end, ray, rec, changed);
      if (changed) {
        store_hit(hits, s, rec);
        hit = true;
      } else if (kFirst) {
        stnt(&hits.tp[s], make_float4(-1.0f, 0.f, 0.f, 0.f));
      }
    }
  }
  const uint64_t live = __ballot(hit);
  if ((threadIdx.x & 63u) == 0u) chunk_counts[s / kChunk] = (uint32_t)__popcll(live);
}

// Exclusive scan of the per-chunk live counts (one workgroup; <= ~32k chunks at 1080p).
// Writes live[bounce+1] (0 after the last bounce: nothing survives the cap) and the ray counter.
__global__ __launch_bounds__(1024) void k_scan(int bounce, int last_bounce, const uint32_t* chunk_counts,
                                               uint32_t* chunk_offsets, DeviceCounters* counters, DBatchInfo bi)
{
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  const uint32_t frame = blockIdx.x;  // one workgroup per frame of the batch
  chunk_counts += (size_t)frame * bi.chunk_stride;
  chunk_offsets += (size_t)frame * bi.chunk_stride;
  counters += frame;
  const uint32_t n = counters->live[bounce];
  const uint32_t chunks = (n + kChunk - 1u) / kChunk;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0u) s_carry = 0u;
  __syncthreads();
  for (uint32_t base = 0; base < chunks; base += 1024u) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < chunks ? chunk_counts[i] : 0u;
    // inclusive scan inside the wavefront
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off, 64);
      if (lane >= (uint32_t)off) x += y;
    }
    if (lane == 63u) s_wave[wave] = x;
    __syncthreads();
    uint32_t wave_prefix = 0u;
    for (uint32_t w = 0; w < wave; ++w) wave_prefix += s_wave[w];
    const uint32_t carry = s_carry;
    if (i < chunks) chunk_offsets[i] = carry + wave_prefix + x - v;
    __syncthreads();
    if (threadIdx.x == 1023u) s_carry = carry + wave_prefix + x;
    __syncthreads();
  }
  if (threadIdx.x == 0u) {
    counters->live[bounce + 1] = last_bounce ? 0u : s_carry;
    counters->rays_total += n;
    counters->paths[bounce] += n;
  }
}

// material_kernel (path_tracer.cu:292-315) + the stable compaction scatter + the final gather of
// every path that ends at this bounce.
// staged: `fb` is the slot's staging buffer (one sample per frame of the batch, plain stores); k_accumulate then
// folds the staged samples into the real framebuffer in iteration order.  Otherwise the running mean goes
// straight into `fb`.
__global__ __launch_bounds__(256) void k_shade(DScene sc, DPaths in, DPaths out, DHits hits, int staged, int bounce,
                                               int last_bounce, const uint32_t* slot_base, const uint32_t* chunk_offsets,
                                               DFrame fb, DBand band, DeviceCounters* counters, uint8_t* octs, DBatchInfo bi)
{
  const uint32_t frame = blockIdx.y;  // see DBatchInfo
  const uint32_t iteration = bi.iteration[frame];
  const uint32_t acc_iteration = staged ? 0u : iteration;
  if (octs) octs += (size_t)frame * bi.stride;
  in.o4 += (size_t)frame * bi.stride;
  in.d4 += (size_t)frame * bi.stride;
  in.t4 += (size_t)frame * bi.stride;
  out.o4 += (size_t)frame * bi.stride;
  out.d4 += (size_t)frame * bi.stride;
  out.t4 += (size_t)frame * bi.stride;
  hits.tp += (size_t)frame * bi.stride;
  hits.nm += (size_t)frame * bi.stride;
  chunk_offsets += (size_t)frame * bi.chunk_stride;
  if (staged) {
    fb.color4 += (size_t)frame * bi.stride;
    fb.nd4 += (size_t)frame * bi.stride;
  }
  counters += frame;
  const uint32_t n = counters->live[bounce];
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (blockIdx.x * 256u >= n) return;
  const bool active = s < n;
  bool survives = false;
  f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 0), color = mk3(0, 0, 0);
  uint32_t pixbits = 0u;
  if (active) {
    const float4 o4 = ldnt(&in.o4[s]);
    const float4 d4 = ldnt(&in.d4[s]);
    const float4 t4 = bounce == 0 ? make_float4(1.0f, 1.0f, 1.0f, 0.0f) : ldnt(&in.t4[s]);  // (k_raygen does not write it)
    const float4 tp = ldnt(&hits.tp[s]);
    ro = xyz(o4);
    rd = xyz(d4);
    color = xyz(t4);
    pixbits = __float_as_uint(o4.w);
    const uint32_t pixel = pixbits & 0x7fffffffu;
    const uint32_t local_pixel = band_local(band, pixel);
    bool tmin_flag = (pixbits >> 31) != 0u;

    if (tp.x < 0.0f) {
      // miss: throughput *= sky; the path ends (path_tracer.cu:304-307, 283-289)
      color = color * background(rd);
      if (bounce == 0) accumulate_nd(fb.nd4, local_pixel, acc_iteration, -rd, 1e6f);  // raygen defaults, ray_gen.cu:26-28
      accumulate_color(fb.color4, local_pixel, acc_iteration, color);
    } else {
      const float4 nm = ldnt(&hits.nm[s]);
      const f3 hn = xyz(nm);
      if (bounce == 0) accumulate_nd(fb.nd4, local_pixel, acc_iteration, hn, tp.x);  // path_tracer.cu:308-311
      const uint32_t ms = __float_as_uint(nm.w);
      const DMaterial m = sc.materials[ms & 0x7fffffffu];
      // RNG re-seeded from the global slot index, then discard(bounce) (path_tracer.cu:300-301)
      const uint32_t slot = (slot_base ? *slot_base : 0u) + s;
      Minstd rng;
      rng.seed(path_seed(slot, iteration));
      rng.discard((uint32_t)bounce);
      const f3 hp = mk3(tp.y, tp.z, tp.w);
      evaluate_material(ro, rd, tmin_flag, hp, hn, ms >> 31, m, rng, color);
      if (last_bounce) {
        accumulate_color(fb.color4, local_pixel, acc_iteration, color);  // capped paths deposit raw throughput
      } else {
        survives = true;
        pixbits = pixel | (tmin_flag ? 0x80000000u : 0u);
      }
    }
  }
  const uint64_t live = __ballot(survives);
  if (survives) {
    const uint32_t dst = chunk_offsets[s / kChunk] + rank_below(live);
    stnt(&out.o4[dst], make_float4(ro.x, ro.y, ro.z, __uint_as_float(pixbits)));
    stnt(&out.d4[dst], make_float4(rd.x, rd.y, rd.z, 0.0f));
    stnt(&out.t4[dst], make_float4(color.x, color.y, color.z, 0.0f));
    // direction octant of the new ray, for the coherence sort of the next bounce (k_sort_octant)
    if (octs) octs[dst] = (uint8_t)((rd.x < 0.0f ? 1u : 0u) | (rd.y < 0.0f ? 2u : 0u) | (rd.z < 0.0f ? 4u : 0u));
  }
}

// ------------------------------------------------------------------------------------------------
// the end of a bounce in ONE pass: trailing sphere run + material + stable compaction + final gather
// ------------------------------------------------------------------------------------------------
// k_tail_count -> k_scan -> k_shade read every ray and hit record twice and put a one-workgroup scan between two
// full-width launches.  This kernel does the three jobs in one pass over the slots (path_tracer.cu:292-315 material_kernel,
// :454-457 stable_partition, :317-330 final gather; :78-100 for the sphere run that ends the object list):
//   * a workgroup owns a TILE of 256 x kFuseK consecutive slots (every thread kFuseK of them, 256 apart: coalesced),
//     loads ray + hit, finishes the closest hit (the trailing spheres), and knows from "is there a hit" alone which of
//     its paths survive -- so the tile's survivor count is published after ONE round trip to memory;
//   * the stable offset of the tile = survivors of all tiles before it, found by decoupled look-back over the tile
//     descriptors (aggregate / inclusive prefix, Merrill & Garland): a wavefront reads up to 64 predecessors at once;
//   * tiles are taken in TICKET order (one agent-scope atomic per workgroup on the frame's counter line; blocks are
//     numbered frame-fastest, so neighbouring workgroups of a batch take their tickets on different lines), not in
//     blockIdx order: whoever waits for a tile's descriptor waits for a workgroup that is RUNNING (it has its ticket),
//     whatever else holds the chip's wavefront slots.  Block order is 1-3 % faster (the ticket's round trip sits in
//     front of a workgroup's first load) and is NOT safe: with several streams' kernels on the chip, workgroups of
//     one kernel fill an XCD spinning for a predecessor that waits for a slot on an XCD filled by another kernel's
//     spinners -- seen once in 67 GPU tests x several runs (ten one-frame launches on ten streams), caught by the
//     bounded wait below.  The wait stays bounded all the same: a workgroup that has waited about a second sets
//     kFlagDispatchOrder and ptc_get_stats reports an error instead of an image;
//   * descriptors carry the launch's epoch, so nothing has to be cleared between launches.
// Same arithmetic and same slot order as the three kernels it replaces: images are bit-identical (tests).
// Slots per thread: 2 (round 4; 4 until then).  With four the kernel needs 128 registers and spills 18 of them at four
// wavefronts per SIMD; with two it needs 88, spills nothing and runs five: config 2 +2.2 %, config 3 +0.6 ... 1.2 %;
// one slot (eight wavefronts): -4 ... -6 % (profiles/r04_config2_counters.txt).  Six wavefronts (80 registers) spill 53.
#ifndef PT_FUSE_K
#define PT_FUSE_K 2
#endif

#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 4
#endif
constexpr int kFuseK = PT_FUSE_K;
constexpr uint32_t kFuseTile = 256u * kFuseK;
static_assert(kFuseTile <= 256u * kListPer, "k_raygen / k_spheres scan their work lists on k_shade_fused's tile descriptors");


// exclusive prefix of `tile` (survivors of tiles [0, tile)), by the calling wavefront; every lane returns it
__device__ __forceinline__ uint32_t tile_lookback(const unsigned long long* desc, uint32_t tile, uint32_t epoch, uint32_t* flags)
{
  uint32_t excl = 0u;
  const int lane = (int)lane_id();
  int pos = (int)tile - 1;
  uint32_t spins = 0u;
  while (pos >= 0) {
    const int idx = pos - lane;  // lane 0 = the nearest predecessor
    for (;;) {
      unsigned long long d = 0ull;
      if (idx >= 0) d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool valid = idx >= 0;
      const bool ready = !valid || (uint32_t)(d >> 34) == epoch;
      const bool is_prefix = valid && ready && (d & kDescPrefix) != 0ull;
      const uint64_t rdy = __ballot(ready), pm = __ballot(is_prefix);
      if (pm != 0ull) {
        const int p = __ffsll((unsigned long long)pm) - 1;  // nearest predecessor that knows its inclusive prefix
        const uint64_t need = (p == 63) ? ~0ull : ((1ull << (p + 1)) - 1ull);
        if ((rdy & need) == need) return excl + wave_sum(lane <= p ? (uint32_t)d : 0u);
      } else if (rdy == ~0ull) {
        excl += wave_sum(valid ? (uint32_t)d : 0u);
        break;  // 64 aggregates and no prefix among them: look further back
      }
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 20)) {  // (about a second) a predecessor that never ran: see the kernel's header
        if (lane == 0) atomicOr(flags, kFlagDispatchOrder);
        return excl;
      }
    }
    pos -= kWave;
  }
  return excl;
}

template <bool kSpheres, bool kFirst>
// (occupancy bounds re-measured on the final build: at least 5 or 6 wavefronts per SIMD forces spills, -8 % / -13 % end
// to end; 1 to 3 compile to the same 124 registers)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PT_SHADE_WAVES, 8))) void k_shade_fused(DScene sc, uint32_t obj_begin, uint32_t obj_end, DPaths in, DPaths out, DHits hits,
                                                     int staged, int bounce, int last_bounce, const uint32_t* slot_base,
                                                     unsigned long long* tile_desc, uint32_t tile_stride, uint32_t epoch, DFrame fb,
                                                     DBand band, DeviceCounters* counters, uint8_t* octs, DBatchInfo bi, const uint32_t* list)
{
  __shared__ uint32_t s_excl, s_tile;
  __shared__ uint32_t s_cnt[kFuseK * 4];
  const uint32_t frame = blockIdx.x % bi.count;  // frame-fastest: neighbouring blocks take their tickets on different lines
  const uint32_t iteration = bi.iteration[frame];
  const uint32_t acc_iteration = staged ? 0u : iteration;
  const size_t fo = (size_t)frame * bi.stride;
  if (octs) octs += fo;
  in.o4 += fo;
  in.d4 += fo;
  in.t4 += fo;
  out.o4 += fo;
  out.d4 += fo;
  out.t4 += fo;
  hits.tp += fo;
  hits.nm += fo;
  tile_desc += (size_t)frame * tile_stride;
  if (staged) {
    fb.color4 += fo;
    fb.nd4 += fo;
  }
  counters += frame;
  // list ("filter_rays", bounce 0 of a scene whose whole object list is the bounce's one listed traversal launch): the
  // rays that are not on the launch's work list have been finished by k_raygen, which knew that they hit nothing; this
  // kernel then walks the list (slot order: the survivors land where they would have) instead of all slots
  const uint32_t n_all = counters->live[bounce];
  const uint32_t n = list ? counters->list_count : n_all;
  if (list) list += fo;
  const uint32_t tiles = (n + kFuseTile - 1u) / kFuseTile;
  const uint32_t wave = threadIdx.x >> 6;
  // The grid is sized for a frame of all-live slots; the workgroups the frame has no tile for leave without a ticket:
  // exactly `tiles` tickets are taken per frame.  (Dispatching the empty workgroups costs little: a launch sized by the
  // last batch's live counts, whose workgroups came back for more tiles when there were too few, was slower -- the
  // loop cost 24 spilled registers -- profiles/r03_shade_breakdown.txt.)
  if (blockIdx.x / bi.count >= tiles) {
    if (tiles == 0u && blockIdx.x / bi.count == 0u && threadIdx.x == 0u) {  // nothing alive: nothing follows
      counters->live[bounce + 1] = 0u;
      counters->rays_total += n_all;  // (not zero when k_raygen has finished every ray of the frame, see `list`)
      counters->paths[bounce] += n_all;
    }
    return;
  }
  if (threadIdx.x == 0u) {
    const uint32_t t = __hip_atomic_fetch_add(&counters->shade_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // every tile of the frame is taken once the last ticket is out: the next launch starts from zero
    if (t + 1u == tiles) __hip_atomic_store(&counters->shade_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_tile = t;
  }
  __syncthreads();
  const uint32_t tile = s_tile;

  // ---- phase 1: rays and hits of the tile; the closest hit is final after the trailing sphere run ----
  float4 o4[kFuseK], d4[kFuseK], tp[kFuseK], nm[kFuseK];
  uint32_t have_nm = 0u, hit_mask = 0u;
  uint32_t slot_of[kFuseK];  // position tile * kFuseTile + j * 256 + thread of the walk -> slot (the same unless `list`)
#pragma unroll
  for (int j = 0; j < kFuseK; ++j) {
    const uint32_t at = tile * kFuseTile + (uint32_t)j * 256u + threadIdx.x;
    slot_of[j] = at < n ? (list ? list[at] - (uint32_t)fo : at) : n_all;
  }
#pragma unroll
  for (int j = 0; j < kFuseK; ++j) {
    const uint32_t s = slot_of[j];
    tp[j] = make_float4(-1.0f, 0.f, 0.f, 0.f);
    nm[j] = o4[j] = d4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s < n_all) {
      o4[j] = ldnt(&in.o4[s]);
      d4[j] = ldnt(&in.d4[s]);
      if (!kFirst) tp[j] = ldnt(&hits.tp[s]);
    }
  }
  // the sphere run that ends the object list: every lane with its own candidates, all its slots at once, when the run
  // allows it (sphere_run_lanes); object by object otherwise
  const bool lanes_run = kSpheres && PT_SPHERE_LANES && sc.lanes_run != 0u;
  if (kSpheres && lanes_run) {
    static_assert(kFuseK <= 4, "sphere_run_lanes: at most four slots");
    uint32_t valid = 0u;
#pragma unroll
    for (int j = 0; j < kFuseK; ++j) valid |= slot_of[j] < n_all ? 1u << j : 0u;
    have_nm |= sphere_run_lanes<kFuseK>(sc, obj_begin, obj_end, o4, d4, tp, nm, valid);
  }
#pragma unroll
  for (int j = 0; j < kFuseK; ++j) {
    const uint32_t s = slot_of[j];
    if (kSpheres && !lanes_run && s < n_all) {
      Ray ray;
      ray.o = xyz(o4[j]);
      ray.d = xyz(d4[j]);
      ray.tmin = (__float_as_uint(o4[j].w) >> 31) ? 1e-5f : 1e-4f;
      ray.tmax = tp[j].x >= 0.0f ? tp[j].x : FLT_MAX;
      Hit rec;
      bool changed = false;
      if (PT_FOLD_TAIL && sc.fold_run != 0u) sphere_fold(sc, obj_begin, obj_end, ray, rec, changed);
      else sphere_segment(sc, obj_begin, obj_end, ray, rec, changed);
      if (changed) {  // the record stays in registers: its only reader is this thread, a few lines down
        tp[j] = make_float4(rec.t, rec.p.x, rec.p.y, rec.p.z);
        nm[j] = make_float4(rec.n.x, rec.n.y, rec.n.z, __uint_as_float(rec.mat | (rec.side << 31)));
        have_nm |= 1u << j;
      }
    }
    const bool hit = s < n_all && tp[j].x >= 0.0f;
    hit_mask |= hit ? 1u << j : 0u;
    const uint64_t live = __ballot(hit && !last_bounce);
    if ((threadIdx.x & 63u) == 0u) s_cnt[j * 4 + (int)wave] = (uint32_t)__popcll(live);
  }
  // what phase 2 still needs from memory, requested before anybody waits for anything
  float4 t4[kFuseK];
#pragma unroll
  for (int j = 0; j < kFuseK; ++j) {
    const uint32_t s = slot_of[j];
    t4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s < n_all) t4[j] = bounce == 0 ? make_float4(1.0f, 1.0f, 1.0f, 0.0f) : ldnt(&in.t4[s]);  // (k_raygen does not write it)
    if (!kFirst && (hit_mask >> j & 1u) && !(have_nm >> j & 1u)) nm[j] = ldnt(&hits.nm[s]);
  }
  __syncthreads();
  // ---- the tile's survivor count goes out (one round trip after the workgroup started) ----
  uint32_t agg = 0u;
#pragma unroll
  for (int k = 0; k < kFuseK * 4; ++k) agg += s_cnt[k];
  const unsigned long long tag = (unsigned long long)epoch << 34;
  if (threadIdx.x == 0u)
    __hip_atomic_store(&tile_desc[tile], tag | (tile == 0u ? kDescPrefix : kDescAggregate) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- phase 2: material_kernel per slot, in place: the new ray and throughput of a survivor take the registers of the
  // old ones (nothing is written yet: where to is known only after the look-back, which by then has had the whole of
  // this phase to resolve); paths that end go into the framebuffer ----
  uint32_t surv_mask = 0u;
#pragma unroll
  for (int j = 0; j < kFuseK; ++j) {
    const uint32_t s = slot_of[j];
    f3 ro = xyz(o4[j]), rd = xyz(d4[j]), color = xyz(t4[j]);
    uint32_t pixbits = __float_as_uint(o4[j].w);
    if (s < n_all) {
      const uint32_t pixel = pixbits & 0x7fffffffu;
      const uint32_t local_pixel = band_local(band, pixel);
      bool tmin_flag = (pixbits >> 31) != 0u;
#if PT_SHADE_KINDS
      const bool is_hit = (hit_mask >> j & 1u) != 0u;
      const f3 hn = xyz(nm[j]);
      const uint32_t ms = __float_as_uint(nm[j].w);
      DMaterial m{3, {0.f, 0.f, 0.f, 0.f}};  // (types are 0..2: validate_scene)
      if (is_hit) m = sc.materials[ms & 0x7fffffffu];
      if (bounce == 0) {
        if (is_hit) accumulate_nd(fb.nd4, local_pixel, acc_iteration, hn, tp[j].x);  // path_tracer.cu:308-311
        else accumulate_nd(fb.nd4, local_pixel, acc_iteration, -rd, 1e6f);           // raygen defaults, ray_gen.cu:26-28
      }
      const uint32_t slot = (slot_base ? *slot_base : 0u) + s;
      shade_kinds((uint32_t)m.type, ro, rd, tmin_flag, mk3(tp[j].y, tp[j].z, tp[j].w), hn, ms >> 31, m, slot, iteration, (uint32_t)bounce, color);
      if (!is_hit || last_bounce) {
        // a miss ends the path with throughput * sky (path_tracer.cu:304-307, 283-289); capped paths deposit raw throughput
        accumulate_color(fb.color4, local_pixel, acc_iteration, color);
      } else {
        surv_mask |= 1u << j;
        o4[j] = make_float4(ro.x, ro.y, ro.z, __uint_as_float(pixel | (tmin_flag ? 0x80000000u : 0u)));
        d4[j] = make_float4(rd.x, rd.y, rd.z, 0.0f);
        t4[j] = make_float4(color.x, color.y, color.z, 0.0f);
      }
#else
      if (!(hit_mask >> j & 1u)) {
        // miss: throughput *= sky; the path ends (path_tracer.cu:304-307, 283-289)
        color = color * background(rd);
        if (bounce == 0) accumulate_nd(fb.nd4, local_pixel, acc_iteration, -rd, 1e6f);  // raygen defaults, ray_gen.cu:26-28
        accumulate_color(fb.color4, local_pixel, acc_iteration, color);
      } else {
        const f3 hn = xyz(nm[j]);
        if (bounce == 0) accumulate_nd(fb.nd4, local_pixel, acc_iteration, hn, tp[j].x);  // path_tracer.cu:308-311
        const uint32_t ms = __float_as_uint(nm[j].w);
        const DMaterial m = sc.materials[ms & 0x7fffffffu];
        // RNG re-seeded from the global slot index, then discard(bounce) (path_tracer.cu:300-301)
        const uint32_t slot = (slot_base ? *slot_base : 0u) + s;
        Minstd rng;
        rng.seed(path_seed(slot, iteration));
        rng.discard((uint32_t)bounce);
        const f3 hp = mk3(tp[j].y, tp[j].z, tp[j].w);
        evaluate_material(ro, rd, tmin_flag, hp, hn, ms >> 31, m, rng, color);
        if (last_bounce) {
          accumulate_color(fb.color4, local_pixel, acc_iteration, color);  // capped paths deposit raw throughput
        } else {
          surv_mask |= 1u << j;
          o4[j] = make_float4(ro.x, ro.y, ro.z, __uint_as_float(pixel | (tmin_flag ? 0x80000000u : 0u)));
          d4[j] = make_float4(rd.x, rd.y, rd.z, 0.0f);
          t4[j] = make_float4(color.x, color.y, color.z, 0.0f);
        }
      }
#endif
    }
  }

  // ---- the tile's offset comes in ----
  if (wave == 0u) {
    uint32_t excl = 0u;
    if (tile != 0u) {
      excl = tile_lookback(tile_desc, tile, epoch, &counters->flags);
      if (threadIdx.x == 0u)
        __hip_atomic_store(&tile_desc[tile], tag | kDescPrefix | (excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0u) {
      s_excl = excl;
      if (tile + 1u == tiles) {  // the last tile knows the frame's total (k_scan's epilogue)
        counters->live[bounce + 1] = last_bounce ? 0u : excl + agg;
        counters->rays_total += n_all;
        counters->paths[bounce] += n_all;
      }
    }
  }
  __syncthreads();

  // ---- survivors to their stable place: tile offset + sub-blocks before + wavefronts before + lanes before ----
  uint32_t base = s_excl;
#pragma unroll
  for (int j = 0; j < kFuseK; ++j) {
    uint32_t before = 0u, in_sub = 0u;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t c = s_cnt[j * 4 + w];
      before += (uint32_t)w < wave ? c : 0u;
      in_sub += c;
    }
    const bool survives = (surv_mask >> j & 1u) != 0u;
    const uint64_t live = __ballot(survives);
    if (survives) {
      const uint32_t dst = base + before + rank_below(live);
      stnt(&out.o4[dst], o4[j]);
      stnt(&out.d4[dst], d4[j]);
      stnt(&out.t4[dst], t4[j]);
      if (octs) octs[dst] = (uint8_t)((d4[j].x < 0.0f ? 1u : 0u) | (d4[j].y < 0.0f ? 2u : 0u) | (d4[j].z < 0.0f ? 4u : 0u));
    }
    base += in_sub;
  }
}

// Ray sorting ("ray_sort", BASELINE.json's ray-sorted wavefront; the reference keeps a sort_by_key by material
// commented out, path_tracer.cu:439-446).  The slots -- and with them the random numbers, which are keyed on the
// compacted slot index -- are NOT permuted: what is sorted is the ORDER in which the persistent traversal lanes pick
// their rays up.  Within every block of 4096 consecutive slots (neighbouring pixels: neighbouring ray origins) the
// rays are grouped by direction octant, stably, into an index array the ray feed reads through; a wavefront's 64 rays
// then start close together AND head the same way.  Results cannot change; what it buys is measured in DESIGN.md.
constexpr uint32_t kSortBlock = 4096u;
__global__ __launch_bounds__(1024) void k_sort_octant(const uint8_t* octs, uint32_t* order, int bounce, DeviceCounters* counters,
                                                      DBatchInfo bi)
{
  __shared__ uint32_t s_cnt[8][4][16];  // [octant][round][wavefront]
  __shared__ uint32_t s_base[8][4][16];
  const uint32_t frame = blockIdx.y;
  const uint32_t n = counters[frame].live[bounce];
  const uint32_t block0 = blockIdx.x * kSortBlock;
  if (block0 >= n) return;
  const size_t fbase = (size_t)frame * bi.stride;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t oct[4], rank[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t s = block0 + (uint32_t)r * 1024u + threadIdx.x;
    oct[r] = s < n ? (uint32_t)octs[fbase + s] : 8u;
#pragma unroll
    for (uint32_t o = 0; o < 8u; ++o) {
      const uint64_t m = __ballot(oct[r] == o);
      if (oct[r] == o) rank[r] = rank_below(m);
      if (lane == 0u) s_cnt[o][r][wave] = (uint32_t)__popcll(m);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0u) {  // 512 counters: exclusive scan in (octant, round, wavefront) order
    uint32_t run = 0u;
    for (int o = 0; o < 8; ++o)
      for (int r = 0; r < 4; ++r)
        for (int w = 0; w < 16; ++w) {
          s_base[o][r][w] = run;
          run += s_cnt[o][r][w];
        }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t s = block0 + (uint32_t)r * 1024u + threadIdx.x;
    if (oct[r] < 8u) order[fbase + block0 + s_base[oct[r]][r][wave] + rank[r]] = (uint32_t)fbase + s;
  }
}


// final_gather (path_tracer.cu:203-219) of the batch's staged samples into the accumulated framebuffers, in
// iteration order (running means do not commute)
__global__ __launch_bounds__(256) void k_accumulate(DFrame stage, DFrame fb, uint32_t pix_count, DBatchInfo bi)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= pix_count) return;
  // The running means of a pixel stay in registers over the frames of the batch (one read and one write of the
  // framebuffer per batch; the same operations in the same order as a fold frame by frame), and the staged samples are
  // requested eight frames at a time: one after the other, a thread of a 32-frame batch sat through 32 dependent round
  // trips and the kernel moved 2.3 TB/s.
  const uint32_t first = bi.iteration[0];
  float4 col = make_float4(0.f, 0.f, 0.f, 0.f), nd = make_float4(0.f, 0.f, 0.f, 0.f);
  if (first != 0u) {
    col = ldnt(&fb.color4[i]);
    nd = ldnt(&fb.nd4[i]);
  }
  constexpr uint32_t kAhead = 8u;
  for (uint32_t f0 = 0; f0 < bi.count; f0 += kAhead) {
    float4 c[kAhead], g[kAhead];
#pragma unroll
    for (uint32_t k = 0; k < kAhead; ++k) {
      const uint32_t f = min(f0 + k, bi.count - 1u);
      c[k] = ldnt(&stage.color4[(size_t)f * bi.stride + i]);
      g[k] = ldnt(&stage.nd4[(size_t)f * bi.stride + i]);
    }
#pragma unroll
    for (uint32_t k = 0; k < kAhead; ++k) {
      if (f0 + k >= bi.count) break;
      const uint32_t it = bi.iteration[f0 + k];
      col.x = running_mean(it, col.x, c[k].x);
      col.y = running_mean(it, col.y, c[k].y);
      col.z = running_mean(it, col.z, c[k].z);
      nd.x = running_mean(it, nd.x, g[k].x);
      nd.y = running_mean(it, nd.y, g[k].y);
      nd.z = running_mean(it, nd.z, g[k].z);
      nd.w = running_mean(it, nd.w, g[k].w);
    }
  }
  col.w = 0.0f;
  stnt(&fb.color4[i], col);
  stnt(&fb.nd4[i], nd);
}

// path_tracing_mega_kernel, path_tracer.cu:227-269: the whole path in one thread, one RNG stream per
// pixel (a different image from streaming mode at the same seed -- a property of the reference).
__global__ __launch_bounds__(kWave) void k_megakernel(DScene sc, DCamera cam, uint32_t iteration, DBand band,
                                                      uint32_t pix_count, int max_bounces, DFrame fb,
                                                      DeviceCounters* counters)
{
  __shared__ uint32_t s_stack[kStackDepth * kWave];
  const uint32_t s = blockIdx.x * kWave + threadIdx.x;
  uint32_t rays = 0u, flags = 0u;
  if (s < pix_count) {
    const uint32_t pixel = band_pixel(band, s);
    const uint32_t x = pixel % cam.width, y = pixel / cam.width;
    Minstd rng;
    rng.seed(path_seed(pixel, iteration));
    const float fx = (float)x + rng.uniform();
    const float fy = (float)y + rng.uniform();
    Ray ray;
    generate_ray(cam, fx, fy, ray.o, ray.d);
    ray.tmin = 1e-4f;
    ray.tmax = FLT_MAX;
    f3 color = mk3(1.0f, 1.0f, 1.0f);
    f3 normal = -ray.d;
    float depth = 1e6f;
    for (int i = 0; i < max_bounces; ++i) {
      Hit rec;
      rec.t = 0.0f;
      rec.p = rec.n = mk3(0.f, 0.f, 0.f);
      rec.mat = 0u;
      rec.side = 0u;
      ++rays;
      Tally tally;
      if (!ray_scene<false>(ray, sc, rec, s_stack + threadIdx.x, flags, tally)) {
        color = color * background(ray.d);
        break;
      }
      if (i == 0) {
        normal = rec.n;
        depth = rec.t;
      }
      bool tmin_flag = ray.tmin != 1e-4f;
      evaluate_material(ray.o, ray.d, tmin_flag, rec.p, rec.n, rec.side, sc.materials[rec.mat], rng, color);
      ray.tmin = tmin_flag ? 1e-5f : 1e-4f;
    }
    accumulate_color(fb.color4, s, iteration, color);
    accumulate_nd(fb.nd4, s, iteration, normal, depth);
    if (flags) atomicOr(&counters->flags, flags);
  }
  // one ray-counter atomic per wavefront
  uint32_t sum = rays;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  if (threadIdx.x == 0u && sum) atomicAdd(&counters->rays_total, (unsigned long long)sum);
}

// intersection_kernel on caller-supplied rays (parity tests): rays_o = origin.xyz,t_min ; rays_d = direction.xyz,t_max
template <int kVariant>
__global__ __launch_bounds__(kWave) void k_intersect(DScene sc, const float4* rays_o, const float4* rays_d, uint32_t n,
                                                     DHits hits, DeviceCounters* counters)
{
  __shared__ uint32_t s_stack[kStackDepth * kWave];
  const uint32_t s = blockIdx.x * kWave + threadIdx.x;
  if (s >= n) return;
  const float4 o = rays_o[s], d = rays_d[s];
  Ray ray;
  ray.o = xyz(o);
  ray.tmin = o.w;
  ray.d = xyz(d);
  ray.tmax = d.w;
  Hit rec;
  rec.t = 0.0f;
  rec.p = rec.n = mk3(0.f, 0.f, 0.f);
  rec.mat = 0u;
  rec.side = 0u;
  uint32_t flags = 0u;
  Tally tally;
  const bool hit = kVariant == 0 ? ray_scene<false>(ray, sc, rec, s_stack + threadIdx.x, flags, tally)
                                 : ray_scene_wide<false>(ray, sc, rec, s_stack + threadIdx.x, flags, tally);
  stnt(&hits.tp[s], make_float4(hit ? rec.t : -1.0f, rec.p.x, rec.p.y, rec.p.z));
  stnt(&hits.nm[s], make_float4(rec.n.x, rec.n.y, rec.n.z, __uint_as_float(rec.mat | (rec.side << 31))));
  if (flags) atomicOr(&counters->flags, flags);
}

// preview_kernel / preview_depth_kernel, path_tracer.cu:334-385.
// mode 0: rgb of buf ; 1: normal view (xyz*0.5+0.5) ; 2: depth view (1/w, alpha 1)
__global__ __launch_bounds__(256) void k_preview(const float4* buf, uint32_t pix_count, int mode, uint32_t* rgba)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= pix_count) return;
  const float4 v = buf[i];
  f3 c = xyz(v);
  uint32_t alpha = 255u;
  if (mode == 2) {
    const float d = 1.0f / v.w;
    c = mk3(d, d, d);
    alpha = 1u;
  } else if (mode == 1) {
    c = c * 0.5f + mk3(0.5f, 0.5f, 0.5f);
  }
  const float g = 1.f / 2.2f;
  c = mk3(powf(c.x, g), powf(c.y, g), powf(c.z, g));
  auto to255 = [](float x) -> uint32_t { return (uint32_t)(unsigned char)(sel_min(sel_max(x, 0.f), 1.f) * 255.99f); };
  rgba[i] = to255(c.x) | (to255(c.y) << 8) | (to255(c.z) << 16) | (alpha << 24);
}

// Multi-GPU gather on the root, ONE launch for all ranks: blockIdx.y = source band.  A band is a rank's packed rows
// (channels floats per pixel); src is the root's own buffer or a peer's buffer mapped through HIP IPC, read where it
// lies -- over xGMI when the peer is another GPU, every peer -> root link busy at once, no staging copy.  Thread i of a
// band moves float i (consecutive threads read consecutive floats; a row of the band is a run of the frame).
__global__ __launch_bounds__(256) void k_gather_bands(DGatherBands bands, int channels, uint32_t frame_pixels, float* frame)
{
  const DGatherBands::Src& b = bands.src[blockIdx.y];
  const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (i >= (uint64_t)b.pix_count * (uint32_t)channels) return;
  const uint32_t s = (uint32_t)(i / (uint32_t)channels), c = (uint32_t)(i % (uint32_t)channels);
  const uint32_t pixel = band_pixel(b.band, s);
  if (pixel >= frame_pixels) return;  // (ptc_band_import has checked the geometry; a stray handle must not write outside)
  frame[(size_t)pixel * (size_t)channels + c] = __builtin_nontemporal_load(&b.src[i]);
}

// preview_kernel / preview_depth_kernel on a packed frame (the gathered frame of a multi-GPU run)
__global__ __launch_bounds__(256) void k_preview_packed(const float* buf, uint32_t pix_count, int channels, int mode, uint32_t* rgba)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= pix_count) return;
  f3 c;
  uint32_t alpha = 255u;
  if (mode == 2) {
    const float d = 1.0f / buf[(size_t)i * (size_t)channels];
    c = mk3(d, d, d);
    alpha = 1u;
  } else {
    c = mk3(buf[3u * (size_t)i], buf[3u * (size_t)i + 1u], buf[3u * (size_t)i + 2u]);
    if (mode == 1) c = c * 0.5f + mk3(0.5f, 0.5f, 0.5f);
  }
  const float g = 1.f / 2.2f;
  c = mk3(powf(c.x, g), powf(c.y, g), powf(c.z, g));
  auto to255 = [](float x) -> uint32_t { return (uint32_t)(unsigned char)(sel_min(sel_max(x, 0.f), 1.f) * 255.99f); };
  rgba[i] = to255(c.x) | (to255(c.y) << 8) | (to255(c.z) << 16) | (alpha << 24);
}

// float4 framebuffer -> packed vec3 (which 0) or the w channel (which 1)
__global__ __launch_bounds__(256) void k_pack(const float4* buf, uint32_t pix_count, int which, float* dst)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= pix_count) return;
  const float4 v = buf[i];
  if (which == 0) {
    dst[3u * (size_t)i] = v.x;
    dst[3u * (size_t)i + 1u] = v.y;
    dst[3u * (size_t)i + 2u] = v.z;
  } else {
    dst[i] = v.w;
  }
}

// denoising_kernel, denoising/edge_avoiding_a_trous_denoiser.cu:24-86, in two kernels.
// The reference rebuilds the view ray of every one of the 25 taps in every pass (generate_ray: a normalise and
// a matrix product each); the tap positions depend only on (pixel, accumulated depth), so k_denoise_positions
// computes them once per denoise call and the four passes read them back (16 B per tap instead of ~40
// instructions).  The reference clamps taps to [0,W] x [0,H] INCLUSIVE (cu:39-42): column W aliases the next
// row's column 0 but keeps its own view ray, and row H is out of bounds; taps on column W / row H therefore
// rebuild their ray here (edge pixels only), and an index beyond the array reads element W*H-1.
__global__ __launch_bounds__(256) void k_denoise_positions(DCamera cam, uint32_t pix_count, const float4* nd, float4* pos)
{
  const uint32_t index = blockIdx.x * 256u + threadIdx.x;
  if (index >= pix_count) return;
  const int x = (int)(index % cam.width), y = (int)(index / cam.width);
  f3 ro, rd;
  generate_ray(cam, (float)x + 0.5f, (float)y + 0.5f, ro, rd);
  const f3 p = ro + rd * nd[index].w;
  pos[index] = make_float4(p.x, p.y, p.z, 0.0f);
}

// Arithmetic of the pass: the three edge-stopping weights min(exp(-d/phi), 1) of a tap (cu:63-77) multiply to
// exp(-(dc/c_phi + dn/(step^2 n_phi) + dp/p_phi)) -- every d is a sum of squares, so no factor exceeds 1 and the
// clamps are inert.  The kernel evaluates that single exponential with v_exp_f32 on a base-2 argument whose
// three reciprocal scale factors are computed once per pass (the reference: three divisions and three expf per
// tap, 75 of each per pixel per pass, which made this kernel VALU-bound).  This stage is outside the random-number
// feedback loop and is compared with the oracle under a tolerance (1e-5 absolute on the radiance,
// tests/test_gpu_parity.py), not bit for bit; contraction into FMAs is allowed here for the same reason.
// kInterior: every tap of the tile is inside the image (no clamp, no off-by-one column / row): the common case,
// decided per tile so that the wavefront does not branch per tap.
template <bool kInterior>
__device__ __forceinline__ void denoise_pixel(const DCamera& cam, const uint32_t pix_count, const float4* color,
                                              const float4* nd, const float4* pos, float4* out, const int step_width,
                                              const DDenoise& prm, const int x, const int y)
{
#pragma clang fp contract(fast)
  const uint32_t W = cam.width, H = cam.height;
  const uint32_t index = (uint32_t)x + (uint32_t)y * W;
  const float kernel[3] = {3.f / 8.f, 1.f / 4.f, 1.f / 16.f};
  const f3 cval = xyz(color[index]);
  const f3 nval = xyz(nd[index]);
  const f3 pval = xyz(pos[index]);
  f3 sum = mk3(0.f, 0.f, 0.f);
  float cum_w = 0.0f;
  const float step2 = (float)(step_width * step_width);
  constexpr float kLog2e = 1.4426950408889634f;
  const float kc = -kLog2e / prm.c_phi, kn = -kLog2e / (step2 * prm.n_phi), kp = -kLog2e / prm.p_phi;
#pragma unroll 1
  for (int dy = -2; dy <= 2; ++dy) {
    int v = y + dy * step_width;
    if (!kInterior) v = v < 0 ? 0 : (v > (int)H ? (int)H : v);
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      int u = x + dx * step_width;
      if (!kInterior) u = u < 0 ? 0 : (u > (int)W ? (int)W : u);
      uint32_t ti = (uint32_t)u + (uint32_t)v * W;
      if (!kInterior && ti >= pix_count) ti = pix_count - 1u;
      const f3 ctemp = xyz(color[ti]);
      const float4 ndt = nd[ti];
      f3 ptmp;
      if (!kInterior && (u == (int)W || v == (int)H)) {  // the reference's off-by-one taps keep their own view ray
        f3 to, td;
        generate_ray(cam, (float)u + 0.5f, (float)v + 0.5f, to, td);
        ptmp = to + td * ndt.w;
      } else {
        ptmp = xyz(pos[ti]);
      }
      const f3 tc = cval - ctemp, tn = nval - xyz(ndt), tp = pval - ptmp;
      const float arg = dot(tc, tc) * kc + dot(tn, tn) * kn + dot(tp, tp) * kp;
      const float weight = __builtin_amdgcn_exp2f(arg);
      const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
      const float wk = weight * kernel[adx < ady ? adx : ady];
      sum = sum + ctemp * wk;
      cum_w += wk;
    }
  }
  const float inv_w = 1.0f / cum_w;
  out[index] = make_float4(sum.x * inv_w, sum.y * inv_w, sum.z * inv_w, 0.0f);
}

__global__ __launch_bounds__(256) void k_denoise(DCamera cam, uint32_t pix_count, const float4* color, const float4* nd,
                                                 const float4* pos, float4* out, int step_width, DDenoise prm)
{
  // 16x16 pixel tiles: neighbouring threads share most of their (dilated) taps in L1/L2
  const int W = (int)cam.width, H = (int)cam.height;
  const uint32_t tiles_x = ((uint32_t)W + 15u) / 16u;
  const int x0 = (int)(blockIdx.x % tiles_x) * 16, y0 = (int)(blockIdx.x / tiles_x) * 16;
  const int x = x0 + (int)(threadIdx.x & 15u), y = y0 + (int)(threadIdx.x >> 4);
  if (x >= W || y >= H) return;
  const int reach = 2 * step_width;
  const bool interior = x0 - reach >= 0 && y0 - reach >= 0 && x0 + 15 + reach < W && y0 + 15 + reach < H;
  if (interior) denoise_pixel<true>(cam, pix_count, color, nd, pos, out, step_width, prm, x, y);
  else denoise_pixel<false>(cam, pix_count, color, nd, pos, out, step_width, prm, x, y);
}

// The same pass with its taps staged in LDS (the default).  The taps of a pixel sit `step` apart: vertically a
// workgroup works on ONE residue class of rows (y mod step): four lattice rows of outputs need eight lattice rows of
// taps, whatever the step; horizontally it is dense (64 consecutive pixels of outputs, 2 * step more on either side),
// so every global load is a coalesced run of pixels and every byte of a fetched cache line is used.  36 bytes per
// staged pixel (colour, normal, position): (64 + 4 step) x 8 of them, 28 KB at step 8.  Every tap then is three LDS
// reads instead of three 16-byte global loads through L1: the pass was bound by the L1 / texture-address rate of its 75
// loads per pixel (110 us at 1080p).  (First attempt, measured: sub-lattices in BOTH directions -- 16x16 outputs from
// 20x20 staged points at any step -- fetch one cache line per point and array at step 8, and the step x step
// workgroups that share those lines are dealt round-robin to the eight XCDs, each with its own L2: 171 us for that
// pass.)  The reference's clamp of a tap coordinate to [0, W] x [0, H] (inclusive: column W aliases the next row, row
// H is out of bounds; see k_denoise_positions) depends only on the tap's coordinate, not on which output uses it, so it
// is applied once, when the pixel is staged.
// kStep is a template parameter (the steps of a denoise call are 1, 2, 4, ...): tap offsets become immediates of the
// LDS reads -- with a run-time step every tap cost three address additions.
constexpr int kDenW = 64, kDenRows = 4, kDenHalo = 2;
template <int kStep>
__global__ __launch_bounds__(256) void k_denoise_lds(DCamera cam, uint32_t pix_count, const float4* color, const float4* nd,
                                                     const float4* pos, float4* out, DDenoise prm)
{
#pragma clang fp contract(fast)
  extern __shared__ float4 s_dyn[];
  constexpr int step = kStep;
  const int W = (int)cam.width, H = (int)cam.height;
  constexpr int row_len = kDenW + 2 * kDenHalo * step, rows = kDenRows + 2 * kDenHalo, points = row_len * rows;
  float4* s_a = s_dyn;                                       // colour.rgb, normal.x
  float4* s_b = s_dyn + points;                              // normal.yz, position.xy
  float* s_c = reinterpret_cast<float*>(s_dyn + 2 * points);  // position.z
  const uint32_t tiles_x = ((uint32_t)W + kDenW - 1u) / kDenW;
  const uint32_t lattice_rows = ((uint32_t)H + (uint32_t)step - 1u) / (uint32_t)step;
  const uint32_t tiles_y = (lattice_rows + kDenRows - 1u) / kDenRows;
  // Workgroups that share staged rows (vertical neighbours of one residue class) must share an L2: workgroup b runs
  // on XCD b % 8 (MI355X_MICROARCH.md, workgroup dispatch), so every XCD gets one contiguous eighth of the tiles in
  // (residue, tile row, tile column) order -- a workgroup's vertical neighbour is 30 workgroups away on the same XCD,
  // about 1 MB of input apart, well inside its 4 MB L2.  Dealt round-robin instead, neighbours land on different
  // XCDs, every halo row is fetched from the Infinity Cache again, and the pass is bound there (2-3x the bytes).
  const uint32_t total = tiles_x * tiles_y * (uint32_t)step, per_xcd = (total + 7u) / 8u;
  uint32_t b = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per_xcd || b >= total) return;
  const int tx = (int)(b % tiles_x);
  b /= tiles_x;
  const int ty = (int)(b % tiles_y), ry = (int)(b / tiles_y);
  const int x0 = tx * kDenW - kDenHalo * step;  // first staged column
  // stage (coordinates may lie outside the image: the reference's clamp).  Three rounds of 256 pixels at a time with
  // all their global loads issued before the first is used: staged one round after the other, the dependent round
  // trips (about 2 us each) were most of a workgroup's life and the pass ran at 65-70 us whatever the step
  constexpr int kRounds = 3;
  for (int base = 0; base < points; base += 256 * kRounds) {
    float4 c[kRounds], g[kRounds], q[kRounds];
    int uu[kRounds], vv[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int k = min(base + r * 256 + (int)threadIdx.x, points - 1);
      const int li = k % row_len, lj = k / row_len;
      int u = x0 + li;
      int v = (ty * kDenRows + lj - kDenHalo) * step + ry;
      u = u < 0 ? 0 : (u > W ? W : u);
      v = v < 0 ? 0 : (v > H ? H : v);
      uint32_t ti = (uint32_t)u + (uint32_t)v * (uint32_t)W;
      if (ti >= pix_count) ti = pix_count - 1u;
      uu[r] = u;
      vv[r] = v;
      c[r] = color[ti];
      g[r] = nd[ti];
      q[r] = pos[ti];
    }
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
      const int k = base + r * 256 + (int)threadIdx.x;
      f3 p = xyz(q[r]);
      if (uu[r] == W || vv[r] == H) {  // the reference's off-by-one taps keep their own view ray
        f3 to, td;
        generate_ray(cam, (float)uu[r] + 0.5f, (float)vv[r] + 0.5f, to, td);
        p = to + td * g[r].w;
      }
      if (k < points) {
        s_a[k] = make_float4(c[r].x, c[r].y, c[r].z, g[r].x);
        s_b[k] = make_float4(g[r].y, g[r].z, p.x, p.y);
        s_c[k] = p.z;
      }
    }
  }
  __syncthreads();
  const int i = (int)(threadIdx.x & 63u), j = (int)(threadIdx.x >> 6);
  const int x = tx * kDenW + i, y = (ty * kDenRows + j) * step + ry;
  if (x >= W || y >= H) return;
  const int centre = (j + kDenHalo) * row_len + i + kDenHalo * step;
  const float4 ca = s_a[centre], cb = s_b[centre];
  const f3 cval = mk3(ca.x, ca.y, ca.z), nval = mk3(ca.w, cb.x, cb.y), pval = mk3(cb.z, cb.w, s_c[centre]);
  const float kernel[3] = {3.f / 8.f, 1.f / 4.f, 1.f / 16.f};
  const float step2 = (float)(step * step);
  constexpr float kLog2e = 1.4426950408889634f;
  const float kc = -kLog2e / prm.c_phi, kn = -kLog2e / (step2 * prm.n_phi), kp = -kLog2e / prm.p_phi;
  f3 sum = mk3(0.f, 0.f, 0.f);
  float cum_w = 0.0f;
  // one tap row per iteration (not unrolled: the fully unrolled 5x5 keeps 130 registers alive -- three wavefronts per
  // SIMD); the tap weight kernel[min(|dx|, |dy|)] of a row depends on |dx| only through three row constants
#pragma unroll 1
  for (int dy = -2; dy <= 2; ++dy) {
    const int ady = dy < 0 ? -dy : dy;
    const float w_by_adx[3] = {kernel[0], kernel[ady < 1 ? ady : 1], kernel[ady]};
    const int row = centre + dy * row_len;
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      const int t = row + dx * step;
      const float4 ta = s_a[t], tb = s_b[t];
      const float tz = s_c[t];
      // (the squared distances written out here, inside the contraction pragma's scope: dot() from pt_math.hpp is
      // compiled under the file's -ffp-contract=off and kept the pass at 37 instead of 27 instructions per tap)
      const float cx = cval.x - ta.x, cy = cval.y - ta.y, cz = cval.z - ta.z;
      const float nx = nval.x - ta.w, ny = nval.y - tb.x, nz = nval.z - tb.y;
      const float px = pval.x - tb.z, py = pval.y - tb.w, pz = pval.z - tz;
      const float dc = cx * cx + cy * cy + cz * cz, dn = nx * nx + ny * ny + nz * nz, dp = px * px + py * py + pz * pz;
      const float arg = dc * kc + dn * kn + dp * kp;
      const float weight = __builtin_amdgcn_exp2f(arg);
      const float wk = weight * w_by_adx[dx < 0 ? -dx : dx];
      sum.x += ta.x * wk;
      sum.y += ta.y * wk;
      sum.z += ta.z * wk;
      cum_w += wk;
    }
  }
  const float inv_w = 1.0f / cum_w;
  out[(uint32_t)x + (uint32_t)y * (uint32_t)W] = make_float4(sum.x * inv_w, sum.y * inv_w, sum.z * inv_w, 0.0f);
}

__global__ void k_selftest(const float* a, const float* b, uint32_t n, float* out_div, float* out_sqrt, float* out_sin,
                           float* out_cos)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out_div[i] = a[i] / b[i];
  out_sqrt[i] = ieee_sqrt(a[i]);
  float s, c;
  det_sincos(a[i], s, c);
  out_sin[i] = s;
  out_cos[i] = c;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }

void launch_raygen(hipStream_t s, const DCameras& cams, const DBatchInfo& bi, DBand band, uint32_t pix_count,
                   DPaths paths, DeviceCounters* counters, const DObject* objects, uint32_t filt_begin, uint32_t filt_end,
                   uint32_t* worklist, DHits hits, unsigned long long* tile_desc, uint32_t tile_stride, uint32_t epoch,
                   bool finish_misses, DFrame fb, bool staged)
{
  const dim3 grid(div_up(pix_count, 256u * kListPer) * bi.count), block(256);
  const DTileScan scan{tile_desc, epoch};
  if (worklist && filt_begin < filt_end && tile_desc) {
    if (finish_misses)
      hipLaunchKernelGGL((k_raygen<true, true>), grid, block, 0, s, cams, bi, band, pix_count, paths, counters, objects, filt_begin,
                         filt_end, worklist, hits, scan, tile_stride, fb, staged ? 1 : 0);
    else
      hipLaunchKernelGGL((k_raygen<true, false>), grid, block, 0, s, cams, bi, band, pix_count, paths, counters, objects, filt_begin,
                         filt_end, worklist, hits, scan, tile_stride, fb, 0);
  } else {
    hipLaunchKernelGGL((k_raygen<false, false>), grid, block, 0, s, cams, bi, band, pix_count, paths, counters, objects, 0u, 0u, worklist,
                       hits, scan, tile_stride, fb, 0);
  }
}
void launch_trace(hipStream_t s, const DScene& scene, DPaths paths, DHits hits, uint32_t max_paths, int bounce,
                  DeviceCounters* counters, bool count_tests, int variant)
{
  if (variant == 1) {
    const dim3 grid(div_up(max_paths, kWave) + 8u);  // room for the per-XCD rounding of the chunk deal
    if (count_tests) hipLaunchKernelGGL(k_trace_wide<true>, grid, dim3(kWave), 0, s, scene, paths, hits, bounce, counters);
    else hipLaunchKernelGGL(k_trace_wide<false>, grid, dim3(kWave), 0, s, scene, paths, hits, bounce, counters);
    return;
  }
  if (count_tests) hipLaunchKernelGGL(k_trace<true>, dim3(div_up(max_paths, kWave)), dim3(kWave), 0, s, scene, paths, hits, bounce, counters);
  else hipLaunchKernelGGL(k_trace<false>, dim3(div_up(max_paths, kWave)), dim3(kWave), 0, s, scene, paths, hits, bounce, counters);
}
void launch_spheres(hipStream_t s, const DScene& scene, uint32_t obj_begin, uint32_t obj_end, bool first, DPaths paths, DHits hits,
                    uint32_t max_paths, int bounce, DeviceCounters* counters, const DBatchInfo& bi, uint32_t filt_begin,
                    uint32_t filt_end, uint32_t* worklist, unsigned long long* tile_desc, uint32_t tile_stride, uint32_t epoch)
{
  const dim3 grid(div_up(max_paths, 256u * kListPer) * bi.count), block(256);
  const DTileScan scan{tile_desc, epoch};
#define PT_SPHERES(FIRST, FILTER)                                                                                              \
  hipLaunchKernelGGL((k_spheres<FIRST, FILTER>), grid, block, 0, s, scene, obj_begin, obj_end, paths, hits, bounce, counters, bi, \
                     filt_begin, filt_end, worklist, scan, tile_stride)
  if (worklist && filt_begin < filt_end && tile_desc) {
    if (first) PT_SPHERES(true, true);
    else PT_SPHERES(false, true);
  } else {
    if (first) PT_SPHERES(true, false);
    else PT_SPHERES(false, false);
  }
#undef PT_SPHERES
}
void launch_tail_count(hipStream_t s, const DScene& scene, uint32_t obj_begin, uint32_t obj_end, bool first, DPaths paths,
                       DHits hits, uint32_t max_paths, int bounce, uint32_t* chunk_counts, DeviceCounters* counters,
                       const DBatchInfo& bi)
{
  const dim3 grid(div_up(max_paths, 256u), bi.count), block(256);
  if (obj_begin < obj_end) {
    if (first) hipLaunchKernelGGL((k_tail_count<true, true>), grid, block, 0, s, scene, obj_begin, obj_end, paths, hits, bounce, chunk_counts, counters, bi);
    else hipLaunchKernelGGL((k_tail_count<true, false>), grid, block, 0, s, scene, obj_begin, obj_end, paths, hits, bounce, chunk_counts, counters, bi);
  } else {
    // (nothing to test: some closest-hit launch has written every record of the bounce)
    hipLaunchKernelGGL((k_tail_count<false, false>), grid, block, 0, s, scene, 0u, 0u, paths, hits, bounce, chunk_counts, counters, bi);
  }
}
void launch_scan(hipStream_t s, int bounce, bool last_bounce, const uint32_t* chunk_counts, uint32_t* chunk_offsets,
                 DeviceCounters* counters, const DBatchInfo& bi)
{
  hipLaunchKernelGGL(k_scan, dim3(bi.count), dim3(1024), 0, s, bounce, last_bounce ? 1 : 0, chunk_counts, chunk_offsets,
                     counters, bi);
}
void launch_traverse_run(hipStream_t s, const DScene& scene, uint32_t obj_begin, uint32_t obj_end, bool first, DPaths paths,
                         DHits hits, int bounce, int work_slot, DeviceCounters* counters, bool count_tests, uint32_t waves,
                         uint32_t* slow_list, const uint32_t* order, const DBatchInfo& bi, bool listed)
{
  const dim3 grid(waves), block(kWave);
  if (count_tests) {
    if (first) hipLaunchKernelGGL((k_traverse4m<true, true>), grid, block, 0, s, scene, obj_begin, obj_end, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
    else hipLaunchKernelGGL((k_traverse4m<true, false>), grid, block, 0, s, scene, obj_begin, obj_end, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
  } else {
    if (first) hipLaunchKernelGGL((k_traverse4m<false, true>), grid, block, 0, s, scene, obj_begin, obj_end, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
    else hipLaunchKernelGGL((k_traverse4m<false, false>), grid, block, 0, s, scene, obj_begin, obj_end, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
  }
}
void launch_beam(hipStream_t s, const DScene& scene, uint32_t obj_index, const DCameras& cams, const uint8_t* cam_of, uint32_t nbeam,
                 uint32_t tiles_x, uint32_t tiles_y, uint32_t node_count4, float4* out)
{
  DBeam geo{};
  geo.tiles_x = tiles_x;
  geo.tiles = tiles_x * tiles_y;
  for (uint32_t b = 0; b < nbeam && b < 32u; ++b) geo.beam_of[b] = cam_of[b];
  const uint32_t threads = nbeam * geo.tiles;
  hipLaunchKernelGGL(k_beam, dim3((threads + 63u) / 64u), dim3(64), 0, s, scene, obj_index, cams, geo, nbeam, node_count4, out);
}
void launch_traverse(hipStream_t s, const DScene& scene, uint32_t obj_index, bool first, DPaths paths, DHits hits,
                     int bounce, int work_slot, DeviceCounters* counters, bool count_tests, uint32_t waves,
                     uint32_t* slow_list, const uint32_t* order, int variant, const DBatchInfo& bi, bool listed)
{
  const dim3 grid(waves), block(kWave);
  if (scene.beam.entries && first) {  // bounce 0's first launch: primary rays start at their tile's entry points
    if (count_tests) hipLaunchKernelGGL((k_traverse4<true, true, true>), grid, block, 0, s, scene, obj_index, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
    else hipLaunchKernelGGL((k_traverse4<false, true, true>), grid, block, 0, s, scene, obj_index, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
    return;
  }
  if (count_tests) {
    if (first) hipLaunchKernelGGL((k_traverse4<true, true>), grid, block, 0, s, scene, obj_index, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
    else hipLaunchKernelGGL((k_traverse4<true, false>), grid, block, 0, s, scene, obj_index, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
  } else {
    if (first) hipLaunchKernelGGL((k_traverse4<false, true>), grid, block, 0, s, scene, obj_index, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
    else hipLaunchKernelGGL((k_traverse4<false, false>), grid, block, 0, s, scene, obj_index, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
  }
}
void launch_shade(hipStream_t s, const DScene& scene, DPaths in, DPaths out, DHits hits, uint32_t max_paths,
                  bool staged, int bounce, bool last_bounce, const uint32_t* slot_base,
                  const uint32_t* chunk_offsets, DFrame fb, DBand band, DeviceCounters* counters, uint8_t* octs,
                  const DBatchInfo& bi)
{
  hipLaunchKernelGGL(k_shade, dim3(div_up(max_paths, 256u), bi.count), dim3(256), 0, s, scene, in, out, hits,
                     staged ? 1 : 0, bounce, last_bounce ? 1 : 0, slot_base, chunk_offsets, fb, band, counters, octs, bi);
}
void launch_shade_fused(hipStream_t s, const DScene& scene, uint32_t obj_begin, uint32_t obj_end, bool first, DPaths in, DPaths out,
                        DHits hits, uint32_t max_paths, bool staged, int bounce, bool last_bounce, const uint32_t* slot_base,
                        unsigned long long* tile_desc, uint32_t tile_stride, uint32_t epoch, DFrame fb, DBand band,
                        DeviceCounters* counters, uint8_t* octs, const DBatchInfo& bi, const uint32_t* list)
{
  const dim3 grid(div_up(max_paths, kFuseTile) * bi.count), block(256);
#define PT_FUSED(SPH, FIRST)                                                                                                   \
  hipLaunchKernelGGL((k_shade_fused<SPH, FIRST>), grid, block, 0, s, scene, obj_begin, obj_end, in, out, hits, staged ? 1 : 0, \
                     bounce, last_bounce ? 1 : 0, slot_base, tile_desc, tile_stride, epoch, fb, band, counters, octs, bi, list)
  if (obj_begin < obj_end) {
    if (first) PT_FUSED(true, true);
    else PT_FUSED(true, false);
  } else if (first) {
    PT_FUSED(false, true);   // a scene without objects: every ray misses
  } else {
    PT_FUSED(false, false);  // (some closest-hit launch has written every record of the bounce)
  }
#undef PT_FUSED
}
uint32_t shade_tiles_per_frame(uint32_t max_paths) { return div_up(max_paths, kFuseTile); }
void launch_sort_octant(hipStream_t s, const uint8_t* octs, uint32_t* order, uint32_t max_paths, int bounce,
                        DeviceCounters* counters, const DBatchInfo& bi)
{
  hipLaunchKernelGGL(k_sort_octant, dim3(div_up(max_paths, kSortBlock), bi.count), dim3(1024), 0, s, octs, order, bounce, counters, bi);
}
void launch_accumulate(hipStream_t s, DFrame stage, DFrame fb, uint32_t pix_count, const DBatchInfo& bi)
{
  hipLaunchKernelGGL(k_accumulate, dim3(div_up(pix_count, 256u)), dim3(256), 0, s, stage, fb, pix_count, bi);
}
void launch_megakernel(hipStream_t s, const DScene& scene, const DCamera& cam, uint32_t iteration, DBand band,
                       uint32_t pix_count, int max_bounces, DFrame fb, DeviceCounters* counters)
{
  hipLaunchKernelGGL(k_megakernel, dim3(div_up(pix_count, kWave)), dim3(kWave), 0, s, scene, cam, iteration, band,
                     pix_count, max_bounces, fb, counters);
}
void launch_preview(hipStream_t s, const float4* buf, uint32_t pix_count, int mode, uint32_t* rgba)
{
  hipLaunchKernelGGL(k_preview, dim3(div_up(pix_count, 256u)), dim3(256), 0, s, buf, pix_count, mode, rgba);
}
void launch_gather_bands(hipStream_t s, const DGatherBands& bands, uint32_t count, uint32_t max_pix, int channels,
                         uint32_t frame_pixels, float* frame)
{
  const uint64_t floats = (uint64_t)max_pix * (uint32_t)channels;
  hipLaunchKernelGGL(k_gather_bands, dim3((uint32_t)((floats + 255u) / 256u), count), dim3(256), 0, s, bands, channels,
                     frame_pixels, frame);
}
void launch_preview_packed(hipStream_t s, const float* buf, uint32_t pix_count, int channels, int mode, uint32_t* rgba)
{
  hipLaunchKernelGGL(k_preview_packed, dim3(div_up(pix_count, 256u)), dim3(256), 0, s, buf, pix_count, channels, mode, rgba);
}
void launch_pack(hipStream_t s, const float4* buf, uint32_t pix_count, int which, float* dst)
{
  hipLaunchKernelGGL(k_pack, dim3(div_up(pix_count, 256u)), dim3(256), 0, s, buf, pix_count, which, dst);
}
void launch_denoise_positions(hipStream_t s, const DCamera& cam, uint32_t pix_count, const float4* nd, float4* pos)
{
  hipLaunchKernelGGL(k_denoise_positions, dim3(div_up(pix_count, 256u)), dim3(256), 0, s, cam, pix_count, nd, pos);
}
void launch_denoise_pass(hipStream_t s, const DCamera& cam, uint32_t pix_count, const float4* color, const float4* nd,
                         const float4* pos, float4* out, int step_width, DDenoise params)
{
  // (beyond step 32 the staged tile outgrows 64 KB of LDS: such filter sizes take the L1 / L2 kernel, and so does a
  // step that is no power of two -- ptc_denoise only issues 1, 2, 4, ...)
  if (params.variant == 1 || step_width > 32 || (step_width & (step_width - 1)) != 0) {  // taps through L1 / L2 (cross-check of the default)
    const uint32_t tiles = div_up(cam.width, 16u) * div_up(cam.height, 16u);
    hipLaunchKernelGGL(k_denoise, dim3(tiles), dim3(256), 0, s, cam, pix_count, color, nd, pos, out, step_width, params);
    return;
  }
  const uint32_t st = (uint32_t)step_width;
  const uint32_t total = div_up(cam.width, (uint32_t)kDenW) * div_up(div_up(cam.height, st), (uint32_t)kDenRows) * st;
  const dim3 grid(div_up(total, 8u) * 8u), block(256);  // one contiguous eighth of the tiles per XCD (see the kernel)
  const size_t lds = (size_t)(kDenW + 2 * kDenHalo * step_width) * (size_t)(kDenRows + 2 * kDenHalo) * 36u;
  switch (step_width) {
  case 1: hipLaunchKernelGGL(k_denoise_lds<1>, grid, block, lds, s, cam, pix_count, color, nd, pos, out, params); break;
  case 2: hipLaunchKernelGGL(k_denoise_lds<2>, grid, block, lds, s, cam, pix_count, color, nd, pos, out, params); break;
  case 4: hipLaunchKernelGGL(k_denoise_lds<4>, grid, block, lds, s, cam, pix_count, color, nd, pos, out, params); break;
  case 8: hipLaunchKernelGGL(k_denoise_lds<8>, grid, block, lds, s, cam, pix_count, color, nd, pos, out, params); break;
  case 16: hipLaunchKernelGGL(k_denoise_lds<16>, grid, block, lds, s, cam, pix_count, color, nd, pos, out, params); break;
  default: hipLaunchKernelGGL(k_denoise_lds<32>, grid, block, lds, s, cam, pix_count, color, nd, pos, out, params); break;
  }
}
void launch_intersect(hipStream_t s, const DScene& scene, const float4* rays_o, const float4* rays_d, uint32_t n,
                      DHits hits, DeviceCounters* counters, int variant)
{
  if (variant == 1)
    hipLaunchKernelGGL(k_intersect<1>, dim3(div_up(n, kWave)), dim3(kWave), 0, s, scene, rays_o, rays_d, n, hits, counters);
  else
    hipLaunchKernelGGL(k_intersect<0>, dim3(div_up(n, kWave)), dim3(kWave), 0, s, scene, rays_o, rays_d, n, hits, counters);
}
void launch_selftest(hipStream_t s, const float* a, const float* b, uint32_t n, float* out_div, float* out_sqrt,
                     float* out_sin, float* out_cos)
{
  hipLaunchKernelGGL(k_selftest, dim3(div_up(n, 256u)), dim3(256), 0, s, a, b, n, out_div, out_sqrt, out_sin, out_cos);
}

}  // namespace pt
