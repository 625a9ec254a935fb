// pt_bvh_rules.hpp -- the decisions of the reference's BVH builder (accelerators/bvh.cpp:60-209), as functions of
// the SET of triangles in a node.  One source for the host builder (pt_bvh.cpp) and the device builder
// (pt_bvh_gpu.hip): both are compiled with -ffp-contract=off and IEEE division, so they decide alike bit for bit.
#pragma once

#include <cfloat>

#include "pt_math.hpp"

namespace pt {
namespace bvh_rules {

constexpr int kBuckets = 12;  // bvh.cpp: bucket_count

struct Box {
  f3 lo, hi;
};
PT_HD Box empty_box() { return Box{mk3(FLT_MAX, FLT_MAX, FLT_MAX), mk3(-FLT_MAX, -FLT_MAX, -FLT_MAX)}; }
PT_HD Box grow(Box b, f3 p) { return Box{min3(b.lo, p), max3(b.hi, p)}; }
PT_HD Box merge(Box a, Box b) { return Box{min3(a.lo, b.lo), max3(a.hi, b.hi)}; }
PT_HD float area(const Box& b)  // AABB::surface_area, aabb.hpp
{
  const f3 d = b.hi - b.lo;
  return 2.0f * (d.x * d.y + d.x * d.z + d.y * d.z);
}
PT_HD int widest_axis(const Box& b)  // AABB::max_extent
{
  const f3 e = b.hi - b.lo;
  return (e.x > e.y && e.x > e.z) ? 0 : (e.y > e.z) ? 1 : 2;
}
PT_HD float comp(f3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

// triangle box and centroid (bvh.cpp:213-224: the centroid is the centre of the box)
PT_HD Box triangle_box(f3 p0, f3 p1, f3 p2) { return grow(grow(grow(empty_box(), p0), p1), p2); }
PT_HD f3 box_center(const Box& b) { return (b.lo + b.hi) / 2.0f; }

// normalised position of a centroid inside the centroid bounds along `axis` (AABB::offset, aabb.hpp:73-80)
PT_HD float offset_along(const Box& cb, f3 c, int axis)
{
  const float lo = comp(cb.lo, axis), hi = comp(cb.hi, axis);
  float o = comp(c, axis) - lo;
  if (hi > lo) o /= hi - lo;
  return o;
}
// SAH bucket of a centroid; outside [0, 11] only for non-finite input (the caller reports PTC_ERR_BVH)
PT_HD int bucket_of(const Box& cb, f3 c, int axis)
{
  int b = (int)((float)kBuckets * offset_along(cb, c, axis));
  if (b == kBuckets) b = kBuckets - 1;
  return b;
}
// the split "buckets 0..s | s+1..11" of least cost .125 + (n0*A0 + n1*A1)/A, first minimum (bvh.cpp:148-176)
PT_HD int sah_best_split(const int* count, const Box* bounds, const Box& all)
{
  const float all_area = area(all);
  int best = 0;
  float best_cost = 0.0f;
  for (int s = 0; s < kBuckets - 1; ++s) {
    Box b0 = empty_box(), b1 = empty_box();
    int c0 = 0, c1 = 0;
    for (int j = 0; j <= s; ++j) {
      b0 = merge(b0, bounds[j]);
      c0 += count[j];
    }
    for (int j = s + 1; j < kBuckets; ++j) {
      b1 = merge(b1, bounds[j]);
      c1 += count[j];
    }
    const float cost = .125f + ((float)c0 * area(b0) + (float)c1 * area(b1)) / all_area;
    if (s == 0 || cost < best_cost) {
      best_cost = cost;
      best = s;
    }
  }
  return best;
}
// order of the triangles of a node of <= 4 (split at the median, n / 2): by centroid along the axis, equal centroids
// by triangle index (the reference leaves ties to std::nth_element)
PT_HD bool small_before(float key_a, uint32_t a, float key_b, uint32_t b) { return key_a < key_b || (key_a == key_b && a < b); }

}  // namespace bvh_rules
}  // namespace pt
