// pt_bvh.cpp -- host-side BVH builder of the render core.
//
// Produces exactly the node array the reference's bvh_from_mesh (accelerators/bvh.cpp:211-253) produces:
// one triangle per leaf; split axis = largest extent of the centroid bounds; 2 leaves ordered by centroid;
// <= 4 leaves split at the median; otherwise 12-bucket SAH (cost .125 + (n0*A0 + n1*A1)/A, first minimum,
// partition by bucket <= split); nodes numbered breadth-first, root 0, children adjacent (left, left+1).
// Unlike the reference (one shared_ptr node per triangle, serial), this builder works on index ranges
// in place and builds large subtrees on worker threads; the result does not depend on the thread count
// because every decision is a function of the SET of triangles in a range.
// Where the reference leaves the order of equal centroids to std::nth_element, ties go to the lower
// triangle index.
#include "pt_bvh_rules.hpp"
#include "pt_host.hpp"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cstring>
#include <thread>
#include <vector>

namespace pt {
using namespace bvh_rules;
namespace {

struct TmpNode {
  Box box;
  int32_t left, right;  // -1: leaf
  uint32_t tri;
};

struct Builder {
  const std::vector<Box>& tri_box;
  const std::vector<f3>& tri_center;
  std::vector<TmpNode>& nodes;     // preallocated 2T-1
  std::atomic<uint32_t>& next;     // node allocator
  std::atomic<int>& error;

  uint32_t leaf(uint32_t t)
  {
    const uint32_t id = next.fetch_add(1u);
    nodes[id] = TmpNode{tri_box[t], -1, -1, t};
    return id;
  }
  uint32_t inner(uint32_t l, uint32_t r)
  {
    const uint32_t id = next.fetch_add(1u);
    nodes[id] = TmpNode{merge(nodes[l].box, nodes[r].box), (int32_t)l, (int32_t)r, 0u};
    return id;
  }

  // Builds the subtree over tris[0..n); returns its node id or -1.
  int32_t build(uint32_t* tris, uint32_t n, int spawn_depth)
  {
    if (error.load(std::memory_order_relaxed)) return -1;
    if (n == 0) {
      error = PTC_ERR_BVH;
      return -1;
    }
    if (n == 1) return (int32_t)leaf(tris[0]);

    Box cb = empty_box();
    for (uint32_t i = 0; i < n; ++i) cb = grow(cb, tri_center[tris[i]]);
    const int axis = widest_axis(cb);

    if (n == 2) {
      uint32_t a = tris[0], b = tris[1];
      const float ka = comp(tri_center[a], axis), kb = comp(tri_center[b], axis);
      if (ka > kb || (ka == kb && a > b)) std::swap(a, b);  // tie: lower triangle index goes left
      const uint32_t l = leaf(a), r = leaf(b);
      return (int32_t)inner(l, r);
    }

    uint32_t mid;
    if (n <= 4) {
      std::sort(tris, tris + n, [&](uint32_t a, uint32_t b) {
        return small_before(comp(tri_center[a], axis), a, comp(tri_center[b], axis), b);
      });
      mid = n / 2;
    } else {
      int count[kBuckets] = {};
      Box bounds[kBuckets];
      for (auto& b : bounds) b = empty_box();
      Box all = empty_box();
      for (uint32_t i = 0; i < n; ++i) {
        const int b = bucket_of(cb, tri_center[tris[i]], axis);
        if (b < 0 || b >= kBuckets) {
          error = PTC_ERR_BVH;
          return -1;
        }
        ++count[b];
        bounds[b] = merge(bounds[b], tri_box[tris[i]]);
        all = merge(all, tri_box[tris[i]]);
      }
      const int best = sah_best_split(count, bounds, all);
      uint32_t* split = std::partition(tris, tris + n, [&](uint32_t t) { return bucket_of(cb, tri_center[t], axis) <= best; });
      mid = (uint32_t)(split - tris);
      if (mid == 0 || mid == n) {  // reference: panic("Shouldn't happen!"), bvh.cpp:84-85
        error = PTC_ERR_BVH;
        return -1;
      }
    }

    int32_t l = -1, r = -1;
    if (spawn_depth > 0 && n > 32768u) {
      std::thread worker([&] { l = build(tris, mid, spawn_depth - 1); });
      r = build(tris + mid, n - mid, spawn_depth - 1);
      worker.join();
    } else {
      l = build(tris, mid, 0);
      r = build(tris + mid, n - mid, 0);
    }
    if (l < 0 || r < 0) return -1;
    return (int32_t)inner((uint32_t)l, (uint32_t)r);
  }
};

}  // namespace

int build_bvh(const float* positions, uint32_t vertex_count, const uint32_t* indices, uint32_t index_count,
              ptc_bvh_node* out, uint32_t* max_depth)
{
  const uint32_t T = index_count / 3u;
  if (T == 0) return PTC_ERR_BVH;
  for (uint32_t i = 0; i < T * 3u; ++i)
    if (indices[i] >= vertex_count) return PTC_ERR_INVALID;

  std::vector<Box> tri_box(T);
  std::vector<f3> tri_center(T);
  std::vector<uint32_t> tris(T);
  for (uint32_t t = 0; t < T; ++t) {
    const float* p0 = positions + 3u * (size_t)indices[3u * t];
    const float* p1 = positions + 3u * (size_t)indices[3u * t + 1u];
    const float* p2 = positions + 3u * (size_t)indices[3u * t + 2u];
    tri_box[t] = triangle_box(mk3(p0[0], p0[1], p0[2]), mk3(p1[0], p1[1], p1[2]), mk3(p2[0], p2[1], p2[2]));
    tri_center[t] = box_center(tri_box[t]);
    tris[t] = t;
  }

  std::vector<TmpNode> nodes(2u * (size_t)T);
  std::atomic<uint32_t> next{0u};
  std::atomic<int> error{0};
  Builder builder{tri_box, tri_center, nodes, next, error};
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  int spawn_depth = 0;
  while ((1u << spawn_depth) < hw && spawn_depth < 6) ++spawn_depth;
  const int32_t root = builder.build(tris.data(), T, spawn_depth);
  if (root < 0) return error.load() ? error.load() : PTC_ERR_BVH;

  // breadth-first numbering (bvh.cpp:228-250): node k of the queue gets linear index k
  const uint32_t total = 2u * T - 1u;
  std::vector<uint32_t> queue(total), depth(total);
  uint32_t head = 0, tail = 0, deepest = 0;
  queue[tail] = (uint32_t)root;
  depth[tail] = 0;
  ++tail;
  auto emit = [&](uint32_t slot, const TmpNode& n) {
    ptc_bvh_node& o = out[slot];
    o.aabb_min[0] = n.box.lo.x; o.aabb_min[1] = n.box.lo.y; o.aabb_min[2] = n.box.lo.z;
    o.aabb_max[0] = n.box.hi.x; o.aabb_max[1] = n.box.hi.y; o.aabb_max[2] = n.box.hi.z;
    o.first_child_or_primitive = n.left < 0 ? n.tri * 3u : 0u;
    o.primitive_count = n.left < 0 ? 1u : 0u;
  };
  emit(0, nodes[root]);
  while (head < tail) {
    const TmpNode& cur = nodes[queue[head]];
    deepest = std::max(deepest, depth[head]);
    if (cur.left >= 0) {
      out[head].first_child_or_primitive = tail;
      emit(tail, nodes[cur.left]);
      queue[tail] = (uint32_t)cur.left;
      depth[tail] = depth[head] + 1;
      ++tail;
      emit(tail, nodes[cur.right]);
      queue[tail] = (uint32_t)cur.right;
      depth[tail] = depth[head] + 1;
      ++tail;
    }
    ++head;
  }
  if (max_depth) *max_depth = deepest;
  return (int)tail;
}

}  // namespace pt
