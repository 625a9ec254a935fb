// pt_wide8.cpp -- the reference BVH collapsed to EIGHT children per node for the persistent traversal (k_traverse8):
// which reference nodes survive (dynamic programme over summed surface area), which child sits in which of the
// eight slots (by octant, so that a ray's visiting order is slot XOR its direction signs), where children live in
// memory (inner children of a node consecutive, leaf children's triangle records consecutive: the node names them by
// two base indices and two 8-bit masks), and the 80-byte quantised record.  Host code, no GPU.
//
// Why this is allowed to differ from the reference tree in every respect but its leaves: the walk only has to be
// conservative -- the winner's reachability in the REFERENCE tree is tested afterwards, exactly,
// against the box of its reference parent (leaf_parent), and ties are broken by the reference's depth-first rank,
// which every triangle record carries.
#include "pt_device.hpp"
#include "pt_host.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace pt {

namespace {

constexpr int K = 8;

float area_of(const ptc_bvh_node& n)
{
  const float dx = n.aabb_max[0] - n.aabb_min[0], dy = n.aabb_max[1] - n.aabb_min[1], dz = n.aabb_max[2] - n.aabb_min[2];
  return 2.0f * (dx * dy + dx * dz + dy * dz);
}

struct Builder {
  const ptc_bvh_node* nodes;
  uint32_t count;
  std::vector<float> best;  // [x * K + (j - 1)]: least summed area of kept inner nodes, subtree of x as at most j roots

  float cost(uint32_t x, int j) const { return best[(size_t)x * K + (size_t)(j - 1)]; }

  void solve()
  {
    best.assign((size_t)count * K, 0.0f);
    for (uint32_t x = count; x-- > 0u;) {
      if (nodes[x].primitive_count != 0u) continue;  // a leaf costs nothing here (its box test is paid in its parent)
      const uint32_t l = nodes[x].first_child_or_primitive;
      float comb[K + 1];
      for (int j = 2; j <= K; ++j) {
        comb[j] = FLT_MAX;
        for (int a = 1; a < j; ++a) comb[j] = std::min(comb[j], cost(l, a) + cost(l + 1u, j - a));
      }
      float* b = &best[(size_t)x * K];
      b[0] = area_of(nodes[x]) + comb[K];  // x as a node of its own
      for (int j = 2; j <= K; ++j) b[j - 1] = std::min(b[j - 2], comb[j]);
    }
  }
  void expand(uint32_t x, int j, std::vector<uint32_t>& out) const
  {
    if (j <= 1 || nodes[x].primitive_count != 0u || cost(x, j) >= cost(x, 1)) {
      out.push_back(x);
      return;
    }
    const uint32_t l = nodes[x].first_child_or_primitive;
    int bj = 1;
    for (int a = 2; a < j; ++a)
      if (cost(l, a) + cost(l + 1u, j - a) < cost(l, bj) + cost(l + 1u, j - bj)) bj = a;
    expand(l, bj, out);
    expand(l + 1u, j - bj, out);
  }
  // the (at most eight) reference nodes that become the children of inner reference node x
  std::vector<uint32_t> children_of(uint32_t x) const
  {
    std::vector<uint32_t> kids;
    const uint32_t l = nodes[x].first_child_or_primitive;
    int bj = 1;
    for (int a = 2; a < K; ++a)
      if (cost(l, a) + cost(l + 1u, K - a) < cost(l, bj) + cost(l + 1u, K - bj)) bj = a;
    expand(l, bj, kids);
    expand(l + 1u, K - bj, kids);
    return kids;
  }
};

// children -> slots: slot s stands for the octant (bit 0: +x, bit 1: +y, bit 2: +z) of the node the child lies in; a ray
// visits slots in ascending (slot XOR direction-sign bits).  Greedy: best remaining (child, slot) affinity first.
void assign_slots(const ptc_bvh_node* nodes, const std::vector<uint32_t>& kids, int slot_of[K])
{
  float centre[3] = {0, 0, 0};
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t k : kids)
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::min(lo[a], nodes[k].aabb_min[a]);
      hi[a] = std::max(hi[a], nodes[k].aabb_max[a]);
    }
  for (int a = 0; a < 3; ++a) centre[a] = 0.5f * (lo[a] + hi[a]);
  const int n = (int)kids.size();
  float aff[K][K];
  for (int c = 0; c < n; ++c)
    for (int s = 0; s < K; ++s) {
      float v = 0.0f;
      for (int a = 0; a < 3; ++a) {
        const float rel = 0.5f * (nodes[kids[c]].aabb_min[a] + nodes[kids[c]].aabb_max[a]) - centre[a];
        v += ((s >> a) & 1) ? rel : -rel;
      }
      aff[c][s] = v;
    }
  bool child_done[K] = {}, slot_done[K] = {};
  for (int round = 0; round < n; ++round) {
    int bc = -1, bs = -1;
    for (int c = 0; c < n; ++c)
      if (!child_done[c])
        for (int s = 0; s < K; ++s)
          if (!slot_done[s] && (bc < 0 || aff[c][s] > aff[bc][bs])) {
            bc = c;
            bs = s;
          }
    child_done[bc] = true;
    slot_done[bs] = true;
    slot_of[bc] = bs;
  }
}

}  // namespace

int build_wide8(const ptc_bvh_node* nodes, uint32_t count, Wide8Accel& out)
{
  out = Wide8Accel{};
  if (count == 0u) return PTC_OK;
  // depth-first (left-first) leaf ranks and parents in the reference tree
  std::vector<uint32_t> leaf_rank(count, 0u), parent(count, 0xffffffffu);
  for (uint32_t i = 0; i < count; ++i)
    if (nodes[i].primitive_count == 0u) {
      if ((uint64_t)nodes[i].first_child_or_primitive + 1u >= count) return PTC_ERR_BVH;
      parent[nodes[i].first_child_or_primitive] = i;
      parent[nodes[i].first_child_or_primitive + 1u] = i;
    }
  {
    uint32_t leaves = 0;
    std::vector<uint32_t> stack{0u};
    while (!stack.empty()) {
      const uint32_t i = stack.back();
      stack.pop_back();
      if (nodes[i].primitive_count != 0u) leaf_rank[i] = leaves++;
      else {
        stack.push_back(nodes[i].first_child_or_primitive + 1u);
        stack.push_back(nodes[i].first_child_or_primitive);
      }
    }
  }
  Builder b{nodes, count, {}};
  b.solve();

  // breadth-first emission: a node's inner children get consecutive node indices, its leaf children consecutive records
  struct Pending {
    uint32_t ref_node;  // reference inner node this wide node stands for (0xffffffff: the synthetic root of a one-leaf tree)
    uint32_t level;
  };
  std::vector<Pending> queue;
  queue.push_back({nodes[0].primitive_count != 0u ? 0xffffffffu : 0u, 1u});
  const float big = 3.402823466e+38f;
  for (size_t qi = 0; qi < queue.size(); ++qi) {
    const Pending p = queue[qi];
    out.depth = std::max(out.depth, p.level);
    std::vector<uint32_t> kids;
    if (p.ref_node == 0xffffffffu) kids.push_back(0u);
    else kids = b.children_of(p.ref_node);
    int slot_of[K];
    assign_slots(nodes, kids, slot_of);
    uint32_t in_slot[K];
    for (int s = 0; s < K; ++s) in_slot[s] = 0xffffffffu;
    for (size_t c = 0; c < kids.size(); ++c) in_slot[slot_of[c]] = kids[c];
    const uint32_t child_base = (uint32_t)queue.size(), tri_base = (uint32_t)out.tri_of_record.size();
    uint32_t imask = 0u, lmask = 0u;
    float lo[3][K], hi[3][K];
    for (int s = 0; s < K; ++s) {
      for (int a = 0; a < 3; ++a) lo[a][s] = hi[a][s] = 0.0f;
      const uint32_t x = in_slot[s];
      if (x == 0xffffffffu) continue;
      for (int a = 0; a < 3; ++a) {
        lo[a][s] = nodes[x].aabb_min[a];
        hi[a][s] = nodes[x].aabb_max[a];
      }
      if (nodes[x].primitive_count != 0u) {
        lmask |= 1u << s;
        out.tri_of_record.push_back(nodes[x].first_child_or_primitive / 3u);
        out.rank_of_record.push_back(leaf_rank[x]);
        const bool has_parent = parent[x] != 0xffffffffu;
        const ptc_bvh_node& pp = nodes[has_parent ? parent[x] : x];
        // (a single-triangle mesh has no inner node at all: every box test of the reference is vacuous)
        out.leaf_parent.push_back(has_parent ? make_float4(pp.aabb_min[0], pp.aabb_min[1], pp.aabb_min[2], 0.f) : make_float4(-big, -big, -big, 0.f));
        out.leaf_parent.push_back(has_parent ? make_float4(pp.aabb_max[0], pp.aabb_max[1], pp.aabb_max[2], 0.f) : make_float4(big, big, big, 0.f));
      } else {
        imask |= 1u << s;
        queue.push_back({x, p.level + 1u});
      }
    }
    // the 80-byte record (DScene::bvh8): grid origin = the node's lower corner, step 2^e per axis with 255 steps
    // covering the extent; a child's lower planes round down, its upper planes up (checked in double precision)
    uint32_t q[kNode8Dwords];
    std::memset(q, 0, sizeof q);
    uint32_t plane_lo[3][2] = {}, plane_hi[3][2] = {}, exps = 0u;
    for (int a = 0; a < 3; ++a) {
      float origin = std::numeric_limits<float>::infinity(), top = -std::numeric_limits<float>::infinity();
      for (int s = 0; s < K; ++s)
        if ((imask | lmask) >> s & 1u) {
          origin = std::min(origin, lo[a][s]);
          top = std::max(top, hi[a][s]);
        }
      const double extent = (double)top - (double)origin;
      int e = -126;
      if (extent > 0.0) e = std::max(-126, std::min(127, (int)std::ceil(std::log2(extent / 255.0))));
      for (;;) {  // grow the step until every plane fits 0..255 (one pass almost always)
        const double step = std::ldexp(1.0, e);
        bool ok = true;
        plane_lo[a][0] = plane_lo[a][1] = plane_hi[a][0] = plane_hi[a][1] = 0u;
        for (int s = 0; s < K; ++s) {
          uint32_t ql = 255u, qh = 0u;  // unused slot: inside-out
          if ((imask | lmask) >> s & 1u) {
            const double fl = std::floor(((double)lo[a][s] - (double)origin) / step);
            const double ce = std::ceil(((double)hi[a][s] - (double)origin) / step);
            if (fl < 0.0 || ce > 255.0 || fl > 255.0) ok = false;
            ql = (uint32_t)std::max(0.0, std::min(255.0, fl));
            qh = (uint32_t)std::max(0.0, std::min(255.0, ce));
            if ((double)origin + ql * step > (double)lo[a][s] || (double)origin + qh * step < (double)hi[a][s]) ok = false;
          }
          plane_lo[a][s >> 2] |= ql << (8 * (s & 3));
          plane_hi[a][s >> 2] |= qh << (8 * (s & 3));
        }
        if (ok || e >= 127) break;
        ++e;
      }
      std::memcpy(&q[a], &origin, 4);
      exps |= (uint32_t)(e + 127) << (8 * a);
    }
    q[3] = exps | (imask << 24);
    q[4] = child_base;
    q[5] = tri_base;
    q[6] = lmask;
    for (int a = 0; a < 3; ++a) {
      q[7 + 2 * a] = plane_lo[a][0];
      q[8 + 2 * a] = plane_lo[a][1];
      q[13 + 2 * a] = plane_hi[a][0];
      q[14 + 2 * a] = plane_hi[a][1];
    }
    out.nodes.insert(out.nodes.end(), q, q + kNode8Dwords);
  }
  out.node_count = (uint32_t)queue.size();
  if (out.node_count > 0x00ffffffu || out.tri_of_record.size() > 0x7fffffffu) return PTC_ERR_INVALID;
  return PTC_OK;
}

void build_instance_triangles8(const m4& m, const float* positions, const uint32_t* indices, const Wide8Accel& w8, float4* out)
{
  for (size_t k = 0; k < w8.tri_of_record.size(); ++k) {
    const uint32_t* idx = indices + 3u * (size_t)w8.tri_of_record[k];
    f3 p[3];
    for (int v = 0; v < 3; ++v) {
      const float* q = positions + 3u * (size_t)idx[v];
      p[v] = xform_point(m, mk3(q[0], q[1], q[2]));
    }
    const f3 e1 = p[1] - p[0];
    const f3 e2 = p[2] - p[0];
    float rank_bits;
    std::memcpy(&rank_bits, &w8.rank_of_record[k], 4);
    out[3u * k] = make_float4(p[0].x, p[0].y, p[0].z, e1.x);
    out[3u * k + 1u] = make_float4(e1.y, e1.z, e2.x, e2.y);
    out[3u * k + 2u] = make_float4(e2.z, rank_bits, 0.0f, 0.0f);
  }
}

}  // namespace pt
