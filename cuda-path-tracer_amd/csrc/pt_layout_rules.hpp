// pt_layout_rules.hpp -- how the reference BVH becomes the four-wide quantised tree of the persistent traversal:
// the cost recurrence of the optimal collapse, the choice of a node's children, and the 64-byte node encoding.  One
// source for the host layout (pt_scene_host.cpp) and the device layout (pt_bvh_gpu.hip), so both emit the same bytes.
#pragma once

#include <cmath>

#include "pt_math.hpp"

namespace pt {
namespace layout_rules {

// best[0..3] of inner node x from its children's: least summed area of four-wide nodes when the subtree of x is
// represented by at most 1, 2, 3, 4 roots (a root = a leaf, or an inner node that becomes a four-wide node of its own);
// a leaf's four entries are 0 (its test is paid in its parent)
PT_HD void collapse_costs(float area_x, const float* bl, const float* br, float* b)
{
  float node_cost = bl[0] + br[2];
  node_cost = sel_min(node_cost, bl[1] + br[1]);
  node_cost = sel_min(node_cost, bl[2] + br[0]);
  b[0] = area_x + node_cost;                                       // x as a four-wide node of its own
  b[1] = sel_min(b[0], bl[0] + br[0]);                             // ... or dissolved into 2, 3, 4 roots
  b[2] = sel_min(b[1], sel_min(bl[0] + br[1], bl[1] + br[0]));
  b[3] = sel_min(b[2], node_cost);
}

// The children of four-wide node x: the cut of its subtree into at most four reference nodes that the recurrence
// chose, left to right (= depth-first order).  Tree: is_leaf(x), first(x) (left child; right = first + 1);
// best: 4 floats per node.
template <class Tree>
PT_HD int choose_children(const Tree& tree, const float* best, uint32_t x, uint32_t kids[4])
{
  int nk = 0;
  // pending (node, budget) pairs, leftmost on top; the root entry is opened unconditionally
  uint32_t stack_x[4];
  int stack_k[4];
  int sp = 0;
  stack_x[sp] = x;
  stack_k[sp++] = 4;
  bool top = true;
  while (sp > 0) {
    const uint32_t y = stack_x[--sp];
    const int k = stack_k[sp];
    const float* by = best + 4u * (size_t)y;
    if (!top && (k <= 1 || tree.is_leaf(y) || by[k - 1] >= by[0])) {
      kids[nk++] = y;
      continue;
    }
    top = false;
    const uint32_t l = tree.first(y);
    const float* bl = best + 4u * (size_t)l;
    const float* br = bl + 4;
    int pick = 1;
    for (int j = 2; j < k; ++j)
      if (bl[j - 1] + br[k - j - 1] < bl[pick - 1] + br[k - pick - 1]) pick = j;
    stack_x[sp] = l + 1u;
    stack_k[sp++] = k - pick;
    stack_x[sp] = l;
    stack_k[sp++] = pick;
  }
  return nk;
}

// 64-byte form of a four-wide node (Wide4Accel::nodes_q, DScene::bvh4q).  Grid: origin = the node's lower corner,
// step 2^e per axis with 255 steps covering the node's extent; a child's lower planes round down, its upper planes up
// (checked in double precision, where origin + q * step is exact), so the walk over these boxes stays conservative.
// lo / hi: [axis][child]; unused slots (k >= nk) get an inside-out box (lower planes at the top of the grid, upper
// planes at its bottom): its slab interval is empty for every ray unless the whole node is smaller than the walk's
// error bound, and then the slot's reference leads to the dummy triangle, which no ray hits -- the kernel needs no
// "is this slot used" test.
PT_HD void quantise_node(uint32_t* q, int nk, const float lo[3][4], const float hi[3][4], const uint32_t refs[4])
{
  uint32_t exps = 0u;
  for (int a = 0; a < 3; ++a) {
    float origin = __builtin_inff(), top = -__builtin_inff();
    for (int k = 0; k < nk; ++k) {
      origin = lo[a][k] < origin ? lo[a][k] : origin;
      top = hi[a][k] > top ? hi[a][k] : top;
    }
    const double extent = (double)top - (double)origin;
    int e = -126;
    if (extent > 0.0) {
      // ceil(log2(extent / 255)) without a logarithm (exact, the same on host and device)
      int k2 = 0;
      const double mant = frexp(extent / 255.0, &k2);
      e = mant == 0.5 ? k2 - 1 : k2;
      e = e < -126 ? -126 : (e > 127 ? 127 : e);
    }
    for (;;) {  // grow the step until every plane fits 0..255 (one pass almost always)
      const double step = ldexp(1.0, e);
      bool ok = true;
      uint32_t lo_q = 0u, hi_q = 0u;
      for (int k = 0; k < 4; ++k) {
        uint32_t ql = 255u, qh = 0u;
        if (k < nk) {
          const double fl = floor(((double)lo[a][k] - (double)origin) / step);
          const double ce = ceil(((double)hi[a][k] - (double)origin) / step);
          if (fl < 0.0 || ce > 255.0 || fl > 255.0) ok = false;
          ql = (uint32_t)(fl < 0.0 ? 0.0 : (fl > 255.0 ? 255.0 : fl));
          qh = (uint32_t)(ce < 0.0 ? 0.0 : (ce > 255.0 ? 255.0 : ce));
          if ((double)origin + ql * step > (double)lo[a][k] || (double)origin + qh * step < (double)hi[a][k]) ok = false;
        }
        lo_q |= ql << (8 * k);
        hi_q |= qh << (8 * k);
      }
      if (ok || e >= 127) {
        q[4 + a] = lo_q;
        q[7 + a] = hi_q;
        break;
      }
      ++e;
    }
    q[a] = __builtin_bit_cast(uint32_t, origin);
    exps |= (uint32_t)(e + 127) << (8 * a);
  }
  // the three grid steps as ready-made floats (exponent field only): dword 3 = step x, dwords 10, 11 = step y, z
  q[3] = (exps & 0xffu) << 23;
  q[10] = ((exps >> 8) & 0xffu) << 23;
  q[11] = ((exps >> 16) & 0xffu) << 23;
  for (int k = 0; k < 4; ++k) q[12 + k] = refs[k];
}

// world-space record of one triangle of an instance (DScene::tris): {p0.xyz, e1.x} {e1.yz, e2.xy} {e2.z, n.xyz}
PT_HD void instance_triangle(const m4& m, f3 q0, f3 q1, f3 q2, float4* out)
{
  const f3 p0 = xform_point(m, q0), p1 = xform_point(m, q1), p2 = xform_point(m, q2);
  const f3 e1 = p1 - p0;
  const f3 e2 = p2 - p0;
  const f3 n = normalize(cross(e1, e2));
  out[0] = make_float4(p0.x, p0.y, p0.z, e1.x);
  out[1] = make_float4(e1.y, e1.z, e2.x, e2.y);
  out[2] = make_float4(e2.z, n.x, n.y, n.z);
}

}  // namespace layout_rules
}  // namespace pt
