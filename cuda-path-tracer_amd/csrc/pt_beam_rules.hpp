// pt_beam_rules.hpp -- entry points for primary rays ("beam", DBeam in pt_device.hpp): what k_beam computes per tile,
// as __host__ __device__ functions so that the host can check it without a GPU (ptc_check_beam, tests/test_abi_cpu.py):
// the tile's frustum, "is this box out of its reach", and the frontier of at most kBeamEntries nodes of the four-wide
// quantised tree (pt_layout_rules.hpp: 16 dwords per node) the frustum overlaps.
#pragma once

#include "pt_math.hpp"

namespace pt {
namespace beam_rules {

constexpr uint32_t kLeaf = 0x80000000u;  // == pt::kLeafBit
constexpr int kEntries = 4;              // == pt::kBeamEntries

PT_HD float as_float(uint32_t u) { return __builtin_bit_cast(float, u); }
PT_HD bool finite1(float x) { return __builtin_fabsf(x) < __builtin_inff(); }

// Four planes through `eye`, each spanned by two neighbouring corner directions of the tile's pixel rectangle and turned
// so that the centre direction `dc` lies inside (n . dc >= 0).  Inside = the forward pyramid only.
struct Frustum {
  f3 eye;
  f3 n[4];
  bool usable;  // false (a degenerate camera): nothing is ever out of reach
};
PT_HD Frustum make_frustum(const f3 eye, const f3 d00, const f3 d10, const f3 d01, const f3 d11, const f3 dc)
{
  Frustum f;
  f.eye = eye;
  f.n[0] = cross(d00, d10);
  f.n[1] = cross(d10, d11);
  f.n[2] = cross(d11, d01);
  f.n[3] = cross(d01, d00);
  f.usable = finite1(eye.x + eye.y + eye.z);
  for (int k = 0; k < 4; ++k) {
    if (dot(f.n[k], dc) < 0.0f) f.n[k] = -f.n[k];
    f.usable = f.usable && finite1(f.n[k].x + f.n[k].y + f.n[k].z) && dot(f.n[k], f.n[k]) > 0.0f;
  }
  return f;
}
// is the box surely out of the frustum's reach?  Its corner farthest along a plane's normal is outside that plane by more
// than the margin (1e-4 of the terms: the rays test boxes with tolerant slabs of 4e-6, and a ray of the tile lies inside
// the pyramid up to the rounding of its own direction).
PT_HD bool outside(const Frustum& f, const f3 lo, const f3 hi)
{
  if (!f.usable) return false;
  bool out_ = false;
  for (int k = 0; k < 4; ++k) {
    const f3 n = f.n[k];
    const f3 p = mk3((n.x >= 0.0f ? hi.x : lo.x) - f.eye.x, (n.y >= 0.0f ? hi.y : lo.y) - f.eye.y, (n.z >= 0.0f ? hi.z : lo.z) - f.eye.z);
    const float tx = n.x * p.x, ty = n.y * p.y, tz = n.z * p.z;
    const float v = (tx + ty) + tz;
    const float m = 1e-4f * ((__builtin_fabsf(tx) + __builtin_fabsf(ty)) + __builtin_fabsf(tz)) + 1e-30f;
    out_ = out_ || v < -m;
  }
  return out_;
}

// child c of a four-wide node: its quantised box, a millionth larger; false for an unused slot (inside-out box)
PT_HD bool child_box(const uint32_t* q, int c, f3& lo, f3& hi)
{
  const uint32_t lx = (q[4] >> (8 * c)) & 0xffu, ly = (q[5] >> (8 * c)) & 0xffu, lz = (q[6] >> (8 * c)) & 0xffu;
  const uint32_t hx = (q[7] >> (8 * c)) & 0xffu, hy = (q[8] >> (8 * c)) & 0xffu, hz = (q[9] >> (8 * c)) & 0xffu;
  if (lx > hx || ly > hy || lz > hz) return false;
  const f3 org = mk3(as_float(q[0]), as_float(q[1]), as_float(q[2]));
  const f3 step = mk3(as_float(q[3]), as_float(q[10]), as_float(q[11]));
  lo = mk3(org.x + (float)lx * step.x, org.y + (float)ly * step.y, org.z + (float)lz * step.z);
  hi = mk3(org.x + (float)hx * step.x, org.y + (float)hy * step.y, org.z + (float)hz * step.z);
  const f3 pad = mk3(1e-6f * (__builtin_fabsf(lo.x) + __builtin_fabsf(hi.x)) + 1e-30f, 1e-6f * (__builtin_fabsf(lo.y) + __builtin_fabsf(hi.y)) + 1e-30f,
                     1e-6f * (__builtin_fabsf(lo.z) + __builtin_fabsf(hi.z)) + 1e-30f);
  lo = lo - pad;
  hi = hi + pad;
  return true;
}

// The tile's entries: start with the root; while there is room, the largest inner entry is replaced by those of its
// children the frustum reaches.  nodes_q: 16 dwords per node; node_count: for the range check of a reference (a tree
// that fails it is left at the entries found so far).  Returns the number of entries (0: the frustum reaches nothing).
PT_HD int tile_entries(const uint32_t* nodes_q, uint32_t node_count, uint32_t root_ref, const f3 root_lo, const f3 root_hi, const Frustum& fr,
                       f3* lo4, f3* hi4, uint32_t* ref4)
{
  int n = 0;
  if (!outside(fr, root_lo, root_hi)) {
    lo4[0] = root_lo;
    hi4[0] = root_hi;
    ref4[0] = root_ref;
    n = 1;
  }
  for (int round = 0; round < 64; ++round) {
    int pick = -1;
    float big = -1.0f;
    for (int k = 0; k < kEntries; ++k)
      if (k < n && (ref4[k] & kLeaf) == 0u) {
        const f3 e = hi4[k] - lo4[k];
        const float area = (e.x * e.y + e.x * e.z) + e.y * e.z;
        if (area > big) {
          big = area;
          pick = k;
        }
      }
    if (pick < 0 || ref4[pick] >= node_count) break;
    const uint32_t* q = nodes_q + 16u * (size_t)ref4[pick];
    f3 klo[4], khi[4];
    uint32_t kref[4];
    int kids = 0;
    for (int c = 0; c < 4; ++c) {
      f3 lo, hi;
      if (!child_box(q, c, lo, hi) || outside(fr, lo, hi)) continue;
      klo[kids] = lo;
      khi[kids] = hi;
      kref[kids++] = q[12 + c];
    }
    if (n - 1 + kids > kEntries) break;
    for (int k = 0; k < kEntries - 1; ++k)
      if (k >= pick && k + 1 < n) {
        lo4[k] = lo4[k + 1];
        hi4[k] = hi4[k + 1];
        ref4[k] = ref4[k + 1];
      }
    --n;
    for (int c = 0; c < 4; ++c)
      if (c < kids) {
        lo4[n] = klo[c];
        hi4[n] = khi[c];
        ref4[n] = kref[c];
        ++n;
      }
  }
  return n;
}

}  // namespace beam_rules
}  // namespace pt
