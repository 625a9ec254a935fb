// walk_sim.cpp -- CPU model of the conservative closest-hit walk over k-wide collapses of the reference BVH:
// how many node visits / box tests / triangle tests per ray does a given width and child order cost?
// (Design tool for the traversal kernel; reads the case written by tools/sim/dump_case.py.  No GPU, no parity claim:
// float slabs without quantisation, Moeller-Trumbore on the raw vertices.)
//   g++ -O2 -std=c++17 -o /tmp/walk_sim tools/sim/walk_sim.cpp && /tmp/walk_sim /tmp/sim_case.bin
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Node { float mn[3], mx[3]; uint32_t first, count; };
struct Ray { float o[3], tmin, d[3], tmax; };

static std::vector<Node> nodes;
static std::vector<uint32_t> indices;
static std::vector<float> positions;
static std::vector<Ray> rays;
static uint32_t per_bounce[8];

struct WNode {           // k-wide node: children = reference node ids
  int nk;
  uint32_t kid[8];
};
static std::vector<WNode> wnodes;            // indexed by wide id
static std::vector<int> wide_of;             // reference inner node -> wide id (or -1)

static float area(const Node& n)
{
  const float dx = n.mx[0] - n.mn[0], dy = n.mx[1] - n.mn[1], dz = n.mx[2] - n.mn[2];
  return 2.0f * (dx * dy + dx * dz + dy * dz);
}

// least summed surface area of the kept inner nodes, subtree of x represented by at most j roots (j = 1..K)
static int K;
static std::vector<float> best;  // [x * K + (j-1)]
static float cost(uint32_t x, int j) { return best[(size_t)x * K + (j - 1)]; }

static void expand(uint32_t x, int j, std::vector<uint32_t>& out)
{
  if (j <= 1 || nodes[x].count != 0 || cost(x, j) >= cost(x, 1)) { out.push_back(x); return; }
  const uint32_t l = nodes[x].first;
  int bj = 1;
  for (int a = 2; a < j; ++a)
    if (cost(l, a) + cost(l + 1, j - a) < cost(l, bj) + cost(l + 1, j - bj)) bj = a;
  expand(l, bj, out);
  expand(l + 1, j - bj, out);
}

static void collapse(int k)
{
  K = k;
  const size_t n = nodes.size();
  best.assign(n * K, 0.0f);
  for (size_t x = n; x-- > 0;) {
    if (nodes[x].count != 0) continue;
    const uint32_t l = nodes[x].first;
    // split of j roots between the two children
    std::vector<float> comb(K + 1, FLT_MAX);
    for (int j = 2; j <= K; ++j)
      for (int a = 1; a < j; ++a) comb[j] = std::min(comb[j], cost(l, a) + cost(l + 1, j - a));
    float* b = &best[x * K];
    b[0] = area(nodes[x]) + comb[K];
    for (int j = 2; j <= K; ++j) b[j - 1] = std::min(b[j - 2], comb[j]);
  }
  wnodes.clear();
  wide_of.assign(n, -1);
  std::vector<uint32_t> stack{0};
  while (!stack.empty()) {
    const uint32_t x = stack.back();
    stack.pop_back();
    if (nodes[x].count != 0) continue;
    std::vector<uint32_t> kids;
    const uint32_t l = nodes[x].first;
    int bj = 1;
    for (int a = 2; a < K; ++a)
      if (cost(l, a) + cost(l + 1, K - a) < cost(l, bj) + cost(l + 1, K - bj)) bj = a;
    expand(l, bj, kids);
    expand(l + 1, K - bj, kids);
    WNode w{};
    w.nk = (int)kids.size();
    for (int c = 0; c < w.nk; ++c) { w.kid[c] = kids[c]; stack.push_back(kids[c]); }
    wide_of[x] = (int)wnodes.size();
    wnodes.push_back(w);
  }
}

static bool tri_hit(const Ray& r, uint32_t t, float tmax, float& tout)
{
  const float* p0 = &positions[3 * indices[3 * t]];
  const float* p1 = &positions[3 * indices[3 * t + 1]];
  const float* p2 = &positions[3 * indices[3 * t + 2]];
  const float e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
  const float h[3] = {r.d[1] * e2[2] - r.d[2] * e2[1], r.d[2] * e2[0] - r.d[0] * e2[2], r.d[0] * e2[1] - r.d[1] * e2[0]};
  const float a = e1[0] * h[0] + e1[1] * h[1] + e1[2] * h[2];
  if (a > -1e-7f && a < 1e-7f) return false;
  const float f = 1.0f / a;
  const float s[3] = {r.o[0] - p0[0], r.o[1] - p0[1], r.o[2] - p0[2]};
  const float u = f * (s[0] * h[0] + s[1] * h[1] + s[2] * h[2]);
  if (u < 0 || u > 1) return false;
  const float q[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
  const float v = f * (r.d[0] * q[0] + r.d[1] * q[1] + r.d[2] * q[2]);
  if (v < 0 || u + v > 1) return false;
  const float t_ = f * (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]);
  if (t_ < r.tmin || t_ > tmax) return false;
  tout = t_;
  return true;
}

struct Stats { double nodes = 0, boxes = 0, tris = 0, iters = 0, blocks = 0, valu = 0; uint32_t max_nodes = 0; };
// leaf blocks (model of an idea): a wide node all of whose children are single-triangle leaves is
// walked as ONE iteration that tests all its triangles unconditionally, instead of one node iteration plus one
// iteration per child whose box the ray passes.  valu: modelled VALU instructions (node step 140, triangle test 60,
// 30 per iteration for the loop, the fetch and the stack).
static bool g_leaf_blocks = false;
static int g_block_per_iter = 4;  // triangles of a block one iteration can test (register budget of the loads)

// order: 0 = children sorted by entry distance; 1 = by the ray's direction octant (centroid projected on sign(d));
//        2 = nearest child first, the rest in slot order; 3 = as stored
static void walk(const Ray& r, int order, Stats& st)
{
  float inv[3], best_t = r.tmax;
  for (int a = 0; a < 3; ++a) inv[a] = 1.0f / r.d[a];
  uint32_t stack[256];
  int sp = 0;
  stack[sp++] = 0;
  uint32_t my_nodes = 0;
  while (sp > 0) {
    const uint32_t x = stack[--sp];
    ++st.iters;
    if (nodes[x].count != 0) {
      st.valu += 30 + 60;
      ++st.tris;
      float t;
      if (tri_hit(r, nodes[x].first / 3, best_t, t)) best_t = t;
      continue;
    }
    const WNode& w = wnodes[wide_of[x]];
    if (g_leaf_blocks) {
      bool all_leaves = true;
      for (int c = 0; c < w.nk; ++c) all_leaves = all_leaves && nodes[w.kid[c]].count != 0;
      if (all_leaves) {
        ++st.blocks;
        st.valu += 30 * ((w.nk + g_block_per_iter - 1) / g_block_per_iter) + 60 * w.nk;
        st.iters += (w.nk + g_block_per_iter - 1) / g_block_per_iter - 1;
        for (int c = 0; c < w.nk; ++c) {
          ++st.tris;
          float t;
          if (tri_hit(r, nodes[w.kid[c]].first / 3, best_t, t)) best_t = t;
        }
        continue;
      }
    }
    st.valu += 30 + 140;
    ++st.nodes;
    ++my_nodes;
    float key[8];
    uint32_t ref[8];
    int nh = 0;
    for (int c = 0; c < w.nk; ++c) {
      const Node& n = nodes[w.kid[c]];
      ++st.boxes;
      float tn = 0.0f, tf = best_t;
      for (int a = 0; a < 3; ++a) {
        const float t0 = (n.mn[a] - r.o[a]) * inv[a], t1 = (n.mx[a] - r.o[a]) * inv[a];
        tn = std::max(tn, std::min(t0, t1));
        tf = std::min(tf, std::max(t0, t1));
      }
      if (tn <= tf * 1.0000001f) {
        float k = tn;
        if (order == 1) {
          k = 0.0f;
          for (int a = 0; a < 3; ++a) k += 0.5f * (n.mn[a] + n.mx[a]) * (r.d[a] < 0 ? -1.0f : 1.0f);
        }
        key[nh] = k;
        ref[nh++] = w.kid[c];
      }
    }
    if (order == 3) {
      // children as stored (no ordering at all)
    } else if (order == 2) {
      int m = 0;
      for (int c = 1; c < nh; ++c)
        if (key[c] < key[m]) m = c;
      if (nh) { std::swap(key[0], key[m]); std::swap(ref[0], ref[m]); }
    } else {
      for (int i = 1; i < nh; ++i)
        for (int j = i; j > 0 && key[j] < key[j - 1]; --j) { std::swap(key[j], key[j - 1]); std::swap(ref[j], ref[j - 1]); }
    }
    for (int c = nh - 1; c >= 0; --c) stack[sp++] = ref[c];
  }
  st.max_nodes = std::max(st.max_nodes, my_nodes);
}

int main(int argc, char** argv)
{
  FILE* f = fopen(argc > 1 ? argv[1] : "/tmp/sim_case.bin", "rb");
  if (!f) return 1;
  uint32_t hdr[4];
  if (fread(hdr, 4, 4, f) != 4) return 1;
  nodes.resize(hdr[0]); indices.resize(3 * (size_t)hdr[1]); positions.resize(3 * (size_t)hdr[2]); rays.resize(hdr[3]);
  if (fread(nodes.data(), sizeof(Node), nodes.size(), f) != nodes.size()) return 1;
  if (fread(indices.data(), 4, indices.size(), f) != indices.size()) return 1;
  if (fread(positions.data(), 4, positions.size(), f) != positions.size()) return 1;
  if (fread(rays.data(), sizeof(Ray), rays.size(), f) != rays.size()) return 1;
  if (fread(per_bounce, 4, 8, f) != 8) return 1;
  fclose(f);
  const char* names[4] = {"sorted by entry distance", "octant order (centroids)", "nearest first, rest unsorted", "as stored (no order)"};
  const bool blocks_only = argc > 2 && !strcmp(argv[2], "blocks");
  if (argc > 3) g_block_per_iter = atoi(argv[3]);
  for (int k : {2, 4, 8}) {
    if (blocks_only && k != 4) continue;
    collapse(k);
    for (int order = 0; order < (blocks_only ? 2 : 4); ++order) {
      g_leaf_blocks = blocks_only && order == 1;
      if (blocks_only) {
        size_t nb = 0;
        for (const WNode& w : wnodes) {
          bool al = true;
          for (int c = 0; c < w.nk; ++c) al = al && nodes[w.kid[c]].count != 0;
          nb += al;
        }
        printf("k=4 sorted by entry distance, leaf blocks %s: %zu of %zu wide nodes have only leaves\n", g_leaf_blocks ? "ON" : "off", nb, wnodes.size());
      }
      const int order_used = blocks_only ? 0 : order;
      size_t at = 0;
      Stats all;
      if (!blocks_only) printf("k=%d  %-30s  wide nodes %zu\n", k, names[order], wnodes.size());
      for (int b = 0; b < 8 && per_bounce[b]; ++b) {
        Stats st;
        for (uint32_t i = 0; i < per_bounce[b]; ++i) walk(rays[at + i], order_used, st);
        at += per_bounce[b];
        const double n = per_bounce[b];
        printf("   bounce %d: nodes/ray %6.2f  boxes/ray %6.2f  tris/ray %5.2f  iterations/ray %6.2f  blocks/ray %5.2f  valu/ray %7.1f  longest %u nodes\n", b,
               st.nodes / n, st.boxes / n, st.tris / n, st.iters / n, st.blocks / n, st.valu / n, st.max_nodes);
        all.nodes += st.nodes; all.boxes += st.boxes; all.tris += st.tris; all.iters += st.iters;
      }
      const double n = (double)at;
      printf("   all     : nodes/ray %6.2f  boxes/ray %6.2f  tris/ray %5.2f  iterations/ray %6.2f\n", all.nodes / n, all.boxes / n,
             all.tris / n, all.iters / n);
    }
  }
  return 0;
}
