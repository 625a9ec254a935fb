// wave_sim.cpp -- CPU model of one 64-lane wavefront of the persistent traversal kernel walking the 4-wide collapse:
// how many loop iterations run the node branch / the triangle branch under a given branch-vote policy?
// (Design tool for the traversal kernel; reads the case written by tools/sim/dump_case.py.  No GPU, no parity claim:
// float slabs without quantisation, Moeller-Trumbore on the raw vertices.)
//   g++ -O2 -std=c++17 -o /tmp/wave_sim tools/sim/wave_sim.cpp && /tmp/wave_sim /tmp/sim_case.bin
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
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


// ---- one wavefront -------------------------------------------------------------------------------------------------
// A lane's item is a node or a leaf (triangle).  Every iteration the wavefront runs the node branch if some lane steps
// a node, and the triangle branch if some lane steps a triangle: divergence makes an iteration cost both.  Policy:
//   tri_min   lanes holding a triangle wait until at least this many hold one (or no lane has a node to step)
//   node_min  in an iteration that runs the triangle branch, node lanes wait unless at least this many have a node
struct Lane {
  bool active = false;
  uint32_t ray = 0, cur = 0;
  float best_t = 0, inv[3];
  int sp = 0, np = 0;
  uint32_t stack[128], pend[8];
};
struct Cost { double iters = 0, node_runs = 0, tri_runs = 0, fetch_runs = 0, node_lanes = 0, tri_lanes = 0, lane_nodes = 0, lane_tris = 0; };

static void step_node(Lane& l, const Ray& r)
{
  const WNode& w = wnodes[wide_of[l.cur]];
  float key[8]; uint32_t ref[8]; int nh = 0;
  for (int c = 0; c < w.nk; ++c) {
    const Node& n = nodes[w.kid[c]];
    float tn = 0.0f, tf = l.best_t;
    for (int a = 0; a < 3; ++a) {
      const float t0 = (n.mn[a] - r.o[a]) * l.inv[a], t1 = (n.mx[a] - r.o[a]) * l.inv[a];
      tn = std::max(tn, std::min(t0, t1));
      tf = std::min(tf, std::max(t0, t1));
    }
    if (tn <= tf * 1.0000001f) { key[nh] = tn; ref[nh++] = w.kid[c]; }
  }
  for (int i = 1; i < nh; ++i)
    for (int j = i; j > 0 && key[j] < key[j - 1]; --j) { std::swap(key[j], key[j - 1]); std::swap(ref[j], ref[j - 1]); }
  for (int c = nh - 1; c >= 1; --c) l.stack[l.sp++] = ref[c];
  if (nh) l.cur = ref[0];
  else if (l.sp) l.cur = l.stack[--l.sp];
  else l.active = false;
}
static void step_tri(Lane& l, const Ray& r)
{
  float t;
  if (tri_hit(r, nodes[l.cur].first / 3, l.best_t, t)) l.best_t = t;
  if (l.sp) l.cur = l.stack[--l.sp];
  else l.active = false;
}

static void run_wave(size_t begin, size_t end, int tri_min, int node_min, int refill, Cost& c)
{
  static Lane lanes[64];
  for (auto& l : lanes) l.active = false;
  size_t next = begin;
  for (;;) {
    int idle = 0;
    for (auto& l : lanes) idle += !l.active;
    if (next < end && (idle == 64 || idle >= refill)) {
      ++c.fetch_runs;
      for (auto& l : lanes)
        if (!l.active && next < end) {
          l.ray = (uint32_t)next++;
          const Ray& r = rays[l.ray];
          for (int a = 0; a < 3; ++a) l.inv[a] = 1.0f / r.d[a];
          l.best_t = r.tmax; l.cur = 0; l.sp = 0; l.active = true;
        }
    }
    int n_node = 0, n_tri = 0;
    for (auto& l : lanes)
      if (l.active) (nodes[l.cur].count ? n_tri : n_node)++;
    if (n_node + n_tri == 0) { if (next >= end) break; continue; }
    const bool do_tri = n_tri > 0 && (n_tri >= tri_min || n_node == 0);
    const bool do_node = n_node > 0 && (!do_tri || n_node >= node_min);
    ++c.iters;
    c.node_runs += do_node; c.tri_runs += do_tri;
    if (do_node) c.node_lanes += n_node;
    if (do_tri) c.tri_lanes += n_tri;
    for (auto& l : lanes) {
      if (!l.active) continue;
      const Ray& r = rays[l.ray];
      if (nodes[l.cur].count) { if (do_tri) { step_tri(l, r); ++c.lane_tris; } }
      else if (do_node) { step_node(l, r); ++c.lane_nodes; }
    }
  }
}

// Variant "stash": a lane whose next item is a triangle puts it aside (up to `cap` per lane) and goes on with its
// stack; the triangle branch runs when at least tri_min lanes hold one (or nothing else is left), and a lane that
// holds one tests ONE triangle in that iteration instead of stepping its node.
static void settle(Lane& l, int cap)
{
  while (l.active && nodes[l.cur].count && l.np < cap) {
    l.pend[l.np++] = l.cur;
    if (l.sp) l.cur = l.stack[--l.sp];
    else { l.active = false; }
  }
}
static void run_wave_stash(size_t begin, size_t end, int tri_min, int cap, int refill, Cost& c)
{
  static Lane lanes[64];
  for (auto& l : lanes) { l.active = false; l.np = 0; }
  size_t next = begin;
  for (;;) {
    int idle = 0;
    for (auto& l : lanes) idle += !l.active && l.np == 0;
    if (next < end && (idle == 64 || idle >= refill)) {
      ++c.fetch_runs;
      for (auto& l : lanes)
        if (!l.active && l.np == 0 && next < end) {
          l.ray = (uint32_t)next++;
          const Ray& r = rays[l.ray];
          for (int a = 0; a < 3; ++a) l.inv[a] = 1.0f / r.d[a];
          l.best_t = r.tmax; l.cur = 0; l.sp = 0; l.active = true;
        }
    }
    int n_node = 0, n_tri = 0, n_blocked = 0;
    for (auto& l : lanes) {
      if (l.np) ++n_tri;
      if (l.active && !nodes[l.cur].count) ++n_node;
      if (l.active && nodes[l.cur].count) ++n_blocked;  // pending list full and another triangle on top
    }
    if (n_node + n_tri == 0) { if (next >= end) break; continue; }
    const bool do_tri = n_tri > 0 && (n_tri >= tri_min || n_node == 0);
    ++c.iters;
    c.tri_runs += do_tri;
    int stepped_nodes = 0;
    for (auto& l : lanes) {
      const Ray& r = rays[l.ray];
      if (do_tri && l.np) {
        float t;
        if (tri_hit(r, nodes[l.pend[--l.np]].first / 3, l.best_t, t)) l.best_t = t;
        ++c.lane_tris; ++c.tri_lanes;
      } else if (l.active && !nodes[l.cur].count) {
        step_node(l, r); ++c.lane_nodes; ++stepped_nodes;
      }
      settle(l, cap);
    }
    c.node_runs += stepped_nodes > 0;
    c.node_lanes += stepped_nodes;
  }
}

// Variant "queue": triangles go to a per-wavefront queue (lane, leaf) and the lane walks on at once; when the queue
// holds at least `flush` entries (or nothing else is left to do) the whole wavefront tests up to 64 of them in one
// dense iteration -- any lane tests any entry (the ray comes from the owner's registers), results go back to the owner.
// A lane whose walk has ended waits until its queued triangles are done.
struct QEntry { int lane; uint32_t leaf; };
static void run_wave_queue(size_t begin, size_t end, int flush, int refill, Cost& c)
{
  static Lane lanes[64];
  static int pending[64];
  std::vector<QEntry> queue;
  for (int i = 0; i < 64; ++i) { lanes[i].active = false; pending[i] = 0; }
  size_t next = begin;
  for (;;) {
    int idle = 0;
    for (int i = 0; i < 64; ++i) idle += !lanes[i].active && pending[i] == 0;
    if (next < end && (idle == 64 || idle >= refill)) {
      ++c.fetch_runs;
      for (int i = 0; i < 64; ++i)
        if (!lanes[i].active && pending[i] == 0 && next < end) {
          Lane& l = lanes[i];
          l.ray = (uint32_t)next++;
          const Ray& r = rays[l.ray];
          for (int a = 0; a < 3; ++a) l.inv[a] = 1.0f / r.d[a];
          l.best_t = r.tmax; l.cur = 0; l.sp = 0; l.active = true;
        }
    }
    // leaves on top go to the queue (the lane pops on)
    for (int i = 0; i < 64; ++i) {
      Lane& l = lanes[i];
      while (l.active && nodes[l.cur].count) {
        queue.push_back({i, l.cur});
        ++pending[i];
        if (l.sp) l.cur = l.stack[--l.sp];
        else l.active = false;
      }
    }
    int n_node = 0;
    for (int i = 0; i < 64; ++i) n_node += lanes[i].active;
    if (n_node == 0 && queue.empty()) { if (next >= end) break; continue; }
    ++c.iters;
    if ((int)queue.size() >= flush || n_node == 0) {
      const size_t take = std::min<size_t>(64, queue.size());
      ++c.tri_runs;
      c.tri_lanes += take;
      for (size_t k = 0; k < take; ++k) {
        const QEntry e = queue[k];
        Lane& l = lanes[e.lane];
        float t;
        if (tri_hit(rays[l.ray], nodes[e.leaf].first / 3, l.best_t, t)) l.best_t = t;
        --pending[e.lane];
        ++c.lane_tris;
      }
      queue.erase(queue.begin(), queue.begin() + take);
    } else {
      ++c.node_runs;
      c.node_lanes += n_node;
      for (int i = 0; i < 64; ++i)
        if (lanes[i].active) { step_node(lanes[i], rays[lanes[i].ray]); ++c.lane_nodes; }
    }
  }
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
  collapse(4);
  // instruction counts of the branches in the compiled kernel (k_traverse4<false,false>, round 2): loop head and
  // loads ~60, node branch ~140, triangle branch ~85, ray fetch ~250
  const double c_head = 60, c_node = 140, c_tri = 85, c_fetch = 250;
  const int refill = argc > 2 ? atoi(argv[2]) : 16;
  const size_t per_wave = argc > 3 ? atoi(argv[3]) : 512;  // rays one wavefront walks (its share of a launch)
  struct P { int tri_min, node_min; } ps[] = {{1, 1}, {4, 1}, {8, 1}, {12, 1}, {16, 1}, {24, 1}, {32, 1}, {16, 16}, {16, 32}, {24, 24}, {32, 32}, {64, 1}};
  for (const P& p : ps) {
    Cost all;
    size_t at = 0;
    for (int b = 0; b < 8 && per_bounce[b]; ++b) {
      for (size_t s = 0; s < per_bounce[b]; s += per_wave) run_wave(at + s, at + std::min<size_t>(per_bounce[b], s + per_wave), p.tri_min, p.node_min, refill, all);
      at += per_bounce[b];
    }
    const double n = (double)at;
    const double cost = all.iters * c_head + all.node_runs * c_node + all.tri_runs * c_tri + all.fetch_runs * c_fetch;
    printf("tri_min %2d node_min %2d: iterations/ray*64 %6.2f  node runs %6.2f  tri runs %6.2f  lanes/node run %5.1f  lanes/tri run %5.1f  "
           "nodes/ray %5.2f tris/ray %4.2f  instr/ray %7.1f\n", p.tri_min, p.node_min, all.iters / n * 64, all.node_runs / n * 64,
           all.tri_runs / n * 64, all.node_lanes / std::max(1.0, all.node_runs), all.tri_lanes / std::max(1.0, all.tri_runs),
           all.lane_nodes / n, all.lane_tris / n, cost / n);
  }
  struct Q { int tri_min, cap; } qs[] = {{1, 1}, {8, 1}, {16, 1}, {24, 1}, {32, 1}, {16, 2}, {24, 2}, {32, 2}, {40, 2}, {24, 4}, {32, 4}, {40, 4}, {48, 4}};
  for (const Q& q : qs) {
    Cost all;
    size_t at = 0;
    for (int b = 0; b < 8 && per_bounce[b]; ++b) {
      for (size_t s = 0; s < per_bounce[b]; s += per_wave) run_wave_stash(at + s, at + std::min<size_t>(per_bounce[b], s + per_wave), q.tri_min, q.cap, refill, all);
      at += per_bounce[b];
    }
    const double n = (double)at;
    const double cost = all.iters * c_head + all.node_runs * c_node + all.tri_runs * c_tri + all.fetch_runs * c_fetch;
    printf("stash tri_min %2d cap %d: iterations/ray*64 %6.2f  node runs %6.2f  tri runs %6.2f  lanes/node run %5.1f  lanes/tri run %5.1f  "
           "nodes/ray %5.2f tris/ray %4.2f  instr/ray %7.1f\n", q.tri_min, q.cap, all.iters / n * 64, all.node_runs / n * 64,
           all.tri_runs / n * 64, all.node_lanes / std::max(1.0, all.node_runs), all.tri_lanes / std::max(1.0, all.tri_runs),
           all.lane_nodes / n, all.lane_tris / n, cost / n);
  }
  // dense triangle iteration: the test itself (85) + fetching the owner's ray and writing the result back (~35);
  // node iteration with the enqueue-and-pop-on path (~20 more than today's)
  const double c_tri_dense = 120, c_node_q = c_node + 20;
  for (int flush : {16, 24, 32, 48, 64}) {
    Cost all;
    size_t at = 0;
    for (int b = 0; b < 8 && per_bounce[b]; ++b) {
      for (size_t s = 0; s < per_bounce[b]; s += per_wave) run_wave_queue(at + s, at + std::min<size_t>(per_bounce[b], s + per_wave), flush, refill, all);
      at += per_bounce[b];
    }
    const double n = (double)at;
    const double cost = all.node_runs * (c_head + c_node_q) + all.tri_runs * (c_head + c_tri_dense) + all.fetch_runs * c_fetch;
    printf("queue flush %2d: iterations/ray*64 %6.2f  node runs %6.2f  tri runs %6.2f  lanes/node run %5.1f  lanes/tri run %5.1f  "
           "nodes/ray %5.2f tris/ray %4.2f  instr/ray %7.1f\n", flush, all.iters / n * 64, all.node_runs / n * 64,
           all.tri_runs / n * 64, all.node_lanes / std::max(1.0, all.node_runs), all.tri_lanes / std::max(1.0, all.tri_runs),
           all.lane_nodes / n, all.lane_tris / n, cost / n);
  }
  return 0;
}
