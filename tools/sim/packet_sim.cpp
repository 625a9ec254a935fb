// packet_sim.cpp -- CPU model of a WAVE-COHERENT first bounce (VERDICT r2 item 4 ii): the 64 primary rays of a wavefront
// walk the four-wide tree TOGETHER -- one node per wavefront and step (a scalar load), a child is entered when ANY ray's
// slab interval is non-empty, a leaf's triangle is tested by all 64 lanes -- against the kernel as it is (every lane
// walks its own ray; a wavefront step advances each busy lane by one node or triangle).  Counts loop iterations per
// wavefront's worth of 64 rays, for packets of 64 x 1 pixels (the order k_traverse4 fetches primary rays in) and of
// 8 x 8 pixels.  Same scene, camera and tree as the benchmark (1,000,000 triangles, 1920 x 1080); primary rays through
// the pixel centres.  No GPU, no parity claim (float slabs without quantisation).
//   python tools/sim/dump_case.py /tmp/sim_case.bin full && g++ -O2 -std=c++17 -o /tmp/packet_sim tools/sim/packet_sim.cpp && /tmp/packet_sim /tmp/sim_case.bin
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#define main walk_sim_main
#include "walk_sim.cpp"
#undef main

struct PStats { double node_steps = 0, tri_steps = 0, packets = 0, lane_node = 0, lane_tri = 0, kids_node = 0, kids_tri = 0; };

static void packet_walk(const Ray* r, int n, PStats& st)
{
  float inv[64][3], best_t[64];
  for (int i = 0; i < n; ++i) {
    best_t[i] = r[i].tmax;
    for (int a = 0; a < 3; ++a) inv[i][a] = 1.0f / r[i].d[a];
  }
  uint32_t stack[512];
  int sp = 0;
  stack[sp++] = 0;
  while (sp > 0) {
    const uint32_t x = stack[--sp];
    if (nodes[x].count != 0) {
      st.tri_steps += 1;
      for (int i = 0; i < n; ++i) {
        float t;
        if (tri_hit(r[i], nodes[x].first / 3, best_t[i], t)) best_t[i] = t;
      }
      continue;
    }
    // (a node whose box every ray has meanwhile culled is still popped: the push decision was taken earlier)
    const WNode& w = wnodes[wide_of[x]];
    st.node_steps += 1;
    float key[8];
    uint32_t ref[8];
    int nh = 0;
    for (int c = 0; c < w.nk; ++c) {
      const Node& nd = nodes[w.kid[c]];
      float nearest = FLT_MAX;
      int lanes = 0;
      for (int i = 0; i < n; ++i) {
        float tn = 0.0f, tf = best_t[i];
        for (int a = 0; a < 3; ++a) {
          const float t0 = (nd.mn[a] - r[i].o[a]) * inv[i][a], t1 = (nd.mx[a] - r[i].o[a]) * inv[i][a];
          tn = std::max(tn, std::min(t0, t1));
          tf = std::min(tf, std::max(t0, t1));
        }
        if (tn <= tf * 1.0000001f) {
          ++lanes;
          nearest = std::min(nearest, tn);
        }
      }
      if (lanes) {
        key[nh] = nearest;
        ref[nh++] = w.kid[c];
        if (nd.count != 0) { st.lane_tri += lanes; st.kids_tri += 1; } else { st.lane_node += lanes; st.kids_node += 1; }
      }
    }
    for (int i = 1; i < nh; ++i)
      for (int j = i; j > 0 && key[j] < key[j - 1]; --j) { std::swap(key[j], key[j - 1]); std::swap(ref[j], ref[j - 1]); }
    for (int c = nh - 1; c >= 0; --c) stack[sp++] = ref[c];
  }
  st.packets += 1;
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
  const int W = 1920, H = 1080;
  if (per_bounce[0] != (uint32_t)(W * H)) { fprintf(stderr, "need the full-resolution case (dump_case.py <file> full)\n"); return 1; }
  collapse(4);
  // every 5th packet row / column band: 1/5 of the frame is plenty
  for (int shape = 0; shape < 2; ++shape) {
    const int pw = shape == 0 ? 64 : 8, ph = shape == 0 ? 1 : 8;
    PStats st;
    Stats own;
    std::vector<Ray> pk(64);
    for (int y0 = 0; y0 + ph <= H; y0 += ph)
      for (int x0 = 0; x0 + pw <= W; x0 += pw) {
        if (((y0 / ph) * 31 + x0 / pw) % 5 != 0) continue;
        int k = 0;
        for (int y = 0; y < ph; ++y)
          for (int x = 0; x < pw; ++x) pk[k++] = rays[(size_t)(y0 + y) * W + x0 + x];
        packet_walk(pk.data(), 64, st);
        for (int i = 0; i < 64; ++i) walk(pk[i], 0, own);
      }
    const double p = st.packets;
    const double own_iters = own.iters / p;  // lane-steps per 64 rays
    printf("packets of %2d x %d pixels (%.0f packets):\n", pw, ph, p);
    printf("   each lane its own ray:  %.1f node + %.1f triangle lane-steps per 64 rays = %.1f; at the kernel's measured 35 busy lanes per\n"
           "                           instruction that is %.1f wavefront iterations per 64 rays\n",
           own.nodes / p, own.tris / p, own_iters, own_iters / 35.0);
    printf("   one walk per wavefront: %.1f node steps + %.1f triangle steps = %.1f wavefront iterations per 64 rays\n", st.node_steps / p,
           st.tri_steps / p, (st.node_steps + st.tri_steps) / p);
    printf("                           (of the 64 lanes, %.1f have a reason to enter an inner child the wavefront enters, %.1f a triangle)\n",
           st.lane_node / std::max(1.0, st.kids_node), st.lane_tri / std::max(1.0, st.kids_tri));
  }
  return 0;
}
