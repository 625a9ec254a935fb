// beam_sim.cpp -- CPU model of "entry points for primary rays" (round-3 review, item 8): a pre-pass walks the four-wide
// tree once per TILE of pixels with the tile's frustum and hands every ray of the tile the deepest nodes the frustum
// overlaps; the rays then start there instead of at the root.  How many of bounce 0's node visits does that remove?
// (Design tool: reads the FULL case of tools/sim/dump_case.py -- every primary ray of the 1920 x 1080 frame; float slabs
// without quantisation, so the counts are those of walk_sim.cpp, slightly below the kernel's.)
//   python3 tools/sim/dump_case.py /tmp/sim_full.bin full
//   g++ -O2 -std=c++17 -o /tmp/beam_sim tools/sim/beam_sim.cpp && /tmp/beam_sim /tmp/sim_full.bin
// Model.  Frustum of a tile = the four planes through the camera and two neighbouring corner rays of the tile's pixel
// rectangle (grown by one pixel: the jitter moves a ray by up to a pixel).  A box is outside when all eight corners are
// on the outer side of one plane (conservative: what is kept may still be missed by every ray).  The pre-pass keeps a
// frontier, starting with the root's children, and replaces its largest inner node by that node's overlapping children
// while the frontier has room (at most E entries): the frontier is the tile's entry list, stored WITH the boxes, so a
// ray pays one shared fetch for the list (E * 28 bytes, one per tile and wavefront) and E box tests, then walks the
// entries it hits, nearest first, exactly like children of one wide node.  Counted per walked ray (rays that hit the
// root box): node records fetched by the ray itself, box tests, and the pre-pass's own node visits divided by the
// tile's rays.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define main walk_sim_main
#include "walk_sim.cpp"
#undef main

struct Plane { float n[3]; };  // through the camera position; inside: n . (p - eye) >= 0

static bool box_outside(const Node& b, const float eye[3], const Plane* pl, int np)
{
  for (int k = 0; k < np; ++k) {
    // the corner farthest along n
    float p[3];
    for (int a = 0; a < 3; ++a) p[a] = (pl[k].n[a] >= 0 ? b.mx[a] : b.mn[a]) - eye[a];
    if (pl[k].n[0] * p[0] + pl[k].n[1] * p[1] + pl[k].n[2] * p[2] < 0) return true;
  }
  return false;
}

static void cross3(const float* a, const float* b, float* c)
{
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

struct WalkCount { double nodes = 0, boxes = 0, tris = 0; };

// the walk of walk_sim.cpp (children sorted by entry distance), started from a list of entry nodes whose boxes the ray
// has tested itself
static void walk_from(const Ray& r, const std::vector<uint32_t>& entries, WalkCount& wc, bool untested = false)
{
  float inv[3], best_t = r.tmax;
  for (int a = 0; a < 3; ++a) inv[a] = 1.0f / r.d[a];
  uint32_t stack[512];
  float key[64];
  uint32_t ref[64];
  int nh = 0;
  for (uint32_t e : entries) {
    const Node& n = nodes[e];
    ++wc.boxes;
    float tn = 0.0f, tf = best_t;
    for (int a = 0; a < 3; ++a) {
      const float t0 = (n.mn[a] - r.o[a]) * inv[a], t1 = (n.mx[a] - r.o[a]) * inv[a];
      tn = std::max(tn, std::min(t0, t1));
      tf = std::min(tf, std::max(t0, t1));
    }
    if (untested) { --wc.boxes; tn = (float)nh; tf = FLT_MAX; }  // (entered in list order, nothing known about them)
    if (tn <= tf * 1.0000001f && nh < 64) { key[nh] = tn; ref[nh++] = e; }
  }
  for (int i = 1; i < nh; ++i)
    for (int j = i; j > 0 && key[j] < key[j - 1]; --j) { std::swap(key[j], key[j - 1]); std::swap(ref[j], ref[j - 1]); }
  int sp = 0;
  for (int c = nh - 1; c >= 0; --c) stack[sp++] = ref[c];
  while (sp > 0) {
    const uint32_t x = stack[--sp];
    if (nodes[x].count != 0) {
      ++wc.tris;
      float t;
      if (tri_hit(r, nodes[x].first / 3, best_t, t)) best_t = t;
      continue;
    }
    // (an entry whose box lies beyond a hit found meanwhile is still fetched: the kernel culls at push time only)
    const WNode& w = wnodes[wide_of[x]];
    ++wc.nodes;
    float k2[8];
    uint32_t r2[8];
    int n2 = 0;
    for (int c = 0; c < w.nk; ++c) {
      const Node& n = nodes[w.kid[c]];
      ++wc.boxes;
      float tn = 0.0f, tf = best_t;
      for (int a = 0; a < 3; ++a) {
        const float t0 = (n.mn[a] - r.o[a]) * inv[a], t1 = (n.mx[a] - r.o[a]) * inv[a];
        tn = std::max(tn, std::min(t0, t1));
        tf = std::min(tf, std::max(t0, t1));
      }
      if (tn <= tf * 1.0000001f) { k2[n2] = tn; r2[n2++] = w.kid[c]; }
    }
    for (int i = 1; i < n2; ++i)
      for (int j = i; j > 0 && k2[j] < k2[j - 1]; --j) { std::swap(k2[j], k2[j - 1]); std::swap(r2[j], r2[j - 1]); }
    for (int c = n2 - 1; c >= 0; --c) stack[sp++] = r2[c];
  }
}

int main(int argc, char** argv)
{
  FILE* f = fopen(argc > 1 ? argv[1] : "/tmp/sim_full.bin", "rb");
  if (!f) { fprintf(stderr, "no case file\n"); return 1; }
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
  if (per_bounce[0] != (uint32_t)W * H) { fprintf(stderr, "expected the full case (%d x %d primary rays), got %u\n", W, H, per_bounce[0]); return 1; }
  collapse(4);
  const float eye[3] = {rays[0].o[0], rays[0].o[1], rays[0].o[2]};
  auto dir = [&](int x, int y) { return rays[(size_t)std::min(std::max(y, 0), H - 1) * W + std::min(std::max(x, 0), W - 1)].d; };
  auto hits_root = [&](const Ray& r) {
    float tn = 0.0f, tf = FLT_MAX;
    for (int a = 0; a < 3; ++a) {
      const float i = 1.0f / r.d[a], t0 = (nodes[0].mn[a] - r.o[a]) * i, t1 = (nodes[0].mx[a] - r.o[a]) * i;
      tn = std::max(tn, std::min(t0, t1));
      tf = std::min(tf, std::max(t0, t1));
    }
    return tn <= tf;
  };
  // baseline: every ray from the root
  WalkCount base;
  size_t walked = 0;
  {
    std::vector<uint32_t> root{0};
    for (size_t i = 0; i < (size_t)W * H; ++i)
      if (hits_root(rays[i])) { ++walked; walk_from(rays[i], root, base); }
    base.boxes -= (double)walked;  // (the root's own box is the world-box test the kernel makes anyway)
  }
  printf("primary rays %d, walked (hit the mesh's box) %zu\n", W * H, walked);
  printf("from the root              : nodes/ray %6.2f  box tests/ray %6.2f  tris/ray %5.2f\n", base.nodes / walked, base.boxes / walked, base.tris / walked);
  struct Shape { int w, h; bool untested; };
  for (const Shape sh : {Shape{8, 8, false}, Shape{16, 16, false}, Shape{64, 1, false}, Shape{32, 2, false}, Shape{16, 4, false}, Shape{8, 8, true}, Shape{64, 1, true}, Shape{16, 4, true}}) {
    const int tw = sh.w, th = sh.h;
    for (int E : {4, 8, 16, 32}) {
      WalkCount wc;
      double pre_nodes = 0, entries_sum = 0, tiles = 0, empty_tiles = 0;
      for (int ty = 0; ty < H; ty += th)
        for (int tx = 0; tx < W; tx += tw) {
          // frustum planes from the tile's corner rays, one pixel of slack
          const int x0 = tx - 1, x1 = tx + tw, y0 = ty - 1, y1 = ty + th;
          const float* c00 = dir(x0, y0); const float* c10 = dir(x1, y0); const float* c01 = dir(x0, y1); const float* c11 = dir(x1, y1);
          Plane pl[4];
          // image rows grow downwards: order the cross products so that the tile's centre ray is inside
          cross3(c00, c10, pl[0].n);  // top
          cross3(c10, c11, pl[1].n);  // right
          cross3(c11, c01, pl[2].n);  // bottom
          cross3(c01, c00, pl[3].n);  // left
          const float* cc = dir(tx + tw / 2, ty + th / 2);
          for (int k = 0; k < 4; ++k)
            if (pl[k].n[0] * cc[0] + pl[k].n[1] * cc[1] + pl[k].n[2] * cc[2] < 0)
              for (int a = 0; a < 3; ++a) pl[k].n[a] = -pl[k].n[a];
          ++tiles;
          // frontier
          std::vector<uint32_t> fr;
          if (!box_outside(nodes[0], eye, pl, 4)) fr.push_back(0);
          for (;;) {
            int pick = -1;
            float big = -1.0f;
            for (size_t k = 0; k < fr.size(); ++k)
              if (nodes[fr[k]].count == 0 && area(nodes[fr[k]]) > big) { big = area(nodes[fr[k]]); pick = (int)k; }
            if (pick < 0) break;
            const WNode& w = wnodes[wide_of[fr[pick]]];
            std::vector<uint32_t> kids;
            for (int c = 0; c < w.nk; ++c)
              if (!box_outside(nodes[w.kid[c]], eye, pl, 4)) kids.push_back(w.kid[c]);
            if (fr.size() - 1 + kids.size() > (size_t)E) break;
            ++pre_nodes;
            fr.erase(fr.begin() + pick);
            fr.insert(fr.end(), kids.begin(), kids.end());
          }
          entries_sum += (double)fr.size();
          empty_tiles += fr.empty();
          for (int y = ty; y < std::min(ty + th, H); ++y)
            for (int x = tx; x < std::min(tx + tw, W); ++x) {
              const Ray& r = rays[(size_t)y * W + x];
              if (hits_root(r)) walk_from(r, fr, wc, sh.untested);
            }
        }
      const double per_ray_pre = pre_nodes / (double)walked;
      printf("tile %2dx%-2d %s entries <= %2d : nodes/ray %6.2f (%+5.1f %%)  box tests/ray %6.2f (%+5.1f %%)  tris/ray %5.2f | list %4.1f entries on average, "
             "%4.1f %% of the tiles empty | pre-pass %5.3f node visits per walked ray\n",
             tw, th, sh.untested ? "refs only " : "with boxes", E, wc.nodes / walked, 100.0 * (wc.nodes / base.nodes - 1.0), wc.boxes / walked, 100.0 * (wc.boxes / base.boxes - 1.0),
             wc.tris / walked, entries_sum / tiles, 100.0 * empty_tiles / tiles, per_ray_pre);
    }
  }
  return 0;
}
