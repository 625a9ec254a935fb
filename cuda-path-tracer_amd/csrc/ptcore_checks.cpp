// ptcore_checks.cpp -- host-side checks of the rules the kernels share with the host (ptc_check_beam, ptc_check_feed,
// ptc_check_traversal_layout), ptc_debug_beam_entries, ptc_selftest_math.  Part of libptcore.so (ptcore_ctx.hpp).
#include "ptcore_ctx.hpp"

using namespace pt;
using namespace ptcd;

extern "C" {

// Entry points for primary rays (pt_beam_rules.hpp) checked on the host: for a mesh, an object matrix, a camera and a
// resolution, every tile's entries as k_beam computes them (same functions), then for sample rays of the tile -- the
// corners and the centre of the jitter range of every `stride`-th pixel -- the closest hit of a plain walk over the
// four-wide quantised tree started at the ROOT against the same walk started at the tile's ENTRIES: triangle and t must
// agree.  Returns the number of rays that disagree (0 = sound), or a negative status; stats (may be NULL): tiles, tiles
// without entries, entries in total, rays checked, rays that hit.
int ptc_check_beam(const float* positions, uint32_t vertex_count, const uint32_t* indices, uint32_t index_count, const float* object_m16,
                   const ptc_camera* camera, uint32_t width, uint32_t height, uint32_t stride, uint64_t* stats5, float* entries_out)
{
  if (!positions || !indices || !camera || index_count == 0u || index_count % 3u || width < 2u || height < 2u || stride == 0u) return PTC_ERR_INVALID;
  std::vector<ptc_bvh_node> nodes((size_t)index_count / 3u * 2u);
  uint32_t depth = 0;
  const int rc = build_bvh(positions, vertex_count, indices, index_count, nodes.data(), &depth);
  if (rc < 0) return rc;
  WideAccel wide;
  if (int r = build_wide(nodes.data(), (uint32_t)rc, wide)) return r;
  Wide4Accel w4;
  if (int r = build_wide4(nodes.data(), (uint32_t)rc, w4)) return r;
  m4 m{};
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) m.c[c][r] = object_m16 ? object_m16[4 * c + r] : (c == r ? 1.0f : 0.0f);
  const m4 inv_m = inverse(m);
  const DCamera cam = make_camera(*camera, width, height);
  auto gen = [&](float fx, float fy, f3& o, f3& d) {  // generate_ray (pt_kernels.hip), the same operations
    const float u = fx / (float)(cam.width - 1u);
    const float v = ((float)cam.height - fy) / (float)(cam.height - 1u);
    const float dx = cam.llx + cam.vw * u;
    const float dy = cam.lly + cam.vh * v;
    o = cam.origin;
    d = normalize(xform_vector(cam.cam, mk3(dx, dy, -1.0f)));
  };
  const uint32_t* nq = w4.nodes_q.data();
  const uint32_t n4 = (uint32_t)(w4.nodes_q.size() / 16u);
  const f3 root_lo = mk3(wide.root_min[0], wide.root_min[1], wide.root_min[2]), root_hi = mk3(wide.root_max[0], wide.root_max[1], wide.root_max[2]);
  // closest hit of the plain walk from a set of start references (boxes tested first)
  struct HitRec { bool hit; uint32_t rank; float t; };
  auto slab = [](const f3 lo, const f3 hi, const f3 o, const f3 inv, float tmax) {
    float tn = 0.0f, tf = tmax;
    const float lo_[3] = {lo.x, lo.y, lo.z}, hi_[3] = {hi.x, hi.y, hi.z}, o_[3] = {o.x, o.y, o.z}, i_[3] = {inv.x, inv.y, inv.z};
    for (int a = 0; a < 3; ++a) {
      const float t0 = (lo_[a] - o_[a]) * i_[a], t1 = (hi_[a] - o_[a]) * i_[a];
      tn = std::max(tn, std::min(t0, t1));
      tf = std::min(tf, std::max(t0, t1));
    }
    return tn <= tf * 1.000001f + 1e-6f;
  };
  auto walk = [&](const f3 o, const f3 d, const f3* lo4, const f3* hi4, const uint32_t* ref4, int n) {
    HitRec best{false, 0u, 3.0e38f};
    const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    std::vector<uint32_t> stack;
    for (int k = 0; k < n; ++k)
      if (slab(lo4[k], hi4[k], o, inv, best.t)) stack.push_back(ref4[k]);
    while (!stack.empty()) {
      const uint32_t ref = stack.back();
      stack.pop_back();
      if (ref & kLeafBit) {
        const uint32_t rank = ref & ~kLeafBit;
        if (rank >= wide.tri_order.size()) continue;  // the dummy
        const uint32_t tri = wide.tri_order[rank];
        const f3 p0 = mk3(positions[3 * indices[3 * tri]], positions[3 * indices[3 * tri] + 1], positions[3 * indices[3 * tri] + 2]);
        const f3 p1 = mk3(positions[3 * indices[3 * tri + 1]], positions[3 * indices[3 * tri + 1] + 1], positions[3 * indices[3 * tri + 1] + 2]);
        const f3 p2 = mk3(positions[3 * indices[3 * tri + 2]], positions[3 * indices[3 * tri + 2] + 1], positions[3 * indices[3 * tri + 2] + 2]);
        const f3 e1 = p1 - p0, e2 = p2 - p0, h = cross(d, e2);
        const float a = dot(e1, h);
        if (a > -1e-12f && a < 1e-12f) continue;
        const float f = 1.0f / a;
        const f3 sv = o - p0;
        const float u = f * dot(sv, h);
        if (u < 0.0f || u > 1.0f) continue;
        const f3 q = cross(sv, e1);
        const float v = f * dot(d, q);
        if (v < 0.0f || u + v > 1.0f) continue;
        const float t = f * dot(e2, q);
        if (t < 1e-4f) continue;
        if (t < best.t || (t == best.t && rank > best.rank)) best = HitRec{true, rank, t};
        continue;
      }
      if (ref >= n4) return HitRec{true, 0xffffffffu, -1.0f};  // a reference out of range: reported as a disagreement
      const uint32_t* q = nq + 16u * (size_t)ref;
      for (int c = 0; c < 4; ++c) {
        f3 lo, hi;
        if (!beam_rules::child_box(q, c, lo, hi)) continue;
        if (slab(lo, hi, o, inv, best.t)) stack.push_back(q[12 + c]);
      }
    }
    return best;
  };
  uint64_t tiles = 0, empty = 0, entries = 0, rays = 0, hits = 0;
  int bad = 0;
  const uint32_t tiles_x = (width + kBeamTile - 1u) / kBeamTile, tiles_y = (height + kBeamTile - 1u) / kBeamTile;
  const uint32_t root_ref4[1] = {w4.root_ref};
  for (uint32_t ty = 0; ty < tiles_y; ++ty)
    for (uint32_t tx = 0; tx < tiles_x; ++tx) {
      const float x0 = (float)(tx * kBeamTile) - 0.05f, x1 = (float)((tx + 1u) * kBeamTile) + 0.05f;
      const float y0 = (float)(ty * kBeamTile) - 0.05f, y1 = (float)((ty + 1u) * kBeamTile) + 0.05f;
      f3 o, d00, d10, d01, d11, dc;
      gen(x0, y0, o, d00);
      gen(x1, y0, o, d10);
      gen(x0, y1, o, d01);
      gen(x1, y1, o, d11);
      gen(0.5f * (x0 + x1), 0.5f * (y0 + y1), o, dc);
      const beam_rules::Frustum fr = beam_rules::make_frustum(xform_point(inv_m, o), xform_vector(inv_m, d00), xform_vector(inv_m, d10),
                                                              xform_vector(inv_m, d01), xform_vector(inv_m, d11), xform_vector(inv_m, dc));
      f3 lo4[4], hi4[4];
      uint32_t ref4[4];
      const int n = beam_rules::tile_entries(nq, n4, w4.root_ref, root_lo, root_hi, fr, lo4, hi4, ref4);
      if (entries_out) {  // as k_beam stores them: {box min, reference bits} {box max, 0}; unused: an inside-out box
        float* e = entries_out + ((size_t)ty * tiles_x + tx) * 32u;
        for (int k = 0; k < 4; ++k) {
          const float inf = __builtin_inff();
          uint32_t ref = k < n ? ref4[k] : kNoChild;
          float refbits;
          std::memcpy(&refbits, &ref, 4);
          const float rec[8] = {k < n ? lo4[k].x : inf, k < n ? lo4[k].y : inf, k < n ? lo4[k].z : inf, refbits,
                                k < n ? hi4[k].x : -inf, k < n ? hi4[k].y : -inf, k < n ? hi4[k].z : -inf, 0.0f};
          std::memcpy(e + 8 * k, rec, sizeof rec);
        }
      }
      ++tiles;
      empty += n == 0;
      entries += (uint64_t)n;
      for (int k = 0; k < n; ++k)
        if (!(ref4[k] & kLeafBit) && ref4[k] >= n4) ++bad;
      for (uint32_t py = ty * kBeamTile; py < std::min(height, (ty + 1u) * kBeamTile); py += stride)
        for (uint32_t px = tx * kBeamTile; px < std::min(width, (tx + 1u) * kBeamTile); px += stride) {
          const float jit[5][2] = {{0.0f, 0.0f}, {1.0f, 0.0f}, {0.0f, 1.0f}, {1.0f, 1.0f}, {0.5f, 0.5f}};  // (uniform_real can round to 1)
          for (const auto& j : jit) {
            f3 ro, rd;
            gen((float)px + j[0], (float)py + j[1], ro, rd);
            const f3 oo = xform_point(inv_m, ro), od = xform_vector(inv_m, rd);
            const HitRec a = walk(oo, od, &root_lo, &root_hi, root_ref4, 1);
            const HitRec b = walk(oo, od, lo4, hi4, ref4, n);
            ++rays;
            hits += a.hit;
            if (a.hit != b.hit || (a.hit && (a.rank != b.rank || a.t != b.t))) ++bad;
          }
        }
    }
  if (stats5) {
    stats5[0] = tiles;
    stats5[1] = empty;
    stats5[2] = entries;
    stats5[3] = rays;
    stats5[4] = hits;
  }
  return bad;
}

// The ray feed of the persistent traversal launches (BatchFeed; pt_feed_rules.hpp) checked on the host: a frame of n rays,
// its eight regions, each dealt as static_eighths / 8 static batches of 64 followed by dynamic batches of dyn_batch (64 or
// 128) rays.  Every ray of the frame must be handed out exactly once, every batch must be contiguous in the frame's order
// and inside the frame.  Returns the number of violations (0 = sound) or a negative status.
int ptc_check_feed(uint32_t n, uint32_t static_eighths, uint32_t dyn_batch)
{
  if (static_eighths > 8u || (dyn_batch != 64u && dyn_batch != 128u) || n > (1u << 28)) return PTC_ERR_INVALID;
  std::vector<uint8_t> seen(n, 0);
  int bad = 0;
  auto hand_out = [&](uint32_t begin, uint32_t end) {
    if (end > n || begin >= end) { ++bad; return; }
    for (uint32_t q = begin; q < end; ++q) {
      if (seen[q]) ++bad;
      seen[q] = 1;
    }
  };
  const uint32_t rs = feed_rules::region_size_of(n);
  uint64_t total = 0;
  for (uint32_t r = 0; r < 8u; ++r) {
    const uint32_t len = feed_rules::region_len_of(n, rs, r);
    total += len;
    const uint32_t stat = feed_rules::static_batches_of(len, static_eighths);
    if ((uint64_t)stat * 64u > len) { ++bad; continue; }
    for (uint32_t k = 0; k < stat; ++k) {  // BatchFeed::acquire, static part: full batches
      const uint32_t begin = feed_rules::pos_of(rs, r, k * 64u);
      hand_out(begin, begin + 64u);
    }
    for (uint32_t b = stat * 64u; b < len; b += dyn_batch) {  // ... dynamic part: the cursor advances by dyn_batch
      const uint32_t begin = feed_rules::pos_of(rs, r, b);
      const uint32_t count = std::min(len, b + dyn_batch) - b;
      hand_out(begin, begin + count);
      // a batch of two must be contiguous: its second half where the map puts it
      if (count > 64u && feed_rules::pos_of(rs, r, b + 64u) != begin + 64u) ++bad;
    }
  }
  if (total != n) ++bad;
  for (uint32_t q = 0; q < n; ++q)
    if (!seen[q]) ++bad;
  return bad;
}

// Test hook: the entries k_beam computes on the GPU for the uploaded scene's first traversal launch (its mesh object) and
// `camera` at the context's resolution, [tiles][4][8 floats] as ptc_check_beam lays them out.
int ptc_debug_beam_entries(ptc_ctx* ctx, const ptc_camera* camera, float* entries_out, uint64_t capacity_floats)
{
  if (!ctx || !camera || !entries_out) return PTC_ERR_INVALID;
  if (!ctx->has_scene || !ctx->pix_capacity || ctx->launches.empty()) return fail(ctx, PTC_ERR_INVALID, "no scene / frame / mesh launch");
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = sync_frames(ctx)) return rc;
  const uint32_t tx = (ctx->width + kBeamTile - 1u) / kBeamTile, ty = (ctx->height + kBeamTile - 1u) / kBeamTile;
  const size_t floats = (size_t)tx * ty * 32u;
  if (capacity_floats < floats) return fail(ctx, PTC_ERR_INVALID, "entries_out too small");
  float4* dev = nullptr;
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&dev), floats * sizeof(float)));
  DCameras cams{};
  cams.c[0] = make_camera(*camera, ctx->width, ctx->height);
  const uint8_t cam_of[1] = {0};
  DScene scene = ctx->scene;
  const uint32_t mesh_obj = ctx->launches[0].mesh;
  scene.cur = ctx->mesh_views[ctx->object_mesh[mesh_obj]];
  launch_beam(ctx->stream, scene, mesh_obj, cams, cam_of, 1u, tx, ty, ctx->mesh_nodes4[ctx->object_mesh[mesh_obj]], dev);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = hipMemcpy(entries_out, dev, floats * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  if (e != hipSuccess) return fail(ctx, PTC_ERR_HIP, std::string("beam entries: ") + hipGetErrorString(e));
  return PTC_OK;
}

int ptc_check_traversal_layout(const ptc_bvh_node* nodes, uint32_t node_count, uint64_t* checked_boxes)
{
  if (!nodes || node_count == 0u) return PTC_ERR_INVALID;
  Wide4Accel w4;
  if (int rc = build_wide4(nodes, node_count, w4)) return rc;
  // the reference tree: leaf of every depth-first rank, parents, and every node's range of leaf ranks
  std::vector<uint32_t> leaf_of_rank, parent(node_count, 0xffffffffu), first_rank(node_count, 0u), last_rank(node_count, 0u);
  {
    std::vector<uint32_t> stack{0u};
    while (!stack.empty()) {
      const uint32_t i = stack.back();
      stack.pop_back();
      if (nodes[i].primitive_count != 0u) {
        first_rank[i] = last_rank[i] = (uint32_t)leaf_of_rank.size();
        leaf_of_rank.push_back(i);
      } else {
        const uint32_t l = nodes[i].first_child_or_primitive;
        if (l + 1u >= node_count) return PTC_ERR_BVH;
        parent[l] = parent[l + 1u] = i;
        stack.push_back(l + 1u);
        stack.push_back(l);
      }
    }
    for (uint32_t i = node_count; i-- > 0u;)  // children come after their parent in the reference's array
      if (nodes[i].primitive_count == 0u) {
        first_rank[i] = first_rank[nodes[i].first_child_or_primitive];
        last_rank[i] = last_rank[nodes[i].first_child_or_primitive + 1u];
      }
  }
  std::unordered_map<uint64_t, uint32_t> node_of_range;
  node_of_range.reserve(node_count * 2u);
  for (uint32_t i = 0; i < node_count; ++i) node_of_range[((uint64_t)first_rank[i] << 32) | last_rank[i]] = i;

  int bad = 0;
  uint64_t boxes = 0;
  const uint32_t triangles = (uint32_t)leaf_of_rank.size();
  std::vector<uint32_t> seen(triangles, 0u);
  if (w4.root_ref & pt::kLeafBit) {
    if (triangles != 1u || (w4.root_ref & ~pt::kLeafBit) != 0u) ++bad;
    else seen[0] = 1u;
  } else {
    const uint32_t n4 = (uint32_t)(w4.nodes_q.size() / 16u);
    // rank range of every four-wide node: children are in depth-first order and nodes in depth-first preorder,
    // so a child node has a larger index than its parent
    std::vector<uint32_t> first4(n4, 0u), last4(n4, 0u);
    for (uint32_t n = n4; n-- > 0u;) {
      const uint32_t* q = &w4.nodes_q[(size_t)n * 16u];
      bool any = false;
      for (int c = 0; c < 4; ++c) {
        const uint32_t ref = q[12 + c];
        if (ref == w4.dummy_ref) {  // unused slot: must carry the inside-out box on every axis
          for (int ax = 0; ax < 3; ++ax)
            if (((q[4 + ax] >> (8 * c)) & 0xffu) != 255u || ((q[7 + ax] >> (8 * c)) & 0xffu) != 0u) ++bad;
          continue;
        }
        uint32_t a, b;
        if (ref & pt::kLeafBit) {
          a = b = ref & ~pt::kLeafBit;
        } else {
          if (ref <= n || ref >= n4) return PTC_ERR_BVH;
          a = first4[ref];
          b = last4[ref];
        }
        if (!any) first4[n] = a;
        else if (a != last4[n] + 1u) ++bad;  // the children tile their parent's leaves in order
        last4[n] = b;
        any = true;
      }
      if (!any) ++bad;
    }
    if (w4.root_ref >= n4 || first4[w4.root_ref] != 0u || last4[w4.root_ref] + 1u != triangles) ++bad;
    for (uint32_t n = 0; n < n4 && w4.root_ref < n4; ++n) {
      const uint32_t* q = &w4.nodes_q[(size_t)n * 16u];
      float origin[3];
      std::memcpy(origin, q, 12);
      for (int c = 0; c < 4; ++c) {
        const uint32_t ref = q[12 + c];
        if (ref == w4.dummy_ref) continue;
        uint32_t a, b;
        if (ref & pt::kLeafBit) {
          a = b = ref & ~pt::kLeafBit;
          if (a < triangles) ++seen[a];
        } else {
          a = first4[ref];
          b = last4[ref];
        }
        const auto it = node_of_range.find(((uint64_t)a << 32) | b);
        if (it == node_of_range.end()) {  // the child does not stand for a node of the reference tree
          ++bad;
          continue;
        }
        const ptc_bvh_node& x = nodes[it->second];
        for (int ax = 0; ax < 3; ++ax) {
          const uint32_t step_bits = q[ax == 0 ? 3 : 9 + ax];
          if (step_bits & 0x807fffffu) ++bad;  // a power of two
          const double step = std::ldexp(1.0, (int)(step_bits >> 23) - 127);
          const double lo = (double)origin[ax] + (double)((q[4 + ax] >> (8 * c)) & 0xffu) * step;
          const double hi = (double)origin[ax] + (double)((q[7 + ax] >> (8 * c)) & 0xffu) * step;
          if (lo > (double)x.aabb_min[ax] || hi < (double)x.aabb_max[ax]) ++bad;
        }
        ++boxes;
      }
    }
  }
  for (uint32_t r = 0; r < triangles; ++r) {
    if (seen[r] != 1u) ++bad;  // every triangle is a child of exactly one four-wide node
    const uint32_t leaf = leaf_of_rank[r];
    if (parent[leaf] != 0xffffffffu) {
      const float4 p0 = w4.leaf_parent[2u * (size_t)r], p1 = w4.leaf_parent[2u * (size_t)r + 1u];
      const ptc_bvh_node& p = nodes[parent[leaf]];
      if (p0.x != p.aabb_min[0] || p0.y != p.aabb_min[1] || p0.z != p.aabb_min[2] || p1.x != p.aabb_max[0] ||
          p1.y != p.aabb_max[1] || p1.z != p.aabb_max[2])
        ++bad;
    }
  }
  if (checked_boxes) *checked_boxes = boxes;
  return bad;
}

int ptc_selftest_math(ptc_ctx* ctx, const float* a, const float* b, uint32_t n, float* out_div, float* out_sqrt,
                      float* out_sin, float* out_cos)
{
  if (!ctx || !a || !b || !out_div || !out_sqrt || !out_sin || !out_cos) return PTC_ERR_INVALID;
  if (n == 0) return PTC_OK;
  if (int rc = bind_device(ctx)) return rc;
  std::vector<void*> pool;
  float* d[6] = {};
  for (auto& p : d)
    if (int rc = dev_alloc(ctx, pool, &p, n)) {
      free_pool(pool);
      return rc;
    }
  hipError_t e = hipMemcpyAsync(d[0], a, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d[1], b, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    launch_selftest(ctx->stream, d[0], d[1], n, d[2], d[3], d[4], d[5]);
    e = hipGetLastError();
  }
  float* outs[4] = {out_div, out_sqrt, out_sin, out_cos};
  for (int k = 0; k < 4 && e == hipSuccess; ++k)
    e = hipMemcpyAsync(outs[k], d[2 + k], n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  free_pool(pool);
  if (e != hipSuccess) return fail(ctx, PTC_ERR_HIP, std::string("selftest: ") + hipGetErrorString(e));
  return PTC_OK;
}

/* diagnostic (tools/debug): the persistent launch's state block of slot `slot`, after a synchronisation */
int ptc_debug_persist(ptc_ctx* ctx, int slot, void* dst, uint64_t bytes)
{
  if (!ctx || !dst || slot < 0 || slot >= (int)ctx->slots.size() || !ctx->slots[(size_t)slot].persist) return PTC_ERR_INVALID;
  if (int rc = bind_device(ctx)) return rc;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(dst, ctx->slots[(size_t)slot].persist, std::min<uint64_t>(bytes, sizeof(DPersist)), hipMemcpyDeviceToHost));
  return PTC_OK;
}

}  // extern "C"

