// ptcore_scene.cpp -- ptc_upload_scene and what it needs: validation, the world-space balls of sphere objects, the reference
// BVH and the traversal layouts (host or device side), ptc_build_bvh*, ptc_make_object.  Part of libptcore.so (ptcore_ctx.hpp).
#include "ptcore_ctx.hpp"

using namespace pt;
using namespace ptcd;

namespace {

// The world-space ball around a sphere object (DScene::sphere_ball), in double precision with the roundings of the
// float copies charged to the radius: centre = M (c, 1), radius = r * (largest singular value of M's 3 x 3 part).
// A matrix whose last row is not (0, 0, 0, 1), anything non-finite, a mesh object: radius -1 (no ball, never skipped).
static void sphere_ball_of(const ptc_object& o, const ptc_sphere* spheres, uint32_t sphere_count, uint32_t material, float4* out)
{
  out[0] = make_float4(0.f, 0.f, 0.f, -1.0f);
  for (uint32_t k = 1; k < kSphereTab; ++k) out[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (o.type != 0u || o.index >= sphere_count) return;
  const float* m = o.m;  // column-major: m[4 * col + row]
  if (!(m[3] == 0.0f && m[7] == 0.0f && m[11] == 0.0f && m[15] == 1.0f)) return;
  const ptc_sphere& sp = spheres[o.index];
  double a[3][3];  // a[row][col]
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) a[r][c] = (double)m[4 * c + r];
  // largest eigenvalue of A^T A by power iteration from three starts (symmetric positive semi-definite 3 x 3), then
  // bounded from above by the Frobenius norm and pushed up by 1e-6 relative: an upper bound is all that is needed
  double g[3][3];
  double frob2 = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      g[i][j] = 0.0;
      for (int k = 0; k < 3; ++k) g[i][j] += a[k][i] * a[k][j];
      frob2 += a[i][j] * a[i][j];
    }
  if (!std::isfinite(frob2) || frob2 <= 0.0) return;
  double lam = 0.0;
  for (int start = 0; start < 3; ++start) {
    double v[3] = {start == 0 ? 1.0 : 0.3, start == 1 ? 1.0 : 0.2, start == 2 ? 1.0 : 0.1};
    double l = 0.0;
    for (int it = 0; it < 200; ++it) {
      double w[3];
      for (int i = 0; i < 3; ++i) w[i] = g[i][0] * v[0] + g[i][1] * v[1] + g[i][2] * v[2];
      const double n = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      if (!(n > 0.0)) break;
      for (int i = 0; i < 3; ++i) v[i] = w[i] / n;
      l = n;
    }
    lam = std::max(lam, l);
  }
  // power iteration approaches the eigenvalue from below: the Gershgorin bound of G is a true upper bound; take the
  // smaller of it and the Frobenius norm, but never less than the iterate
  double gersh = 0.0;
  for (int i = 0; i < 3; ++i) gersh = std::max(gersh, std::fabs(g[i][0]) + std::fabs(g[i][1]) + std::fabs(g[i][2]));
  // (only proven bounds: the iterate approaches from below and is no bound, however close -- round 4 took "iterate plus
  // 1 %" when that was smaller, which an anisotropic matrix with a slowly converging iteration could undercut)
  double lam_up = std::min(gersh, frob2);
  lam_up = std::max(lam_up, lam);
  const double sigma = std::sqrt(lam_up) * (1.0 + 1e-6);
  const double cx = a[0][0] * sp.center[0] + a[0][1] * sp.center[1] + a[0][2] * sp.center[2] + (double)m[12];
  const double cy = a[1][0] * sp.center[0] + a[1][1] * sp.center[1] + a[1][2] * sp.center[2] + (double)m[13];
  const double cz = a[2][0] * sp.center[0] + a[2][1] * sp.center[1] + a[2][2] * sp.center[2] + (double)m[14];
  const double rad = std::fabs((double)sp.radius) * sigma;
  if (!std::isfinite(cx + cy + cz + rad)) return;
  const float fx = (float)cx, fy = (float)cy, fz = (float)cz;
  // What separates the ball the kernels compute with from the sphere the reference's float sequence sees, as a length:
  // the rounding of the centre to float (slack); the one rounding of `origin + inverse translation` in inverse_transform_ray,
  // which is relative to the OBJECT-space origin and so carries 2^-24 of the sphere's own centre; and the rounding of the hit
  // point back in world space (2^-24 of its coordinates), which moves the distance the reference records against the root.
  // The OUTER ball (radius + that) contains what the reference can hit: it decides "missed" and the lower bounds; the INNER
  // ball (radius - that, row 1 .z) lies inside it: "surely hit" and the upper bounds come from it (round 4 took the outer
  // radius for both, which is the wrong way round for the latter -- a small sphere far from the origin).
  const double slack = std::fabs(cx - fx) + std::fabs(cy - fy) + std::fabs(cz - fz);
  const double coord = std::fabs(cx) + std::fabs(cy) + std::fabs(cz) + std::fabs((double)sp.center[0]) + std::fabs((double)sp.center[1]) +
                       std::fabs((double)sp.center[2]) + 3.0 * rad;
  const double cerr = slack + coord * (1.0 / 4194304.0);  // 2^-22
  float fr = (float)((rad + cerr) * (1.0 + 1e-6));
  fr = std::nextafter(fr, INFINITY);
  float fin = (float)(std::max(0.0, std::fabs((double)sp.radius) * (1.0 - 1e-6) - cerr) * (1.0 - 1e-6));
  fin = fin > 0.0f ? std::nextafter(fin, 0.0f) : 0.0f;
  float inv_sigma = (float)((1.0 / sigma) * (1.0 - 1e-6));
  inv_sigma = std::nextafter(inv_sigma, 0.0f);
  out[0] = make_float4(fx, fy, fz, fr);
  // "simple": both matrices are a pure translation -- diagonal 1.0f, everything else outside the translation column a
  // zero of either sign (a cofactor inverse leaves -0.0f in a checkerboard).  The reference's matrix arithmetic then has
  // the same operands for every such object of a run except the translation, and a lane can fetch what differs for
  // itself (sphere_run_lanes): box, inverse translation, sphere, translation, material
  auto bits = [](float v) { uint32_t u; std::memcpy(&u, &v, 4); return u; };
  bool simple = true;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      if (c == 3 && r < 3) continue;  // the translation column
      for (const float* mat : {o.m, o.inv_m}) {
        const uint32_t u = bits(mat[4 * c + r]);
        simple = simple && (c == r ? u == 0x3f800000u : (u & 0x7fffffffu) == 0u);
      }
    }
  for (int r = 0; r < 3; ++r) simple = simple && std::isfinite(o.m[12 + r]) && std::isfinite(o.inv_m[12 + r]);
  out[1] = make_float4(inv_sigma, simple ? 1.0f : 0.0f, simple ? fin : 0.0f, 0.f);  // (.z: a simple object does not stretch)
  float mat_f;
  std::memcpy(&mat_f, &material, 4);
  out[2] = make_float4(o.aabb_min[0], o.aabb_min[1], o.aabb_min[2], o.inv_m[12]);
  out[3] = make_float4(o.aabb_max[0], o.aabb_max[1], o.aabb_max[2], o.inv_m[13]);
  out[4] = make_float4(sp.center[0], sp.center[1], sp.center[2], o.inv_m[14]);
  out[5] = make_float4(o.m[12], o.m[13], o.m[14], sp.radius);
  out[6] = make_float4(mat_f, 0.f, 0.f, 0.f);
}

int validate_scene(ptc_ctx* ctx, const ptc_scene_desc* s)
{
  if (s->object_count && (!s->objects || !s->object_material_indices)) return fail(ctx, PTC_ERR_INVALID, "objects missing");
  if (s->sphere_count && !s->spheres) return fail(ctx, PTC_ERR_INVALID, "spheres missing");
  if (s->material_count && !s->materials) return fail(ctx, PTC_ERR_INVALID, "materials missing");
  if (s->index_count % 3u) return fail(ctx, PTC_ERR_INVALID, "index_count is not a multiple of 3");
  if (s->object_count > 0xffffu) return fail(ctx, PTC_ERR_INVALID, "more than 65535 objects");
  if (s->index_count && (!s->indices || !s->positions)) return fail(ctx, PTC_ERR_INVALID, "mesh arrays missing");
  if (!s->meshes) {
    for (uint32_t i = 0; i < s->index_count; ++i)
      if (s->indices[i] >= s->vertex_count) return fail(ctx, PTC_ERR_INVALID, "vertex index out of range");
  } else {
    if (s->mesh_count > 0xffffu) return fail(ctx, PTC_ERR_INVALID, "more than 65535 meshes");
    for (uint32_t m = 0; m < s->mesh_count; ++m) {
      const ptc_mesh_range& r = s->meshes[m];
      if ((uint64_t)r.first_vertex + r.vertex_count > s->vertex_count || (uint64_t)r.first_index + r.index_count > s->index_count ||
          r.index_count % 3u)
        return fail(ctx, PTC_ERR_INVALID, "mesh range outside the vertex / index arrays");
      if (r.bvh_node_count && (!s->bvh || (uint64_t)r.first_bvh_node + r.bvh_node_count > s->bvh_node_count))
        return fail(ctx, PTC_ERR_INVALID, "mesh range outside the BVH array");
      for (uint32_t i = 0; i < r.index_count; ++i)
        if (s->indices[r.first_index + i] >= r.vertex_count) return fail(ctx, PTC_ERR_INVALID, "vertex index out of range");
    }
  }
  for (uint32_t i = 0; i < s->object_count; ++i) {
    const ptc_object& o = s->objects[i];
    if (o.type > 1u) return fail(ctx, PTC_ERR_INVALID, "unknown object type");
    if (o.type == 0u && o.index >= s->sphere_count) return fail(ctx, PTC_ERR_INVALID, "sphere index out of range");
    if (o.type == 1u && s->meshes && o.index >= s->mesh_count) return fail(ctx, PTC_ERR_INVALID, "mesh index out of range");
    if (s->object_material_indices[i] >= s->material_count) return fail(ctx, PTC_ERR_INVALID, "material index out of range");
  }
  for (uint32_t i = 0; i < s->material_count; ++i)
    if (s->materials[i].type < 0 || s->materials[i].type > 2) return fail(ctx, PTC_ERR_INVALID, "unknown material type");
  return PTC_OK;
}

int validate_bvh(ptc_ctx* ctx, const ptc_bvh_node* nodes, uint32_t count, uint32_t index_count)
{
  for (uint32_t i = 0; i < count; ++i) {
    const ptc_bvh_node& n = nodes[i];
    if (n.primitive_count != 0u) {
      if ((uint64_t)n.first_child_or_primitive + 2u >= index_count) return fail(ctx, PTC_ERR_INVALID, "BVH leaf out of range");
    } else if ((uint64_t)n.first_child_or_primitive + 1u >= count || n.first_child_or_primitive <= i) {
      return fail(ctx, PTC_ERR_INVALID, "BVH child out of range");
    }
    for (int k = 0; k < 3; ++k)
      if (!(n.aabb_min[k] <= n.aabb_max[k])) return fail(ctx, PTC_ERR_INVALID, "BVH node with an empty or NaN box");
  }
  return PTC_OK;
}

// depth of a tree numbered children-after-parents; level_base (optional) gets the first node of every depth plus the
// node count when the nodes are stored depth by depth (the reference's breadth-first numbering), else it is left empty
uint32_t bvh_depth_of(const ptc_bvh_node* nodes, uint32_t count, std::vector<uint32_t>* level_base = nullptr)
{
  std::vector<uint32_t> depth(count, 0u);
  uint32_t deepest = 0;
  bool by_level = true;
  if (level_base) level_base->assign(1, 0u);
  for (uint32_t i = 0; i < count; ++i) {
    if (depth[i] < deepest) by_level = false;
    if (depth[i] > deepest && level_base) level_base->push_back(i);
    deepest = std::max(deepest, depth[i]);
    if (nodes[i].primitive_count == 0u) {
      depth[nodes[i].first_child_or_primitive] = depth[i] + 1;
      depth[nodes[i].first_child_or_primitive + 1] = depth[i] + 1;
    }
  }
  if (level_base) {
    level_base->push_back(count);
    if (!by_level || level_base->size() != (size_t)deepest + 2u) level_base->clear();
  }
  return deepest;
}

}  // namespace

namespace {
// The reference BVH of a mesh built on the device.  nodes_host gets the 2T-1 nodes in the reference's layout;
// *packed_out (when asked for) keeps the device copy in DScene::bvh's layout, owned by the caller.
int bvh_on_device(ptc_ctx* ctx, const float* positions, uint32_t vertex_count, const uint32_t* indices, uint32_t index_count,
                  ptc_bvh_node* nodes_host, uint32_t* max_depth, float4** packed_out, std::vector<uint32_t>* level_base = nullptr)
{
  const uint32_t T = index_count / 3u;
  if (T == 0u) return fail(ctx, PTC_ERR_BVH, "empty mesh");
  for (uint32_t i = 0; i < T * 3u; ++i)
    if (indices[i] >= vertex_count) return fail(ctx, PTC_ERR_INVALID, "vertex index out of range");
  std::vector<void*> pool;
  const float* d_pos = nullptr;
  const uint32_t* d_idx = nullptr;
  float4* d_packed = nullptr;
  ptc_bvh_node* d_nodes = nullptr;
  const size_t count = 2u * (size_t)T - 1u;
  int rc = upload(ctx, pool, &d_pos, positions, (size_t)vertex_count * 3u);
  if (!rc) rc = upload(ctx, pool, &d_idx, indices, (size_t)T * 3u);
  if (!rc) rc = dev_alloc(ctx, pool, &d_packed, 2u * count);
  if (!rc && nodes_host) rc = dev_alloc(ctx, pool, &d_nodes, count);
  uint32_t built = 0u;
  if (!rc) {
    rc = build_bvh_device(ctx->stream, d_pos, d_idx, T * 3u, d_packed, d_nodes, &built, max_depth, level_base);
    if (rc) fail(ctx, rc, rc == PTC_ERR_BVH ? "BVH build failed (empty SAH side: coincident centroids?)" : "device BVH build failed");
  }
  if (!rc && nodes_host && hipMemcpy(nodes_host, d_nodes, count * sizeof(ptc_bvh_node), hipMemcpyDeviceToHost) != hipSuccess)
    rc = fail(ctx, PTC_ERR_HIP, "device BVH download failed");
  for (void* p : pool)
    if (p != d_packed || rc || !packed_out) (void)hipFree(p);
  if (!rc && packed_out) *packed_out = d_packed;
  return rc ? rc : (int)built;
}
}  // namespace

namespace {

// one mesh of the scene on its way to the device
struct MeshWork {
  // input slice
  const float* positions = nullptr;
  uint32_t vertex_count = 0;
  const uint32_t* indices = nullptr;
  uint32_t index_count = 0;
  const ptc_bvh_node* caller_bvh = nullptr;
  uint32_t caller_nodes = 0;
  // reference BVH
  std::vector<ptc_bvh_node> built;   // host copy of a BVH built here (only when something on the host needs it)
  const ptc_bvh_node* nodes = nullptr;
  std::vector<uint32_t> level_base;  // first node of every depth + the node count, when the nodes are stored depth by depth
  float4* dev_packed = nullptr;      // the device builder's output, already in DMeshView::bvh's layout
  uint32_t node_count = 0, depth = 0;
  // layouts
  DMeshView view{};
  const uint32_t* tri_order_dev = nullptr;  // depth-first rank -> triangle (device layouts)
  std::vector<uint32_t> tri_order_host;     // ... (host layouts)
  uint32_t triangles = 0, w4_depth = 0, w4_nodes = 0;
  bool layouts_on_device = false;
  ~MeshWork() { if (dev_packed) (void)hipFree(dev_packed); }
};

}  // namespace

extern "C" {

int ptc_upload_scene(ptc_ctx* ctx, const ptc_scene_desc* s)
{
  if (!ctx || !s) return PTC_ERR_INVALID;
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = validate_scene(ctx, s)) return rc;

  ptc_upload_times times{};
  auto t_start = std::chrono::steady_clock::now(), t_lap = t_start;
  auto lap = [&](float& into) {
    const auto now = std::chrono::steady_clock::now();
    into += std::chrono::duration<float, std::milli>(now - t_lap).count();
    t_lap = now;
  };

  // The meshes of the scene.  The reference keeps ONE mesh whatever the scene file says (scene_description.cpp:42,95),
  // which is what a description without a mesh table means here; with a table (ptc_mesh_range) every mesh object
  // instantiates the mesh its `index` names.
  const uint32_t mesh_count = s->meshes ? s->mesh_count : (s->index_count ? 1u : 0u);
  std::vector<MeshWork> meshes(mesh_count);
  for (uint32_t m = 0; m < mesh_count; ++m) {
    MeshWork& w = meshes[m];
    if (s->meshes) {
      const ptc_mesh_range& r = s->meshes[m];
      w.positions = s->positions + 3u * (size_t)r.first_vertex;
      w.vertex_count = r.vertex_count;
      w.indices = s->indices + r.first_index;
      w.index_count = r.index_count;
      w.caller_bvh = s->bvh && r.bvh_node_count ? s->bvh + r.first_bvh_node : nullptr;
      w.caller_nodes = w.caller_bvh ? r.bvh_node_count : 0u;
    } else {
      w.positions = s->positions;
      w.vertex_count = s->vertex_count;
      w.indices = s->indices;
      w.index_count = s->index_count;
      w.caller_bvh = s->bvh;
      w.caller_nodes = s->bvh ? s->bvh_node_count : 0u;
    }
  }

  // ---- phase 1: the reference BVH of every mesh (scene_description.cpp:99-101), unless the caller brought it.  The
  // old scene is still intact: a failure here leaves the context as it was.
  uint32_t deepest = 0u, total_nodes = 0u, total_triangles = 0u;
  for (MeshWork& w : meshes) {
    if (w.index_count == 0u) continue;  // (the reference panics on an empty mesh, bvh.cpp:200; here: a mesh nobody can hit)
    if (!w.caller_bvh) {
      int rc;
      if (ctx->bvh_on_device) {
        const bool host_copy = !ctx->layout_on_device;
        if (host_copy) w.built.resize((size_t)w.index_count / 3u * 2u);
        rc = bvh_on_device(ctx, w.positions, w.vertex_count, w.indices, w.index_count, host_copy ? w.built.data() : nullptr,
                           &w.depth, &w.dev_packed, &w.level_base);
        if (rc < 0) return rc;
        times.bvh_on_device = 1u;
        w.nodes = host_copy ? w.built.data() : nullptr;
      } else {
        w.built.resize((size_t)w.index_count / 3u * 2u);
        rc = build_bvh(w.positions, w.vertex_count, w.indices, w.index_count, w.built.data(), &w.depth);
        if (rc < 0) return fail(ctx, rc, "BVH build failed (empty SAH side: coincident centroids?)");
        w.nodes = w.built.data();
        (void)bvh_depth_of(w.nodes, (uint32_t)rc, &w.level_base);
      }
      w.node_count = (uint32_t)rc;
      lap(times.bvh_build_ms);
    } else {
      w.nodes = w.caller_bvh;
      w.node_count = w.caller_nodes;
      if (int rc = validate_bvh(ctx, w.nodes, w.node_count, w.index_count)) return rc;
      w.depth = bvh_depth_of(w.nodes, w.node_count, &w.level_base);
      lap(times.copy_ms);
    }
    // depth-first traversal pushes two children per inner node popped: stack need = depth + 1
    if (w.node_count && w.depth + 2u > (uint32_t)kStackDepth)
      return fail(ctx, PTC_ERR_STACK, "BVH depth " + std::to_string(w.depth) + " exceeds the traversal stack");
    // the layouts come from the device when the nodes are stored depth by depth (the reference's breadth-first order:
    // always, unless the caller brought a tree numbered some other way)
    w.layouts_on_device = ctx->layout_on_device && w.node_count != 0u && !w.level_base.empty();
    if (!w.layouts_on_device && w.node_count != 0u && !w.nodes) return fail(ctx, PTC_ERR_INVALID, "internal: no host copy of the BVH");
    deepest = std::max(deepest, w.depth);
    total_nodes += w.node_count;
    total_triangles += w.index_count / 3u;
  }

  // ---- phase 2: the old scene goes.  Iterations queued or in flight were asked for against it: trace them first
  if (int rc = sync_frames(ctx)) return rc;
  free_pool(ctx->scene_allocs);
  ctx->has_scene = false;
  ++ctx->scene_serial;
  DScene d{};
  const DObject* objects = nullptr;
  if (int rc = upload(ctx, ctx->scene_allocs, &objects, reinterpret_cast<const DObject*>(s->objects), s->object_count)) return rc;
  d.objects = objects;
  if (int rc = upload(ctx, ctx->scene_allocs, &d.object_material, s->object_material_indices, s->object_count)) return rc;
  if (int rc = upload(ctx, ctx->scene_allocs, &d.spheres, reinterpret_cast<const float4*>(s->spheres), s->sphere_count)) return rc;
  const DMaterial* mats = nullptr;
  if (int rc = upload(ctx, ctx->scene_allocs, &mats, reinterpret_cast<const DMaterial*>(s->materials), s->material_count)) return rc;
  d.materials = mats;
  {
    std::vector<float4> balls((size_t)s->object_count * kSphereTab);
    ctx->sphere_class.assign(s->object_count, 0u);
    std::vector<uint32_t> class_first;  // first object of every class
    for (uint32_t i = 0; i < s->object_count; ++i) {
      sphere_ball_of(s->objects[i], s->spheres, s->sphere_count, s->object_material_indices[i], &balls[(size_t)kSphereTab * i]);
      if (balls[(size_t)kSphereTab * i + 1u].y == 0.0f) continue;
      auto same = [&](const ptc_object& a, const ptc_object& b) {
        for (int c = 0; c < 4; ++c)
          for (int r = 0; r < 4; ++r) {
            if (c == 3 && r < 3) continue;
            if (std::memcmp(&a.m[4 * c + r], &b.m[4 * c + r], 4) != 0 || std::memcmp(&a.inv_m[4 * c + r], &b.inv_m[4 * c + r], 4) != 0) return false;
          }
        return true;
      };
      uint32_t k = 0;
      while (k < class_first.size() && !same(s->objects[class_first[k]], s->objects[i])) ++k;
      if (k == class_first.size()) class_first.push_back(i);
      ctx->sphere_class[i] = k + 1u;
    }
    if (int rc = upload(ctx, ctx->scene_allocs, &d.sphere_ball, balls.data(), balls.size())) return rc;
  }
  lap(times.copy_ms);

  // ---- phase 3: per mesh, the arrays of the reference layout and the layouts for the fast traversals (wide inner
  // records, the four-wide quantised tree, the depth-first leaf order)
  for (MeshWork& w : meshes) {
    DMeshView& v = w.view;
    w.triangles = w.node_count ? (w.node_count + 1u) / 2u : 0u;
    if (int rc = upload(ctx, ctx->scene_allocs, &v.positions, w.positions, (size_t)w.vertex_count * 3u)) return rc;
    if (int rc = upload(ctx, ctx->scene_allocs, &v.indices, w.indices, w.index_count)) return rc;
    if (w.dev_packed) {
      ctx->scene_allocs.push_back(w.dev_packed);
      v.bvh = w.dev_packed;
      w.dev_packed = nullptr;
    } else {
      // node -> two float4: {min.xyz, first}, {max.xyz, count}
      std::vector<float4> packed((size_t)w.node_count * 2u);
      for (uint32_t i = 0; i < w.node_count; ++i) {
        const ptc_bvh_node& n = w.nodes[i];
        float fbits, cbits;
        std::memcpy(&fbits, &n.first_child_or_primitive, 4);
        std::memcpy(&cbits, &n.primitive_count, 4);
        packed[2u * i] = make_float4(n.aabb_min[0], n.aabb_min[1], n.aabb_min[2], fbits);
        packed[2u * i + 1u] = make_float4(n.aabb_max[0], n.aabb_max[1], n.aabb_max[2], cbits);
      }
      if (int rc = upload(ctx, ctx->scene_allocs, &v.bvh, packed.data(), packed.size())) return rc;
    }
    v.bvh_node_count = w.node_count;
    lap(times.copy_ms);
    if (w.layouts_on_device) {
      DeviceLayouts lay;
      const int rc = build_layouts_device(ctx->stream, v.bvh, w.node_count, w.level_base, &lay);
      for (void* q : {(void*)lay.nodes_q, (void*)lay.leaf_parent, (void*)lay.tri_order, (void*)lay.wide})
        if (q) ctx->scene_allocs.push_back(q);
      if (rc) return fail(ctx, rc, "traversal layouts failed on the device");
      v.wide = lay.wide;
      v.leaf_parent = lay.leaf_parent;
      v.bvh4q = reinterpret_cast<const uint4*>(lay.nodes_q);
      v.bvh4_root = lay.root_ref4;
      v.dummy_ref = lay.dummy_ref;
      v.root_ref = lay.root_ref2;
      std::memcpy(v.root_min, lay.root_min, sizeof v.root_min);
      std::memcpy(v.root_max, lay.root_max, sizeof v.root_max);
      w.tri_order_dev = lay.tri_order;
      w.w4_depth = lay.wide4_depth;
      w.w4_nodes = lay.wide4_nodes;
      times.layout_on_device = 1u;
      lap(times.layout_ms);
    } else {
      WideAccel wa;
      if (int rc = build_wide(w.nodes, w.node_count, wa)) return fail(ctx, rc, "wide BVH layout failed");
      Wide4Accel w4;
      if (int rc = build_wide4(w.nodes, w.node_count, w4)) return fail(ctx, rc, "four-wide BVH layout failed");
      lap(times.layout_ms);
      if (int rc = upload(ctx, ctx->scene_allocs, &v.wide, wa.wide.data(), wa.wide.size())) return rc;
      if (int rc = upload(ctx, ctx->scene_allocs, &v.leaf_parent, w4.leaf_parent.data(), w4.leaf_parent.size())) return rc;
      {
        const uint32_t* q = nullptr;
        if (int rc = upload(ctx, ctx->scene_allocs, &q, w4.nodes_q.data(), w4.nodes_q.size())) return rc;
        v.bvh4q = reinterpret_cast<const uint4*>(q);
      }
      v.bvh4_root = w4.root_ref;
      v.dummy_ref = w4.dummy_ref;
      v.root_ref = wa.root_ref;
      std::memcpy(v.root_min, wa.root_min, sizeof v.root_min);
      std::memcpy(v.root_max, wa.root_max, sizeof v.root_max);
      w.tri_order_host = std::move(wa.tri_order);
      w.w4_depth = w4.depth;
      w.w4_nodes = w4.node_count;
      lap(times.copy_ms);
    }
  }

  // ---- phase 4: per mesh OBJECT (instance), its world-space triangle records in depth-first order (+ one all-zero
  // record: the dummy triangle of the four-wide tree's unused slots), and the object -> mesh table
  std::vector<uint32_t> object_mesh(s->object_count, 0u), tri_base(s->object_count, 0u);
  size_t tri_records = 0;
  for (uint32_t i = 0; i < s->object_count; ++i) {
    if (s->objects[i].type != 1u) continue;
    const uint32_t m = s->meshes ? s->objects[i].index : 0u;
    object_mesh[i] = m;
    tri_base[i] = (uint32_t)tri_records;
    if (m < mesh_count) tri_records += (size_t)meshes[m].triangles + 1u;
    if (tri_records > 0x7fffffffull) return fail(ctx, PTC_ERR_OOM, "too many instance triangles");
  }
  {
    float4* tris = nullptr;
    if (int rc = dev_alloc(ctx, ctx->scene_allocs, &tris, tri_records * kTriVec4)) return rc;
    if (tri_records) HIP_TRY(ctx, hipMemsetAsync(tris, 0, tri_records * kTriVec4 * sizeof(float4), ctx->stream));
    std::vector<float4> host_tris;
    for (uint32_t i = 0; i < s->object_count; ++i) {
      if (s->objects[i].type != 1u || object_mesh[i] >= mesh_count) continue;
      const MeshWork& w = meshes[object_mesh[i]];
      if (w.triangles == 0u) continue;
      m4 m;
      std::memcpy(&m, s->objects[i].m, sizeof m);
      float4* dst = tris + (size_t)tri_base[i] * kTriVec4;
      if (w.tri_order_dev) {
        launch_instance_triangles(ctx->stream, m, w.view.positions, w.view.indices, w.tri_order_dev, w.triangles, dst);
      } else {
        host_tris.assign((size_t)w.triangles * kTriVec4, make_float4(0.f, 0.f, 0.f, 0.f));
        build_instance_triangles(m, w.positions, w.indices, w.tri_order_host, host_tris.data());
        HIP_TRY(ctx, hipMemcpyAsync(dst, host_tris.data(), host_tris.size() * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      }
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    d.tris = tris;
    if (int rc = upload(ctx, ctx->scene_allocs, &d.object_tri_base, tri_base.data(), tri_base.size())) return rc;
    lap(times.triangles_ms);
  }
  ctx->mesh_views.clear();
  ctx->mesh_nodes4.clear();
  for (const MeshWork& w : meshes) {
    ctx->mesh_views.push_back(w.view);
    ctx->mesh_nodes4.push_back(w.w4_nodes);
  }
  ctx->object_mesh = object_mesh;
  if (int rc = upload(ctx, ctx->scene_allocs, &d.mesh_views, ctx->mesh_views.data(), ctx->mesh_views.size())) return rc;
  if (int rc = upload(ctx, ctx->scene_allocs, &d.object_mesh, object_mesh.data(), object_mesh.size())) return rc;
  if (!ctx->mesh_views.empty()) d.cur = ctx->mesh_views[0];
  lap(times.copy_ms);

  // sizes of mesh 0's arrays (ptc_download_layout)
  {
    const MeshWork* w0 = meshes.empty() ? nullptr : &meshes[0];
    const uint64_t t0 = w0 ? w0->triangles : 0u, n0 = w0 ? w0->node_count : 0u;
    ctx->layout_counts[0] = (uint64_t)(w0 ? w0->w4_nodes : 0u) * 64u;         // bvh4q
    ctx->layout_counts[1] = n0 ? (t0 + 1u) * 32u : 0u;                          // leaf_parent
    ctx->layout_counts[2] = (uint64_t)tri_records * 16u * kTriVec4;            // tris (all instances)
    ctx->layout_counts[3] = n0 ? (t0 - 1u) * 64u : 0u;                          // wide
    ctx->layout_counts[4] = n0 * 32u;                                           // bvh
  }
  d.refill_lanes = ctx->refill_lanes;
  d.split_idle = ctx->split_idle;
  d.static_eighths = ctx->static_eighths;
  d.force_slow = (uint32_t)ctx->force_slow;
  d.spill = nullptr;
  d.spill_stride = ctx->traverse_waves * kWave;
  // stack need of the four-wide walk: up to three entries per level; whatever exceeds the LDS part (24 entries)
  // goes to this per-thread overflow area (the areas themselves belong to the frame slots, batch_begin)
  d.spill_cap = 0;
  d.lds_cap = std::min<uint32_t>(ctx->lds_entries, (uint32_t)kLds4);
  uint32_t w4_depth = 0u, w4_nodes = 0u;
  for (const MeshWork& w : meshes) {
    if (!w.node_count) continue;
    // four-wide walk: up to three refs per level
    const uint32_t need4 = 3u * w.w4_depth + 2u > d.lds_cap ? 3u * w.w4_depth + 2u - d.lds_cap : 0u;
    d.spill_cap = std::max(d.spill_cap, need4);
    w4_depth = std::max(w4_depth, w.w4_depth);
    w4_nodes += w.w4_nodes;
  }
  ctx->bvh4_nodes = w4_nodes;
  ctx->bvh4_depth = w4_depth;
  d.object_count = s->object_count;
  // launches of the persistent pipeline (TraceLaunch).  A mesh object without nodes (empty mesh) is no launch; the
  // sphere code skips non-sphere objects, so the runs on both sides of it merge.
  ctx->launches.clear();
  {
    auto has_sphere = [&](uint32_t b, uint32_t e) {
      for (uint32_t i = b; i < e; ++i)
        if (s->objects[i].type == 0u) return true;
      return false;
    };
    auto mesh_nodes = [&](uint32_t i) { return object_mesh[i] < mesh_count ? meshes[object_mesh[i]].node_count : 0u; };
    uint32_t run_begin = 0;  // objects [run_begin, i) hold the spheres seen since the last mesh launch
    for (uint32_t i = 0; i < s->object_count; ++i)
      if (s->objects[i].type == 1u && mesh_nodes(i)) {
        const bool any = has_sphere(run_begin, i);
        ctx->launches.push_back({i, any ? run_begin : 0u, any ? i : 0u});
        run_begin = i + 1u;
      }
    const bool any = has_sphere(run_begin, s->object_count);
    ctx->tail_begin = any ? run_begin : 0u;
    ctx->tail_end = any ? s->object_count : 0u;
  }
  ctx->scene = d;
  ctx->has_scene = true;
  ctx->bvh_nodes = total_nodes;
  ctx->bvh_depth = deepest;
  ctx->triangles = total_triangles;
  if (hipDeviceSynchronize() != hipSuccess) return fail(ctx, PTC_ERR_HIP, "scene upload failed");
  lap(times.copy_ms);
  times.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_start).count();
  ctx->upload_times = times;
  return PTC_OK;
}

int ptc_build_bvh_device(ptc_ctx* ctx, const float* positions, uint32_t vertex_count, const uint32_t* indices,
                         uint32_t index_count, ptc_bvh_node* nodes, uint32_t* max_depth)
{
  if (!ctx || !positions || !indices || !nodes || index_count % 3u != 0u) return fail(ctx, PTC_ERR_INVALID, "bad arguments");
  if (int rc = bind_device(ctx)) return rc;
  return bvh_on_device(ctx, positions, vertex_count, indices, index_count, nodes, max_depth, nullptr);
}

int ptc_download_layout(ptc_ctx* ctx, int which, void* host, uint64_t capacity, uint64_t* bytes)
{
  if (!ctx || which < 0 || which > 4) return fail(ctx, PTC_ERR_INVALID, "layout: 0 bvh4q, 1 leaf_parent, 2 tris, 3 wide, 4 bvh");
  if (!ctx->has_scene) return fail(ctx, PTC_ERR_NO_SCENE, "no scene uploaded");
  if (int rc = bind_device(ctx)) return rc;
  const void* src[5] = {ctx->scene.cur.bvh4q, ctx->scene.cur.leaf_parent, ctx->scene.tris, ctx->scene.cur.wide, ctx->scene.cur.bvh};
  const uint64_t n = ctx->layout_counts[which];
  if (bytes) *bytes = n;
  if (!host) return PTC_OK;
  if (capacity < n) return fail(ctx, PTC_ERR_INVALID, "buffer too small");
  if (n) HIP_TRY(ctx, hipMemcpy(host, src[which], n, hipMemcpyDeviceToHost));
  return PTC_OK;
}

int ptc_get_upload_times(const ptc_ctx* ctx, ptc_upload_times* out)
{
  if (!ctx || !out) return PTC_ERR_INVALID;
  *out = ctx->upload_times;
  return PTC_OK;
}

int ptc_build_bvh(const float* positions, uint32_t vertex_count, const uint32_t* indices, uint32_t index_count,
                  ptc_bvh_node* nodes, uint32_t* max_depth)
{
  if (!positions || !indices || !nodes || index_count % 3u) return PTC_ERR_INVALID;
  return build_bvh(positions, vertex_count, indices, index_count, nodes, max_depth);
}

int ptc_make_object(uint32_t type, uint32_t index, const float* m16, const ptc_sphere* sphere, const float* mesh_aabb6,
                    ptc_object* out)
{
  return make_object(type, index, m16, sphere, mesh_aabb6, out);
}

}  // extern "C"

