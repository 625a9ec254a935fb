// ptcore.cpp -- the C ABI of libptcore.so (include/ptcore.h): context, buffers, frame loop.
//
// One ptc_ctx owns what the reference's PathTracer owns (path_tracer.hpp:68-84): the device scene, the
// path state, the hit records, the accumulated framebuffers, the two denoise ping-pong buffers and the
// iteration counter.  Differences that matter for speed, not results:
//   - no per-bounce host synchronisation: live-path counts stay in a device counter block and the
//     kernels of bounce b read live[b] themselves (the reference reads the Thrust partition result
//     back every bounce, path_tracer.cu:457);
//   - path state is ping-ponged between two buffers by the fused shade+compaction kernel instead of
//     being partitioned in place through a Thrust temporary.
#include "../../include/ptcore.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "pt_device.hpp"
#include "pt_host.hpp"
#include "pt_beam_rules.hpp"
#include "pt_feed_rules.hpp"

using namespace pt;

static_assert(sizeof(ptc_object) == sizeof(DObject), "ptc_object must match the device object");
static_assert(sizeof(ptc_material) == sizeof(DMaterial), "ptc_material must match the device material");
static_assert(sizeof(ptc_bvh_node) == 32, "BVH node is 32 bytes (bvh.hpp:30)");
static_assert(PTC_MAX_BOUNCES_CAP == kMaxBounces, "bounce cap mismatch");

static thread_local std::string g_create_error;

// in-flight path state a context allocates when the caller has not chosen frames_in_flight
constexpr uint64_t kAutoFrameBytes = 24ull << 30;

// The frames in flight run on separate HIP streams, and streams only overlap when they sit on different
// hardware queues; the runtime's default is 4 queues per process.  Ask for more before this library's first HIP
// call (no effect if the application has set the variable or has already initialised HIP itself: such an
// application exports GPU_MAX_HW_QUEUES on its own, see ptcore.h).
static void request_hw_queues()
{
  static const int once = setenv("GPU_MAX_HW_QUEUES", "24", 0);
  (void)once;
}

struct ptc_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;

  // scene
  std::vector<void*> scene_allocs;
  DScene scene{};
  bool has_scene = false;
  uint32_t bvh_nodes = 0, bvh_depth = 0, triangles = 0, bvh4_nodes = 0, bvh4_depth = 0;
  ptc_upload_times upload_times{};
  std::vector<DMeshView> mesh_views;   // host copy of DScene::mesh_views: a traversal launch gets its object's mesh as DScene::cur
  std::vector<uint32_t> object_mesh;
  std::vector<uint32_t> mesh_nodes4;   // four-wide nodes of every mesh (k_beam's range check)

  // frame
  uint32_t width = 0, height = 0;
  uint32_t pix_begin = 0, pix_count = 0, pix_capacity = 0;
  DBand band{0, 0, 0, 1, 0};
  std::vector<void*> frame_allocs;
  // Frames in flight: consecutive iterations are independent until they are folded into the framebuffer, and
  // the tail of every bounce is a handful of long rays (latency-bound), so iteration i runs on stream i % F
  // with its own path state and staging buffers; k_accumulate folds the staged samples in iteration order.
  struct FrameSlot {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DPaths paths[2]{};
    DHits hits{};
    uint32_t* chunk_counts = nullptr;   // "fused_shade" 0: k_tail_count -> k_scan -> k_shade
    uint32_t* chunk_offsets = nullptr;
    unsigned long long* tile_desc = nullptr;  // k_shade_fused: look-back descriptors, tile_stride per frame of the batch
    uint32_t tile_stride = 0;
    uint32_t shade_epoch = 0;           // look-back launches on these descriptors so far (1 .. 2^30 - 1, then round again)
    float4* beam_entries = nullptr; // "beam": entry points of the batch's cameras (DBeam), capacity x tiles x 8 float4
    DBeam beam{};                   // ... as bounce 0's first traversal launch gets them (entries null: off for this batch)
    DCameras beam_cams{};           // the cameras (of the beams) the entries in beam_entries were computed for ...
    uint32_t beam_count = 0;        // ... how many, for which scene upload and mesh object: the next batch of this slot with
    uint64_t beam_scene = 0;        //     the same cameras (a viewer that accumulates, the benchmark) skips k_beam
    uint32_t beam_obj = 0;
    uint32_t* slow_list = nullptr;  // slots of rays set aside for the exact redo at the end of a traversal launch
    uint32_t* slow_stack = nullptr; // that redo's traversal stack, [kStackDepth][kWave]
    uint8_t* octs = nullptr;        // "ray_sort": direction octant per slot of the rays of the next bounce
    uint32_t* order = nullptr;      // ... and the order in which the traversal lanes pick them up
    bool primary_finished = false;  // ... and finished the others itself: bounce 0's shade walks the list (launch_raygen)
    bool first_listed = false;      // ... k_raygen has listed the rays of bounce 0's first traversal launch (this batch)
    uint32_t* worklist = nullptr;   // "filter_rays": the rays of the next traversal launch that may hit one of its objects (k_spheres)
    uint2* spill = nullptr;         // traversal stack overflow area of this slot's launches (DScene::spill)
    size_t spill_elems = 0;
    DFrame stage{};
    DeviceCounters* counters = nullptr;  // one per frame of the batch
    hipEvent_t done = nullptr;  // after this slot's last accumulate
    uint32_t* live_host = nullptr;  // pinned: live[] of the slot's last batch (frame 0), copied back after `done`
    bool live_pending = false;      // ... and not yet looked at (launch sizing, see traverse_waves_for)
    int cur = 0;
    int work_slot = 0;
    int bounces_done = 0;
    DBatchInfo bi{};            // the batch being traced / traced last
    int capacity = 1;           // frames the slot's arrays hold
  };
  std::vector<FrameSlot> slots;
  int frames_in_flight = 64;
  bool frames_auto = true;  // not set by the caller: ptc_resize caps it so that the in-flight state stays under kAutoFrameBytes
  // Batches: up to `batch` consecutive iterations share the launches of a slot (DBatchInfo).  ptc_trace only
  // queues the iteration; the batch is enqueued when it is full or when anything else looks at the context.
  int batch_frames = 32;  // requested (ptc_set_param, before ptc_resize)
  int batch = 1;          // allocated per slot
  bool staged = false;    // samples go through staging buffers and k_accumulate
  struct Pending {
    DCamera cam;
    uint32_t iteration;
  };
  std::vector<Pending> pending;
  uint64_t batches_issued = 0;
  // Slots [0, big_slots) hold `batch` frames each; slots [big_slots, slots.size()) hold ONE frame: a batch of a
  // single iteration (a viewer that presents after every iteration, the stepwise calls) goes to one of those, so
  // that many such launches can be in flight on their own streams without the memory of full-size slots.
  int big_slots = 0;
  uint64_t singles_issued = 0;
  int active_slot = -1;          // slot of the frame being built by ptc_trace_begin/bounce/end
  int last_slot = 0;             // slot of the most recent finished frame
  hipEvent_t order_event = nullptr;  // last accumulate enqueued (accumulates run in iteration order)
  bool order_valid = false;
  hipEvent_t main_event = nullptr;   // last main-stream consumer that read the framebuffers asynchronously
  bool main_valid = false;
  DFrame fb{};
  float4* den_a = nullptr;
  float4* den_b = nullptr;
  float4* den_pos = nullptr;     // per-pixel view-space hit position of the accumulated depth (denoiser)
  const float4* result = nullptr;
  float* pack_buf = nullptr;     // 3 floats / pixel staging for downloads
  uint32_t* rgba_buf = nullptr;  // staging for host presents
  DeviceCounters* misc_counters = nullptr;  // flags of kernels outside the frame loop (ptc_intersect_rays)
  uint32_t slot_offset = 0;                 // "slot_offset" (multi-GPU: distinct random streams per rank)
  uint32_t* slot_offset_dev = nullptr;
  hipEvent_t xstream_event = nullptr;       // orders the stepwise calls between a frame's stream and ctx->stream
  // several GPUs (ptc_band_*): this rank's exported band buffer, and on the root the peers' mapped buffers
  float* band_buf = nullptr;                // 3 floats per pixel of pix_capacity
  struct Peer {
    void* mapped = nullptr;                 // hipIpcOpenMemHandle of the peer process's band buffer
    bool opened = false;
    ptc_band_handle h{};
  };
  std::vector<Peer> peers;                  // by rank
  float* gather_frame = nullptr;            // root: the whole frame, 3 floats per pixel
  uint32_t* gather_rgba = nullptr;
  hipEvent_t gather_ev[2] = {nullptr, nullptr};  // around the most recent gather launch (ptc_gather_last_us)
  bool gather_timed = false;

  int iteration = 0;
  int max_iterations = 1;
  int method = PTC_METHOD_STREAMING;
  int max_bounces = 50;
  ptc_denoiser_params den{10, 0.45f, 0.30f, 0.25f};
  DCamera cam{};
  bool have_cam = false;
  uint64_t frames = 0;

  int trace_variant = 3;  // 3: persistent lanes over the four-wide collapse, conservative FMA slabs, exact check of the winner (default); 0: reference-order traversal; 1: culled near-first traversal with exact box decisions
  // The closest-hit stage of the default variant, in object order: per mesh object a k_spheres launch for the run of
  // spheres in front of it ([pre_begin, pre_end), if it holds any) and a persistent traversal launch; the run that
  // ends the object list ([tail_begin, tail_end): everything, in a scene without a mesh) is tested by the kernel that ends the bounce (k_shade_fused; k_tail_count in the three-kernel form).
  struct TraceLaunch {
    uint32_t mesh, pre_begin, pre_end;
  };
  std::vector<TraceLaunch> launches;
  uint32_t tail_begin = 0, tail_end = 0;
  // per object: 0, or the class of a "simple" sphere object (sphere_ball_of) -- objects of one class have the same
  // matrix entries outside the translation columns; a run of one class (at most eight objects) takes sphere_run_lanes
  std::vector<uint32_t> sphere_class;
  bool sphere_lanes = true;   // "sphere_lanes"
  bool sphere_fold = true;    // "sphere_fold"
  bool beam = true;           // "beam": primary rays start at their tile's entry points (k_beam)
  uint64_t scene_serial = 0;  // counts ptc_upload_scene calls (entry points computed for another scene are stale)
  uint32_t beam_tiles_x = 0, beam_tiles_y = 0;
  uint32_t traverse_waves = 5120;
  uint32_t refill_lanes = 32;   // (20 until round 4: re-swept on its final code, profiles/r04_schedules.txt)
  uint32_t static_eighths = 4;  // (3 until round 4)
  bool merge_instances = true;  // "merge_instances": consecutive instances of one mesh walked by one launch (k_traverse4m)
  bool bvh_on_device = true;  // "bvh_build_on_device": the reference BVH of ptc_upload_scene from pt_bvh_gpu.hip
  bool layout_on_device = true;  // "layout_on_device": the traversal layouts derived from it, too
  uint64_t layout_counts[5] = {0, 0, 0, 0, 0};  // bytes of bvh4q, leaf_parent, tris, wide, bvh (ptc_download_layout)
  uint32_t split_idle = 8;    // "split_idle"
  uint32_t min_waves = 1024;  // "min_waves": fewest persistent wavefronts of a traversal launch
  uint32_t small_waves = 3072;        // "small_waves": ... of a launch with fewer than small_rays_per_lane rays per lane of a full one
  uint32_t small_rays_per_lane = 4;   // "small_rays_per_lane" (8 until round 3: bounces 5 and 6 of a 20-frame batch -- 5 to 8 rays
                                      // per lane -- are 12-14 % faster on all 5120 wavefronts than on 3072)
  // live paths entering each bounce of one recent frame (what a frame of this scene / camera looks like): the host
  // never waits for them, they only size the traversal launches
  uint32_t est_live[2 * (kMaxBounces + 1)] = {};  // live[], then listed_now[] (DeviceCounters) of a recent batch's first frame
  bool est_valid = false;
  bool filter_rays = true;    // "filter_rays": a sphere run in front of a mesh launch also lists the rays that launch has to walk
  bool fused_shade = true;    // "fused_shade": the end of a bounce in one pass (k_shade_fused); 0: k_tail_count -> k_scan -> k_shade
  int ray_sort = 0;           // "ray_sort": 1 = traversal lanes pick their rays up grouped by direction octant (bounces >= 1)
  int denoise_variant = 0;    // "denoise_variant": 0 = taps staged in LDS (default), 1 = taps through L1 / L2
  uint32_t lds_entries = kLds4;  // the kernels' LDS stack (pt_device.hpp); fewer only through "debug_lds_entries"
  int force_slow = 0;

  // measurement
  bool time_trace = false;
  bool count_tests = false;
  struct TimedLaunch {
    hipEvent_t start, stop;
    int bounce;
  };
  bool staging() const { return staged; }
  std::vector<TimedLaunch> timed;        // recorded, not yet read
  std::vector<hipEvent_t> free_events;
  double trace_ms[kMaxBounces] = {};
  uint32_t trace_launches[kMaxBounces] = {};
  uint64_t intersect_redone = 0;         // rays ptc_intersect_rays redid exactly (reported as slow_rays[0])
  double denoise_ms = 0.0;               // A-Trous passes (TimedLaunch::bounce == -1)
  uint32_t denoise_passes = 0;
};

namespace {

int fail(ptc_ctx* ctx, int code, const std::string& msg)
{
  if (ctx) ctx->err = msg;
  else g_create_error = msg;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail(ctx, e_ == hipErrorOutOfMemory ? PTC_ERR_OOM : PTC_ERR_HIP,                   \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                           \
  } while (0)

int check_last(ptc_ctx* ctx, const char* what)
{
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ctx, PTC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  return PTC_OK;
}

int bind_device(ptc_ctx* ctx)
{
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return PTC_OK;
}

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

template <typename T>
int dev_alloc(ptc_ctx* ctx, std::vector<void*>& pool, T** out, size_t count)
{
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(count * sizeof(T), 256);
  HIP_TRY(ctx, hipMalloc(&p, bytes));
  pool.push_back(p);
  *out = static_cast<T*>(p);
  return PTC_OK;
}

void free_pool(std::vector<void*>& pool)
{
  for (void* p : pool) (void)hipFree(p);
  pool.clear();
}

template <typename T>
int upload(ptc_ctx* ctx, std::vector<void*>& pool, const T** out, const T* host, size_t count)
{
  T* d = nullptr;
  int rc = dev_alloc(ctx, pool, &d, count);
  if (rc) return rc;
  if (count) HIP_TRY(ctx, hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
  *out = d;
  return PTC_OK;
}

// Camera::to_gpu_camera (camera.cpp:5-13) + the frame-invariant part of generate_ray (ray_gen.cu:37-47)
DCamera make_camera(const ptc_camera& c, uint32_t w, uint32_t h)
{
  DCamera d;
  d.cam = camera_matrix(c.position, c.rotation_wxyz);
  const f4 o = mul(d.cam, 0.0f, 0.0f, 0.0f, 1.0f);
  d.origin = mk3(o.x, o.y, o.z);
  const float aspect = (float)w / (float)h;
  d.vh = 2.0f * tanf(c.vfov / 2);
  d.vw = aspect * d.vh;
  // lower_left_corner = origin - horizontal/2 - vertical/2 - (0,0,focal)
  d.llx = ((0.0f - d.vw / 2.f) - 0.0f / 2.f) - 0.0f;
  d.lly = ((0.0f - 0.0f / 2.f) - d.vh / 2.f) - 0.0f;
  d.width = w;
  d.height = h;
  return d;
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

int flush_pending(ptc_ctx* ctx);

// wait (host-side) until every frame in flight has been folded into the framebuffers
int sync_frames(ptc_ctx* ctx)
{
  if (int rc = flush_pending(ctx)) return rc;
  for (auto& sl : ctx->slots)
    if (sl.stream) HIP_TRY(ctx, hipStreamSynchronize(sl.stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->order_valid = false;
  ctx->main_valid = false;
  return PTC_OK;
}

void free_slots(ptc_ctx* ctx)
{
  for (auto& sl : ctx->slots) {
    if (sl.own_stream && sl.stream) (void)hipStreamDestroy(sl.stream);
    if (sl.done) (void)hipEventDestroy(sl.done);
    if (sl.live_host) (void)hipHostFree(sl.live_host);
    if (sl.spill) (void)hipFree(sl.spill);
  }
  ctx->slots.clear();
  ctx->pending.clear();
  ctx->active_slot = -1;
  ctx->est_valid = false;
}

}  // namespace

extern "C" {

int ptc_abi_version(void) { return PTC_ABI_VERSION; }

int ptc_device_count(int* count)
{
  if (!count) return PTC_ERR_INVALID;
  request_hw_queues();
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    (void)hipGetLastError();
    return fail(nullptr, PTC_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = n;
  return PTC_OK;
}

const char* ptc_last_error(const ptc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int ptc_create(const ptc_config* config, ptc_ctx** out)
{
  if (!out) return fail(nullptr, PTC_ERR_INVALID, "out is NULL");
  *out = nullptr;
  request_hw_queues();
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(nullptr, PTC_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  }
  const int device = config ? config->device : 0;
  if (device < 0 || device >= n) return fail(nullptr, PTC_ERR_NO_DEVICE, "device ordinal out of range");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, PTC_ERR_HIP, "hipGetDeviceProperties failed");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, PTC_ERR_NO_DEVICE, std::string("kernels are built for gfx950 only; device is ") + prop.gcnArchName);

  ptc_ctx* ctx = new (std::nothrow) ptc_ctx();
  if (!ctx) return fail(nullptr, PTC_ERR_OOM, "out of host memory");
  ctx->device = device;
  if (config) {
    if (config->max_bounces > 0) ctx->max_bounces = std::min(config->max_bounces, (int)kMaxBounces);
    if (config->method == PTC_METHOD_MEGAKERNEL) ctx->method = PTC_METHOD_MEGAKERNEL;
  }
  auto bail = [&](int rc) {
    g_create_error = ctx->err;
    ptc_destroy(ctx);
    return rc;
  };
  if (hipSetDevice(device) != hipSuccess) return bail(fail(ctx, PTC_ERR_HIP, "hipSetDevice failed"));
  if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(ctx, PTC_ERR_HIP, "hipStreamCreate failed"));
  ctx->stream = ctx->own_stream;
  void* p = nullptr;
  if (hipMalloc(&p, sizeof(DeviceCounters)) != hipSuccess) return bail(fail(ctx, PTC_ERR_OOM, "hipMalloc(counters) failed"));
  ctx->misc_counters = static_cast<DeviceCounters*>(p);
  if (hipMemset(ctx->misc_counters, 0, sizeof(DeviceCounters)) != hipSuccess) return bail(fail(ctx, PTC_ERR_HIP, "hipMemset failed"));
  if (hipEventCreateWithFlags(&ctx->order_event, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->main_event, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->xstream_event, hipEventDisableTiming) != hipSuccess)
    return bail(fail(ctx, PTC_ERR_HIP, "hipEventCreate failed"));
  if (hipMalloc(&p, 256) != hipSuccess) return bail(fail(ctx, PTC_ERR_OOM, "hipMalloc(slot offset) failed"));
  ctx->slot_offset_dev = static_cast<uint32_t*>(p);
  if (hipMemset(ctx->slot_offset_dev, 0, 256) != hipSuccess) return bail(fail(ctx, PTC_ERR_HIP, "hipMemset failed"));
  *out = ctx;
  return PTC_OK;
}

void ptc_destroy(ptc_ctx* ctx)
{
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (auto& sl : ctx->slots)
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
  if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
  free_slots(ctx);
  free_pool(ctx->scene_allocs);
  free_pool(ctx->frame_allocs);
  for (auto& tl : ctx->timed) {
    (void)hipEventDestroy(tl.start);
    (void)hipEventDestroy(tl.stop);
  }
  for (hipEvent_t e : ctx->free_events) (void)hipEventDestroy(e);
  if (ctx->order_event) (void)hipEventDestroy(ctx->order_event);
  if (ctx->main_event) (void)hipEventDestroy(ctx->main_event);
  if (ctx->xstream_event) (void)hipEventDestroy(ctx->xstream_event);
  for (auto& peer : ctx->peers)
    if (peer.opened && peer.mapped) (void)hipIpcCloseMemHandle(peer.mapped);
  for (void* q : {(void*)ctx->band_buf, (void*)ctx->gather_frame, (void*)ctx->gather_rgba, (void*)ctx->slot_offset_dev})
    if (q) (void)hipFree(q);
  for (hipEvent_t e : ctx->gather_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->misc_counters) (void)hipFree(ctx->misc_counters);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int ptc_set_stream(ptc_ctx* ctx, void* hip_stream)
{
  if (!ctx) return PTC_ERR_INVALID;
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = sync_frames(ctx)) return rc;
  ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  if (!ctx->staged && !ctx->slots.empty()) ctx->slots[0].stream = ctx->stream;  // one frame in flight: trace on the caller's stream
  return PTC_OK;
}

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

int ptc_resize(ptc_ctx* ctx, uint32_t width, uint32_t height)
{
  if (!ctx || width < 2u || height < 2u) return fail(ctx, PTC_ERR_INVALID, "resolution must be at least 2x2");
  if ((uint64_t)width * height > 0x7fffffffull) return fail(ctx, PTC_ERR_INVALID, "too many pixels");
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = sync_frames(ctx)) return rc;
  free_slots(ctx);
  free_pool(ctx->frame_allocs);
  // nothing below is usable until this call has succeeded (frame_ready / ptc_download / ptc_present check these)
  ctx->pix_capacity = ctx->pix_count = 0;
  ctx->width = ctx->height = 0;
  ctx->fb = DFrame{};
  ctx->den_a = ctx->den_b = ctx->den_pos = nullptr;
  ctx->result = nullptr;
  ctx->pack_buf = nullptr;
  ctx->rgba_buf = nullptr;
  ctx->have_cam = false;
  // band / gather buffers belong to the old frame size (an exported handle dies with its buffer: export again)
  for (auto& peer : ctx->peers)
    if (peer.opened && peer.mapped) (void)hipIpcCloseMemHandle(peer.mapped);
  ctx->peers.clear();
  for (float** q : {&ctx->band_buf, &ctx->gather_frame}) {
    if (*q) (void)hipFree(*q);
    *q = nullptr;
  }
  if (ctx->gather_rgba) (void)hipFree(ctx->gather_rgba);
  ctx->gather_rgba = nullptr;
  ctx->gather_timed = false;
  const size_t P = (size_t)width * height;
  auto& pool = ctx->frame_allocs;
  const size_t chunks = (P + kChunk - 1) / kChunk;
  int frames = std::max(1, ctx->frames_in_flight);
  if (ctx->frames_auto) {
    // path state, hit records, staging: 164 bytes per pixel and frame in flight
    const uint64_t per_frame = 164ull * P;
    frames = (int)std::min<uint64_t>((uint64_t)frames, std::max<uint64_t>(1ull, kAutoFrameBytes / per_frame));
  }
  const int B = std::min({std::max(1, ctx->batch_frames), frames, kMaxBatch});
  if (ctx->frames_auto) frames -= frames % B;
  const int F = (frames + B - 1) / B;  // slots (streams); each holds a batch of B frames
  ctx->batch = B;
  ctx->staged = frames > 1;
  ctx->batches_issued = 0;
  ctx->singles_issued = 0;
  ctx->big_slots = F;
  const int singles = (ctx->staged && B > 1) ? 8 : 0;
  ctx->slots.resize((size_t)(F + singles));
  if (int rc = dev_alloc(ctx, pool, &ctx->fb.color4, P)) return rc;
  if (int rc = dev_alloc(ctx, pool, &ctx->fb.nd4, P)) return rc;
  for (int f = 0; f < F + singles; ++f) {
    auto& sl = ctx->slots[(size_t)f];
    const int B = f < F ? ctx->batch : 1;  // this slot's capacity (shadows the batch size above)
    sl.capacity = B;
    if (!ctx->staged) {
      sl.stream = ctx->stream;
      sl.own_stream = false;
    } else {
      HIP_TRY(ctx, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
      sl.own_stream = true;
    }
    HIP_TRY(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&sl.live_host), sizeof(uint32_t) * 2 * (kMaxBounces + 1), hipHostMallocDefault));
    const size_t BP = (size_t)B * P;  // frame f of the batch at element offset f * P (DBatchInfo::stride)
    for (int k = 0; k < 2; ++k) {
      if (int rc = dev_alloc(ctx, pool, &sl.paths[k].o4, BP)) return rc;
      if (int rc = dev_alloc(ctx, pool, &sl.paths[k].d4, BP)) return rc;
      if (int rc = dev_alloc(ctx, pool, &sl.paths[k].t4, BP)) return rc;
    }
    if (int rc = dev_alloc(ctx, pool, &sl.hits.tp, BP)) return rc;
    if (int rc = dev_alloc(ctx, pool, &sl.hits.nm, BP)) return rc;
    if (int rc = dev_alloc(ctx, pool, &sl.chunk_counts, (size_t)B * chunks)) return rc;
    if (int rc = dev_alloc(ctx, pool, &sl.chunk_offsets, (size_t)B * chunks)) return rc;
    sl.tile_stride = shade_tiles_per_frame((uint32_t)P);
    if (int rc = dev_alloc(ctx, pool, &sl.tile_desc, (size_t)B * sl.tile_stride)) return rc;
    HIP_TRY(ctx, hipMemsetAsync(sl.tile_desc, 0, sizeof(unsigned long long) * (size_t)B * sl.tile_stride, ctx->stream));
    sl.shade_epoch = 0;
    if (int rc = dev_alloc(ctx, pool, &sl.slow_list, BP)) return rc;
    if (int rc = dev_alloc(ctx, pool, &sl.slow_stack, (size_t)kStackDepth * kWave)) return rc;
    if (ctx->beam) {
      ctx->beam_tiles_x = (width + kBeamTile - 1u) / kBeamTile;    // (ctx->width is set when everything has been allocated)
      ctx->beam_tiles_y = (height + kBeamTile - 1u) / kBeamTile;
      if (int rc = dev_alloc(ctx, pool, &sl.beam_entries, (size_t)B * ctx->beam_tiles_x * ctx->beam_tiles_y * 2u * kBeamEntries)) return rc;
    }
    if (int rc = dev_alloc(ctx, pool, &sl.worklist, BP)) return rc;
    if (ctx->ray_sort) {
      if (int rc = dev_alloc(ctx, pool, &sl.octs, BP)) return rc;
      if (int rc = dev_alloc(ctx, pool, &sl.order, BP)) return rc;
    }
    if (int rc = dev_alloc(ctx, pool, &sl.counters, (size_t)B)) return rc;
    HIP_TRY(ctx, hipMemsetAsync(sl.counters, 0, sizeof(DeviceCounters) * (size_t)B, ctx->stream));
    sl.bi = DBatchInfo{};
    sl.bi.stride = (uint32_t)P;
    sl.bi.chunk_stride = (uint32_t)chunks;
    sl.bi.count = 1u;
    if (!ctx->staged) {
      sl.stage = ctx->fb;  // shade accumulates straight into the framebuffers
    } else {
      if (int rc = dev_alloc(ctx, pool, &sl.stage.color4, BP)) return rc;
      if (int rc = dev_alloc(ctx, pool, &sl.stage.nd4, BP)) return rc;
    }
  }
  if (int rc = dev_alloc(ctx, pool, &ctx->den_a, P)) return rc;
  if (int rc = dev_alloc(ctx, pool, &ctx->den_b, P)) return rc;
  if (int rc = dev_alloc(ctx, pool, &ctx->den_pos, P)) return rc;
  if (int rc = dev_alloc(ctx, pool, &ctx->pack_buf, P * 3u)) return rc;
  if (int rc = dev_alloc(ctx, pool, &ctx->rgba_buf, P)) return rc;
  HIP_TRY(ctx, hipMemsetAsync(ctx->fb.color4, 0, P * sizeof(float4), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(ctx->fb.nd4, 0, P * sizeof(float4), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(ctx->den_a, 0, P * sizeof(float4), ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(ctx->den_b, 0, P * sizeof(float4), ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->width = width;
  ctx->height = height;
  ctx->pix_begin = 0;
  ctx->band = DBand{0u, width, 0u, 1u, 0u};
  ctx->pix_count = (uint32_t)P;
  ctx->pix_capacity = (uint32_t)P;
  ctx->result = ctx->fb.color4;
  ctx->have_cam = false;
  ctx->last_slot = 0;
  return ptc_restart(ctx);
}

int ptc_set_rows(ptc_ctx* ctx, uint32_t row_begin, uint32_t row_end)
{
  if (!ctx || !ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "ptc_resize first");
  if (row_begin >= row_end || row_end > ctx->height) return fail(ctx, PTC_ERR_INVALID, "bad row range");
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = sync_frames(ctx)) return rc;
  ctx->pix_begin = row_begin * ctx->width;
  ctx->pix_count = (row_end - row_begin) * ctx->width;
  ctx->band = DBand{ctx->pix_begin, ctx->width, 0u, 1u, 0u};
  return ptc_restart(ctx);
}

int ptc_set_interleave(ptc_ctx* ctx, uint32_t rank, uint32_t nranks, uint32_t block_rows)
{
  if (!ctx || !ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "ptc_resize first");
  if (nranks == 0 || rank >= nranks || block_rows == 0) return fail(ctx, PTC_ERR_INVALID, "bad interleave");
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = sync_frames(ctx)) return rc;
  uint32_t rows = 0;
  const uint32_t blocks = (ctx->height + block_rows - 1u) / block_rows;
  for (uint32_t gb = rank; gb < blocks; gb += nranks) rows += std::min(block_rows, ctx->height - gb * block_rows);
  if (rows == 0) return fail(ctx, PTC_ERR_INVALID, "this rank gets no rows");
  ctx->pix_begin = 0;
  ctx->pix_count = rows * ctx->width;
  ctx->band = DBand{0u, ctx->width, rank, nranks, block_rows};
  return ptc_restart(ctx);
}

int ptc_restart(ptc_ctx* ctx)
{
  if (!ctx) return PTC_ERR_INVALID;
  // Iterations still queued are traced first, not dropped: the reference has rendered them by the time restart()
  // runs (path_trace is synchronous there), and a present between restart and the next path_trace shows them.
  // A viewer restarts after it has presented, i.e. with an empty queue, so this costs nothing where it matters.
  if (!ctx->pending.empty())
    if (int rc = flush_pending(ctx)) return rc;
  ctx->iteration = 0;
  return PTC_OK;
}

int ptc_iteration(const ptc_ctx* ctx) { return ctx ? ctx->iteration : PTC_ERR_INVALID; }

int ptc_set_iteration(ptc_ctx* ctx, int iteration)
{
  if (!ctx || iteration < 0) return PTC_ERR_INVALID;
  if (int rc = flush_pending(ctx)) return rc;
  ctx->iteration = iteration;
  return PTC_OK;
}

int ptc_set_max_iterations(ptc_ctx* ctx, int max_iterations)
{
  if (!ctx) return PTC_ERR_INVALID;
  ctx->max_iterations = max_iterations;
  return PTC_OK;
}

int ptc_set_method(ptc_ctx* ctx, int method)
{
  if (!ctx || (method != PTC_METHOD_MEGAKERNEL && method != PTC_METHOD_STREAMING)) return fail(ctx, PTC_ERR_INVALID, "unknown method");
  if (method == ctx->method) return PTC_OK;
  if (int rc = flush_pending(ctx)) return rc;
  ctx->method = method;
  return PTC_OK;
}

int ptc_set_max_bounces(ptc_ctx* ctx, int max_bounces)
{
  if (!ctx || max_bounces < 1 || max_bounces > (int)kMaxBounces) return fail(ctx, PTC_ERR_INVALID, "max_bounces must be in [1,64]");
  if (max_bounces == ctx->max_bounces) return PTC_OK;
  if (int rc = flush_pending(ctx)) return rc;
  ctx->max_bounces = max_bounces;
  return PTC_OK;
}

int ptc_set_trace_variant(ptc_ctx* ctx, int variant)
{
  const bool known = variant == 0 || variant == 1 || variant == 3;
  if (!ctx || !known) return fail(ctx, PTC_ERR_INVALID, "unknown trace variant (0, 1 or 3)");
  if (variant == ctx->trace_variant) return PTC_OK;
  if (int rc = flush_pending(ctx)) return rc;
  ctx->trace_variant = variant;
  return PTC_OK;
}

int ptc_set_param(ptc_ctx* ctx, const char* name, int value)
{
  if (!ctx || !name) return PTC_ERR_INVALID;
  if (int rc = flush_pending(ctx)) return rc;
  if (std::strcmp(name, "batch_frames") == 0) {
    if (value < 1 || value > kMaxBatch) return fail(ctx, PTC_ERR_INVALID, "batch_frames must be in [1,32]");
    if (ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "set batch_frames before ptc_resize");
    ctx->batch_frames = value;
    return PTC_OK;
  }
  if (std::strcmp(name, "traverse_waves") == 0) {
    if (value < 8 || value > 65536) return fail(ctx, PTC_ERR_INVALID, "traverse_waves out of range");
    if (ctx->has_scene) return fail(ctx, PTC_ERR_INVALID, "set traverse_waves before ptc_upload_scene");
    ctx->traverse_waves = (uint32_t)value;
    return PTC_OK;
  }
  if (std::strcmp(name, "debug_lds_entries") == 0) {
    if (value < 1 || value > kLds4) return fail(ctx, PTC_ERR_INVALID, "debug_lds_entries must be in [1," + std::to_string(kLds4) + "]");
    if (ctx->has_scene) return fail(ctx, PTC_ERR_INVALID, "set debug_lds_entries before ptc_upload_scene");
    ctx->lds_entries = (uint32_t)value;
    return PTC_OK;
  }
  if (std::strcmp(name, "debug_force_slow") == 0) {
    ctx->scene.force_slow = (uint32_t)value;  // 1: every ray at fetch time, 2: every winner at verification time
    ctx->force_slow = value;
    return PTC_OK;
  }
  if (std::strcmp(name, "layout_on_device") == 0) {
    if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "layout_on_device must be 0 or 1");
    ctx->layout_on_device = value != 0;
    return PTC_OK;
  }
  if (std::strcmp(name, "filter_rays") == 0) {
    if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "filter_rays must be 0 or 1");
    ctx->filter_rays = value != 0;
    return PTC_OK;
  }
  if (std::strcmp(name, "fused_shade") == 0) {
    if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "fused_shade must be 0 or 1");
    ctx->fused_shade = value != 0;
    return PTC_OK;
  }
  if (std::strcmp(name, "merge_instances") == 0) {
    if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "merge_instances must be 0 or 1");
    ctx->merge_instances = value != 0;
    return PTC_OK;
  }
  if (std::strcmp(name, "bvh_build_on_device") == 0) {
    if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "bvh_build_on_device must be 0 or 1");
    ctx->bvh_on_device = value != 0;
    return PTC_OK;
  }
  if (std::strcmp(name, "static_eighths") == 0) {
    if (value < 0 || value > 8) return fail(ctx, PTC_ERR_INVALID, "static_eighths must be in [0,8]");
    ctx->static_eighths = (uint32_t)value;
    ctx->scene.static_eighths = ctx->static_eighths;
    return PTC_OK;
  }
  if (std::strcmp(name, "small_waves") == 0) {
    if (value < 8 || value > 65536) return fail(ctx, PTC_ERR_INVALID, "small_waves out of range");
    ctx->small_waves = (uint32_t)value;
    return PTC_OK;
  }
  if (std::strcmp(name, "small_rays_per_lane") == 0) {
    if (value < 0 || value > 1024) return fail(ctx, PTC_ERR_INVALID, "small_rays_per_lane out of range");
    ctx->small_rays_per_lane = (uint32_t)value;
    return PTC_OK;
  }
  if (std::strcmp(name, "min_waves") == 0) {
    if (value < 8 || value > 65536) return fail(ctx, PTC_ERR_INVALID, "min_waves out of range");
    ctx->min_waves = (uint32_t)value;
    return PTC_OK;
  }
  if (std::strcmp(name, "beam") == 0) {
    if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "beam must be 0 or 1");
    if (ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "set beam before ptc_resize");
    ctx->beam = value != 0;
    return PTC_OK;
  }
  if (std::strcmp(name, "sphere_fold") == 0) {
    if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "sphere_fold must be 0 or 1");
    ctx->sphere_fold = value != 0;
    return PTC_OK;
  }
  if (std::strcmp(name, "sphere_lanes") == 0) {
    if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "sphere_lanes must be 0 or 1");
    ctx->sphere_lanes = value != 0;
    return PTC_OK;
  }
  if (std::strcmp(name, "split_idle") == 0) {
    if (value < 0 || value > 64) return fail(ctx, PTC_ERR_INVALID, "split_idle must be in [0,64]");
    ctx->split_idle = (uint32_t)value;
    ctx->scene.split_idle = ctx->split_idle;
    return PTC_OK;
  }
  if (std::strcmp(name, "refill_lanes") == 0) {
    if (value < 1 || value > 64) return fail(ctx, PTC_ERR_INVALID, "refill_lanes must be in [1,64]");
    ctx->refill_lanes = (uint32_t)value;
    ctx->scene.refill_lanes = ctx->refill_lanes;
    return PTC_OK;
  }
  if (std::strcmp(name, "ray_sort") == 0) {
    if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "ray_sort must be 0 or 1");
    if (ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "set ray_sort before ptc_resize");
    ctx->ray_sort = value;
    return PTC_OK;
  }
  if (std::strcmp(name, "denoise_variant") == 0) {
    if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "denoise_variant must be 0 or 1");
    ctx->denoise_variant = value;
    return PTC_OK;
  }
  if (std::strcmp(name, "slot_offset") == 0) {
    if (value < 0) return fail(ctx, PTC_ERR_INVALID, "slot_offset must not be negative");
    if (int rc = bind_device(ctx)) return rc;
    if (int rc = sync_frames(ctx)) return rc;  // frames in flight read the offset when they execute
    ctx->slot_offset = (uint32_t)value;
    HIP_TRY(ctx, hipMemcpy(ctx->slot_offset_dev, &ctx->slot_offset, sizeof(uint32_t), hipMemcpyHostToDevice));
    return PTC_OK;
  }
  if (std::strcmp(name, "frames_in_flight") == 0) {
    if (value < 1 || value > 256) return fail(ctx, PTC_ERR_INVALID, "frames_in_flight must be in [1,256]");
    if (ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "set frames_in_flight before ptc_resize");
    ctx->frames_in_flight = value;
    ctx->frames_auto = false;
    return PTC_OK;
  }
  return fail(ctx, PTC_ERR_INVALID, std::string("unknown parameter ") + name);
}

int ptc_set_denoiser_params(ptc_ctx* ctx, const ptc_denoiser_params* p)
{
  if (!ctx || !p) return PTC_ERR_INVALID;
  ctx->den = *p;
  return PTC_OK;
}

static int frame_ready(ptc_ctx* ctx)
{
  if (!ctx) return PTC_ERR_INVALID;
  if (!ctx->has_scene) return fail(ctx, PTC_ERR_NO_SCENE, "no scene uploaded");
  if (!ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "ptc_resize first");
  return bind_device(ctx);
}

namespace {

// Persistent wavefronts of a traversal launch.  A launch ends with its longest ray (about 100 us however few rays it
// carries), so a small launch wants about one ray per lane -- rays / 64 wavefronts, at least 1024, at most 3072 -- and
// only a launch with eight or more rays per lane fills every wavefront slot of the chip (traverse_waves: what is
// resident at five per SIMD).  Measured on single 1080p frames (2 M rays at the first bounce, 65 k at the eighth): one size
// for all bounces 2.60 ms per frame, sized per bounce 2.1 ms.  The ray count of a bounce lives on the device; the
// host sizes with the counts of a recent frame (FrameSlot::live_host), a bounce it knows nothing about with its cap.
// the epoch of the next look-back launch on a slot's tile descriptors (k_shade_fused, the listing k_raygen / k_spheres)
uint32_t next_epoch(ptc_ctx::FrameSlot& sl)
{
  sl.shade_epoch = sl.shade_epoch >= 0x3fffffffu ? 1u : sl.shade_epoch + 1u;
  return sl.shade_epoch;
}

uint32_t traverse_waves_for(ptc_ctx* ctx, uint32_t frames, int bounce, bool listed)
{
  for (auto& sl : ctx->slots)
    if (sl.live_pending && sl.done && hipEventQuery(sl.done) == hipSuccess) {
      std::memcpy(ctx->est_live, sl.live_host, sizeof ctx->est_live);
      ctx->est_valid = true;
      sl.live_pending = false;
    }
  (void)hipGetLastError();  // hipEventQuery's "not ready" is no error
  uint64_t per_frame = ctx->pix_count;
  if (ctx->est_valid && ctx->est_live[0] == ctx->pix_count) per_frame = std::min<uint64_t>(ctx->pix_count, ctx->est_live[bounce]);
  // a launch that walks a work list carries the listed rays only (a single frame's primary rays: 0.85 of 2.07 M on the
  // benchmark scene -- 2.6 rays per lane of a full launch, and the smaller launch ends sooner)
  if (listed && ctx->est_valid && ctx->est_live[0] == ctx->pix_count && ctx->est_live[kMaxBounces + 1 + bounce] != 0u)
    per_frame = std::min<uint64_t>(per_frame, (uint64_t)ctx->est_live[kMaxBounces + 1 + bounce] * 9u / 8u + 64u);
  const uint64_t rays = per_frame * frames;
  // (fewer than four rays per lane at full size: 3072 wavefronts do as well or a little better -- single frames)
  const uint64_t cap = rays >= (uint64_t)ctx->traverse_waves * kWave * ctx->small_rays_per_lane ? ctx->traverse_waves
                                                                                                 : std::min<uint32_t>(ctx->traverse_waves, ctx->small_waves);
  const uint64_t want = ((rays + kWave - 1u) / kWave + 7u) & ~7ull;
  return (uint32_t)std::min<uint64_t>(cap, std::max<uint64_t>(std::min<uint32_t>(ctx->min_waves, ctx->traverse_waves), want));
}

// Enqueue raygen for `count` consecutive iterations on the next slot (round robin) and make it the active batch.
// launches [k, k + run) of the plan are one traversal launch: consecutive objects that instantiate the same mesh, with
// nothing between them (k_traverse4m; "merge_instances")
size_t launch_run(const ptc_ctx* ctx, size_t k)
{
  size_t run = 1;
  const auto& l = ctx->launches[k];
  if (ctx->trace_variant == 3 && ctx->merge_instances)
    while (k + run < ctx->launches.size() && ctx->launches[k + run].pre_begin == ctx->launches[k + run].pre_end &&
           ctx->launches[k + run].mesh == l.mesh + (uint32_t)run &&
           ctx->object_mesh[ctx->launches[k + run].mesh] == ctx->object_mesh[l.mesh])
      ++run;
  return run;
}

int batch_begin(ptc_ctx* ctx, const ptc_ctx::Pending* items, int count)
{
  const int single_slots = (int)ctx->slots.size() - ctx->big_slots;
  const int f = (count == 1 && single_slots > 0) ? ctx->big_slots + (int)(ctx->singles_issued++ % (uint64_t)single_slots)
                                                 : (int)(ctx->batches_issued++ % (uint64_t)ctx->big_slots);
  auto& sl = ctx->slots[(size_t)f];
  // the slot's previous batch has been enqueued on the same stream, so its buffers are free in stream order.
  // A main-stream consumer that still reads the framebuffers (denoise) must finish before anything is folded
  // in: with staging that is only the accumulate at the end of the batch (so tracing overlaps the denoise of
  // the previous frame); without staging the shade kernels write the framebuffers directly.
  if (ctx->main_valid && !ctx->staging()) HIP_TRY(ctx, hipStreamWaitEvent(sl.stream, ctx->main_event, 0));
  // launches of different slots run at the same time: each slot has its own stack overflow area
  const size_t spill_need = (size_t)ctx->scene.spill_cap * ctx->scene.spill_stride;
  if (spill_need > sl.spill_elems) {
    HIP_TRY(ctx, hipStreamSynchronize(sl.stream));
    if (sl.spill) HIP_TRY(ctx, hipFree(sl.spill));
    sl.spill = nullptr;
    sl.spill_elems = 0;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&sl.spill), spill_need * sizeof(uint2)));
    sl.spill_elems = spill_need;
  }
  sl.cur = 0;
  sl.work_slot = 0;
  sl.bounces_done = 0;
  sl.bi.count = (uint32_t)count;
  DCameras cams{};
  for (int k = 0; k < count; ++k) {
    cams.c[k] = items[k].cam;
    sl.bi.iteration[k] = items[k].iteration;
  }
  // "filter_rays" at bounce 0: when the bounce opens with a traversal launch (no sphere run in front of the first mesh),
  // raygen lists the rays that may hit that launch's world boxes and writes the others' miss records itself
  sl.first_listed = ctx->filter_rays && ctx->trace_variant == 3 && !ctx->launches.empty() &&
                    ctx->launches[0].pre_begin == ctx->launches[0].pre_end;
  const uint32_t first_mesh = sl.first_listed ? ctx->launches[0].mesh : 0u;
  uint32_t filt_end = sl.first_listed ? first_mesh + (uint32_t)launch_run(ctx, 0) : 0u;
  // ... and when that launch is the scene's whole mesh part -- only the sphere run that ends the object list, if any,
  // follows it -- the filter also takes the world boxes of those spheres (the reference tests a sphere's box before the
  // sphere, path_tracer.cu:84): a ray it does not list then hits nothing at all, raygen ends its path, and bounce 0's
  // k_shade_fused walks the list.  (The few rays listed for a sphere's box alone leave the traversal launch at its root.)
  const bool tail = ctx->tail_begin < ctx->tail_end;
  sl.primary_finished = sl.first_listed && ctx->fused_shade && launch_run(ctx, 0) == ctx->launches.size() &&
                        (!tail || ctx->tail_begin == filt_end);
  if (sl.primary_finished && tail) filt_end = ctx->tail_end;
  launch_raygen(sl.stream, cams, sl.bi, ctx->band, ctx->pix_count, sl.paths[0], sl.counters, ctx->scene.objects, first_mesh, filt_end,
                sl.first_listed ? sl.worklist : nullptr, sl.hits, sl.tile_desc, sl.tile_stride, next_epoch(sl), sl.primary_finished,
                sl.stage, ctx->staging());
  if (int rc = check_last(ctx, "raygen")) return rc;
  // "beam": when bounce 0 opens with a launch over ONE mesh object (k_traverse4), its primary rays start at entry points
  // computed per tile and distinct camera of the batch
  sl.beam = DBeam{};
  if (ctx->beam && sl.beam_entries && ctx->trace_variant == 3 && !ctx->launches.empty() && ctx->launches[0].pre_begin == ctx->launches[0].pre_end &&
      launch_run(ctx, 0) == 1 && ctx->width >= 2u && ctx->height >= 2u) {
    uint8_t cam_of_beam[kMaxBatch];
    uint32_t nbeam = 0;
    for (int k = 0; k < count; ++k) {
      uint32_t b = 0;
      while (b < nbeam && std::memcmp(&cams.c[cam_of_beam[b]], &cams.c[k], sizeof(DCamera)) != 0) ++b;
      if (b == nbeam) cam_of_beam[nbeam++] = (uint8_t)k;
      sl.beam.beam_of[k] = (uint8_t)b;
    }
    DScene scene = ctx->scene;
    const uint32_t mesh_obj = ctx->launches[0].mesh;
    scene.cur = ctx->mesh_views[ctx->object_mesh[mesh_obj]];
    bool cached = sl.beam_count == nbeam && sl.beam_scene == ctx->scene_serial && sl.beam_obj == mesh_obj;
    for (uint32_t b = 0; b < nbeam && cached; ++b) cached = std::memcmp(&sl.beam_cams.c[b], &cams.c[cam_of_beam[b]], sizeof(DCamera)) == 0;
    if (!cached) {
      launch_beam(sl.stream, scene, mesh_obj, cams, cam_of_beam, nbeam, ctx->beam_tiles_x, ctx->beam_tiles_y, ctx->mesh_nodes4[ctx->object_mesh[mesh_obj]],
                  sl.beam_entries);
      if (int rc = check_last(ctx, "beam")) return rc;
      for (uint32_t b = 0; b < nbeam; ++b) sl.beam_cams.c[b] = cams.c[cam_of_beam[b]];
      sl.beam_count = nbeam;
      sl.beam_scene = ctx->scene_serial;
      sl.beam_obj = mesh_obj;
    }
    sl.beam.entries = sl.beam_entries;
    sl.beam.tiles_x = ctx->beam_tiles_x;
    sl.beam.tiles = ctx->beam_tiles_x * ctx->beam_tiles_y;
    sl.beam.width = ctx->width;
    sl.beam.band = ctx->band;
  }
  ctx->active_slot = f;
  return PTC_OK;
}

// may the sphere run [begin, end) take the per-lane path (sphere_run_lanes)?
static uint32_t lanes_run_of(const ptc_ctx* ctx, uint32_t begin, uint32_t end)
{
  if (!ctx->sphere_lanes || end <= begin || end - begin > 8u || end > ctx->sphere_class.size()) return 0u;
  const uint32_t k = ctx->sphere_class[begin];
  if (k == 0u) return 0u;
  for (uint32_t i = begin; i < end; ++i)
    if (ctx->sphere_class[i] != k) return 0u;
  return 1u;
}

// may k_spheres take sphere_fold for the run [begin, end)?  Every object a "simple" sphere (sphere_ball_of)
static uint32_t fold_run_of(const ptc_ctx* ctx, uint32_t begin, uint32_t end)
{
  if (!ctx->sphere_fold || end <= begin || end > ctx->sphere_class.size()) return 0u;
  for (uint32_t i = begin; i < end; ++i)
    if (ctx->sphere_class[i] == 0u) return 0u;
  return 1u;
}

int batch_bounce(ptc_ctx* ctx, int bounce, const uint32_t* slot_base_dev)
{
  auto& sl = ctx->slots[(size_t)ctx->active_slot];
  const bool last = bounce == ctx->max_bounces - 1;
  DPaths in = sl.paths[sl.cur], out = sl.paths[sl.cur ^ 1];
  DScene scene = ctx->scene;
  scene.spill = sl.spill;
  scene.slow_stack = sl.slow_stack;
  // HIP events around each launch of the dominant (closest-hit) kernel, on the stream it runs on
  auto timed_begin = [&](ptc_ctx::TimedLaunch& tl) -> int {
    if (!ctx->time_trace) return PTC_OK;
    for (hipEvent_t* e : {&tl.start, &tl.stop}) {
      if (!ctx->free_events.empty()) {
        *e = ctx->free_events.back();
        ctx->free_events.pop_back();
      } else {
        HIP_TRY(ctx, hipEventCreate(e));
      }
    }
    HIP_TRY(ctx, hipEventRecord(tl.start, sl.stream));
    return PTC_OK;
  };
  auto timed_end = [&](ptc_ctx::TimedLaunch& tl) -> int {
    if (!ctx->time_trace) return PTC_OK;
    HIP_TRY(ctx, hipEventRecord(tl.stop, sl.stream));
    ctx->timed.push_back(tl);
    return PTC_OK;
  };
  bool wrote = false;  // some launch of this bounce has written the hit records
  // ray sorting: the shade kernel of the previous bounce has tagged its surviving rays with their direction octant
  const bool persistent = ctx->trace_variant == 3;
  const bool sorted = ctx->ray_sort && ctx->trace_variant == 3 && bounce >= 1 && sl.order && !ctx->launches.empty();
  if (sorted) launch_sort_octant(sl.stream, sl.octs, sl.order, ctx->pix_count, bounce, sl.counters, sl.bi);
  if (persistent) {
    // closest hit = the object list walked by the launches of TraceLaunch
    for (size_t k = 0; k < ctx->launches.size(); ++k) {
      const auto& l = ctx->launches[k];
      const size_t run = launch_run(ctx, k);
      // a sphere run in front of the launch reads every ray anyway: it also lists the rays that may hit one of the launch's
      // objects at all ("filter_rays"), and the launch fetches through that list
      const bool by_spheres = ctx->filter_rays && l.pre_begin < l.pre_end && !sorted && ctx->trace_variant == 3;
      const bool listed = by_spheres || (bounce == 0 && k == 0 && sl.first_listed);  // (bounce 0's first launch: listed by k_raygen)
      if (l.pre_begin < l.pre_end) {
        scene.lanes_run = lanes_run_of(ctx, l.pre_begin, l.pre_end);
        scene.fold_run = fold_run_of(ctx, l.pre_begin, l.pre_end);
        launch_spheres(sl.stream, scene, l.pre_begin, l.pre_end, !wrote, in, sl.hits, ctx->pix_count, bounce, sl.counters, sl.bi,
                       by_spheres ? l.mesh : 0u, by_spheres ? l.mesh + (uint32_t)run : 0u, by_spheres ? sl.worklist : nullptr,
                       sl.tile_desc, sl.tile_stride, by_spheres ? next_epoch(sl) : 0u);
        wrote = true;
      }
      ptc_ctx::TimedLaunch tl{nullptr, nullptr, bounce};
      if (int rc = timed_begin(tl)) return rc;
      const uint32_t waves = traverse_waves_for(ctx, sl.bi.count, bounce, listed);
      scene.cur = ctx->mesh_views[ctx->object_mesh[l.mesh]];  // this object's mesh
      const uint32_t* pick = listed ? sl.worklist : (sorted ? sl.order : nullptr);
      if (run > 1) {
        launch_traverse_run(sl.stream, scene, l.mesh, l.mesh + (uint32_t)run, !wrote, in, sl.hits, bounce, sl.work_slot++ % kWorkSlots, sl.counters,
                            ctx->count_tests, waves, sl.slow_list, pick, sl.bi, listed);
        k += run - 1;
      } else {
        const int kernel = ctx->trace_variant;
        scene.beam = (bounce == 0 && k == 0) ? sl.beam : DBeam{};
        launch_traverse(sl.stream, scene, l.mesh, !wrote, in, sl.hits, bounce, sl.work_slot++ % kWorkSlots, sl.counters, ctx->count_tests, waves,
                        sl.slow_list, pick, kernel, sl.bi, listed);
      }
      wrote = true;
      if (int rc = timed_end(tl)) return rc;
    }
  } else {
    ptc_ctx::TimedLaunch tl{nullptr, nullptr, bounce};
    if (int rc = timed_begin(tl)) return rc;
    launch_trace(sl.stream, scene, in, sl.hits, ctx->pix_count, bounce, sl.counters, ctx->count_tests, ctx->trace_variant);
    wrote = true;
    if (int rc = timed_end(tl)) return rc;
  }
  // the sphere run that ends the object list (variant 3 only) + the live counts; their scan
  const bool tail = persistent && ctx->tail_begin < ctx->tail_end;
  uint8_t* octs = ctx->ray_sort && !last ? sl.octs : nullptr;
  if (ctx->fused_shade) {
    // one pass: trailing spheres + material + stable compaction (decoupled look-back) + final gather
    next_epoch(sl);
    scene.lanes_run = tail ? lanes_run_of(ctx, ctx->tail_begin, ctx->tail_end) : 0u;
    scene.fold_run = tail && !scene.lanes_run ? fold_run_of(ctx, ctx->tail_begin, ctx->tail_end) : 0u;
    launch_shade_fused(sl.stream, scene, tail ? ctx->tail_begin : 0u, tail ? ctx->tail_end : 0u, !wrote, in, out, sl.hits, ctx->pix_count,
                       ctx->staging(), bounce, last, slot_base_dev, sl.tile_desc, sl.tile_stride, sl.shade_epoch, sl.stage, ctx->band,
                       sl.counters, octs, sl.bi, bounce == 0 && sl.primary_finished ? sl.worklist : nullptr);
  } else {
    launch_tail_count(sl.stream, scene, tail ? ctx->tail_begin : 0u, tail ? ctx->tail_end : 0u, !wrote, in, sl.hits, ctx->pix_count,
                      bounce, sl.chunk_counts, sl.counters, sl.bi);
    launch_scan(sl.stream, bounce, last, sl.chunk_counts, sl.chunk_offsets, sl.counters, sl.bi);
    launch_shade(sl.stream, scene, in, out, sl.hits, ctx->pix_count, ctx->staging(), bounce, last, slot_base_dev,
                 sl.chunk_offsets, sl.stage, ctx->band, sl.counters, octs, sl.bi);
  }
  sl.cur ^= 1;
  sl.bounces_done = bounce + 1;
  return check_last(ctx, "bounce");
}

int batch_end(ptc_ctx* ctx)
{
  auto& sl = ctx->slots[(size_t)ctx->active_slot];
  if (ctx->staging()) {
    // fold these samples in after the previous iteration's fold (running means do not commute)
    if (ctx->order_valid) HIP_TRY(ctx, hipStreamWaitEvent(sl.stream, ctx->order_event, 0));
    if (ctx->main_valid) HIP_TRY(ctx, hipStreamWaitEvent(sl.stream, ctx->main_event, 0));
    launch_accumulate(sl.stream, sl.stage, ctx->fb, ctx->pix_count, sl.bi);
    if (int rc = check_last(ctx, "accumulate")) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->order_event, sl.stream));
    ctx->order_valid = true;
  }
  // the live counts of this batch's first frame, for the sizing of later launches (nobody waits for the copy)
  if (sl.live_host && sl.bounces_done == ctx->max_bounces) {
    HIP_TRY(ctx, hipMemcpyAsync(sl.live_host, &sl.counters[0].live[0], sizeof(uint32_t) * 2 * (kMaxBounces + 1), hipMemcpyDeviceToHost, sl.stream));
    sl.live_pending = true;
  }
  HIP_TRY(ctx, hipEventRecord(sl.done, sl.stream));
  ctx->last_slot = ctx->active_slot;
  ctx->active_slot = -1;
  return PTC_OK;
}

// frames per batch ptc_trace may use right now (only the default traversal kernel reads DBatchInfo)
int batch_limit(const ptc_ctx* ctx) { return ctx->staged && ctx->trace_variant == 3 ? ctx->batch : 1; }

// enqueue the iterations ptc_trace has queued
int flush_pending(ptc_ctx* ctx)
{
  if (ctx->pending.empty()) return PTC_OK;
  if (int rc = bind_device(ctx)) return rc;
  std::vector<ptc_ctx::Pending> items;
  items.swap(ctx->pending);
  if (int rc = batch_begin(ctx, items.data(), (int)items.size())) return rc;
  for (int b = 0; b < ctx->max_bounces; ++b)
    if (int rc = batch_bounce(ctx, b, ctx->slot_offset ? ctx->slot_offset_dev : nullptr)) {
      ctx->active_slot = -1;
      return rc;
    }
  return batch_end(ctx);
}

}  // namespace

int ptc_trace_begin(ptc_ctx* ctx, const ptc_camera* camera)
{
  if (int rc = frame_ready(ctx)) return rc;
  if (!camera) return fail(ctx, PTC_ERR_INVALID, "camera is NULL");
  if (ctx->active_slot >= 0) return fail(ctx, PTC_ERR_INVALID, "ptc_trace_end missing");
  if (int rc = flush_pending(ctx)) return rc;
  ctx->cam = make_camera(*camera, ctx->width, ctx->height);
  ctx->have_cam = true;
  const ptc_ctx::Pending one{ctx->cam, (uint32_t)ctx->iteration};
  return batch_begin(ctx, &one, 1);
}

int ptc_trace_bounce(ptc_ctx* ctx, int bounce, const uint32_t* slot_base_dev)
{
  if (int rc = frame_ready(ctx)) return rc;
  if (ctx->active_slot < 0) return fail(ctx, PTC_ERR_INVALID, "ptc_trace_begin missing");
  if (bounce < 0 || bounce >= ctx->max_bounces) return fail(ctx, PTC_ERR_INVALID, "bounce out of range");
  if (slot_base_dev) {
    // the slot base was produced by work on the context's stream (the caller's collective): the frame's stream waits
    auto& sl = ctx->slots[(size_t)ctx->active_slot];
    if (sl.stream != ctx->stream) {
      HIP_TRY(ctx, hipEventRecord(ctx->xstream_event, ctx->stream));
      HIP_TRY(ctx, hipStreamWaitEvent(sl.stream, ctx->xstream_event, 0));
    }
  }
  return batch_bounce(ctx, bounce, slot_base_dev);
}

int ptc_trace_end(ptc_ctx* ctx)
{
  if (!ctx || ctx->active_slot < 0) return fail(ctx, PTC_ERR_INVALID, "ptc_trace_begin missing");
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = batch_end(ctx)) return rc;
  ++ctx->iteration;
  ++ctx->frames;
  ctx->result = ctx->fb.color4;  // path_tracer.cu:476
  return PTC_OK;
}

int ptc_live_count_dev(ptc_ctx* ctx, int bounce, const uint32_t** dev_ptr)
{
  if (!ctx || !dev_ptr || bounce < 0 || bounce > (int)kMaxBounces || ctx->slots.empty()) return PTC_ERR_INVALID;
  if (int rc = flush_pending(ctx)) return rc;
  const auto& sl = ctx->slots[(size_t)(ctx->active_slot >= 0 ? ctx->active_slot : ctx->last_slot)];
  *dev_ptr = &sl.counters[sl.bi.count - 1u].live[bounce];  // the most recent iteration of the batch
  return PTC_OK;
}

int ptc_copy_live_count(ptc_ctx* ctx, int bounce, void* dst_dev)
{
  if (!ctx || !dst_dev || bounce < 0 || bounce > (int)kMaxBounces) return PTC_ERR_INVALID;
  if (ctx->active_slot < 0) return fail(ctx, PTC_ERR_INVALID, "ptc_trace_begin missing");
  if (int rc = bind_device(ctx)) return rc;
  auto& sl = ctx->slots[(size_t)ctx->active_slot];
  HIP_TRY(ctx, hipMemcpyAsync(dst_dev, &sl.counters->live[bounce], sizeof(uint32_t), hipMemcpyDeviceToDevice, sl.stream));
  if (sl.stream != ctx->stream) {  // what the caller enqueues on the context's stream next (an all-gather) sees the value
    HIP_TRY(ctx, hipEventRecord(ctx->xstream_event, sl.stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->xstream_event, 0));
  }
  return PTC_OK;
}

int ptc_read_live_count(ptc_ctx* ctx, int bounce, uint32_t* host_out)
{
  if (!ctx || !host_out || bounce < 0 || bounce > (int)kMaxBounces) return PTC_ERR_INVALID;
  if (ctx->active_slot < 0) return fail(ctx, PTC_ERR_INVALID, "ptc_trace_begin missing");
  if (int rc = bind_device(ctx)) return rc;
  auto& sl = ctx->slots[(size_t)ctx->active_slot];
  HIP_TRY(ctx, hipMemcpyAsync(host_out, &sl.counters->live[bounce], sizeof(uint32_t), hipMemcpyDeviceToHost, sl.stream));
  HIP_TRY(ctx, hipStreamSynchronize(sl.stream));
  return PTC_OK;
}

int ptc_trace(ptc_ctx* ctx, const ptc_camera* camera)
{
  if (int rc = frame_ready(ctx)) return rc;
  if (!camera) return fail(ctx, PTC_ERR_INVALID, "camera is NULL");
  if (ctx->active_slot >= 0) return fail(ctx, PTC_ERR_INVALID, "ptc_trace_end missing");
  if (ctx->iteration >= ctx->max_iterations) {  // path_tracer.cu:391
    ctx->result = ctx->fb.color4;
    return PTC_OK;
  }
  if (ctx->method == PTC_METHOD_MEGAKERNEL) {
    // one kernel per sample, accumulating in place: frames are serialised on slot 0's stream
    if (ctx->staging()) {
      if (int rc = sync_frames(ctx)) return rc;
    } else if (int rc = flush_pending(ctx)) {
      return rc;
    }
    auto& sl = ctx->slots[0];
    ctx->cam = make_camera(*camera, ctx->width, ctx->height);
    ctx->have_cam = true;
    launch_megakernel(sl.stream, ctx->scene, ctx->cam, (uint32_t)ctx->iteration, ctx->band, ctx->pix_count,
                      ctx->max_bounces, ctx->fb, sl.counters);
    if (int rc = check_last(ctx, "megakernel")) return rc;
    HIP_TRY(ctx, hipEventRecord(sl.done, sl.stream));
    if (ctx->staging()) {
      HIP_TRY(ctx, hipEventRecord(ctx->order_event, sl.stream));
      ctx->order_valid = true;
    }
    ctx->last_slot = 0;
    sl.bi.count = 1u;
    ++ctx->iteration;
    ++ctx->frames;
    ctx->result = ctx->fb.color4;
    return PTC_OK;
  }
  // streaming mode: queue the iteration; a full batch goes to the GPU
  ctx->cam = make_camera(*camera, ctx->width, ctx->height);
  ctx->have_cam = true;
  ctx->pending.push_back(ptc_ctx::Pending{ctx->cam, (uint32_t)ctx->iteration});
  ++ctx->iteration;
  ++ctx->frames;
  ctx->result = ctx->fb.color4;  // path_tracer.cu:476
  if ((int)ctx->pending.size() >= batch_limit(ctx)) return flush_pending(ctx);
  return PTC_OK;
}

int ptc_denoise(ptc_ctx* ctx)
{
  if (int rc = frame_ready(ctx)) return rc;
  if (!ctx->have_cam) return fail(ctx, PTC_ERR_INVALID, "denoise needs a traced frame (it reuses the last camera)");
  if (ctx->pix_count != ctx->width * ctx->height) return fail(ctx, PTC_ERR_INVALID, "denoise needs the full frame in one context");
  if (int rc = flush_pending(ctx)) return rc;
  // every sample must be folded in before the framebuffers are read (stream order, no host sync)
  if (ctx->order_valid) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->order_event, 0));
  else if (!ctx->slots.empty() && ctx->slots[(size_t)ctx->last_slot].stream != ctx->stream)
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->slots[(size_t)ctx->last_slot].done, 0));
  const DDenoise prm{ctx->den.color_weight, ctx->den.normal_weight, ctx->den.position_weight, ctx->denoise_variant};
  // edge_avoiding_a_trous_denoiser.cu:102-108: (color, back, front) <- (back, front, back) after each pass
  const float4* color = ctx->fb.color4;
  float4* back = ctx->den_a;
  float4* front = ctx->den_b;
  if (ctx->den.filter_size >= 1) launch_denoise_positions(ctx->stream, ctx->cam, ctx->pix_count, ctx->fb.nd4, ctx->den_pos);
  for (int step = 1; step <= ctx->den.filter_size; step *= 2) {
    ptc_ctx::TimedLaunch tl{nullptr, nullptr, -1};
    if (ctx->time_trace) {
      for (hipEvent_t* e : {&tl.start, &tl.stop}) {
        if (!ctx->free_events.empty()) {
          *e = ctx->free_events.back();
          ctx->free_events.pop_back();
        } else {
          HIP_TRY(ctx, hipEventCreate(e));
        }
      }
      HIP_TRY(ctx, hipEventRecord(tl.start, ctx->stream));
    }
    launch_denoise_pass(ctx->stream, ctx->cam, ctx->pix_count, color, ctx->fb.nd4, ctx->den_pos, back, step, prm);
    if (ctx->time_trace) {
      HIP_TRY(ctx, hipEventRecord(tl.stop, ctx->stream));
      ctx->timed.push_back(tl);
    }
    const float4* new_color = back;
    float4* new_back = front;
    float4* new_front = back;
    color = new_color;
    back = new_back;
    front = new_front;
  }
  ctx->result = front;
  if (int rc = check_last(ctx, "denoise")) return rc;
  HIP_TRY(ctx, hipEventRecord(ctx->main_event, ctx->stream));
  ctx->main_valid = true;
  return PTC_OK;
}

int ptc_present_rgba8(ptc_ctx* ctx, void* dst, int dst_is_device, int display_type)
{
  if (!ctx || !dst) return PTC_ERR_INVALID;
  if (!ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "ptc_resize first");
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = flush_pending(ctx)) return rc;
  const float4* src = nullptr;
  int mode = 0;
  switch (display_type) {
  case PTC_DISPLAY_FINAL: src = ctx->result; break;
  case PTC_DISPLAY_COLOR: src = ctx->fb.color4; break;
  case PTC_DISPLAY_NORMAL: src = ctx->fb.nd4; mode = 1; break;
  case PTC_DISPLAY_DEPTH: src = ctx->fb.nd4; mode = 2; break;
  default: return fail(ctx, PTC_ERR_INVALID, "unknown display type");
  }
  if (int rc = sync_frames(ctx)) return rc;
  uint32_t* out = dst_is_device ? static_cast<uint32_t*>(dst) : ctx->rgba_buf;
  launch_preview(ctx->stream, src, ctx->pix_count, mode, out);
  if (int rc = check_last(ctx, "preview")) return rc;
  if (!dst_is_device)
    HIP_TRY(ctx, hipMemcpyAsync(dst, ctx->rgba_buf, (size_t)ctx->pix_count * 4u, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // path_tracer.cu:519
  return PTC_OK;
}

int ptc_download(ptc_ctx* ctx, int which, void* dst, int dst_is_device)
{
  if (!ctx || !dst) return PTC_ERR_INVALID;
  if (!ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "ptc_resize first");
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = flush_pending(ctx)) return rc;
  const float4* src = nullptr;
  int sel = 0;
  size_t floats = (size_t)ctx->pix_count * 3u;
  switch (which) {
  case PTC_BUF_COLOR: src = ctx->fb.color4; break;
  case PTC_BUF_NORMAL: src = ctx->fb.nd4; break;
  case PTC_BUF_DEPTH: src = ctx->fb.nd4; sel = 1; floats = ctx->pix_count; break;
  case PTC_BUF_FINAL: src = ctx->result; break;
  default: return fail(ctx, PTC_ERR_INVALID, "unknown buffer");
  }
  if (int rc = sync_frames(ctx)) return rc;
  float* out = dst_is_device ? static_cast<float*>(dst) : ctx->pack_buf;
  launch_pack(ctx->stream, src, ctx->pix_count, sel, out);
  if (int rc = check_last(ctx, "pack")) return rc;
  if (!dst_is_device) HIP_TRY(ctx, hipMemcpyAsync(dst, ctx->pack_buf, floats * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PTC_OK;
}

// ---- several GPUs: bands over HIP inter-process memory (include/ptcore.h) -------------------------------------------
static int band_pack(ptc_ctx* ctx, int which, float* dst, size_t* floats)
{
  const float4* src = nullptr;
  int sel = 0;
  *floats = (size_t)ctx->pix_count * 3u;
  switch (which) {
  case PTC_BUF_COLOR: src = ctx->fb.color4; break;
  case PTC_BUF_NORMAL: src = ctx->fb.nd4; break;
  case PTC_BUF_DEPTH: src = ctx->fb.nd4; sel = 1; *floats = ctx->pix_count; break;
  case PTC_BUF_FINAL: src = ctx->result; break;
  default: return fail(ctx, PTC_ERR_INVALID, "unknown buffer");
  }
  if (int rc = sync_frames(ctx)) return rc;
  launch_pack(ctx->stream, src, ctx->pix_count, sel, dst);
  return check_last(ctx, "pack");
}

int ptc_band_export(ptc_ctx* ctx, ptc_band_handle* out)
{
  if (!ctx || !out) return PTC_ERR_INVALID;
  if (!ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "ptc_resize first");
  if (int rc = bind_device(ctx)) return rc;
  if (!ctx->band_buf) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->band_buf), (size_t)ctx->pix_capacity * 3u * sizeof(float)));
  std::memset(out, 0, sizeof *out);
  hipIpcMemHandle_t h;
  static_assert(sizeof h <= sizeof out->ipc_mem, "ipc handle size");
  HIP_TRY(ctx, hipIpcGetMemHandle(&h, ctx->band_buf));
  std::memcpy(out->ipc_mem, &h, sizeof h);
  out->pix_count = ctx->pix_count;
  out->pix_begin = ctx->band.pix_begin;
  out->width = ctx->width;
  out->rank = ctx->band.rank;
  out->nranks = ctx->band.nranks;
  out->block_rows = ctx->band.block_rows;
  return PTC_OK;
}

int ptc_band_import(ptc_ctx* root, uint32_t rank, const ptc_band_handle* handle)
{
  if (!root || !handle || rank > 0xffffu) return PTC_ERR_INVALID;
  if (!root->pix_capacity) return fail(root, PTC_ERR_INVALID, "ptc_resize first");
  if (handle->width != root->width) return fail(root, PTC_ERR_INVALID, "band of another frame width");
  // The handle arrives from another process: its geometry decides where k_scatter_band writes, so it must describe a
  // band of THIS frame (a peer that resized or re-partitioned after exporting sends a stale one).
  {
    const uint64_t P = (uint64_t)root->width * root->height;
    if (handle->pix_count == 0u || (uint64_t)handle->pix_count > P) return fail(root, PTC_ERR_INVALID, "band larger than the frame");
    if (handle->nranks <= 1u) {
      if ((uint64_t)handle->pix_begin + handle->pix_count > P) return fail(root, PTC_ERR_INVALID, "band reaches beyond the frame");
    } else {
      if (handle->block_rows == 0u || handle->rank >= handle->nranks) return fail(root, PTC_ERR_INVALID, "bad interleave in the band handle");
      uint64_t rows = 0;  // as ptc_set_interleave counts them
      const uint32_t blocks = (root->height + handle->block_rows - 1u) / handle->block_rows;
      for (uint32_t gb = handle->rank; gb < blocks; gb += handle->nranks)
        rows += std::min(handle->block_rows, root->height - gb * handle->block_rows);
      if (rows * root->width != handle->pix_count) return fail(root, PTC_ERR_INVALID, "band handle does not match this frame's interleave");
    }
  }
  if (int rc = bind_device(root)) return rc;
  if (root->peers.size() <= rank) root->peers.resize((size_t)rank + 1u);
  auto& peer = root->peers[rank];
  if (peer.opened && peer.mapped) (void)hipIpcCloseMemHandle(peer.mapped);
  peer = ptc_ctx::Peer{};
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle->ipc_mem, sizeof h);
  HIP_TRY(root, hipIpcOpenMemHandle(&peer.mapped, h, hipIpcMemLazyEnablePeerAccess));
  peer.opened = true;
  peer.h = *handle;
  return PTC_OK;
}

int ptc_band_publish(ptc_ctx* ctx, int which)
{
  if (!ctx) return PTC_ERR_INVALID;
  if (!ctx->band_buf) return fail(ctx, PTC_ERR_INVALID, "ptc_band_export first");
  if (int rc = bind_device(ctx)) return rc;
  size_t floats = 0;
  if (int rc = band_pack(ctx, which, ctx->band_buf, &floats)) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // the rows are in the exported buffer when this returns
  return PTC_OK;
}

static int gather_rows(ptc_ctx* root, int which, int channels)
{
  const size_t P = (size_t)root->width * root->height;
  if (!root->gather_frame) HIP_TRY(root, hipMalloc(reinterpret_cast<void**>(&root->gather_frame), P * 3u * sizeof(float)));
  if (!root->band_buf) HIP_TRY(root, hipMalloc(reinterpret_cast<void**>(&root->band_buf), (size_t)root->pix_capacity * 3u * sizeof(float)));
  if (!root->gather_ev[0]) {
    HIP_TRY(root, hipEventCreate(&root->gather_ev[0]));
    HIP_TRY(root, hipEventCreate(&root->gather_ev[1]));
  }
  // the root's own rows, packed like a peer's
  size_t floats = 0;
  if (int rc = band_pack(root, which, root->band_buf, &floats)) return rc;
  // One launch pulls every band -- the root's own and every imported rank's, straight out of the peers' mapped
  // buffers -- into row order: all peer -> root xGMI links carry their band at the same time, nothing is staged.
  HIP_TRY(root, hipEventRecord(root->gather_ev[0], root->stream));
  DGatherBands bands{};
  uint32_t n = 0, max_pix = 0;
  auto flush = [&]() {
    if (n) launch_gather_bands(root->stream, bands, n, max_pix, channels, (uint32_t)P, root->gather_frame);
    n = 0;
    max_pix = 0;
  };
  auto add = [&](const float* src, const DBand& band, uint32_t pix_count) {
    bands.src[n].src = src;
    bands.src[n].band = band;
    bands.src[n].pix_count = pix_count;
    max_pix = std::max(max_pix, pix_count);
    if (++n == (uint32_t)kGatherBands) flush();
  };
  add(root->band_buf, root->band, root->pix_count);
  for (size_t r = 0; r < root->peers.size(); ++r) {
    const auto& peer = root->peers[r];
    if (!peer.mapped) continue;
    add(static_cast<const float*>(peer.mapped), DBand{peer.h.pix_begin, peer.h.width, peer.h.rank, peer.h.nranks, peer.h.block_rows},
        peer.h.pix_count);
  }
  flush();
  HIP_TRY(root, hipEventRecord(root->gather_ev[1], root->stream));
  root->gather_timed = true;
  return check_last(root, "gather");
}

int ptc_gather_last_us(ptc_ctx* root, float* microseconds)
{
  if (!root || !microseconds) return PTC_ERR_INVALID;
  *microseconds = 0.0f;
  if (!root->gather_timed) return fail(root, PTC_ERR_INVALID, "no gather has run");
  if (int rc = bind_device(root)) return rc;
  float ms = 0.0f;
  HIP_TRY(root, hipEventSynchronize(root->gather_ev[1]));
  HIP_TRY(root, hipEventElapsedTime(&ms, root->gather_ev[0], root->gather_ev[1]));
  *microseconds = ms * 1e3f;
  return PTC_OK;
}

int ptc_gather_frame(ptc_ctx* root, int which, void* dst, int dst_is_device)
{
  if (!root || !dst) return PTC_ERR_INVALID;
  if (!root->pix_capacity) return fail(root, PTC_ERR_INVALID, "ptc_resize first");
  if (which < PTC_BUF_COLOR || which > PTC_BUF_FINAL) return fail(root, PTC_ERR_INVALID, "unknown buffer");
  if (int rc = bind_device(root)) return rc;
  const int channels = which == PTC_BUF_DEPTH ? 1 : 3;
  if (int rc = gather_rows(root, which, channels)) return rc;
  const size_t bytes = (size_t)root->width * root->height * (size_t)channels * sizeof(float);
  HIP_TRY(root, hipMemcpyAsync(dst, root->gather_frame, bytes, dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, root->stream));
  HIP_TRY(root, hipStreamSynchronize(root->stream));
  return PTC_OK;
}

int ptc_gather_present_rgba8(ptc_ctx* root, void* dst, int dst_is_device, int display_type)
{
  if (!root || !dst) return PTC_ERR_INVALID;
  if (!root->pix_capacity) return fail(root, PTC_ERR_INVALID, "ptc_resize first");
  if (int rc = bind_device(root)) return rc;
  int which = PTC_BUF_COLOR, mode = 0;
  switch (display_type) {
  case PTC_DISPLAY_FINAL:
  case PTC_DISPLAY_COLOR: break;
  case PTC_DISPLAY_NORMAL: which = PTC_BUF_NORMAL; mode = 1; break;
  case PTC_DISPLAY_DEPTH: which = PTC_BUF_DEPTH; mode = 2; break;
  default: return fail(root, PTC_ERR_INVALID, "unknown display type");
  }
  const int channels = which == PTC_BUF_DEPTH ? 1 : 3;
  if (int rc = gather_rows(root, which, channels)) return rc;
  const uint32_t P = root->width * root->height;
  if (!root->gather_rgba) HIP_TRY(root, hipMalloc(reinterpret_cast<void**>(&root->gather_rgba), (size_t)P * 4u));
  uint32_t* out = dst_is_device ? static_cast<uint32_t*>(dst) : root->gather_rgba;
  launch_preview_packed(root->stream, root->gather_frame, P, channels, mode, out);
  if (int rc = check_last(root, "preview")) return rc;
  if (!dst_is_device) HIP_TRY(root, hipMemcpyAsync(dst, root->gather_rgba, (size_t)P * 4u, hipMemcpyDeviceToHost, root->stream));
  HIP_TRY(root, hipStreamSynchronize(root->stream));
  return PTC_OK;
}

int ptc_synchronize(ptc_ctx* ctx)
{
  if (!ctx) return PTC_ERR_INVALID;
  if (int rc = bind_device(ctx)) return rc;
  return sync_frames(ctx);
}

int ptc_get_stats(ptc_ctx* ctx, ptc_stats* out)
{
  if (!ctx || !out) return PTC_ERR_INVALID;
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = sync_frames(ctx)) return rc;
  std::memset(out, 0, sizeof *out);
  uint32_t flags = 0;
  const size_t head = offsetof(DeviceCounters, work);  // everything but the fetch cursors
  std::vector<char> buf(sizeof(DeviceCounters));
  for (size_t f = 0; f < ctx->slots.size(); ++f)
    for (int k = 0; k < ctx->slots[f].capacity; ++k) {
      HIP_TRY(ctx, hipMemcpy(buf.data(), ctx->slots[f].counters + k, head, hipMemcpyDeviceToHost));
      const DeviceCounters& host = *reinterpret_cast<const DeviceCounters*>(buf.data());
      out->rays_total += host.rays_total;
      flags |= host.flags;
      if ((int)f == ctx->last_slot && k + 1 == (int)ctx->slots[f].bi.count)
        for (int i = 0; i < PTC_MAX_BOUNCES_CAP; ++i) out->last_live[i] = i < ctx->max_bounces ? host.live[i] : 0u;
    }
  HIP_TRY(ctx, hipMemcpy(buf.data(), ctx->misc_counters, head, hipMemcpyDeviceToHost));
  flags |= reinterpret_cast<const DeviceCounters*>(buf.data())->flags;
  out->frames = ctx->frames;
  out->bvh_node_count = ctx->bvh_nodes;
  out->bvh_max_depth = ctx->bvh_depth;
  out->triangle_count = ctx->triangles;
  out->stack_capacity = kStackDepth;
  if (flags & kFlagStackOverflow) return fail(ctx, PTC_ERR_STACK, "traversal stack overflow during rendering");
  if (flags & kFlagDispatchOrder)
    return fail(ctx, PTC_ERR_HIP, "k_shade_fused gave up waiting for a predecessor tile's survivor count: the image is invalid "
                                  "(set the parameter \"fused_shade\" to 0 to use the three-kernel path)");
  return PTC_OK;
}

static int drain_timed(ptc_ctx* ctx)
{
  for (auto& tl : ctx->timed) {
    float ms = 0.0f;
    HIP_TRY(ctx, hipEventSynchronize(tl.stop));
    HIP_TRY(ctx, hipEventElapsedTime(&ms, tl.start, tl.stop));
    if (tl.bounce < 0) {
      ctx->denoise_ms += ms;
      ctx->denoise_passes += 1u;
    } else {
      ctx->trace_ms[tl.bounce] += ms;
      ctx->trace_launches[tl.bounce] += 1u;
    }
    ctx->free_events.push_back(tl.start);
    ctx->free_events.push_back(tl.stop);
  }
  ctx->timed.clear();
  return PTC_OK;
}

int ptc_set_profiling(ptc_ctx* ctx, int time_trace_kernel, int count_tests)
{
  if (!ctx) return PTC_ERR_INVALID;
  ctx->time_trace = time_trace_kernel != 0;
  ctx->count_tests = count_tests != 0;
  return PTC_OK;
}

int ptc_reset_profile(ptc_ctx* ctx)
{
  if (!ctx) return PTC_ERR_INVALID;
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = sync_frames(ctx)) return rc;
  if (int rc = drain_timed(ctx)) return rc;
  std::memset(ctx->trace_ms, 0, sizeof ctx->trace_ms);
  std::memset(ctx->trace_launches, 0, sizeof ctx->trace_launches);
  ctx->denoise_ms = 0.0;
  ctx->denoise_passes = 0;
  ctx->intersect_redone = 0;
  const size_t off = offsetof(DeviceCounters, paths), end = offsetof(DeviceCounters, work);
  for (auto& sl : ctx->slots)
    for (int k = 0; k < sl.capacity; ++k)
      HIP_TRY(ctx, hipMemset(reinterpret_cast<char*>(sl.counters + k) + off, 0, end - off));
  return PTC_OK;
}

int ptc_get_profile(ptc_ctx* ctx, ptc_profile* out)
{
  if (!ctx || !out) return PTC_ERR_INVALID;
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = sync_frames(ctx)) return rc;
  if (int rc = drain_timed(ctx)) return rc;
  std::memset(out, 0, sizeof *out);
  const size_t head = offsetof(DeviceCounters, work);
  std::vector<char> buf(sizeof(DeviceCounters));
  for (auto& sl : ctx->slots)
    for (int k = 0; k < sl.capacity; ++k) {
    HIP_TRY(ctx, hipMemcpy(buf.data(), sl.counters + k, head, hipMemcpyDeviceToHost));
    const DeviceCounters& host = *reinterpret_cast<const DeviceCounters*>(buf.data());
    for (int b = 0; b < PTC_MAX_BOUNCES_CAP; ++b) {
      out->paths[b] += host.paths[b];
      out->box_tests[b] += host.box_tests[b];
      out->tri_tests[b] += host.tri_tests[b];
      out->max_box_tests[b] = std::max(out->max_box_tests[b], host.max_box_tests[b]);
      out->listed_rays[b] += host.listed_rays[b];
      out->slow_rays[b] += host.slow_rays[b];
      out->node_visits[b] += host.node_visits[b];
    }
  }
  for (int b = 0; b < PTC_MAX_BOUNCES_CAP; ++b) {
    out->trace_ms[b] = ctx->trace_ms[b];
    out->trace_launches[b] = ctx->trace_launches[b];
  }
  out->slow_rays[0] += ctx->intersect_redone;
  out->denoise_ms = ctx->denoise_ms;
  out->denoise_passes = ctx->denoise_passes;
  return PTC_OK;
}

int ptc_intersect_rays(ptc_ctx* ctx, const float* rays, uint32_t n, float* hit_t, float* hit_normal, uint32_t* hit_material,
                       uint8_t* hit_side)
{
  if (!ctx || !rays || !hit_t || !hit_normal || !hit_material || !hit_side) return PTC_ERR_INVALID;
  if (!ctx->has_scene) return fail(ctx, PTC_ERR_NO_SCENE, "no scene uploaded");
  if (n == 0) return PTC_OK;
  if (n > 0x7fffffffu) return fail(ctx, PTC_ERR_INVALID, "too many rays");
  if (int rc = bind_device(ctx)) return rc;
  if (int rc = flush_pending(ctx)) return rc;
  // The default schedule (variant 3) is the production closest-hit stage itself: the object list walked by the
  // traversal launches (k_traverse4 with its sphere runs and its exact redo), fed with the caller's rays instead of
  // path state.
  // Path rays know two t_min values (1e-4, and 1e-5 after a dielectric: a flag bit) and start every bounce with
  // t_max = FLT_MAX; a caller's t_max enters as the "closest hit so far" the segments carry in the hit record.
  // Rays with another t_min take the one-wavefront-per-64-rays kernel with exact box decisions (variant 1).
  bool path_like = ctx->trace_variant == 3;
  for (uint32_t i = 0; i < n && path_like; ++i) {
    const float tmin = rays[8u * (size_t)i + 3u], tmax = rays[8u * (size_t)i + 7u];
    path_like = (tmin == 1e-4f || tmin == 1e-5f) && tmax >= 0.0f;
  }
  constexpr uint32_t kUntouched = 0x7fffffffu;  // material field of a record no segment has written: a miss
  std::vector<void*> pool;
  float4 *ro = nullptr, *rd = nullptr;
  DHits hits{};
  uint32_t *chunk_counts = nullptr, *slow_list = nullptr, *slow_stack = nullptr;
  uint2* spill = nullptr;
  DeviceCounters* counters = nullptr;
  int rc = dev_alloc(ctx, pool, &ro, n);
  if (!rc) rc = dev_alloc(ctx, pool, &rd, n);
  if (!rc) rc = dev_alloc(ctx, pool, &hits.tp, n);
  if (!rc) rc = dev_alloc(ctx, pool, &hits.nm, n);
  if (!rc && path_like) {
    rc = dev_alloc(ctx, pool, &chunk_counts, (size_t)n / kChunk + 1u);
    if (!rc) rc = dev_alloc(ctx, pool, &slow_list, n);
    if (!rc) rc = dev_alloc(ctx, pool, &slow_stack, (size_t)kStackDepth * kWave);
    if (!rc) rc = dev_alloc(ctx, pool, &spill, (size_t)ctx->scene.spill_cap * ctx->scene.spill_stride);
    if (!rc) rc = dev_alloc(ctx, pool, &counters, 1);
  }
  if (rc) {
    free_pool(pool);
    return rc;
  }
  std::vector<float4> ho(n), hd(n), tp(n), nm(n);
  float untouched_bits;
  std::memcpy(&untouched_bits, &kUntouched, 4);
  for (uint32_t i = 0; i < n; ++i) {
    const float* r = rays + 8u * (size_t)i;
    if (path_like) {
      const uint32_t flag = r[3] == 1e-5f ? 0x80000000u : 0u;
      float fbits;
      std::memcpy(&fbits, &flag, 4);
      ho[i] = make_float4(r[0], r[1], r[2], fbits);
      hd[i] = make_float4(r[4], r[5], r[6], 0.0f);
      tp[i] = make_float4(r[7], 0.0f, 0.0f, 0.0f);
      nm[i] = make_float4(0.0f, 0.0f, 0.0f, untouched_bits);
    } else {
      ho[i] = make_float4(r[0], r[1], r[2], r[3]);
      hd[i] = make_float4(r[4], r[5], r[6], r[7]);
    }
  }
  hipError_t e = hipMemcpyAsync(ro, ho.data(), n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(rd, hd.data(), n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess && path_like) {
    e = hipMemcpyAsync(hits.tp, tp.data(), n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hits.nm, nm.data(), n * sizeof(float4), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(counters, 0, sizeof(DeviceCounters), ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&counters->live[0], &n, sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      DScene scene = ctx->scene;
      scene.spill = spill;
      scene.slow_stack = slow_stack;
      DPaths paths{ro, rd, nullptr};
      DBatchInfo bi{};
      bi.stride = n;
      bi.chunk_stride = n / kChunk + 1u;
      bi.count = 1u;
      const uint32_t waves = std::min<uint32_t>(ctx->traverse_waves, std::max<uint32_t>(8u, ((n / (4u * kWave)) + 7u) & ~7u));
      int work_slot = 0;
      for (size_t k = 0; k < ctx->launches.size(); ++k) {
        const auto& l = ctx->launches[k];
        if (l.pre_begin < l.pre_end) {
          scene.fold_run = fold_run_of(ctx, l.pre_begin, l.pre_end);
          launch_spheres(ctx->stream, scene, l.pre_begin, l.pre_end, false, paths, hits, n, 0, counters, bi);
        }
        scene.cur = ctx->mesh_views[ctx->object_mesh[l.mesh]];
        launch_traverse(ctx->stream, scene, l.mesh, false, paths, hits, 0, work_slot++ % kWorkSlots, counters, false, waves, slow_list,
                        nullptr, ctx->trace_variant, bi);
      }
      launch_tail_count(ctx->stream, scene, ctx->tail_begin, ctx->tail_end, false, paths, hits, n, 0, chunk_counts, counters, bi);
      e = hipGetLastError();
    }
  } else if (e == hipSuccess) {
    launch_intersect(ctx->stream, ctx->scene, ro, rd, n, hits, ctx->misc_counters, ctx->trace_variant == 0 ? 0 : 1);
    e = hipGetLastError();
  }
  uint32_t dev_flags = 0u;
  unsigned long long redone = 0ull;
  if (e == hipSuccess && path_like)
    e = hipMemcpyAsync(&redone, &counters->slow_rays[0], sizeof redone, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(tp.data(), hits.tp, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(nm.data(), hits.nm, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess && path_like)
    e = hipMemcpyAsync(&dev_flags, &counters->flags, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  free_pool(pool);
  if (e != hipSuccess) return fail(ctx, PTC_ERR_HIP, std::string("intersect_rays: ") + hipGetErrorString(e));
  if (dev_flags & kFlagStackOverflow) return fail(ctx, PTC_ERR_STACK, "traversal stack overflow in ptc_intersect_rays");
  ctx->intersect_redone += redone;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t ms;
    std::memcpy(&ms, &nm[i].w, 4);
    const bool miss = path_like ? (ms & 0x7fffffffu) == kUntouched : tp[i].x < 0.0f;
    hit_t[i] = miss ? -1.0f : tp[i].x;
    hit_normal[3u * i] = nm[i].x;
    hit_normal[3u * i + 1u] = nm[i].y;
    hit_normal[3u * i + 2u] = nm[i].z;
    hit_material[i] = miss ? 0u : (ms & 0x7fffffffu);
    hit_side[i] = miss ? (uint8_t)0 : (uint8_t)(ms >> 31);
  }
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

}  // extern "C"
