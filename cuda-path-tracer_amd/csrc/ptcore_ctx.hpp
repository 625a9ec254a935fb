// ptcore_ctx.hpp -- private to libptcore.so: the context behind include/ptcore.h and the helpers its translation units share
// (ptcore.cpp: context, frame slots, parameters, views; ptcore_scene.cpp: scene upload; ptcore_trace.cpp: the launch plan of a
// batch of frames; ptcore_bands.cpp: several GPUs; ptcore_checks.cpp: host-side checks of the schedule helpers).
#pragma once

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

// in-flight path state a context allocates when the caller has not chosen frames_in_flight
constexpr uint64_t kAutoFrameBytes = 24ull << 30;

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
    DHits hits_other{};             // "prefold": the second set of hit records (the shade kernel of bounce b writes bounce b + 1's while other
                                    // tiles still read bounce b's); `hits` is always the set the current bounce reads
    uint8_t* next_flags = nullptr;  // "prefold": per slot of the next bounce, does the ray go on the mesh launch's work list
    bool prefolded = false;         // ... the previous bounce's shade kernel has walked this bounce's leading sphere run (k_list_flags follows)
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
    DPersist* persist = nullptr;    // "persist": the state of this slot's bounce-spanning launch (k_persist)
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
  // "pair_batches" (round 5, measured: see DESIGN section 4d, the alternative that needs no hand-over inside a launch): a full
  // batch is HELD until the next one is full (or anything else looks at the context); the two are then enqueued bounce by
  // bounce on their two slots, and events make their traversal launches take turns -- A.T(0), B.T(0), A.T(1), ... -- so that the
  // kernels that end a bounce of one batch run beside the traversal launch of the other instead of beside nothing
  std::vector<Pending> held;
  int pair_batches = 0;
  hipEvent_t turn_event[2] = {nullptr, nullptr};  // recorded behind the traversal launches of a pair's two batches
  int turn_wait = -1;                             // which of them the next traversal launch of the pair waits for (-1: none)
  int turn_mine = -1;                             // which one the batch being enqueued records (-1: not a pair)
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
  // "persist": bounces >= 1 of a batch and every shade pass as ONE launch (k_persist) when the launch plan is
  // one mesh object per bounce with nothing in front of it (ptcore_trace.cpp, persist_ok); 0: one launch per kernel and bounce
  int persist = 0;  // (off: measured 3 x slower than the per-bounce launches, DESIGN section 4d -- the service wavefronts are latency-bound)
  uint32_t persist_service_every = 5;  // "persist_service_every": one wavefront in this many shades, the others walk
  uint32_t persist_help_tiles = 8;     // "persist_help_tiles": tiles a walking wavefront without rays shades before it looks for rays again (0: it sleeps)
  uint32_t persist_min_frames = 2;     // "persist_min_frames": batches of fewer frames keep the per-bounce launches
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
  uint32_t run_waves = 3072;          // "run_waves" (round 5): most persistent wavefronts of a launch that walks a run of instances (k_traverse4m).
                                      // Config 2 -- 4.5 M listed rays per launch, two batches in flight -- 14.0 -> 14.8 Grays/s on 1536 ... 3584
                                      // wavefronts against 5120: the other batch's HBM-bound kernels get on the chip (profiles/r05_run_waves.txt)
  // live paths entering each bounce of one recent frame (what a frame of this scene / camera looks like): the host
  // never waits for them, they only size the traversal launches
  uint32_t est_live[2 * (kMaxBounces + 1)] = {};  // live[], then listed_now[] (DeviceCounters) of a recent batch's first frame
  bool est_valid = false;
  bool filter_rays = true;    // "filter_rays": a sphere run in front of a mesh launch also lists the rays that launch has to walk
  bool prefold = true;        // "prefold": the kernel that ends a bounce also walks the NEXT bounce's leading sphere run for its survivors (config 2's shape)
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
  uint32_t persist_launches = 0;         // batches traced through k_persist since ptc_reset_profile
};


namespace ptcd {

extern thread_local std::string g_create_error;
void request_hw_queues();
int fail(ptc_ctx* ctx, int code, const std::string& msg);

#define HIP_TRY(ctx, expr)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return fail(ctx, e_ == hipErrorOutOfMemory ? PTC_ERR_OOM : PTC_ERR_HIP,                   \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                           \
  } while (0)

int check_last(ptc_ctx* ctx, const char* what);
int bind_device(ptc_ctx* ctx);
void free_pool(std::vector<void*>& pool);
DCamera make_camera(const ptc_camera& c, uint32_t w, uint32_t h);
int flush_pending(ptc_ctx* ctx, bool from_trace = false);  // enqueue the iterations ptc_trace has queued (ptcore_trace.cpp)
int sync_frames(ptc_ctx* ctx);
void free_slots(ptc_ctx* ctx);
int frame_ready(ptc_ctx* ctx);

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

}  // namespace ptcd

