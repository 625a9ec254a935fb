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
//
// This unit: context, frame slots (ptc_resize), parameters, denoise / present / download, statistics.  Scene upload:
// ptcore_scene.cpp; the launch plan of a batch: ptcore_trace.cpp; several GPUs: ptcore_bands.cpp; host-side checks: ptcore_checks.cpp.
#include "ptcore_ctx.hpp"

using namespace pt;
using namespace ptcd;

namespace ptcd {

thread_local std::string g_create_error;

// The frames in flight run on separate HIP streams, and streams only overlap when they sit on different
// hardware queues; the runtime's default is 4 queues per process.  Ask for more before this library's first HIP
// call (no effect if the application has set the variable or has already initialised HIP itself: such an
// application exports GPU_MAX_HW_QUEUES on its own, see ptcore.h).
void request_hw_queues()
{
  static const int once = setenv("GPU_MAX_HW_QUEUES", "24", 0);
  (void)once;
}

int fail(ptc_ctx* ctx, int code, const std::string& msg)
{
  if (ctx) ctx->err = msg;
  else g_create_error = msg;
  return code;
}

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

void free_pool(std::vector<void*>& pool)
{
  for (void* p : pool) (void)hipFree(p);
  pool.clear();
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
  ctx->held.clear();
  ctx->active_slot = -1;
  ctx->est_valid = false;
}

int frame_ready(ptc_ctx* ctx)
{
  if (!ctx) return PTC_ERR_INVALID;
  if (!ctx->has_scene) return fail(ctx, PTC_ERR_NO_SCENE, "no scene uploaded");
  if (!ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "ptc_resize first");
  return bind_device(ctx);
}

}  // namespace ptcd

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
  for (hipEvent_t e : ctx->turn_event)
    if (e) (void)hipEventDestroy(e);
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
    // path state, hit records (two sets with "prefold"), staging: 148 (181) bytes per pixel and frame in flight
    const uint64_t per_frame = (ctx->prefold ? 181ull : 148ull) * P;
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
      if (int rc = dev_alloc(ctx, pool, &sl.paths[k].t2, BP)) return rc;
    }
    if (int rc = dev_alloc(ctx, pool, &sl.hits.tp, BP)) return rc;
    if (int rc = dev_alloc(ctx, pool, &sl.hits.nm, BP)) return rc;
    if (ctx->prefold) {
      if (int rc = dev_alloc(ctx, pool, &sl.hits_other.tp, BP)) return rc;
      if (int rc = dev_alloc(ctx, pool, &sl.hits_other.nm, BP)) return rc;
      if (int rc = dev_alloc(ctx, pool, &sl.next_flags, BP)) return rc;
    }
    sl.prefolded = false;
    if (int rc = dev_alloc(ctx, pool, &sl.chunk_counts, (size_t)B * chunks)) return rc;
    if (int rc = dev_alloc(ctx, pool, &sl.chunk_offsets, (size_t)B * chunks)) return rc;
    // (tile descriptors: k_shade_fused's 512-slot tiles, or the persistent launch's 128-slot ones)
    sl.tile_stride = std::max(shade_tiles_per_frame((uint32_t)P), persist_tiles_per_frame((uint32_t)P));
    if (int rc = dev_alloc(ctx, pool, &sl.tile_desc, (size_t)B * sl.tile_stride)) return rc;
    HIP_TRY(ctx, hipMemsetAsync(sl.tile_desc, 0, sizeof(unsigned long long) * (size_t)B * sl.tile_stride, ctx->stream));
    sl.shade_epoch = 0;
    if (int rc = dev_alloc(ctx, pool, &sl.slow_list, BP)) return rc;
    if (ctx->staged) {
      if (int rc = dev_alloc(ctx, pool, &sl.persist, 1)) return rc;
      HIP_TRY(ctx, hipMemsetAsync(sl.persist, 0, sizeof(DPersist), ctx->stream));
    }
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

int ptc_restart(ptc_ctx* ctx)
{
  if (!ctx) return PTC_ERR_INVALID;
  // Iterations still queued are traced first, not dropped: the reference has rendered them by the time restart()
  // runs (path_trace is synchronous there), and a present between restart and the next path_trace shows them.
  // A viewer restarts after it has presented, i.e. with an empty queue, so this costs nothing where it matters.
  if (!ctx->pending.empty() || !ctx->held.empty())
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
  if (std::strcmp(name, "run_waves") == 0) {
    if (value < 8 || value > 65536) return fail(ctx, PTC_ERR_INVALID, "run_waves out of range");
    ctx->run_waves = (uint32_t)value;
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
  if (std::strcmp(name, "persist") == 0) {
    if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "persist must be 0 or 1");
    if (int rc = flush_pending(ctx)) return rc;
    ctx->persist = value;
    return PTC_OK;
  }
  if (std::strcmp(name, "persist_service_every") == 0) {
    if (value < 2 || value > 64) return fail(ctx, PTC_ERR_INVALID, "persist_service_every must be in [2, 64]");
    if (int rc = flush_pending(ctx)) return rc;
    ctx->persist_service_every = (uint32_t)value;
    return PTC_OK;
  }
  if (std::strcmp(name, "prefold") == 0) {
    if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "prefold must be 0 or 1");
    if (ctx->pix_capacity) return fail(ctx, PTC_ERR_INVALID, "set prefold before ptc_resize");
    ctx->prefold = value != 0;
    return PTC_OK;
  }
  if (std::strcmp(name, "pair_batches") == 0) {
    if (value != 0 && value != 1) return fail(ctx, PTC_ERR_INVALID, "pair_batches must be 0 or 1");
    if (int rc = flush_pending(ctx)) return rc;
    ctx->pair_batches = value;
    return PTC_OK;
  }
  if (std::strcmp(name, "persist_help_tiles") == 0) {
    if (value < 0 || value > 4096) return fail(ctx, PTC_ERR_INVALID, "persist_help_tiles must be in [0, 4096]");
    if (int rc = flush_pending(ctx)) return rc;
    ctx->persist_help_tiles = (uint32_t)value;
    return PTC_OK;
  }
  if (std::strcmp(name, "persist_min_frames") == 0) {
    if (value < 1 || value > kMaxBatch) return fail(ctx, PTC_ERR_INVALID, "persist_min_frames must be in [1, 32]");
    if (int rc = flush_pending(ctx)) return rc;
    ctx->persist_min_frames = (uint32_t)value;
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
  if (flags & kFlagPersistStall)
    return fail(ctx, PTC_ERR_HIP, "the persistent launch (k_persist) gave up waiting for work: a wavefront of its batch never finished (\"persist\" 0 turns it off)");
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
  ctx->persist_launches = 0;
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
  out->persist_launches = ctx->persist_launches;
  return PTC_OK;
}

}  // extern "C"

