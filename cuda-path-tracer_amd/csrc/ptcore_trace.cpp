// ptcore_trace.cpp -- the launch plan of a batch of frames (batch_begin / batch_bounce / batch_end), ptc_trace and its stepwise
// form, the live counts, ptc_intersect_rays.  Part of libptcore.so (ptcore_ctx.hpp).
#include "ptcore_ctx.hpp"

using namespace pt;
using namespace ptcd;

namespace {

// the epoch of the next look-back launch on a slot's tile descriptors (k_shade_fused, the listing k_raygen / k_spheres)
uint32_t next_epoch(ptc_ctx::FrameSlot& sl)
{
  sl.shade_epoch = sl.shade_epoch >= 0x3fffffffu ? 1u : sl.shade_epoch + 1u;
  return sl.shade_epoch;
}

// Persistent wavefronts of a traversal launch.  A launch ends with its longest ray (about 100 us however few rays it
// carries), so a small launch wants about one ray per lane -- rays / 64 wavefronts, at least 1024, at most 3072 -- and
// only a launch with eight or more rays per lane fills every wavefront slot of the chip (traverse_waves: what is
// resident at five per SIMD).  Measured on single 1080p frames (2 M rays at the first bounce, 65 k at the eighth): one size
// for all bounces 2.60 ms per frame, sized per bounce 2.1 ms.  The ray count of a bounce lives on the device; the
// host sizes with the counts of a recent frame (FrameSlot::live_host), a bounce it knows nothing about with its cap.
uint32_t traverse_waves_for(ptc_ctx* ctx, uint32_t frames, int bounce, bool listed, bool run = false)
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
  uint64_t cap = rays >= (uint64_t)ctx->traverse_waves * kWave * ctx->small_rays_per_lane ? ctx->traverse_waves
                                                                                           : std::min<uint32_t>(ctx->traverse_waves, ctx->small_waves);
  // (a launch over a run of instances, k_traverse4m: "run_waves" -- room for the other batch's kernels, so only when there is one)
  if (run && ctx->big_slots >= 2) cap = std::min<uint64_t>(cap, ctx->run_waves);
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
  sl.prefolded = false;
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
  // "prefold": the object list opens with a sphere run in front of a mesh launch that lists its rays -- ray generation walks
  // the run for the primary rays itself, and bounce 0 starts with k_list_flags (batch_bounce) instead of k_spheres
  const bool raygen_folds = ctx->prefold && ctx->trace_variant == 3 && ctx->fused_shade && ctx->filter_rays && !ctx->ray_sort && sl.next_flags &&
                            !ctx->launches.empty() && ctx->launches[0].pre_begin < ctx->launches[0].pre_end;
  if (raygen_folds) {
    const auto& l0 = ctx->launches[0];
    DNextRun next{};
    next.begin = l0.pre_begin;
    next.end = l0.pre_end;
    next.filt_begin = l0.mesh;
    next.filt_end = l0.mesh + (uint32_t)launch_run(ctx, 0);
    next.fold_run = fold_run_of(ctx, l0.pre_begin, l0.pre_end);
    next.hits = sl.hits;
    next.flags = sl.next_flags;
    launch_raygen_next(sl.stream, ctx->scene, cams, sl.bi, ctx->band, ctx->pix_count, sl.paths[0], sl.counters, next);
    sl.prefolded = true;
  } else
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
  const bool prefolded = sl.prefolded;  // ... the previous bounce's shade kernel has, for the leading sphere run ("prefold")
  sl.prefolded = false;
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
      if (l.pre_begin < l.pre_end && k == 0 && prefolded) {
        // "prefold": the shade kernel of the bounce before has walked this run for every survivor and left the hit records
        // (in what is now sl.hits) and one byte per ray; all that is left of k_spheres is its work list
        launch_list_flags(sl.stream, sl.next_flags, ctx->pix_count, bounce, sl.counters, sl.bi, sl.worklist, sl.tile_desc, sl.tile_stride, next_epoch(sl));
        wrote = true;
      } else if (l.pre_begin < l.pre_end) {
        scene.lanes_run = lanes_run_of(ctx, l.pre_begin, l.pre_end);
        scene.fold_run = fold_run_of(ctx, l.pre_begin, l.pre_end);
        launch_spheres(sl.stream, scene, l.pre_begin, l.pre_end, !wrote, in, sl.hits, ctx->pix_count, bounce, sl.counters, sl.bi,
                       by_spheres ? l.mesh : 0u, by_spheres ? l.mesh + (uint32_t)run : 0u, by_spheres ? sl.worklist : nullptr,
                       sl.tile_desc, sl.tile_stride, by_spheres ? next_epoch(sl) : 0u);
        wrote = true;
      }
      // ("pair_batches": this batch's traversal launches wait for the partner's previous one, and say when they are done)
      if (ctx->turn_mine >= 0 && ctx->turn_wait >= 0) HIP_TRY(ctx, hipStreamWaitEvent(sl.stream, ctx->turn_event[ctx->turn_wait], 0));
      ptc_ctx::TimedLaunch tl{nullptr, nullptr, bounce};
      if (int rc = timed_begin(tl)) return rc;
      const uint32_t waves = traverse_waves_for(ctx, sl.bi.count, bounce, listed, run > 1);
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
      if (ctx->turn_mine >= 0) {
        HIP_TRY(ctx, hipEventRecord(ctx->turn_event[ctx->turn_mine], sl.stream));
        ctx->turn_wait = ctx->turn_mine;
      }
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
    // "prefold": the next bounce opens with a sphere run in front of its first traversal launch, which lists its rays -- this
    // kernel walks that run for every survivor (DNextRun).  By k_spheres' own conditions: the listing is on, no ray sorting.
    DNextRun next{};
    const bool prefold = ctx->prefold && persistent && !last && wrote && ctx->filter_rays && !ctx->ray_sort && sl.next_flags && !ctx->launches.empty() &&
                         ctx->launches[0].pre_begin < ctx->launches[0].pre_end;
    if (prefold) {
      const auto& l0 = ctx->launches[0];
      next.begin = l0.pre_begin;
      next.end = l0.pre_end;
      next.filt_begin = l0.mesh;
      next.filt_end = l0.mesh + (uint32_t)launch_run(ctx, 0);
      next.fold_run = fold_run_of(ctx, l0.pre_begin, l0.pre_end);
      next.hits = sl.hits_other;
      next.flags = sl.next_flags;
    }
    launch_shade_fused(sl.stream, scene, tail ? ctx->tail_begin : 0u, tail ? ctx->tail_end : 0u, !wrote, in, out, sl.hits, ctx->pix_count,
                       ctx->staging(), bounce, last, slot_base_dev, sl.tile_desc, sl.tile_stride, sl.shade_epoch, sl.stage, ctx->band,
                       sl.counters, octs, sl.bi, bounce == 0 && sl.primary_finished ? sl.worklist : nullptr, prefold ? &next : nullptr);
    if (prefold) {
      std::swap(sl.hits, sl.hits_other);
      sl.prefolded = true;
    }
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

// May the batch that batch_begin has just opened run as bounce 0's traversal launch + ONE persistent launch (k_persist)?
// The launch plan must be the simplest one -- every bounce is one traversal launch over one mesh object with nothing in front
// of it, then the kernel that ends the bounce (a sphere run may end the object list) --, the shade pass the fused one, no ray
// sorting, staged samples, no instrumentation (the counting runs and the per-launch events keep the per-bounce launches:
// same results), and slots must fit the 26 bits a lane keeps them in.
bool persist_ok(const ptc_ctx* ctx, int count)
{
  if (!ctx->persist || ctx->trace_variant != 3 || !ctx->fused_shade || ctx->ray_sort || !ctx->staging()) return false;
  if (ctx->count_tests || ctx->max_bounces < 2 || count < (int)ctx->persist_min_frames) return false;
  if (ctx->launches.size() != 1u || ctx->launches[0].pre_begin != ctx->launches[0].pre_end || launch_run(ctx, 0) != 1u) return false;
  const auto& sl = ctx->slots[(size_t)ctx->active_slot];
  if (!sl.persist || (uint64_t)sl.bi.stride * (uint64_t)count >= (1ull << kPersistSlotBits)) return false;
  return true;
}

// The batch after raygen: bounce 0's traversal launch as ever (entry points, work list), then the persistent launch for
// everything else -- every shade pass and the traversal of bounces 1 .. max_bounces - 1.
int batch_persist(ptc_ctx* ctx, const uint32_t* slot_base_dev)
{
  auto& sl = ctx->slots[(size_t)ctx->active_slot];
  const int MB = ctx->max_bounces;
  DScene scene = ctx->scene;
  scene.spill = sl.spill;
  scene.slow_stack = sl.slow_stack;
  const auto& l = ctx->launches[0];
  scene.cur = ctx->mesh_views[ctx->object_mesh[l.mesh]];
  // ---- bounce 0's closest hit: batch_bounce's traversal part ----
  ptc_ctx::TimedLaunch tl{nullptr, nullptr, 0};
  if (ctx->time_trace) {
    for (hipEvent_t* e : {&tl.start, &tl.stop}) {
      if (!ctx->free_events.empty()) {
        *e = ctx->free_events.back();
        ctx->free_events.pop_back();
      } else {
        HIP_TRY(ctx, hipEventCreate(e));
      }
    }
    HIP_TRY(ctx, hipEventRecord(tl.start, sl.stream));
  }
  const bool listed = sl.first_listed;
  scene.beam = sl.beam;
  launch_traverse(sl.stream, scene, l.mesh, true, sl.paths[0], sl.hits, 0, sl.work_slot++ % kWorkSlots, sl.counters, false,
                  traverse_waves_for(ctx, sl.bi.count, 0, listed), sl.slow_list, listed ? sl.worklist : nullptr, 3, sl.bi, listed);
  if (ctx->time_trace) {
    HIP_TRY(ctx, hipEventRecord(tl.stop, sl.stream));
    ctx->timed.push_back(tl);
  }
  // ---- everything else ----
  scene.beam = DBeam{};
  const bool tail = ctx->tail_begin < ctx->tail_end;
  scene.lanes_run = tail ? lanes_run_of(ctx, ctx->tail_begin, ctx->tail_end) : 0u;
  scene.fold_run = tail && !scene.lanes_run ? fold_run_of(ctx, ctx->tail_begin, ctx->tail_end) : 0u;
  DPersistArgs pa{};
  pa.st = sl.persist;
  pa.paths[0] = sl.paths[0];
  pa.paths[1] = sl.paths[1];
  pa.max_bounces = MB;
  pa.service_every = ctx->persist_service_every;
  pa.help_tiles = ctx->persist_help_tiles;
  pa.tail_begin = tail ? ctx->tail_begin : 0u;
  pa.tail_end = tail ? ctx->tail_end : 0u;
  pa.staged = ctx->staging() ? 1 : 0;
  pa.slot_base = slot_base_dev;
  pa.tile_desc = sl.tile_desc;
  pa.tile_stride = sl.tile_stride;
  pa.epoch0 = next_epoch(sl);  // bounce b's pass: epoch0 + b
  for (int b = 1; b < MB; ++b) next_epoch(sl);
  if (sl.shade_epoch < pa.epoch0) {  // the epochs wrapped inside this batch: start the batch's run at 1 again
    sl.shade_epoch = 0;
    pa.epoch0 = next_epoch(sl);
    for (int b = 1; b < MB; ++b) next_epoch(sl);
  }
  pa.stage = sl.stage;
  pa.band = ctx->band;
  pa.list0 = sl.primary_finished ? sl.worklist : nullptr;
  pa.slow_list = sl.slow_list;
  ptc_ctx::TimedLaunch tp{nullptr, nullptr, 1};
  if (ctx->time_trace) {
    for (hipEvent_t* e : {&tp.start, &tp.stop}) {
      if (!ctx->free_events.empty()) {
        *e = ctx->free_events.back();
        ctx->free_events.pop_back();
      } else {
        HIP_TRY(ctx, hipEventCreate(e));
      }
    }
    HIP_TRY(ctx, hipEventRecord(tp.start, sl.stream));
  }
  launch_persist(sl.stream, scene, l.mesh, sl.hits, sl.counters, sl.bi, pa, ctx->traverse_waves, tail, pa.list0 != nullptr);
  if (ctx->time_trace) {
    HIP_TRY(ctx, hipEventRecord(tp.stop, sl.stream));
    ctx->timed.push_back(tp);
  }
  sl.cur = MB & 1;
  sl.bounces_done = MB;
  ++ctx->persist_launches;
  return check_last(ctx, "persistent launch");
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


}  // namespace

namespace ptcd {
// one whole batch on the next slot: raygen, the bounces (per-bounce launches or the persistent launch), the fold
static int enqueue_batch(ptc_ctx* ctx, std::vector<ptc_ctx::Pending>& items)
{
  if (int rc = batch_begin(ctx, items.data(), (int)items.size())) return rc;
  const uint32_t* slot_base = ctx->slot_offset ? ctx->slot_offset_dev : nullptr;
  if (persist_ok(ctx, (int)items.size())) {
    if (int rc = batch_persist(ctx, slot_base)) {
      ctx->active_slot = -1;
      return rc;
    }
    return batch_end(ctx);
  }
  for (int b = 0; b < ctx->max_bounces; ++b)
    if (int rc = batch_bounce(ctx, b, slot_base)) {
      ctx->active_slot = -1;
      return rc;
    }
  return batch_end(ctx);
}

// two batches bounce by bounce on two slots, their traversal launches taking turns ("pair_batches")
static int enqueue_pair(ptc_ctx* ctx, std::vector<ptc_ctx::Pending>& a, std::vector<ptc_ctx::Pending>& b)
{
  for (hipEvent_t& e : ctx->turn_event)
    if (!e) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const uint32_t* slot_base = ctx->slot_offset ? ctx->slot_offset_dev : nullptr;
  if (int rc = batch_begin(ctx, a.data(), (int)a.size())) return rc;
  const int slot_a = ctx->active_slot;
  if (int rc = batch_begin(ctx, b.data(), (int)b.size())) return rc;
  const int slot_b = ctx->active_slot;
  ctx->turn_wait = -1;
  int rc = PTC_OK;
  for (int bounce = 0; bounce < ctx->max_bounces && rc == PTC_OK; ++bounce) {
    ctx->active_slot = slot_a;
    ctx->turn_mine = 0;
    rc = batch_bounce(ctx, bounce, slot_base);
    if (rc != PTC_OK) break;
    ctx->active_slot = slot_b;
    ctx->turn_mine = 1;
    rc = batch_bounce(ctx, bounce, slot_base);
  }
  ctx->turn_mine = ctx->turn_wait = -1;
  if (rc != PTC_OK) {
    ctx->active_slot = -1;
    return rc;
  }
  ctx->active_slot = slot_a;
  if (int rc2 = batch_end(ctx)) return rc2;
  ctx->active_slot = slot_b;
  return batch_end(ctx);
}

// enqueue the iterations ptc_trace has queued
int flush_pending(ptc_ctx* ctx, bool from_trace)
{
  if (ctx->pending.empty() && ctx->held.empty()) return PTC_OK;
  if (int rc = bind_device(ctx)) return rc;
  // "pair_batches": a batch that ptc_trace has just filled waits for its partner
  const bool pairing = ctx->pair_batches && ctx->big_slots >= 2 && ctx->staging() && ctx->trace_variant == 3 && !ctx->persist;
  if (pairing && from_trace && ctx->held.empty() && (int)ctx->pending.size() >= batch_limit(ctx) && batch_limit(ctx) > 1) {
    ctx->held.swap(ctx->pending);
    return PTC_OK;
  }
  std::vector<ptc_ctx::Pending> first, second;
  first.swap(ctx->held);
  second.swap(ctx->pending);
  if (first.empty()) first.swap(second);
  if (!second.empty() && pairing && first.size() > 1 && second.size() > 1) return enqueue_pair(ctx, first, second);
  if (int rc = enqueue_batch(ctx, first)) return rc;
  if (!second.empty()) return enqueue_batch(ctx, second);
  return PTC_OK;
}

}  // namespace ptcd

extern "C" {
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
  if ((int)ctx->pending.size() >= batch_limit(ctx)) return flush_pending(ctx, true);
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

}  // extern "C"

