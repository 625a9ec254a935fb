// pt_shade.hip -- ray generation and the end of a bounce (see pt_kernels.hip for the map of the path): k_raygen, k_spheres,
// k_tail_count / k_scan / k_shade (the three-kernel form), k_shade_fused (default), k_sort_octant, k_accumulate.

#include "pt_device.hpp"
#include "pt_rng.hpp"
#include "pt_beam_rules.hpp"
#include "pt_feed_rules.hpp"
static_assert(pt::beam_rules::kLeaf == pt::kLeafBit, "pt_beam_rules.hpp restates the leaf bit of the four-wide node (pt_device.hpp)");
static_assert((uint32_t)pt::beam_rules::kEntries == pt::kBeamEntries, "pt_beam_rules.hpp restates the entries per tile (pt_device.hpp)");
static_assert(pt::feed_rules::kBatch == (uint32_t)pt::kWave, "a feed batch is one wavefront's worth of rays");
#include <float.h>

namespace pt {

#include "pt_kernels_common.inc"

// raygen_kernel, ray_gen.cu:11-32.  Slot s of this context holds pixel band_pixel(band, s).
// kFilter ("filter_rays"): the bounce's first launch is a traversal launch over the mesh objects [filt_begin, filt_end)
// (no sphere run in front of it): the rays that may hit one of their world boxes go on its work list, the others get
// their miss record here (what that launch would have written for them) -- the sky pixels of an outdoor scene never
// reach the traversal kernel.
// kFinish (with kFilter, when that launch walks the scene's WHOLE object list): a ray that is not listed hits nothing at
// all, so its path ends here -- throughput (1, 1, 1) times the sky into the frame, exactly what the shade kernel does
// for a miss at bounce 0 (path_tracer.cu:304-307, ray_gen.cu:26-28) -- and neither its ray nor a miss record is written;
// bounce 0's k_shade_fused then walks the work list instead of all slots.  Per sky pixel and frame: 32 bytes written
// here instead of 48, and 48 bytes the shade kernel no longer reads.
template <bool kFilter, bool kFinish>
__global__ __launch_bounds__(256) void k_raygen(DCameras cams, DBatchInfo bi, DBand band, uint32_t pix_count,
                                                DPaths paths, DeviceCounters* counters, const DObject* objects, uint32_t filt_begin,
                                                uint32_t filt_end, uint32_t* worklist, DHits hits, DTileScan scan, uint32_t tile_stride,
                                                DFrame fb, int staged)
{
  const uint32_t frame = blockIdx.x % bi.count;  // see DBatchInfo; frame-fastest: neighbouring workgroups take their tickets on different lines
  const DCamera& cam = cams.c[frame];
  const uint32_t iteration = bi.iteration[frame];
  paths.o4 += (size_t)frame * bi.stride;
  paths.d4 += (size_t)frame * bi.stride;
  counters += frame;
  scan.desc += (size_t)frame * tile_stride;
  if (kFinish && staged) {
    fb.color4 += (size_t)frame * bi.stride;
    fb.nd4 += (size_t)frame * bi.stride;
  }
  const uint32_t acc_iteration = staged ? 0u : iteration;
  const uint32_t tiles = gridDim.x / bi.count;
  const uint32_t tile = kFilter ? list_tile(counters, tiles) : blockIdx.x / bi.count;
  const uint32_t block_first = tile * (256u * kListPer);  // a workgroup generates kListPer x 256 consecutive slots
  if (tile == 0u) {
    if (threadIdx.x == 0u) counters->live[0] = pix_count;
    // fetch cursors of this frame's persistent traversal launches
    for (uint32_t i = threadIdx.x; i < (uint32_t)kWorkSlots * 8u; i += 256u) (&counters->work[0][0][0])[i * 32u] = 0u;
  }
  uint32_t may_mask = 0u;
#pragma unroll
  for (int j = 0; j < kListPer; ++j) {
    const uint32_t s = block_first + (uint32_t)j * 256u + threadIdx.x;
    if (s >= pix_count) continue;
    const uint32_t pixel = band_pixel(band, s);
    const uint32_t x = pixel % cam.width, y = pixel / cam.width;
    Minstd rng;
    rng.seed(path_seed(pixel, iteration));
    const float fx = (float)x + rng.uniform();
    const float fy = (float)y + rng.uniform();
    f3 o, d;
    generate_ray(cam, fx, fy, o, d);
    bool may_hit = true;
    if (kFilter) {
      may_hit = may_hit_boxes(objects, filt_begin, filt_end, o, d, FLT_MAX);
      may_mask |= may_hit ? 1u << j : 0u;
    }
    if (kFinish && !may_hit) {
      const uint32_t local_pixel = band_local(band, pixel);
      const f3 color = mk3(1.0f, 1.0f, 1.0f) * background(d);
      accumulate_nd(fb.nd4, local_pixel, acc_iteration, -d, 1e6f);
      accumulate_color(fb.color4, local_pixel, acc_iteration, color);
      continue;
    }
    stnt(&paths.o4[s], make_float4(o.x, o.y, o.z, __uint_as_float(pixel)));
    stnt(&paths.d4[s], make_float4(d.x, d.y, d.z, 0.0f));
    // (the throughput of a primary ray is (1, 1, 1), ray_gen.cu:25: the shade kernels know that at bounce 0 and neither
    // is it written here nor read there -- 32 bytes per pixel and frame less)
    if (kFilter && !may_hit) stnt(&hits.tp[(size_t)frame * bi.stride + s], make_float4(-1.0f, 0.f, 0.f, 0.f));
  }
  if (kFilter) list_rays(may_mask, worklist, counters, (size_t)frame * bi.stride, tile, tiles, scan);
}

// "prefold" at bounce 0: ray generation for a scene whose object list opens with a sphere run in front of a mesh (config 2).  The
// primary ray is in registers: k_spheres<first, filter>'s work for it -- the run, the hit or miss record, the byte that says
// whether it goes on the mesh launch's work list (shade_tile<.., kNext> does the same for the later bounces) -- is done here, and
// bounce 0 starts with k_list_flags.
__global__ __launch_bounds__(256) void k_raygen_next(DScene sc, DCameras cams, DBatchInfo bi, DBand band, uint32_t pix_count, DPaths paths,
                                                     DeviceCounters* counters, DNextRun next)
{
  const uint32_t frame = blockIdx.x % bi.count;
  const DCamera& cam = cams.c[frame];
  const uint32_t iteration = bi.iteration[frame];
  const size_t fo = (size_t)frame * bi.stride;
  paths.o4 += fo;
  paths.d4 += fo;
  next.hits.tp += fo;
  next.hits.nm += fo;
  next.flags += fo;
  counters += frame;
  const uint32_t tile = blockIdx.x / bi.count;
  const uint32_t block_first = tile * (256u * kListPer);
  if (tile == 0u) {
    if (threadIdx.x == 0u) counters->live[0] = pix_count;
    for (uint32_t i = threadIdx.x; i < (uint32_t)kWorkSlots * 8u; i += 256u) (&counters->work[0][0][0])[i * 32u] = 0u;
  }
#pragma unroll 1
  for (int j = 0; j < kListPer; ++j) {
    const uint32_t s = block_first + (uint32_t)j * 256u + threadIdx.x;
    if (s >= pix_count) continue;
    const uint32_t pixel = band_pixel(band, s);
    const uint32_t x = pixel % cam.width, y = pixel / cam.width;
    Minstd rng;
    rng.seed(path_seed(pixel, iteration));
    const float fx = (float)x + rng.uniform();
    const float fy = (float)y + rng.uniform();
    Ray ray;
    generate_ray(cam, fx, fy, ray.o, ray.d);
    stnt(&paths.o4[s], make_float4(ray.o.x, ray.o.y, ray.o.z, __uint_as_float(pixel)));
    stnt(&paths.d4[s], make_float4(ray.d.x, ray.d.y, ray.d.z, 0.0f));
    ray.tmin = 1e-4f;  // (load_ray: the sign bit of a primary ray's pixel word is clear)
    ray.tmax = FLT_MAX;
    Hit rec;
    bool changed = false;
    if (next.fold_run != 0u) sphere_fold(sc, next.begin, next.end, ray, rec, changed);
    else sphere_segment<true>(sc, next.begin, next.end, ray, rec, changed);
    if (changed) store_hit(next.hits, s, rec);
    else stnt(&next.hits.tp[s], make_float4(-1.0f, 0.f, 0.f, 0.f));
    next.flags[s] = may_hit_boxes(sc.objects, next.filt_begin, next.filt_end, ray.o, ray.d, ray.tmax) ? (uint8_t)1 : (uint8_t)0;
  }
}

// A run of sphere objects that does not end the object list (the spheres in front of a mesh), continuing from /
// handing on the closest hit in the hit record.  (The run that ENDS the list -- or is the whole list -- is part of
// the kernel that ends the bounce.)  (Taking the sphere runs into the traversal kernel instead was tried in round 2: inlined or as a
// call, their temporaries pushed loop-carried state of the walk into scratch, with reloads inside its hot loop.)
// kFilter: the launch is followed by a traversal launch over the mesh objects [filt_begin, filt_end).  A ray that SURELY
// misses the world boxes of all of them (the same test with the same margin by which that launch skips an instance,
// traverse4m_walk::begin_object), or whose boxes all start beyond the closest hit so far, has nothing to do there: only
// the others are put on the work list (batch-global slots, DeviceCounters::list_count per frame; their order is
// irrelevant -- results are written per slot), and the traversal launch fetches its rays through that list.  In the
// Cornell-box scenes most rays of most bounces never come near the meshes.
template <bool kFirst, bool kFilter>
// (at least six wavefronts per SIMD: 80 registers, 3-14 spilled, against 92 and five wavefronts: config 2 +2.7 %; seven: +1.7 %,
// eight (49-62 spilled): +1.1 %; profiles/r04_config2_counters.txt)
#ifndef PT_SPHERES_WAVES
#define PT_SPHERES_WAVES 6
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PT_SPHERES_WAVES, 8))) void k_spheres(DScene sc, uint32_t obj_begin, uint32_t obj_end, DPaths paths, DHits hits,
                                                 int bounce, DeviceCounters* counters, DBatchInfo bi, uint32_t filt_begin,
                                                 uint32_t filt_end, uint32_t* worklist, DTileScan scan, uint32_t tile_stride)
{
  const uint32_t frame = blockIdx.x % bi.count;  // see DBatchInfo; frame-fastest as in k_raygen
  paths.o4 += (size_t)frame * bi.stride;
  paths.d4 += (size_t)frame * bi.stride;
  hits.tp += (size_t)frame * bi.stride;
  hits.nm += (size_t)frame * bi.stride;
  counters += frame;
  scan.desc += (size_t)frame * tile_stride;
  const uint32_t n = counters->live[bounce];
  const uint32_t tiles = (n + 256u * kListPer - 1u) / (256u * kListPer);
  if (blockIdx.x / bi.count >= tiles) {
    if (kFilter && tiles == 0u && blockIdx.x / bi.count == 0u && threadIdx.x == 0u) counters->list_count = 0u;  // nothing alive: an empty list
    return;
  }
  const uint32_t tile = kFilter ? list_tile(counters, tiles) : blockIdx.x / bi.count;
  const uint32_t block_first = tile * (256u * kListPer);  // a workgroup tests kListPer x 256 consecutive slots
  uint32_t may_mask = 0u;
  // (sphere_run_lanes is for the run that ends the list: here, in front of a mesh, the spheres are typically the walls of a
  // room -- every ray hits every one of them, there is little to rule out, and sphere_segment shares the inverse
  // transform's normalised direction among them: measured 907 us against 1114 for the per-lane form, config 2)
  const bool lanes_run = PT_SPHERE_LANES_LEADING && sc.lanes_run != 0u;
  const bool fold_run = sc.fold_run != 0u;
#pragma unroll 1
  for (int j = 0; j < kListPer; ++j) {
    const uint32_t s = block_first + (uint32_t)j * 256u + threadIdx.x;
    if (s >= n) continue;
    Ray ray = load_ray(paths, s);
    if (!kFirst) {
      const float carried = ldnt(&hits.tp[s]).x;
      if (carried >= 0.0f) ray.tmax = carried;
    }
    Hit rec;
    bool changed = false;
    if (lanes_run) {
      const float4 ro4[1] = {ldnt(&paths.o4[s])}, rd4[1] = {ldnt(&paths.d4[s])};
      float4 rtp[1] = {make_float4(ray.tmax < FLT_MAX ? ray.tmax : -1.0f, 0.f, 0.f, 0.f)}, rnm[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
      changed = sphere_run_lanes<1>(sc, obj_begin, obj_end, ro4, rd4, rtp, rnm, 1u) != 0u;
      if (changed) {
        stnt(&hits.tp[s], rtp[0]);
        stnt(&hits.nm[s], rnm[0]);
        ray.tmax = rtp[0].x;
      }
    } else {
      if (fold_run) sphere_fold(sc, obj_begin, obj_end, ray, rec, changed);
      else sphere_segment<true>(sc, obj_begin, obj_end, ray, rec, changed);
      if (changed) store_hit(hits, s, rec);
    }
    if (!changed && kFirst) stnt(&hits.tp[s], make_float4(-1.0f, 0.f, 0.f, 0.f));
    if (kFilter && may_hit_boxes(sc.objects, filt_begin, filt_end, ray.o, ray.d, ray.tmax)) may_mask |= 1u << j;
  }
  if (kFilter) list_rays(may_mask, worklist, counters, (size_t)frame * bi.stride, tile, tiles, scan);
}

// "prefold": the work list of a bounce whose leading sphere run the shade kernel of the bounce before has walked for every
// survivor (shade_tile<.., kNext>): k_spheres<.., kFilter>'s list -- the flagged slots in slot order, the frame's count in
// DeviceCounters::list_count -- from the bytes that kernel left, one per slot.  A compaction of its own rather than list_rays: a
// workgroup takes 4096 consecutive slots, every thread sixteen of them in ONE 16-byte load (the first form, list_rays' 1024-slot
// tiles with a byte per load, spent 234 us per launch on 28,800 workgroups' chains of ticket, load, look-back and store: as
// long as the k_traverse4m launch behind it took to walk its rays' first nodes).  Tiles by ticket and look-back over the
// slot's descriptors as everywhere (a launch of its own epoch).
constexpr uint32_t kFlagsPer = 16u, kFlagsTile = 256u * kFlagsPer;
__global__ __launch_bounds__(256) void k_list_flags(const uint8_t* flags, int bounce, DeviceCounters* counters, DBatchInfo bi, uint32_t* worklist,
                                                    DTileScan scan, uint32_t tile_stride)
{
  __shared__ uint32_t s_wave[4];
  __shared__ uint32_t s_base, s_tile;
  const uint32_t frame = blockIdx.x % bi.count;
  const size_t fo = (size_t)frame * bi.stride;
  flags += fo;
  counters += frame;
  scan.desc += (size_t)frame * tile_stride;
  const uint32_t n = counters->live[bounce];
  const uint32_t tiles = (n + kFlagsTile - 1u) / kFlagsTile;
  if (blockIdx.x / bi.count >= tiles) {
    if (tiles == 0u && blockIdx.x / bi.count == 0u && threadIdx.x == 0u) counters->list_count = 0u;  // nothing alive: an empty list
    return;
  }
  if (threadIdx.x == 0u) {
    const uint32_t t = __hip_atomic_fetch_add(&counters->shade_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t + 1u == tiles) __hip_atomic_store(&counters->shade_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_tile = t;
  }
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint32_t first = tile * kFlagsTile + threadIdx.x * kFlagsPer;  // this thread's sixteen slots: first .. first + 15
  // (the flag array is a frame's stride long and the stride a multiple of 16? not necessarily: a thread whose sixteen bytes
  // cross the live count reads them one by one)
  uint32_t bits = 0u;
  if (first + kFlagsPer <= n && ((reinterpret_cast<uintptr_t>(flags + first) & 15u) == 0u)) {
    const uint4 w = *reinterpret_cast<const uint4*>(flags + first);
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k) bits |= ((ww[q] >> (8 * k)) & 0xffu) != 0u ? 1u << (4 * q + k) : 0u;
  } else {
    for (uint32_t k = 0u; k < kFlagsPer; ++k)
      if (first + k < n && flags[first + k] != 0u) bits |= 1u << k;
  }
  const uint32_t mine = (uint32_t)__popc(bits);
  // exclusive prefix within the wavefront, then across the workgroup's four
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
    if ((int)(threadIdx.x & 63u) >= off) incl += up;
  }
  const uint32_t wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 63u) s_wave[wave] = incl;
  __syncthreads();
  uint32_t before = 0u, agg = 0u;
#pragma unroll
  for (uint32_t w = 0u; w < 4u; ++w) {
    before += w < wave ? s_wave[w] : 0u;
    agg += s_wave[w];
  }
  if (wave == 0u) {
    const unsigned long long tag = (unsigned long long)scan.epoch << 34;
    if (threadIdx.x == 0u)
      __hip_atomic_store(&scan.desc[tile], tag | (tile == 0u ? kDescPrefix : kDescAggregate) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t excl = 0u;
    if (tile != 0u) {
      excl = tile_lookback(scan.desc, tile, scan.epoch, &counters->flags);
      if (threadIdx.x == 0u)
        __hip_atomic_store(&scan.desc[tile], tag | kDescPrefix | (excl + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0u) {
      s_base = excl;
      if (tile + 1u == tiles) counters->list_count = excl + agg;
    }
  }
  __syncthreads();
  uint32_t at = s_base + before + (incl - mine);
  while (bits != 0u) {
    const uint32_t k = (uint32_t)__ffs((int)bits) - 1u;
    bits &= bits - 1u;
    worklist[fo + at++] = (uint32_t)fo + first + k;
  }
}

// The end of a bounce's closest-hit stage: the sphere run that ends the object list (if any) and the live count of
// every 64-slot chunk (ballot / popcount of "the hit record holds a hit"), for the compaction scan.
// kSpheres: objects [obj_begin, obj_end) are tested; kFirst: nothing has written the hit record in this bounce yet.
// 256-thread workgroups: one wavefront per SIMD fits beside the other stream's persistent traversal wavefronts as soon
// as one of those has left (1024-thread workgroups wait until four per SIMD have: measured 15 % slower end to end,
// together with a scan fused in behind a "last workgroup" sign-off).
template <bool kSpheres, bool kFirst>
__global__ __launch_bounds__(256) void k_tail_count(DScene sc, uint32_t obj_begin, uint32_t obj_end, DPaths paths, DHits hits,
                                                    int bounce, uint32_t* chunk_counts, DeviceCounters* counters, DBatchInfo bi)
{
  const uint32_t frame = blockIdx.y;  // see DBatchInfo
  paths.o4 += (size_t)frame * bi.stride;
  paths.d4 += (size_t)frame * bi.stride;
  hits.tp += (size_t)frame * bi.stride;
  hits.nm += (size_t)frame * bi.stride;
  chunk_counts += (size_t)frame * bi.chunk_stride;
  counters += frame;
  const uint32_t n = counters->live[bounce];
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  // a wavefront beyond the live range owns no chunk: k_scan reads ceil(n / 64) entries, and in a batch the next
  // entries belong to the next frame
  if ((s & ~63u) >= n) return;
  bool hit = false;
  if (s < n) {
    float t_so_far = -1.0f;
    if (!kFirst) t_so_far = ldnt(&hits.tp[s]).x;
    hit = t_so_far >= 0.0f;
    if (kSpheres) {
      Ray ray = load_ray(paths, s);
      if (hit) ray.tmax = t_so_far;
      Hit rec;
      bool changed = false;
      sphere_segment(sc, obj_begin, obj_end, ray, rec, changed);
      if (changed) {
        store_hit(hits, s, rec);
        hit = true;
      } else if (kFirst) {
        stnt(&hits.tp[s], make_float4(-1.0f, 0.f, 0.f, 0.f));
      }
    }
  }
  const uint64_t live = __ballot(hit);
  if ((threadIdx.x & 63u) == 0u) chunk_counts[s / kChunk] = (uint32_t)__popcll(live);
}

// Exclusive scan of the per-chunk live counts (one workgroup; <= ~32k chunks at 1080p).
// Writes live[bounce+1] (0 after the last bounce: nothing survives the cap) and the ray counter.
__global__ __launch_bounds__(1024) void k_scan(int bounce, int last_bounce, const uint32_t* chunk_counts,
                                               uint32_t* chunk_offsets, DeviceCounters* counters, DBatchInfo bi)
{
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  const uint32_t frame = blockIdx.x;  // one workgroup per frame of the batch
  chunk_counts += (size_t)frame * bi.chunk_stride;
  chunk_offsets += (size_t)frame * bi.chunk_stride;
  counters += frame;
  const uint32_t n = counters->live[bounce];
  const uint32_t chunks = (n + kChunk - 1u) / kChunk;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0u) s_carry = 0u;
  __syncthreads();
  for (uint32_t base = 0; base < chunks; base += 1024u) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < chunks ? chunk_counts[i] : 0u;
    // inclusive scan inside the wavefront
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off, 64);
      if (lane >= (uint32_t)off) x += y;
    }
    if (lane == 63u) s_wave[wave] = x;
    __syncthreads();
    uint32_t wave_prefix = 0u;
    for (uint32_t w = 0; w < wave; ++w) wave_prefix += s_wave[w];
    const uint32_t carry = s_carry;
    if (i < chunks) chunk_offsets[i] = carry + wave_prefix + x - v;
    __syncthreads();
    if (threadIdx.x == 1023u) s_carry = carry + wave_prefix + x;
    __syncthreads();
  }
  if (threadIdx.x == 0u) {
    counters->live[bounce + 1] = last_bounce ? 0u : s_carry;
    counters->rays_total += n;
    counters->paths[bounce] += n;
  }
}

// material_kernel (path_tracer.cu:292-315) + the stable compaction scatter + the final gather of
// every path that ends at this bounce.
// staged: `fb` is the slot's staging buffer (one sample per frame of the batch, plain stores); k_accumulate then
// folds the staged samples into the real framebuffer in iteration order.  Otherwise the running mean goes
// straight into `fb`.
__global__ __launch_bounds__(256) void k_shade(DScene sc, DPaths in, DPaths out, DHits hits, int staged, int bounce,
                                               int last_bounce, const uint32_t* slot_base, const uint32_t* chunk_offsets,
                                               DFrame fb, DBand band, DeviceCounters* counters, uint8_t* octs, DBatchInfo bi)
{
  const uint32_t frame = blockIdx.y;  // see DBatchInfo
  const uint32_t iteration = bi.iteration[frame];
  const uint32_t acc_iteration = staged ? 0u : iteration;
  if (octs) octs += (size_t)frame * bi.stride;
  in.o4 += (size_t)frame * bi.stride;
  in.d4 += (size_t)frame * bi.stride;
  in.t2 += (size_t)frame * bi.stride;
  out.o4 += (size_t)frame * bi.stride;
  out.d4 += (size_t)frame * bi.stride;
  out.t2 += (size_t)frame * bi.stride;
  hits.tp += (size_t)frame * bi.stride;
  hits.nm += (size_t)frame * bi.stride;
  chunk_offsets += (size_t)frame * bi.chunk_stride;
  if (staged) {
    fb.color4 += (size_t)frame * bi.stride;
    fb.nd4 += (size_t)frame * bi.stride;
  }
  counters += frame;
  const uint32_t n = counters->live[bounce];
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (blockIdx.x * 256u >= n) return;
  const bool active = s < n;
  bool survives = false;
  f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 0), color = mk3(0, 0, 0);
  uint32_t pixbits = 0u;
  if (active) {
    const float4 o4 = ldnt(&in.o4[s]);
    const float4 d4 = ldnt(&in.d4[s]);
    const float2 t2 = bounce == 0 ? make_float2(1.0f, 1.0f) : ldnt(&in.t2[s]);  // (throughput: d4.w, t2 -- DPaths; k_raygen writes neither)
    const float4 tp = ldnt(&hits.tp[s]);
    ro = xyz(o4);
    rd = xyz(d4);
    color = mk3(bounce == 0 ? 1.0f : d4.w, t2.x, t2.y);
    pixbits = __float_as_uint(o4.w);
    const uint32_t pixel = pixbits & 0x7fffffffu;
    const uint32_t local_pixel = band_local(band, pixel);
    bool tmin_flag = (pixbits >> 31) != 0u;

    if (tp.x < 0.0f) {
      // miss: throughput *= sky; the path ends (path_tracer.cu:304-307, 283-289)
      color = color * background(rd);
      if (bounce == 0) accumulate_nd(fb.nd4, local_pixel, acc_iteration, -rd, 1e6f);  // raygen defaults, ray_gen.cu:26-28
      accumulate_color(fb.color4, local_pixel, acc_iteration, color);
    } else {
      const float4 nm = ldnt(&hits.nm[s]);
      const f3 hn = xyz(nm);
      if (bounce == 0) accumulate_nd(fb.nd4, local_pixel, acc_iteration, hn, tp.x);  // path_tracer.cu:308-311
      const uint32_t ms = __float_as_uint(nm.w);
      const DMaterial m = sc.materials[ms & 0x7fffffffu];
      // RNG re-seeded from the global slot index, then discard(bounce) (path_tracer.cu:300-301)
      const uint32_t slot = (slot_base ? *slot_base : 0u) + s;
      Minstd rng;
      rng.seed(path_seed(slot, iteration));
      rng.discard((uint32_t)bounce);
      const f3 hp = mk3(tp.y, tp.z, tp.w);
      evaluate_material(ro, rd, tmin_flag, hp, hn, ms >> 31, m, rng, color);
      if (last_bounce) {
        accumulate_color(fb.color4, local_pixel, acc_iteration, color);  // capped paths deposit raw throughput
      } else {
        survives = true;
        pixbits = pixel | (tmin_flag ? 0x80000000u : 0u);
      }
    }
  }
  const uint64_t live = __ballot(survives);
  if (survives) {
    const uint32_t dst = chunk_offsets[s / kChunk] + rank_below(live);
    stnt(&out.o4[dst], make_float4(ro.x, ro.y, ro.z, __uint_as_float(pixbits)));
    stnt(&out.d4[dst], make_float4(rd.x, rd.y, rd.z, color.x));
    stnt(&out.t2[dst], make_float2(color.y, color.z));
    // direction octant of the new ray, for the coherence sort of the next bounce (k_sort_octant)
    if (octs) octs[dst] = (uint8_t)((rd.x < 0.0f ? 1u : 0u) | (rd.y < 0.0f ? 2u : 0u) | (rd.z < 0.0f ? 4u : 0u));
  }
}

// ------------------------------------------------------------------------------------------------
// the end of a bounce in ONE pass: trailing sphere run + material + stable compaction + final gather
// ------------------------------------------------------------------------------------------------
// k_tail_count -> k_scan -> k_shade read every ray and hit record twice and put a one-workgroup scan between two
// full-width launches.  This kernel does the three jobs in one pass over the slots (path_tracer.cu:292-315 material_kernel,
// :454-457 stable_partition, :317-330 final gather; :78-100 for the sphere run that ends the object list):
//   * a workgroup owns a TILE of 256 x kFuseK consecutive slots (every thread kFuseK of them, 256 apart: coalesced),
//     loads ray + hit, finishes the closest hit (the trailing spheres), and knows from "is there a hit" alone which of
//     its paths survive -- so the tile's survivor count is published after ONE round trip to memory;
//   * the stable offset of the tile = survivors of all tiles before it, found by decoupled look-back over the tile
//     descriptors (aggregate / inclusive prefix, Merrill & Garland): a wavefront reads up to 64 predecessors at once;
//   * tiles are taken in TICKET order (one agent-scope atomic per workgroup on the frame's counter line; blocks are
//     numbered frame-fastest, so neighbouring workgroups of a batch take their tickets on different lines), not in
//     blockIdx order: whoever waits for a tile's descriptor waits for a workgroup that is RUNNING (it has its ticket),
//     whatever else holds the chip's wavefront slots.  Block order is 1-3 % faster (the ticket's round trip sits in
//     front of a workgroup's first load) and is NOT safe: with several streams' kernels on the chip, workgroups of
//     one kernel fill an XCD spinning for a predecessor that waits for a slot on an XCD filled by another kernel's
//     spinners -- seen once in 67 GPU tests x several runs (ten one-frame launches on ten streams), caught by the
//     bounded wait below.  The wait stays bounded all the same: a workgroup that has waited about a second sets
//     kFlagDispatchOrder and ptc_get_stats reports an error instead of an image;
//   * descriptors carry the launch's epoch, so nothing has to be cleared between launches.
// Same arithmetic and same slot order as the three kernels it replaces: images are bit-identical (tests).
// Slots per thread: 2 (round 4; 4 until then).  With four the kernel needs 128 registers and spills 18 of them at four
// wavefronts per SIMD; with two it needs 88, spills nothing and runs five: config 2 +2.2 %, config 3 +0.6 ... 1.2 %;
// one slot (eight wavefronts): -4 ... -6 % (profiles/r04_config2_counters.txt).  Six wavefronts (80 registers) spill 53.
#include "pt_shade_tile.inc"


template <bool kSpheres, bool kFirst, bool kNext = false>
// (occupancy bounds re-measured on the final build: at least 5 or 6 wavefronts per SIMD forces spills, -8 % / -13 % end
// to end; 1 to 3 compile to the same 124 registers)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PT_SHADE_WAVES, 8))) void k_shade_fused(DScene sc, uint32_t obj_begin, uint32_t obj_end, DPaths in, DPaths out, DHits hits,
                                                     int staged, int bounce, int last_bounce, const uint32_t* slot_base,
                                                     unsigned long long* tile_desc, uint32_t tile_stride, uint32_t epoch, DFrame fb,
                                                     DBand band, DeviceCounters* counters, uint8_t* octs, DBatchInfo bi, const uint32_t* list,
                                                     DNextRun next)
{
  __shared__ uint32_t s_excl, s_tile;
  __shared__ uint32_t s_cnt[kFuseK * 4];
  const uint32_t frame = blockIdx.x % bi.count;  // frame-fastest: neighbouring blocks take their tickets on different lines
  const uint32_t iteration = bi.iteration[frame];
  const size_t fo = (size_t)frame * bi.stride;
  if (octs) octs += fo;
  in.o4 += fo;
  in.d4 += fo;
  in.t2 += fo;
  out.o4 += fo;
  out.d4 += fo;
  out.t2 += fo;
  hits.tp += fo;
  hits.nm += fo;
  tile_desc += (size_t)frame * tile_stride;
  if (staged) {
    fb.color4 += fo;
    fb.nd4 += fo;
  }
  counters += frame;
  // list ("filter_rays", bounce 0 of a scene whose whole object list is the bounce's one listed traversal launch): the
  // rays that are not on the launch's work list have been finished by k_raygen, which knew that they hit nothing; this
  // kernel then walks the list (slot order: the survivors land where they would have) instead of all slots
  const uint32_t n_all = counters->live[bounce];
  const uint32_t n = list ? counters->list_count : n_all;
  if (list) list += fo;
  const uint32_t tiles = (n + kFuseTile - 1u) / kFuseTile;
  // The grid is sized for a frame of all-live slots; the workgroups the frame has no tile for leave without a ticket:
  // exactly `tiles` tickets are taken per frame.  (Dispatching the empty workgroups costs little: a launch sized by the
  // last batch's live counts, whose workgroups came back for more tiles when there were too few, was slower -- the
  // loop cost 24 spilled registers -- profiles/r03_shade_breakdown.txt.)
  if (blockIdx.x / bi.count >= tiles) {
    if (tiles == 0u && blockIdx.x / bi.count == 0u && threadIdx.x == 0u) {  // nothing alive: nothing follows
      counters->live[bounce + 1] = 0u;
      counters->rays_total += n_all;  // (not zero when k_raygen has finished every ray of the frame, see `list`)
      counters->paths[bounce] += n_all;
    }
    return;
  }
  if (threadIdx.x == 0u) {
    const uint32_t t = __hip_atomic_fetch_add(&counters->shade_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // every tile of the frame is taken once the last ticket is out: the next launch starts from zero
    if (t + 1u == tiles) __hip_atomic_store(&counters->shade_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_tile = t;
  }
  __syncthreads();
  const uint32_t tile = s_tile;

  // (the tile itself: pt_shade_tile.inc, shared with the persistent launch's service wavefronts)
  if (kNext) {  // (the next bounce's records and bytes of this frame)
    next.hits.tp += fo;
    next.hits.nm += fo;
    next.flags += fo;
  }
  shade_tile<kSpheres, kFirst, 4, false, kNext>(sc, obj_begin, obj_end, in, out, hits, staged, bounce, last_bounce, slot_base, tile_desc, epoch, fb, band,
                                                counters, octs, iteration, list, fo, tile, tiles, n, n_all, s_cnt, &s_excl, &next);
}

// Ray sorting ("ray_sort", BASELINE.json's ray-sorted wavefront; the reference keeps a sort_by_key by material
// commented out, path_tracer.cu:439-446).  The slots -- and with them the random numbers, which are keyed on the
// compacted slot index -- are NOT permuted: what is sorted is the ORDER in which the persistent traversal lanes pick
// their rays up.  Within every block of 4096 consecutive slots (neighbouring pixels: neighbouring ray origins) the
// rays are grouped by direction octant, stably, into an index array the ray feed reads through; a wavefront's 64 rays
// then start close together AND head the same way.  Results cannot change; what it buys is measured in DESIGN.md.
constexpr uint32_t kSortBlock = 4096u;
__global__ __launch_bounds__(1024) void k_sort_octant(const uint8_t* octs, uint32_t* order, int bounce, DeviceCounters* counters,
                                                      DBatchInfo bi)
{
  __shared__ uint32_t s_cnt[8][4][16];  // [octant][round][wavefront]
  __shared__ uint32_t s_base[8][4][16];
  const uint32_t frame = blockIdx.y;
  const uint32_t n = counters[frame].live[bounce];
  const uint32_t block0 = blockIdx.x * kSortBlock;
  if (block0 >= n) return;
  const size_t fbase = (size_t)frame * bi.stride;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t oct[4], rank[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t s = block0 + (uint32_t)r * 1024u + threadIdx.x;
    oct[r] = s < n ? (uint32_t)octs[fbase + s] : 8u;
#pragma unroll
    for (uint32_t o = 0; o < 8u; ++o) {
      const uint64_t m = __ballot(oct[r] == o);
      if (oct[r] == o) rank[r] = rank_below(m);
      if (lane == 0u) s_cnt[o][r][wave] = (uint32_t)__popcll(m);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0u) {  // 512 counters: exclusive scan in (octant, round, wavefront) order
    uint32_t run = 0u;
    for (int o = 0; o < 8; ++o)
      for (int r = 0; r < 4; ++r)
        for (int w = 0; w < 16; ++w) {
          s_base[o][r][w] = run;
          run += s_cnt[o][r][w];
        }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t s = block0 + (uint32_t)r * 1024u + threadIdx.x;
    if (oct[r] < 8u) order[fbase + block0 + s_base[oct[r]][r][wave] + rank[r]] = (uint32_t)fbase + s;
  }
}


// final_gather (path_tracer.cu:203-219) of the batch's staged samples into the accumulated framebuffers, in
// iteration order (running means do not commute)
__global__ __launch_bounds__(256) void k_accumulate(DFrame stage, DFrame fb, uint32_t pix_count, DBatchInfo bi)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= pix_count) return;
  // The running means of a pixel stay in registers over the frames of the batch (one read and one write of the
  // framebuffer per batch; the same operations in the same order as a fold frame by frame), and the staged samples are
  // requested eight frames at a time: one after the other, a thread of a 32-frame batch sat through 32 dependent round
  // trips and the kernel moved 2.3 TB/s.
  const uint32_t first = bi.iteration[0];
  float4 col = make_float4(0.f, 0.f, 0.f, 0.f), nd = make_float4(0.f, 0.f, 0.f, 0.f);
  if (first != 0u) {
    col = ldnt(&fb.color4[i]);
    nd = ldnt(&fb.nd4[i]);
  }
  constexpr uint32_t kAhead = 8u;
  for (uint32_t f0 = 0; f0 < bi.count; f0 += kAhead) {
    float4 c[kAhead], g[kAhead];
#pragma unroll
    for (uint32_t k = 0; k < kAhead; ++k) {
      const uint32_t f = min(f0 + k, bi.count - 1u);
      c[k] = ldnt(&stage.color4[(size_t)f * bi.stride + i]);
      g[k] = ldnt(&stage.nd4[(size_t)f * bi.stride + i]);
    }
#pragma unroll
    for (uint32_t k = 0; k < kAhead; ++k) {
      if (f0 + k >= bi.count) break;
      const uint32_t it = bi.iteration[f0 + k];
      col.x = running_mean(it, col.x, c[k].x);
      col.y = running_mean(it, col.y, c[k].y);
      col.z = running_mean(it, col.z, c[k].z);
      nd.x = running_mean(it, nd.x, g[k].x);
      nd.y = running_mean(it, nd.y, g[k].y);
      nd.z = running_mean(it, nd.z, g[k].z);
      nd.w = running_mean(it, nd.w, g[k].w);
    }
  }
  col.w = 0.0f;
  stnt(&fb.color4[i], col);
  stnt(&fb.nd4[i], nd);
}


// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }

void launch_raygen(hipStream_t s, const DCameras& cams, const DBatchInfo& bi, DBand band, uint32_t pix_count,
                   DPaths paths, DeviceCounters* counters, const DObject* objects, uint32_t filt_begin, uint32_t filt_end,
                   uint32_t* worklist, DHits hits, unsigned long long* tile_desc, uint32_t tile_stride, uint32_t epoch,
                   bool finish_misses, DFrame fb, bool staged)
{
  const dim3 grid(div_up(pix_count, 256u * kListPer) * bi.count), block(256);
  const DTileScan scan{tile_desc, epoch};
  if (worklist && filt_begin < filt_end && tile_desc) {
    if (finish_misses)
      hipLaunchKernelGGL((k_raygen<true, true>), grid, block, 0, s, cams, bi, band, pix_count, paths, counters, objects, filt_begin,
                         filt_end, worklist, hits, scan, tile_stride, fb, staged ? 1 : 0);
    else
      hipLaunchKernelGGL((k_raygen<true, false>), grid, block, 0, s, cams, bi, band, pix_count, paths, counters, objects, filt_begin,
                         filt_end, worklist, hits, scan, tile_stride, fb, 0);
  } else {
    hipLaunchKernelGGL((k_raygen<false, false>), grid, block, 0, s, cams, bi, band, pix_count, paths, counters, objects, 0u, 0u, worklist,
                       hits, scan, tile_stride, fb, 0);
  }
}
void launch_raygen_next(hipStream_t s, const DScene& scene, const DCameras& cams, const DBatchInfo& bi, DBand band, uint32_t pix_count, DPaths paths,
                        DeviceCounters* counters, const DNextRun& next)
{
  const dim3 grid(div_up(pix_count, 256u * kListPer) * bi.count), block(256);
  hipLaunchKernelGGL(k_raygen_next, grid, block, 0, s, scene, cams, bi, band, pix_count, paths, counters, next);
}
void launch_spheres(hipStream_t s, const DScene& scene, uint32_t obj_begin, uint32_t obj_end, bool first, DPaths paths, DHits hits,
                    uint32_t max_paths, int bounce, DeviceCounters* counters, const DBatchInfo& bi, uint32_t filt_begin,
                    uint32_t filt_end, uint32_t* worklist, unsigned long long* tile_desc, uint32_t tile_stride, uint32_t epoch)
{
  const dim3 grid(div_up(max_paths, 256u * kListPer) * bi.count), block(256);
  const DTileScan scan{tile_desc, epoch};
#define PT_SPHERES(FIRST, FILTER)                                                                                              \
  hipLaunchKernelGGL((k_spheres<FIRST, FILTER>), grid, block, 0, s, scene, obj_begin, obj_end, paths, hits, bounce, counters, bi, \
                     filt_begin, filt_end, worklist, scan, tile_stride)
  if (worklist && filt_begin < filt_end && tile_desc) {
    if (first) PT_SPHERES(true, true);
    else PT_SPHERES(false, true);
  } else {
    if (first) PT_SPHERES(true, false);
    else PT_SPHERES(false, false);
  }
#undef PT_SPHERES
}
void launch_list_flags(hipStream_t s, const uint8_t* flags, uint32_t max_paths, int bounce, DeviceCounters* counters, const DBatchInfo& bi,
                       uint32_t* worklist, unsigned long long* tile_desc, uint32_t tile_stride, uint32_t epoch)
{
  const dim3 grid(div_up(max_paths, kFlagsTile) * bi.count), block(256);
  const DTileScan scan{tile_desc, epoch};
  hipLaunchKernelGGL(k_list_flags, grid, block, 0, s, flags, bounce, counters, bi, worklist, scan, tile_stride);
}
void launch_tail_count(hipStream_t s, const DScene& scene, uint32_t obj_begin, uint32_t obj_end, bool first, DPaths paths,
                       DHits hits, uint32_t max_paths, int bounce, uint32_t* chunk_counts, DeviceCounters* counters,
                       const DBatchInfo& bi)
{
  const dim3 grid(div_up(max_paths, 256u), bi.count), block(256);
  if (obj_begin < obj_end) {
    if (first) hipLaunchKernelGGL((k_tail_count<true, true>), grid, block, 0, s, scene, obj_begin, obj_end, paths, hits, bounce, chunk_counts, counters, bi);
    else hipLaunchKernelGGL((k_tail_count<true, false>), grid, block, 0, s, scene, obj_begin, obj_end, paths, hits, bounce, chunk_counts, counters, bi);
  } else {
    // (nothing to test: some closest-hit launch has written every record of the bounce)
    hipLaunchKernelGGL((k_tail_count<false, false>), grid, block, 0, s, scene, 0u, 0u, paths, hits, bounce, chunk_counts, counters, bi);
  }
}
void launch_scan(hipStream_t s, int bounce, bool last_bounce, const uint32_t* chunk_counts, uint32_t* chunk_offsets,
                 DeviceCounters* counters, const DBatchInfo& bi)
{
  hipLaunchKernelGGL(k_scan, dim3(bi.count), dim3(1024), 0, s, bounce, last_bounce ? 1 : 0, chunk_counts, chunk_offsets,
                     counters, bi);
}
void launch_shade(hipStream_t s, const DScene& scene, DPaths in, DPaths out, DHits hits, uint32_t max_paths,
                  bool staged, int bounce, bool last_bounce, const uint32_t* slot_base,
                  const uint32_t* chunk_offsets, DFrame fb, DBand band, DeviceCounters* counters, uint8_t* octs,
                  const DBatchInfo& bi)
{
  hipLaunchKernelGGL(k_shade, dim3(div_up(max_paths, 256u), bi.count), dim3(256), 0, s, scene, in, out, hits,
                     staged ? 1 : 0, bounce, last_bounce ? 1 : 0, slot_base, chunk_offsets, fb, band, counters, octs, bi);
}
void launch_shade_fused(hipStream_t s, const DScene& scene, uint32_t obj_begin, uint32_t obj_end, bool first, DPaths in, DPaths out,
                        DHits hits, uint32_t max_paths, bool staged, int bounce, bool last_bounce, const uint32_t* slot_base,
                        unsigned long long* tile_desc, uint32_t tile_stride, uint32_t epoch, DFrame fb, DBand band,
                        DeviceCounters* counters, uint8_t* octs, const DBatchInfo& bi, const uint32_t* list, const DNextRun* next)
{
  const dim3 grid(div_up(max_paths, kFuseTile) * bi.count), block(256);
  const DNextRun none{};
#define PT_FUSED(SPH, FIRST, NEXT)                                                                                                 \
  hipLaunchKernelGGL((k_shade_fused<SPH, FIRST, NEXT>), grid, block, 0, s, scene, obj_begin, obj_end, in, out, hits, staged ? 1 : 0, \
                     bounce, last_bounce ? 1 : 0, slot_base, tile_desc, tile_stride, epoch, fb, band, counters, octs, bi, list, next ? *next : none)
  if (next && !first && !last_bounce) {  // "prefold": the next bounce's leading sphere run rides along
    if (obj_begin < obj_end) PT_FUSED(true, false, true);
    else PT_FUSED(false, false, true);
  } else if (obj_begin < obj_end) {
    if (first) PT_FUSED(true, true, false);
    else PT_FUSED(true, false, false);
  } else if (first) {
    PT_FUSED(false, true, false);   // a scene without objects: every ray misses
  } else {
    PT_FUSED(false, false, false);  // (some closest-hit launch has written every record of the bounce)
  }
#undef PT_FUSED
}
uint32_t shade_tiles_per_frame(uint32_t max_paths) { return div_up(max_paths, kFuseTile); }
void launch_sort_octant(hipStream_t s, const uint8_t* octs, uint32_t* order, uint32_t max_paths, int bounce,
                        DeviceCounters* counters, const DBatchInfo& bi)
{
  hipLaunchKernelGGL(k_sort_octant, dim3(div_up(max_paths, kSortBlock), bi.count), dim3(1024), 0, s, octs, order, bounce, counters, bi);
}
void launch_accumulate(hipStream_t s, DFrame stage, DFrame fb, uint32_t pix_count, const DBatchInfo& bi)
{
  hipLaunchKernelGGL(k_accumulate, dim3(div_up(pix_count, 256u)), dim3(256), 0, s, stage, fb, pix_count, bi);
}
}  // namespace pt
