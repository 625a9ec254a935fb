// pt_kernels.hip -- gfx950 kernels of the render core (wave64, LDS traversal stacks, ballot compaction).
//
// Arithmetic contract: every float operation on the path (ray generation, intersection, shading) is a
// single IEEE binary32 operation in a fixed order (built with -ffp-contract=off and correctly rounded
// divide/sqrt), because the reference's streaming mode seeds its RNG from the COMPACTED slot index
// (path_tracer.cu:297-301): one hit/miss decision that differs moves every later path to another slot.
// The parity tests therefore compare with the CPU oracle bit for bit.
//
// Frame pipeline (streaming mode, PathTracer::path_trace path_tracer.cu:413-471):
//   raygen                                 generate_rays / raygen_kernel   (ray_gen.cu:11-32)
//   per bounce b:
//     trace   closest hit per live path    intersection_kernel             (path_tracer.cu:271-290)
//             + per-wavefront live count (ballot/popcount)
//     scan    exclusive scan of the per-wavefront counts -> compaction offsets, live[b+1]
//     shade   material + sky + G-buffer    material_kernel                 (path_tracer.cu:292-315)
//             fused with the stable compaction scatter   thrust::stable_partition (path_tracer.cu:454-457)
//             and with the final gather of paths that end here  final_gathering_kernel (path_tracer.cu:317-330)
// A path that ends (miss, or the bounce cap) is accumulated into the framebuffer at once; only live
// paths are kept, in their original order, so slot indices equal the reference's.



// This unit: the closest-hit stage (k_trace, k_trace_wide, k_beam, k_traverse4, k_megakernel, k_intersect); the launch over a run of
// instances (k_traverse4m): pt_traverse_run.hip; the bounce-spanning persistent launch (k_persist): pt_persist.hip -- all three
// over pt_walk.inc.
// Ray generation and the end of a bounce: pt_shade.hip.  Views, multi-GPU gather, denoiser: pt_post.hip.

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

// intersection_kernel, path_tracer.cu:271-290.  One wavefront per 64-path chunk.
template <bool kCount>
__global__ __launch_bounds__(kWave) void k_trace(DScene sc, DPaths paths, DHits hits, int bounce, DeviceCounters* counters)
{
  __shared__ uint32_t s_stack[kStackDepth * kWave];
  const uint32_t n = counters->live[bounce];
  const uint32_t s = blockIdx.x * kWave + threadIdx.x;
  if (blockIdx.x * kWave >= n) return;
  bool hit = false;
  uint32_t flags = 0u;
  Tally tally;
  if (s < n) {
    const Ray ray = load_ray(paths, s);
    Hit rec;
    rec.t = 0.0f;
    rec.p = rec.n = mk3(0.f, 0.f, 0.f);
    rec.mat = 0u;
    rec.side = 0u;
    hit = ray_scene<kCount>(ray, sc, rec, s_stack + threadIdx.x, flags, tally);
    stnt(&hits.tp[s], make_float4(hit ? rec.t : -1.0f, rec.p.x, rec.p.y, rec.p.z));
    stnt(&hits.nm[s], make_float4(rec.n.x, rec.n.y, rec.n.z, __uint_as_float(rec.mat | (rec.side << 31))));
    if (flags) atomicOr(&counters->flags, flags);
  }
  if (kCount) flush_tally(tally, counters, bounce, true);
}

// The same kernel over the wide layout (default).  Chunks are dealt to workgroups so that workgroups that
// share an XCD (blockIdx % 8, MI355X_MICROARCH.md "Workgroup dispatch") get one contiguous run of
// chunks = one contiguous image region: neighbouring paths walk the same subtrees, which keeps that
// XCD's 4 MiB L2 on one part of the BVH.  Placement only affects speed.
template <bool kCount>
__global__ __launch_bounds__(kWave) void k_trace_wide(DScene sc, DPaths paths, DHits hits, int bounce, DeviceCounters* counters)
{
  __shared__ uint32_t s_stack[kWideStack * kWave];
  const uint32_t n = counters->live[bounce];
  const uint32_t chunks = (n + kChunk - 1u) / kChunk;
  const uint32_t per_xcd = (chunks + 7u) / 8u;
  const uint32_t chunk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per_xcd || chunk >= chunks) return;
  const uint32_t s = chunk * kWave + threadIdx.x;
  bool hit = false;
  uint32_t flags = 0u;
  Tally tally;
  if (s < n) {
    const Ray ray = load_ray(paths, s);
    Hit rec;
    rec.t = 0.0f;
    rec.p = rec.n = mk3(0.f, 0.f, 0.f);
    rec.mat = 0u;
    rec.side = 0u;
    hit = ray_scene_wide<kCount>(ray, sc, rec, s_stack + threadIdx.x, flags, tally);
    stnt(&hits.tp[s], make_float4(hit ? rec.t : -1.0f, rec.p.x, rec.p.y, rec.p.z));
    stnt(&hits.nm[s], make_float4(rec.n.x, rec.n.y, rec.n.z, __uint_as_float(rec.mat | (rec.side << 31))));
    if (flags) atomicOr(&counters->flags, flags);
  }
  if (kCount) flush_tally(tally, counters, bounce, true);
}

#ifdef PT_TAILPROF
__device__ unsigned long long g_tailprof[16][8192][8];
extern "C" int ptc_debug_tailprof(void* dst, size_t bytes)
{
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (!dst) {  // clear
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_tailprof)) != hipSuccess) return -1;
    return hipMemset(p, 0, sizeof(g_tailprof)) == hipSuccess && hipDeviceSynchronize() == hipSuccess ? 0 : -1;
  }
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tailprof), bytes < sizeof(g_tailprof) ? bytes : sizeof(g_tailprof)) == hipSuccess ? 0 : -1;
}
#endif

#include "pt_walk.inc"

// Entry points for primary rays (DBeam), one thread per tile and camera.  The tile's frustum: four planes through the
// camera, each spanned by two neighbouring corner rays of the tile's pixel rectangle (generate_ray at the rectangle's
// corners, a twentieth of a pixel outside: the jitter keeps a ray of pixel x inside [x, x + 1]); everything in the space
// of the launch's object (the walk's space).  A box is out when its corner farthest along a plane's normal is still
// outside that plane by more than the margin; the pyramid is the forward one only, so what lies behind the camera is out
// -- the reference's line-without-range box test would pass such boxes, but no triangle in them can be hit at t >= t_min.
// Frontier: the root; as long as there is room for them, the largest inner entry is replaced by those of its (quantised)
// children the frustum reaches.  Quantised boxes contain the exact ones and are handed on a millionth larger: the rays
// test them with the walk's own tolerant slabs, and the winner is verified exactly (finalize) as for any other walk.
__global__ __launch_bounds__(64) void k_beam(DScene sc, uint32_t obj_index, DCameras cams, DBeam geo, uint32_t nbeam, uint32_t node_count4, float4* out)
{
  const uint32_t id = blockIdx.x * 64u + threadIdx.x;
  if (id >= nbeam * geo.tiles) return;
  const uint32_t beam = id / geo.tiles, tile = id - beam * geo.tiles;
  const uint32_t ty = tile / geo.tiles_x, tx = tile - ty * geo.tiles_x;
  const DCamera& cam = cams.c[geo.beam_of[beam]];  // (here: the camera OF the beam, filled in by launch_beam)
  const DObject* obj = sc.objects + obj_index;
  const float x0 = (float)(tx * kBeamTile) - 0.05f, x1 = (float)((tx + 1u) * kBeamTile) + 0.05f;
  const float y0 = (float)(ty * kBeamTile) - 0.05f, y1 = (float)((ty + 1u) * kBeamTile) + 0.05f;
  f3 o, d00, d10, d01, d11, dc;
  generate_ray(cam, x0, y0, o, d00);
  generate_ray(cam, x1, y0, o, d10);
  generate_ray(cam, x0, y1, o, d01);
  generate_ray(cam, x1, y1, o, d11);
  generate_ray(cam, 0.5f * (x0 + x1), 0.5f * (y0 + y1), o, dc);
  const beam_rules::Frustum fr = beam_rules::make_frustum(xform_point(obj->inv_m, o), xform_vector(obj->inv_m, d00), xform_vector(obj->inv_m, d10),
                                                          xform_vector(obj->inv_m, d01), xform_vector(obj->inv_m, d11), xform_vector(obj->inv_m, dc));
  f3 lo4[4], hi4[4];
  uint32_t ref4[4];
  int n = 0;
  if (sc.cur.bvh_node_count != 0u)
    n = beam_rules::tile_entries(reinterpret_cast<const uint32_t*>(sc.cur.bvh4q), node_count4, sc.cur.bvh4_root, ld3(sc.cur.root_min), ld3(sc.cur.root_max), fr,
                                 lo4, hi4, ref4);
  float4* e = out + (size_t)id * (2u * kBeamEntries);
#pragma unroll
  for (int k = 0; k < (int)kBeamEntries; ++k) {
    if (k < n) {
      e[2 * k] = make_float4(lo4[k].x, lo4[k].y, lo4[k].z, __uint_as_float(ref4[k]));
      e[2 * k + 1] = make_float4(hi4[k].x, hi4[k].y, hi4[k].z, 0.0f);
    } else {  // nothing: a box no ray is inside of
      e[2 * k] = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __uint_as_float(kNoChild));
      e[2 * k + 1] = make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), 0.0f);
    }
  }
}

template <bool kCount, bool kFirst, bool kBeam = false>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(PT_T4_WAVES, PT_T4_WAVES)))
void k_traverse4(DScene sc, uint32_t obj_index, DPaths paths, DHits hits, int bounce, int work_slot,
                 DeviceCounters* counters, uint32_t* slow_list, const uint32_t* order, DBatchInfo bi, int listed)
{
  traverse4_walk<kCount, kFirst, kBeam>(sc, obj_index, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed != 0);
  // Epilogue: every wavefront signs off; the last one redoes the rays that were set aside.  The list entries were
  // written with agent-scope atomic stores; waiting for this wavefront's own stores before the sign-off and reading
  // the list with agent-scope loads orders them without a full L2 write-back per wavefront.
  uint32_t prev = 0u;
  if (threadIdx.x == 0u) {
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): this wavefront's list entries have landed
    prev = __hip_atomic_fetch_add(&counters->waves_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  prev = (uint32_t)__builtin_amdgcn_readfirstlane((int)prev);
  if (prev + 1u != gridDim.x) return;
  const uint32_t count = __hip_atomic_load(&counters->slow_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (count != 0u) redo_slow_rays<kFirst>(sc, obj_index, paths, hits, slow_list, count, counters);
  launch_epilogue(counters, bounce, work_slot, count, bi, listed != 0);
}


// path_tracing_mega_kernel, path_tracer.cu:227-269: the whole path in one thread, one RNG stream per
// pixel (a different image from streaming mode at the same seed -- a property of the reference).
__global__ __launch_bounds__(kWave) void k_megakernel(DScene sc, DCamera cam, uint32_t iteration, DBand band,
                                                      uint32_t pix_count, int max_bounces, DFrame fb,
                                                      DeviceCounters* counters)
{
  __shared__ uint32_t s_stack[kStackDepth * kWave];
  const uint32_t s = blockIdx.x * kWave + threadIdx.x;
  uint32_t rays = 0u, flags = 0u;
  if (s < pix_count) {
    const uint32_t pixel = band_pixel(band, s);
    const uint32_t x = pixel % cam.width, y = pixel / cam.width;
    Minstd rng;
    rng.seed(path_seed(pixel, iteration));
    const float fx = (float)x + rng.uniform();
    const float fy = (float)y + rng.uniform();
    Ray ray;
    generate_ray(cam, fx, fy, ray.o, ray.d);
    ray.tmin = 1e-4f;
    ray.tmax = FLT_MAX;
    f3 color = mk3(1.0f, 1.0f, 1.0f);
    f3 normal = -ray.d;
    float depth = 1e6f;
    for (int i = 0; i < max_bounces; ++i) {
      Hit rec;
      rec.t = 0.0f;
      rec.p = rec.n = mk3(0.f, 0.f, 0.f);
      rec.mat = 0u;
      rec.side = 0u;
      ++rays;
      Tally tally;
      if (!ray_scene<false>(ray, sc, rec, s_stack + threadIdx.x, flags, tally)) {
        color = color * background(ray.d);
        break;
      }
      if (i == 0) {
        normal = rec.n;
        depth = rec.t;
      }
      bool tmin_flag = ray.tmin != 1e-4f;
      evaluate_material(ray.o, ray.d, tmin_flag, rec.p, rec.n, rec.side, sc.materials[rec.mat], rng, color);
      ray.tmin = tmin_flag ? 1e-5f : 1e-4f;
    }
    accumulate_color(fb.color4, s, iteration, color);
    accumulate_nd(fb.nd4, s, iteration, normal, depth);
    if (flags) atomicOr(&counters->flags, flags);
  }
  // one ray-counter atomic per wavefront
  uint32_t sum = rays;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  if (threadIdx.x == 0u && sum) atomicAdd(&counters->rays_total, (unsigned long long)sum);
}

// intersection_kernel on caller-supplied rays (parity tests): rays_o = origin.xyz,t_min ; rays_d = direction.xyz,t_max
template <int kVariant>
__global__ __launch_bounds__(kWave) void k_intersect(DScene sc, const float4* rays_o, const float4* rays_d, uint32_t n,
                                                     DHits hits, DeviceCounters* counters)
{
  __shared__ uint32_t s_stack[kStackDepth * kWave];
  const uint32_t s = blockIdx.x * kWave + threadIdx.x;
  if (s >= n) return;
  const float4 o = rays_o[s], d = rays_d[s];
  Ray ray;
  ray.o = xyz(o);
  ray.tmin = o.w;
  ray.d = xyz(d);
  ray.tmax = d.w;
  Hit rec;
  rec.t = 0.0f;
  rec.p = rec.n = mk3(0.f, 0.f, 0.f);
  rec.mat = 0u;
  rec.side = 0u;
  uint32_t flags = 0u;
  Tally tally;
  const bool hit = kVariant == 0 ? ray_scene<false>(ray, sc, rec, s_stack + threadIdx.x, flags, tally)
                                 : ray_scene_wide<false>(ray, sc, rec, s_stack + threadIdx.x, flags, tally);
  stnt(&hits.tp[s], make_float4(hit ? rec.t : -1.0f, rec.p.x, rec.p.y, rec.p.z));
  stnt(&hits.nm[s], make_float4(rec.n.x, rec.n.y, rec.n.z, __uint_as_float(rec.mat | (rec.side << 31))));
  if (flags) atomicOr(&counters->flags, flags);
}


// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }

void launch_trace(hipStream_t s, const DScene& scene, DPaths paths, DHits hits, uint32_t max_paths, int bounce,
                  DeviceCounters* counters, bool count_tests, int variant)
{
  if (variant == 1) {
    const dim3 grid(div_up(max_paths, kWave) + 8u);  // room for the per-XCD rounding of the chunk deal
    if (count_tests) hipLaunchKernelGGL(k_trace_wide<true>, grid, dim3(kWave), 0, s, scene, paths, hits, bounce, counters);
    else hipLaunchKernelGGL(k_trace_wide<false>, grid, dim3(kWave), 0, s, scene, paths, hits, bounce, counters);
    return;
  }
  if (count_tests) hipLaunchKernelGGL(k_trace<true>, dim3(div_up(max_paths, kWave)), dim3(kWave), 0, s, scene, paths, hits, bounce, counters);
  else hipLaunchKernelGGL(k_trace<false>, dim3(div_up(max_paths, kWave)), dim3(kWave), 0, s, scene, paths, hits, bounce, counters);
}
void launch_beam(hipStream_t s, const DScene& scene, uint32_t obj_index, const DCameras& cams, const uint8_t* cam_of, uint32_t nbeam,
                 uint32_t tiles_x, uint32_t tiles_y, uint32_t node_count4, float4* out)
{
  DBeam geo{};
  geo.tiles_x = tiles_x;
  geo.tiles = tiles_x * tiles_y;
  for (uint32_t b = 0; b < nbeam && b < 32u; ++b) geo.beam_of[b] = cam_of[b];
  const uint32_t threads = nbeam * geo.tiles;
  hipLaunchKernelGGL(k_beam, dim3((threads + 63u) / 64u), dim3(64), 0, s, scene, obj_index, cams, geo, nbeam, node_count4, out);
}
void launch_traverse(hipStream_t s, const DScene& scene, uint32_t obj_index, bool first, DPaths paths, DHits hits,
                     int bounce, int work_slot, DeviceCounters* counters, bool count_tests, uint32_t waves,
                     uint32_t* slow_list, const uint32_t* order, int variant, const DBatchInfo& bi, bool listed)
{
  const dim3 grid(waves), block(kWave);
  if (scene.beam.entries && first) {  // bounce 0's first launch: primary rays start at their tile's entry points
    if (count_tests) hipLaunchKernelGGL((k_traverse4<true, true, true>), grid, block, 0, s, scene, obj_index, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
    else hipLaunchKernelGGL((k_traverse4<false, true, true>), grid, block, 0, s, scene, obj_index, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
    return;
  }
  if (count_tests) {
    if (first) hipLaunchKernelGGL((k_traverse4<true, true>), grid, block, 0, s, scene, obj_index, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
    else hipLaunchKernelGGL((k_traverse4<true, false>), grid, block, 0, s, scene, obj_index, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
  } else {
    if (first) hipLaunchKernelGGL((k_traverse4<false, true>), grid, block, 0, s, scene, obj_index, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
    else hipLaunchKernelGGL((k_traverse4<false, false>), grid, block, 0, s, scene, obj_index, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
  }
}
void launch_megakernel(hipStream_t s, const DScene& scene, const DCamera& cam, uint32_t iteration, DBand band,
                       uint32_t pix_count, int max_bounces, DFrame fb, DeviceCounters* counters)
{
  hipLaunchKernelGGL(k_megakernel, dim3(div_up(pix_count, kWave)), dim3(kWave), 0, s, scene, cam, iteration, band,
                     pix_count, max_bounces, fb, counters);
}
void launch_intersect(hipStream_t s, const DScene& scene, const float4* rays_o, const float4* rays_d, uint32_t n,
                      DHits hits, DeviceCounters* counters, int variant)
{
  if (variant == 1)
    hipLaunchKernelGGL(k_intersect<1>, dim3(div_up(n, kWave)), dim3(kWave), 0, s, scene, rays_o, rays_d, n, hits, counters);
  else
    hipLaunchKernelGGL(k_intersect<0>, dim3(div_up(n, kWave)), dim3(kWave), 0, s, scene, rays_o, rays_d, n, hits, counters);
}
}  // namespace pt

