/*
 * ptcore.h -- C ABI of libptcore.so, the MI355X (gfx950) path-tracing core.
 *
 * This is the drop-in boundary for the render core of LesleyLai/cuda-path-tracer: one `ptc_ctx`
 * replaces one `class PathTracer` (reference src/lib/path_tracer.hpp:60-99) together with the
 * device-side `Scene` it owns (src/lib/scene.hpp:25-67).  Each entry point names the reference
 * interface it replaces.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - every function returns PTC_OK (0) or a negative ptc_status; nothing calls exit()
 *     (the reference's CUDA_CHECK / panic abort the process: cuda_utils/cuda_check.cpp:7-23,
 *     prelude.cpp:5-10);  ptc_last_error() gives the message of the last failure.
 *   - one context = one GPU = one host thread at a time (same as the reference, which is
 *     single-threaded on the default stream).
 *   - matrices are column-major float[16] (glm layout): m[4*col + row].
 *   - framebuffers are row-major, row 0 = top of the view, flat index = x + y*width
 *     (cuda_utils/indices.cuh:20-26).
 *   - there is NO CPU fallback: every call that needs the GPU fails with PTC_ERR_NO_DEVICE /
 *     PTC_ERR_HIP when no gfx950 device is usable.
 */
#ifndef PTCORE_H
#define PTCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTC_ABI_VERSION 3

typedef enum ptc_status {
  PTC_OK = 0,
  PTC_ERR_INVALID = -1,       /* bad argument / bad call order */
  PTC_ERR_NO_DEVICE = -2,     /* no usable HIP device */
  PTC_ERR_HIP = -3,           /* a HIP runtime call or kernel failed */
  PTC_ERR_OOM = -4,
  PTC_ERR_BVH = -5,           /* BVH build failed (empty SAH side: bvh.cpp:84-85) */
  PTC_ERR_STACK = -6,         /* traversal stack overflow (reference: UB beyond depth 24, static_stack.hpp:21-25) */
  PTC_ERR_NO_SCENE = -7
} ptc_status;

/* GPUMethod, path_tracer.hpp:57 */
typedef enum ptc_method { PTC_METHOD_MEGAKERNEL = 0, PTC_METHOD_STREAMING = 1 } ptc_method;

/* DisplayBufferType, path_tracer.hpp:19 */
typedef enum ptc_display { PTC_DISPLAY_FINAL = 0, PTC_DISPLAY_COLOR = 1, PTC_DISPLAY_NORMAL = 2, PTC_DISPLAY_DEPTH = 3 } ptc_display;

/* which accumulated framebuffer ptc_download reads */
typedef enum ptc_buffer { PTC_BUF_COLOR = 0, PTC_BUF_NORMAL = 1, PTC_BUF_DEPTH = 2, PTC_BUF_FINAL = 3 } ptc_buffer;

/* ObjectType + GPUObject, scene.hpp:14-22.  Same 160-byte layout as the reference. */
typedef struct ptc_object {
  uint32_t type;      /* 0 sphere, 1 mesh */
  uint32_t index;     /* sphere index; 0 for meshes (scene_description.cpp:42) */
  float m[16];        /* Transform::m_ */
  float inv_m[16];    /* Transform::inverse_m_ */
  float aabb_min[3];  /* world-space AABB */
  float aabb_max[3];
} ptc_object;

/* Sphere, sphere.hpp:8-11 */
typedef struct ptc_sphere { float center[3]; float radius; } ptc_sphere;

/* Material, material.hpp:19-38 (20 bytes).  type 0 Diffuse {albedo rgb}, 1 Metal {albedo rgb, fuzz},
 * 2 Dielectric {refraction_index} */
typedef struct ptc_material { int32_t type; float p[4]; } ptc_material;

/* BVHNode, accelerators/bvh.hpp:17-28 (32 bytes).  leaf <=> primitive_count != 0; for a leaf
 * first_child_or_primitive is the offset of its triangle in the index array (multiple of 3); for an
 * inner node it is the left child, the right child is +1. */
typedef struct ptc_bvh_node {
  float aabb_min[3];
  float aabb_max[3];
  uint32_t first_child_or_primitive;
  uint32_t primitive_count;
} ptc_bvh_node;

/* One mesh of a scene that has several (an extension: the reference keeps ONE mesh per scene whatever the scene
 * file holds, scene_description.cpp:42,95 -- SURVEY section 8 f2).  Ranges into ptc_scene_desc's positions / indices /
 * bvh arrays; the indices of a mesh count from ITS first vertex, its BVH nodes from ITS first node (each mesh as
 * bvh_from_mesh would return it on its own). */
typedef struct ptc_mesh_range {
  uint32_t first_vertex, vertex_count;
  uint32_t first_index, index_count;
  uint32_t first_bvh_node, bvh_node_count; /* bvh_node_count 0: built by ptc_upload_scene */
} ptc_mesh_range;

/* The flat arrays SceneDescription::build_scene() uploads with six cudaMemcpy calls
 * (scene_description.cpp:54-114).  Host pointers; copied during ptc_upload_scene. */
typedef struct ptc_scene_desc {
  const ptc_object* objects;
  uint32_t object_count;
  const uint32_t* object_material_indices; /* object_count entries */
  const ptc_sphere* spheres;
  uint32_t sphere_count;
  const ptc_material* materials;
  uint32_t material_count;
  const float* positions;                  /* 3 floats per vertex */
  uint32_t vertex_count;
  const uint32_t* indices;                 /* 3 per triangle */
  uint32_t index_count;
  const ptc_bvh_node* bvh;                 /* optional: NULL -> built by ptc_upload_scene */
  uint32_t bvh_node_count;
  /* optional mesh table (NULL: the arrays above are the scene's one mesh, every mesh object instantiates it and
   * ptc_object::index is ignored, like the reference).  With a table, a mesh object's `index` names its mesh. */
  const ptc_mesh_range* meshes;
  uint32_t mesh_count;
} ptc_scene_desc;

/* Camera, camera.hpp:17-23 */
typedef struct ptc_camera {
  float position[3];
  float rotation_wxyz[4]; /* glm::quat, default {1,0,0,0} */
  float vfov;             /* radians */
} ptc_camera;

/* EdgeAvoidingATrousDenoiser's public fields, denoising/edge_avoiding_a_trous_denoiser.hpp:9-12 */
typedef struct ptc_denoiser_params {
  int32_t filter_size;   /* default 10 */
  float color_weight;    /* default 0.45 */
  float normal_weight;   /* default 0.30 */
  float position_weight; /* default 0.25 */
} ptc_denoiser_params;

typedef struct ptc_config {
  int32_t device;       /* HIP device ordinal (the reference never selects one: cli.cpp:71-78) */
  int32_t max_bounces;  /* reference: compile-time 50 (path_tracer.cu:27); <=0 -> 50 */
  int32_t method;       /* ptc_method; default streaming (path_tracer.hpp:64) */
  int32_t reserved;
} ptc_config;

#define PTC_MAX_BOUNCES_CAP 64

typedef struct ptc_stats {
  uint64_t rays_total;                       /* closest-hit queries since ptc_restart / create */
  uint64_t frames;                           /* ptc_trace calls that rendered since then */
  uint32_t last_live[PTC_MAX_BOUNCES_CAP];   /* live paths entering each bounce of the last streaming frame */
  uint32_t bvh_node_count;
  uint32_t bvh_max_depth;
  uint32_t triangle_count;
  uint32_t stack_capacity;                   /* traversal stack entries available per ray */
} ptc_stats;

/* Measurement support (no reference equivalent; the reference only has a wall-clock Stopwatch, cli.cpp:27-60).
 * Per bounce index, summed over every streaming frame traced since ptc_reset_profile:
 *   paths      live paths that entered the bounce (= closest-hit queries = rays)
 *   node_visits BVH node records fetched by the closest-hit kernel (counting runs; one four-child 64-byte record
 *              in the default kernel, one two-child record in variant 1, one 32-byte node in variant 0)
 *   trace_ms   duration of the closest-hit kernel, from HIP events recorded on the context's stream
 *              around each launch (only while events are enabled)
 *   box_tests  ray/AABB tests of BVH nodes, tri_tests  ray/triangle tests (only while counting is
 *              enabled: an instrumented, slower kernel variant -- never time it) */
typedef struct ptc_profile {
  uint64_t paths[PTC_MAX_BOUNCES_CAP];
  uint64_t box_tests[PTC_MAX_BOUNCES_CAP];
  uint64_t tri_tests[PTC_MAX_BOUNCES_CAP];
  double trace_ms[PTC_MAX_BOUNCES_CAP];
  uint32_t trace_launches[PTC_MAX_BOUNCES_CAP];
  uint32_t max_box_tests[PTC_MAX_BOUNCES_CAP]; /* longest single traversal seen (counting runs) */
  uint64_t listed_rays[PTC_MAX_BOUNCES_CAP]; /* rays the traversal launches fetched through a work list ("filter_rays"); 0: they walked all live rays */
  uint64_t slow_rays[PTC_MAX_BOUNCES_CAP];   /* rays redone with exact box decisions at the end of a traversal launch (always counted) */
  uint64_t node_visits[PTC_MAX_BOUNCES_CAP]; /* BVH node records fetched (counting runs): SURVEY 8(d)'s N_node */
  double denoise_ms;                         /* summed duration of the A-Trous passes (HIP events, while events are enabled) */
  uint32_t denoise_passes;
  uint32_t persist_launches;                 /* batches whose bounces >= 1 and shade passes ran as ONE persistent launch ("persist", k_persist) */
} ptc_profile;

typedef struct ptc_ctx ptc_ctx;

/* ---- lifetime ---- */
int ptc_abi_version(void);
int ptc_device_count(int* count);                                   /* cli.cpp:72-73 cudaGetDeviceCount */
int ptc_create(const ptc_config* config, ptc_ctx** out);            /* PathTracer::PathTracer(), path_tracer.cu:387 */
void ptc_destroy(ptc_ctx* ctx);                                     /* ~PathTracer (cuda::Buffer dtors, cuda_buffer.hpp:20) */
const char* ptc_last_error(const ptc_ctx* ctx);                     /* ctx may be NULL: last ptc_create failure */

/* Use an externally owned HIP stream (hipStream_t as void*), e.g. torch's current stream, for all
 * kernels of this context.  NULL restores the context's own stream. */
int ptc_set_stream(ptc_ctx* ctx, void* hip_stream);

/* ---- scene + buffers ---- */
/* Upload half of PathTracer::create_buffers (path_tracer.cu:559-564) = SceneDescription::build_scene's
 * device uploads.  An empty mesh (index_count == 0) is accepted (the reference panics, bvh.cpp:200). */
int ptc_upload_scene(ptc_ctx* ctx, const ptc_scene_desc* scene);
/* PathTracer::resize_image (path_tracer.cu:527-545): (re)allocates all per-pixel buffers, restarts. */
int ptc_resize(ptc_ctx* ctx, uint32_t width, uint32_t height);
/* Render only image rows [row_begin,row_end) of the width x height frame (multi-GPU row bands;
 * no reference equivalent).  Must follow ptc_resize.  Default: all rows. */
int ptc_set_rows(ptc_ctx* ctx, uint32_t row_begin, uint32_t row_end);
/* Load-balanced multi-GPU partition: the frame is cut into blocks of block_rows rows dealt round-robin to
 * nranks contexts; this context renders the blocks rank, rank+nranks, ...  Its framebuffers (ptc_download,
 * ptc_present_rgba8) hold those rows packed in order.  Slot numbering is per context ("local"), so the noise
 * pattern differs from the single-context image; contiguous bands (ptc_set_rows) keep the exact option. */
int ptc_set_interleave(ptc_ctx* ctx, uint32_t rank, uint32_t nranks, uint32_t block_rows);
int ptc_restart(ptc_ctx* ctx);                                      /* PathTracer::restart, path_tracer.cu:522-525 */
int ptc_iteration(const ptc_ctx* ctx);                              /* PathTracer::iteration(), path_tracer.hpp:88 */
int ptc_set_iteration(ptc_ctx* ctx, int iteration);                 /* test hook: continue from a given sample index */

/* public mutable fields of PathTracer (path_tracer.hpp:62-66) */
int ptc_set_max_iterations(ptc_ctx* ctx, int max_iterations);       /* PathTracer::max_iterations (default 1) */
int ptc_set_method(ptc_ctx* ctx, int method);                       /* PathTracer::current_gpu_method */
int ptc_set_max_bounces(ptc_ctx* ctx, int max_bounces);             /* static max_bounces = 50, path_tracer.cu:27 */
int ptc_set_denoiser_params(ptc_ctx* ctx, const ptc_denoiser_params* p); /* PathTracer::atrous_denoiser */
/* Closest-hit kernel variant (speed only; all return the same hits, bit for bit -- same box decisions, same
 * tie rule; 0 and 1 exist to cross-check the default on the GPU):
 *   3 (default) = persistent wavefronts whose lanes fetch the next ray as soon as their own is finished, over the
 *                 tree collapsed to four children per 64-byte quantised node; box decisions only conservative, the
 *                 winning triangle re-checked against its parent's box with the reference's arithmetic
 *                 (sufficient: see DESIGN.md "nesting"); objects walked as sphere / mesh segments; the only
 *                 variant that traces several iterations per launch ("batch_frames")
 *   1           = culled near-first traversal with exact box decisions, one wavefront per 64 fixed paths (the
 *                 walk the default uses for the few rays it sets aside)
 *   0           = traversal in the reference's own order (path_tracer.cu:36-76: depth-first, left first, no
 *                 t culling) */
int ptc_set_trace_variant(ptc_ctx* ctx, int variant);
/* Knobs by name.  Three kinds (round-3 review: "they are ABI now -- mark which are stable"):
 *   STABLE   part of the interface: meaning and default are kept, an application may rely on them
 *              frames_in_flight, batch_frames, slot_offset, traverse_waves
 *   SCHEDULE choose between implementations that return the SAME bits; kept for A/B measurements and as cross-checks of
 *            each other, defaults may move with the hardware, a name may go when its alternative goes
 *              fused_shade, filter_rays, merge_instances, sphere_lanes, sphere_fold, beam, ray_sort, denoise_variant, bvh_build_on_device,
 *              layout_on_device, split_idle, refill_lanes, static_eighths, small_waves, small_rays_per_lane, min_waves
 *   TEST     hooks for the parity tests only: debug_lds_entries, debug_force_slow
 * None of them changes a result, with one exception that is the point of it: slot_offset keys the material RNG
 * (multi-GPU).  Unknown names return PTC_ERR_INVALID.  Known names:
 *   "frames_in_flight" consecutive iterations in flight at once, folded into the framebuffer in iteration order
 *                      (default: 64, fewer when their path state would exceed 24 GiB; 1 = strictly serial on
 *                      the context's stream; 1..256; set before ptc_resize)
 *   "batch_frames"     iterations traced by the same launches (default 32; 1..32; before ptc_resize).  ptc_trace
 *                      queues an iteration and enqueues the batch when it is full or when any other call
 *                      looks at the context; frames_in_flight / batch_frames batches run on separate streams.
 *                      The library asks the HIP runtime for 24 hardware queues (GPU_MAX_HW_QUEUES, default 4:
 *                      streams on one queue serialise) when it is loaded before the runtime starts; an
 *                      application that initialises HIP first should export GPU_MAX_HW_QUEUES=24 itself
 *   "merge_instances"  1 (default): consecutive mesh objects that instantiate the same mesh are walked by one traversal
 *                      launch per bounce (a lane keeps its ray and takes the instances in turn); 0: one launch each
 *   "bvh_build_on_device"  1 (default): a scene without BVH gets the reference BVH from the GPU builder
 *                      (ptc_build_bvh_device); 0: from the threaded host builder.  Same nodes either way
 *   "layout_on_device" 1 (default): the traversal layouts (collapsed four-wide tree, leaf order, per-instance
 *                      triangle records) are derived on the GPU; 0: on the host.  Same bytes either way
 *                      (ptc_download_layout); a caller's BVH that is not numbered depth by depth goes to the host
 *   "fused_shade"      1 (default): the end of a bounce -- trailing sphere run, material, stable compaction, final gather -- is
 *                      ONE kernel (tiles by ticket, decoupled look-back; a bounded wait reports PTC_ERR_HIP from ptc_get_stats
 *                      instead of ever hanging); 0: three kernels (count, scan, shade)
 *   "filter_rays"      1 (default): the kernel in front of a mesh launch (a sphere run; ray generation at bounce 0) also lists the
 *                      rays that may hit one of the launch's world boxes at all -- in slot order, by the look-back scan of
 *                      "fused_shade" -- and the launch walks only those (ptc_profile::listed_rays).  When bounce 0's launch
 *                      covers the scene's whole mesh part (only the sphere run that ends the object list follows), a primary ray
 *                      that is not listed hits nothing: its path ends in ray generation and the bounce's shade kernel walks the
 *                      list.  Images, live counts and ray totals are those of 0: every live ray is fetched by the launch
 *   "traverse_waves"   most persistent wavefronts a traversal launch may use (default 5120 = the number that is
 *                      resident at 5 per SIMD; before ptc_upload_scene).  A launch uses one wavefront per 3072
 *                      primary rays it carries, at least 1024
 *   A batch of a single iteration (an interactive front-end that presents or denoises after every iteration, the
 *   stepwise calls) runs in one of eight extra one-frame slots with streams of their own, so that viewer-style
 *   use keeps eight frames in flight as well (config 5, 1 spp + denoise per 1080p frame: 1.6 ms)
 *   "ray_sort"         1: from the second bounce on, the persistent traversal lanes pick their rays up grouped by
 *                      direction octant within blocks of 4096 neighbouring slots (an index array built by a small
 *                      kernel per bounce; the slots themselves, and so the random numbers, stay where they are).
 *                      Default 0: measured on the benchmark scene it does not pay for itself.  Before
 *                      ptc_resize
 *   "denoise_variant"  0 (default): the A-Trous passes stage their taps in LDS, one sub-lattice of the dilated filter per
 *                      workgroup; 1: taps through L1 / L2 (round-1 kernel, kept as a cross-check).  Same results up to
 *                      summation order (both within 1e-5 of the oracle)
 *   "slot_offset"      added to every compacted slot index before the material RNG is seeded (path_tracer.cu:300).  A
 *                      rank of a multi-GPU run that numbers its paths locally (ptc_set_interleave) sets rank * (pixels of
 *                      the largest share) so that no two ranks draw the same random streams; 0 (default) = the reference
 *   "beam"             1 (default): when a bounce-0 traversal launch walks one mesh object, a small kernel first computes, per
 *                      8 x 8-pixel tile and distinct camera of the batch, the deepest nodes of that object's tree the tile's
 *                      frustum overlaps (at most four, with their boxes), and the primary rays start there instead of at the
 *                      root; 0: at the root.  Before ptc_resize
 *   "sphere_lanes"     1 (default): in the kernel that ends a bounce, a sphere run that ends the object list and consists of
 *                      at most eight spheres whose matrices are pure translations is tested candidate by candidate, every
 *                      lane with the spheres IT cannot rule out ("select approximately, verify exactly": approximate bounds
 *                      on the root the reference would accept, then the reference's own sequence for each candidate);
 *                      0: object by object for the whole wavefront.  Any time
 *   "sphere_fold"      1 (default): a sphere run IN FRONT of a mesh whose objects are all translated spheres (a room's
 *                      walls) is walked with the matrix products that are sums with zeros written as those sums, and the hit
 *                      record's normal finished once, for the hit that remains, instead of for every accepted hit on the way
 *                      (same operations on the same operands; a wavefront with a ray on one of the exceptions -- a -0.0f
 *                      coordinate against a zero translation, a zero direction component, anything non-finite -- takes the
 *                      plain form); 0: the plain form.  Any time
 *   "min_waves"        fewest persistent wavefronts of a traversal launch (default 1024)
 *   "split_idle"       once a traversal launch has handed out its last ray: idle lanes of a persistent wavefront that
 *                      trigger work splitting (an idle lane takes over the bottom of a busy lane's traversal stack
 *                      with a copy of its ray; default 8, 0 = never).  Cuts the latency tail of every launch
 *   "refill_lanes"     idle lanes of a persistent wavefront that trigger the next ray fetch (default 32; 20 until round 4)
 *   "static_eighths"   share of a launch's rays dealt to the wavefronts statically (default 4 = 4/8; 3 until round 4)
 *   "small_waves", "small_rays_per_lane"  a traversal launch with fewer than small_rays_per_lane (default 4) rays per lane of
 *                      "traverse_waves" wavefronts uses at most small_waves (default 3072) of them
 *   "run_waves"        SCHEDULE (round 5; default 3072): most wavefronts of a launch that walks a run of instances of one mesh
 *                      (k_traverse4m) when the context has two or more batch slots: the other batch's HBM-bound kernels get on the
 *                      chip beside it (config 2: +6 % against 5120)
 *   "persist"          SCHEDULE (round 5; default 0): 1 = a batch of at least "persist_min_frames" (2) frames whose launch plan is one
 *                      mesh object per bounce with nothing in front of it (fused shade, staged samples, no ray sorting, not instrumented)
 *                      runs bounce 0's traversal as ever and then ONE launch for the traversal of bounces >= 1 and every shade pass
 *                      (k_persist: one wavefront in "persist_service_every" (5) shades tiles, the others walk; a walking wavefront
 *                      without rays shades "persist_help_tiles" (8) tiles before it looks again).  Bit-identical (tests/test_gpu_persist.py),
 *                      measured 0.58 x the per-bounce launches: off.  Its waits are bounded: a launch that gives up
 *                      makes ptc_get_stats fail with PTC_ERR_HIP instead of hanging.  Any time (queued frames are flushed first)
 *   "prefold"          SCHEDULE (round 5; default 1): when a bounce opens with a sphere run in front of a mesh launch that lists its rays (a
 *                      room's walls), the kernel that ends the bounce BEFORE walks that run for its survivors -- and ray generation for
 *                      the primary rays -- into a second set of hit records, and the bounce starts with the work list (k_list_flags)
 *                      instead of k_spheres.  Bit-identical; config 2 +8-9 %.  0: every bounce runs k_spheres.  Before ptc_resize
 *                      (33 more bytes per pixel and frame in flight)
 *   "pair_batches"     SCHEDULE (round 5; default 0): 1 = a full batch is held until the next one is full (or anything else looks at
 *                      the context); the two are enqueued bounce by bounce on two slots and their traversal launches take turns.
 *                      Bit-identical, measured 6-7 % slower than the default (profiles/r05_pair_batches.txt): off.  Any time
 *   "debug_lds_entries" test hook: keep only this many of the 24 per-lane traversal stack entries in LDS, so that small
 *                      scenes exercise the global overflow area (1..24; before ptc_upload_scene)
 *   "debug_force_slow" test hook: route every ray through the exact redo at the end of the traversal launch */
int ptc_set_param(ptc_ctx* ctx, const char* name, int value);

/* ---- the hot path ---- */
/* PathTracer::path_trace (path_tracer.cu:389-477): one sample per pixel, accumulated as a running
 * mean; no-op once iteration() >= max_iterations.  Asynchronous on the context's stream. */
int ptc_trace(ptc_ctx* ctx, const ptc_camera* camera);

/* The same frame in three steps, for callers that must exchange live-path counts between bounces
 * (multi-GPU row bands keep the reference's global slot numbering this way):
 *   ptc_trace_begin   = generate_rays (ray_gen.cu:63-79)
 *   ptc_trace_bounce  = intersection_kernel + material_kernel + stable_partition for bounce b
 *                       (path_tracer.cu:423-457); slot_base_dev (device pointer to ONE uint32, or
 *                       NULL for 0) is added to every local slot index before RNG seeding
 *   ptc_trace_end     = final_gathering_kernel (path_tracer.cu:460-470) + ++iteration
 * ptc_live_count_dev returns the device address of the uint32 holding the number of live paths
 * entering bounce b of the current frame (valid after bounce b-1 was enqueued). */
int ptc_trace_begin(ptc_ctx* ctx, const ptc_camera* camera);
int ptc_trace_bounce(ptc_ctx* ctx, int bounce, const uint32_t* slot_base_dev);
int ptc_trace_end(ptc_ctx* ctx);
int ptc_live_count_dev(ptc_ctx* ctx, int bounce, const uint32_t** dev_ptr);
/* Enqueue a copy of that counter into a caller-owned device uint32, e.g. a torch tensor that is then all-gathered
 * over RCCL.  Only inside ptc_trace_begin .. ptc_trace_end.  Ordering: the copy runs on the frame's own stream and
 * the context's stream (ptc_set_stream, e.g. torch's current stream) is made to wait for it, so work the caller
 * enqueues there afterwards sees the value; ptc_trace_bounce in turn makes the frame's stream wait for everything
 * enqueued on the context's stream so far before it reads slot_base_dev.  No host synchronisation. */
int ptc_copy_live_count(ptc_ctx* ctx, int bounce, void* dst_dev);
/* The same value on the host (synchronises the frame's stream): the per-bounce read-back the reference does
 * after every thrust::stable_partition (path_tracer.cu:457). */
int ptc_read_live_count(ptc_ctx* ctx, int bounce, uint32_t* host_out);

/* PathTracer::denoise (path_tracer.cu:479-485) */
int ptc_denoise(ptc_ctx* ctx);

/* PathTracer::send_to_preview (path_tracer.cu:487-520): tonemap to RGBA8 (4 bytes/pixel, rows of
 * this context only).  dst is a device pointer if dst_is_device != 0 (the reference always writes a
 * device/managed pointer), else host memory.  Synchronises like the reference. */
int ptc_present_rgba8(ptc_ctx* ctx, void* dst, int dst_is_device, int display_type);

/* Copy an accumulated float framebuffer (rows of this context only) as packed floats: 3 per pixel
 * for COLOR/NORMAL/FINAL (glm::vec3 layout of dev_color_buffer_ etc., path_tracer.hpp:73-81),
 * 1 per pixel for DEPTH.  Synchronises. */
int ptc_download(ptc_ctx* ctx, int which, void* dst, int dst_is_device);

/* ---- several GPUs: one process and one context per GPU, the frame's rows dealt to the ranks ---- */
/* (No reference equivalent: the reference is single-GPU.  SURVEY section 8e: the scene is replicated, every rank
 * traces its rows (ptc_set_interleave / ptc_set_rows) without talking to anyone, and radiance moves only at present
 * time, from every rank straight to the root over xGMI.)
 * The transport is HIP inter-process memory: a rank exports the device buffer that holds its packed rows
 * (ptc_band_export, once), the root maps the peers' buffers (ptc_band_import, once; the handle's geometry is checked
 * against the root's frame) and, whenever a frame is presented (ptc_gather_frame / ptc_gather_present_rgba8), ONE
 * kernel on the root reads every band where it lies -- the peers' over xGMI, all links at once -- and writes it into
 * row order; nothing is staged.  ptc_gather_last_us reports how long that launch took on the root.  The handles
 * and the "my rows are ready" signal travel over whatever channel the application has between its processes --
 * shared memory in hip_pt --gpus N (host/main.cpp), torch.distributed in bench.py and the tests:
 *     every rank but the root:  ptc_band_publish(ctx, which);   then signal / barrier
 *     the root, after that:     ptc_gather_frame(ctx, which, dst, dst_is_device)
 * Works the same when the ranks share one GPU (how it is tested on a one-GPU box). */
typedef struct ptc_band_handle {
  uint8_t ipc_mem[64];     /* hipIpcMemHandle_t of the rank's band buffer (3 floats per pixel of the band) */
  uint32_t pix_count;      /* pixels the rank renders */
  uint32_t pix_begin;      /* DBand of the rank: where its packed pixel s sits in the frame */
  uint32_t width;
  uint32_t rank, nranks, block_rows;
  uint32_t reserved[2];
} ptc_band_handle;
int ptc_band_export(ptc_ctx* ctx, ptc_band_handle* out);
int ptc_band_import(ptc_ctx* root, uint32_t rank, const ptc_band_handle* handle);
/* Pack this rank's rows of buffer `which` (ptc_buffer) into its exported band buffer and wait until they are there. */
int ptc_band_publish(ptc_ctx* ctx, int which);
/* Root: its own rows + the rows every imported rank has published -> the whole frame in row order, packed like
 * ptc_download (3 floats per pixel, 1 for PTC_BUF_DEPTH).  dst holds width*height pixels.  Synchronises. */
int ptc_gather_frame(ptc_ctx* root, int which, void* dst, int dst_is_device);
/* Root: PathTracer::send_to_preview of the gathered frame (width*height RGBA8).  display_type as ptc_present_rgba8;
 * PTC_DISPLAY_FINAL gathers the accumulated colour (a denoised buffer exists only for a context that owns the whole
 * frame). */
int ptc_gather_present_rgba8(ptc_ctx* root, void* dst, int dst_is_device, int display_type);
/* Root: device time of the most recent gather (the root's own pack excluded: from the first to the last band landing
 * in row order), in microseconds.  Waits for that gather. */
int ptc_gather_last_us(ptc_ctx* root, float* microseconds);

int ptc_synchronize(ptc_ctx* ctx);                                  /* cudaDeviceSynchronize at cli.cpp:100 */
int ptc_get_stats(ptc_ctx* ctx, ptc_stats* out);                    /* synchronises */
int ptc_set_profiling(ptc_ctx* ctx, int time_trace_kernel, int count_tests);
int ptc_reset_profile(ptc_ctx* ctx);
int ptc_get_profile(ptc_ctx* ctx, ptc_profile* out);                /* synchronises */

/* ---- pieces exposed for parity tests and for host-side callers ---- */
/* intersection_kernel alone (path_tracer.cu:271-290) on caller-supplied rays (host arrays of
 * 8 floats: origin, t_min, direction, t_max = struct Ray, ray.hpp:8-20).  Outputs (host):
 * hit_t[n] (t, or -1 on miss), hit_normal[3n], hit_material[n], hit_side[n]. */
int ptc_intersect_rays(ptc_ctx* ctx, const float* rays, uint32_t n, float* hit_t, float* hit_normal,
                       uint32_t* hit_material, uint8_t* hit_side);

/* Where the time of the last ptc_upload_scene went (milliseconds of host wall clock; the "Initialization" stage of
 * the reference's Stopwatch, cli.cpp): */
typedef struct ptc_upload_times {
  float bvh_build_ms;   /* the reference BVH (0 when the caller brought one) */
  float layout_ms;      /* collapsed / quantised traversal trees derived from it */
  float triangles_ms;   /* per-instance world-space triangle records */
  float copy_ms;        /* hipMalloc + host-to-device copies (includes packing the reference nodes) */
  float total_ms;
  uint32_t bvh_on_device; /* 1: the reference BVH was built by the GPU builder */
  uint32_t layout_on_device; /* 1: the traversal trees and triangle records were derived on the GPU */
} ptc_upload_times;
int ptc_get_upload_times(const ptc_ctx* ctx, ptc_upload_times* out);
/* The device-resident traversal data of the uploaded scene, for inspection and tests: which = 0 four-wide nodes
 * (64 B each), 1 parent boxes per triangle rank, 2 per-instance triangle records (64 B each), 3 two-child records, 4 the reference
 * nodes as two float4.  *bytes (may be NULL) gets the size; host may be NULL to ask for the size only. */
int ptc_download_layout(ptc_ctx* ctx, int which, void* host, uint64_t capacity, uint64_t* bytes);

/* bvh_from_mesh (accelerators/bvh.cpp:211-253), host-side, no GPU needed.  nodes must hold
 * index_count/3*2-1 entries.  Returns the node count (>0) or a negative ptc_status. */
int ptc_build_bvh(const float* positions, uint32_t vertex_count, const uint32_t* indices,
                  uint32_t index_count, ptc_bvh_node* nodes, uint32_t* max_depth);
/* The same tree built by the context's GPU (pt_bvh_gpu.hip: all nodes of a depth split at once; the decisions are
 * the host builder's, from one source) -- node for node what ptc_build_bvh returns.  This is what ptc_upload_scene
 * runs when the scene brings no BVH (param "bvh_build_on_device", default 1; 0 = the host builder). */
int ptc_build_bvh_device(ptc_ctx* ctx, const float* positions, uint32_t vertex_count, const uint32_t* indices,
                         uint32_t index_count, ptc_bvh_node* nodes, uint32_t* max_depth);

/* Per-object part of SceneDescription::build_scene (scene_description.cpp:17-52): fills inv_m
 * (glm::inverse) and the world AABB.  sphere / mesh_aabb6 (min xyz, max xyz) as the type needs. */
int ptc_make_object(uint32_t type, uint32_t index, const float* m16, const ptc_sphere* sphere,
                    const float* mesh_aabb6, ptc_object* out);

/* Host-side check of the traversal layouts derived from a reference BVH (no GPU needed): builds the collapsed
 * four-wide tree with its 64-byte quantised nodes and verifies, in double precision, that every quantised child
 * box contains the exact box of the reference node it stands for, that every reference leaf is reachable exactly
 * once, and that each triangle's recorded parent box is its reference parent's.  Returns the number of
 * violations (0 = sound), or a negative ptc_status; *checked_boxes (may be NULL) gets the number of child boxes
 * looked at. */
int ptc_check_traversal_layout(const ptc_bvh_node* nodes, uint32_t node_count, uint64_t* checked_boxes);
/* The ray feed of the persistent traversal launches, checked on the host (no GPU; test hook): a frame of n rays dealt in
 * eight regions, static_eighths / 8 of every region as static batches of 64, the rest as dynamic batches of dyn_batch (64
 * or 128) rays, with the region geometry the kernels use (csrc/pt_feed_rules.hpp).  Returns the number of rays handed out
 * twice or never and of batches that are not contiguous or leave the frame (0 = sound), or a negative ptc_status. */
int ptc_check_feed(uint32_t n, uint32_t static_eighths, uint32_t dyn_batch);
/* Entry points for primary rays ("beam"), checked on the host (no GPU): for every 8 x 8-pixel tile of a width x height
 * frame of `camera`, the entries k_beam computes for the mesh under the object matrix object_m16 (NULL: identity;
 * column-major), and for sample rays of the tile (corners and centre of the jitter range of every stride-th pixel) the
 * closest hit of a walk from the tree's root against the same walk from the tile's entries.  Returns the number of rays
 * that disagree (0 = sound) or a negative status; stats5 (may be NULL): tiles, tiles without entries, entries, rays, hits;
 * entries_out (may be NULL): the entries themselves, [tiles][4][8 floats] = {box min, reference bits} {box max, 0}. */
int ptc_check_beam(const float* positions, uint32_t vertex_count, const uint32_t* indices, uint32_t index_count, const float* object_m16,
                   const ptc_camera* camera, uint32_t width, uint32_t height, uint32_t stride, uint64_t* stats5, float* entries_out);
/* Test hook: the same entries as k_beam computes them on the GPU for the uploaded scene's first traversal launch. */
int ptc_debug_beam_entries(ptc_ctx* ctx, const ptc_camera* camera, float* entries_out, uint64_t capacity_floats);
/* TEST / diagnostic: the state block (DPersist, pt_device.hpp) of slot `slot`'s persistent launch ("persist"), copied after a
 * device synchronisation; its per-phase counters are filled in by -DPT_PERSIST_DEBUG builds only (tools/debug/persist_diff.py).
 * No reference equivalent. */
int ptc_debug_persist(ptc_ctx* ctx, int slot, void* dst, uint64_t bytes);

/* Device self-test of the arithmetic contract the parity tests rely on: evaluates IEEE divide,
 * sqrt and the deterministic sin/cos on the GPU for n inputs (host arrays in, host arrays out). */
int ptc_selftest_math(ptc_ctx* ctx, const float* a, const float* b, uint32_t n, float* out_div,
                      float* out_sqrt, float* out_sin, float* out_cos);

#ifdef __cplusplus
}
#endif
#endif /* PTCORE_H */
