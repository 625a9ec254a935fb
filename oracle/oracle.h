/*
 * oracle.h -- CPU restatement ("oracle") of the hot path of LesleyLai/cuda-path-tracer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing outside tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may include, link, load or execute this code.  The
 * shipped product (libptcore.so) does not reference anything under oracle/.
 *
 * Pinning status (see oracle.c header for the details):
 *   - hash / minstd LCG / uniform_real mapping : PINNED (SURVEY KATs + rocThrust 7.2 host run,
 *                                                 tests/golden/rng_kat.json)
 *   - AABB / inverse_transform_ray             : PINNED by the reference's own Catch2 values
 *                                                 (test/aabb_test.cpp, test/transform_test.cpp)
 *   - intersections, BVH, materials, streaming loop, denoiser, tonemap:
 *                                                 PARITY UNPINNED (the reference has no tests,
 *                                                 golden images or runnable build for them)
 *
 * Plain C99, IEEE binary32 everywhere, built with -ffp-contract=off.
 * All reference citations are path:line relative to /root/reference.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } ovec3;
typedef struct { float c[4][4]; } omat4; /* c[column][row], column-major like glm */

/* src/lib/ray.hpp:8-20 (32 B) */
typedef struct { ovec3 origin; float t_min; ovec3 direction; float t_max; } ORay;

/* src/lib/intersection.hpp:6-14 (48 B) */
typedef struct {
  float t;
  ovec3 point;
  ovec3 normal;
  size_t material_id;
  uint8_t side; /* 0 front, 1 back */
} OIntersection;

/* src/lib/aabb.hpp:16-18 */
typedef struct { ovec3 min, max; } OAABB;

/* src/lib/scene.hpp:14-22 (160 B) */
typedef struct {
  uint32_t type;  /* 0 sphere, 1 mesh */
  uint32_t index;
  omat4 m;
  omat4 inv_m;
  OAABB aabb;
} OObject;

/* src/lib/sphere.hpp:8-11 */
typedef struct { ovec3 center; float radius; } OSphere;

/* src/lib/material.hpp:19-38 (20 B): type 0 diffuse {albedo}, 1 metal {albedo,fuzz}, 2 dielectric {ior} */
typedef struct { int32_t type; float p[4]; } OMaterial;

/* src/lib/accelerators/bvh.hpp:17-28 (32 B) */
typedef struct { OAABB aabb; uint32_t first_child_or_primitive; uint32_t primitive_count; } OBVHNode;

/* The flat arrays SceneDescription::build_scene uploads (scene_description.cpp:12-117) */
struct OMeshRange;
typedef struct {
  const OObject* objects;
  uint32_t object_count;
  const uint32_t* object_material_indices;
  const OSphere* spheres;
  uint32_t sphere_count;
  const OMaterial* materials;
  uint32_t material_count;
  const float* positions; /* 3 floats per vertex */
  uint32_t vertex_count;
  const uint32_t* indices;
  uint32_t index_count;
  const OBVHNode* bvh;
  uint32_t bvh_node_count;
  /* Optional mesh table (NULL / 0 = the reference: ONE mesh per scene, every mesh object instantiates it whatever its
   * `index` says, scene_description.cpp:42,95).  With a table, a mesh object's `index` names its mesh: ranges into the
   * concatenated positions / indices / bvh arrays, each mesh as bvh_from_mesh sees it on its own (vertex indices and
   * child indices relative to the mesh's own first vertex / first node).  This extends the reference's rule, it does
   * not restate anything the reference does: results with a table are PARITY UNPINNED by construction. */
  const struct OMeshRange* meshes;
  uint32_t mesh_count;
} OScene;
struct OMeshRange { uint32_t first_vertex, vertex_count, first_index, index_count, first_bvh_node, bvh_node_count; };

/* src/lib/camera.hpp:17-23 */
typedef struct { float position[3]; float rotation_wxyz[4]; float vfov; } OCamera;

/* src/lib/camera.hpp:10-15 */
typedef struct { omat4 camera_matrix; float vfov; uint32_t width, height; } OGPUCamera;

/* ---- scalar pieces (KAT surface) ---- */
uint32_t orc_hash(uint32_t a);                                  /* hash.cuh:4-14 */
uint32_t orc_rng_seed(uint32_t s);                              /* thrust minstd_rand::seed */
uint32_t orc_rng_next(uint32_t* state);                         /* x <- 48271 x mod (2^31-1) */
void     orc_rng_discard(uint32_t* state, uint64_t z);          /* thrust discard */
float    orc_rng_uniform(uint32_t* state);                      /* uniform_real_distribution<float>(0,1) */
uint32_t orc_path_seed(uint32_t index, uint64_t iteration);     /* hash(hash(index) ^ iteration) */
void     orc_sincos(float x, float* s, float* c);               /* deterministic sinf/cosf (see oracle.c) */

void orc_to_gpu_camera(const OCamera* cam, uint32_t w, uint32_t h, OGPUCamera* out); /* camera.cpp:5-13 */
void orc_generate_ray(const OGPUCamera* cam, float x, float y, ORay* out);           /* ray_gen.cu:34-61 */

int  orc_ray_sphere(const ORay* ray, const OSphere* s, OIntersection* rec);          /* intersections.cuh:7-41 */
int  orc_ray_triangle(const ORay* ray, const float* p0, const float* p1, const float* p2,
                      OIntersection* rec);                                           /* intersections.cuh:49-85 */
int  orc_ray_aabb(const ORay* ray, const OAABB* box);                                /* intersections.cuh:87-103 */
void orc_inverse_transform_ray(const omat4* m, const omat4* inv_m, const ORay* ray, ORay* out); /* transform.hpp:51-58 */
void orc_transform_aabb(const omat4* m, const OAABB* in, OAABB* out);                /* transform.hpp:69-88 */
void orc_mat4_inverse(const omat4* m, omat4* out);                                   /* glm::inverse restated */
/* scene_description.cpp:17-52: fills inv_m and the world AABB of one object.
 * sphere may be NULL for meshes; mesh_aabb may be NULL for spheres. */
void orc_make_object(uint32_t type, uint32_t index, const omat4* m, const OSphere* sphere,
                     const OAABB* mesh_aabb, OObject* out);

float    orc_aabb_surface_area(const OAABB* b);                 /* aabb.hpp:62-68 */
void     orc_aabb_extent(const OAABB* b, float* out3);          /* aabb.hpp:41-44 */
int      orc_aabb_max_extent(const OAABB* b);                   /* aabb.hpp:51-55 */
void     orc_aabb_offset(const OAABB* b, const float* p, float* out); /* aabb.hpp:73-80 */

/* accelerators/bvh.cpp:211-253.  nodes must hold 2*T-1 entries.  Returns node count, or <0:
 * -1 empty mesh (bvh.cpp:200), -2 empty SAH side (bvh.cpp:84). max_depth (may be NULL) gets the leaf depth. */
int orc_bvh_build(const float* positions, uint32_t vertex_count, const uint32_t* indices,
                  uint32_t index_count, OBVHNode* nodes, uint32_t* max_depth);

/* path_tracer.cu:110-128, one closest-hit query.  Returns hit flag. */
int orc_scene_intersect(const OScene* scene, const ORay* ray, OIntersection* rec);
/* Batch form used by per-ray parity tests: hit_t[i] = t or -1. */
void orc_intersect_rays(const OScene* scene, const ORay* rays, uint32_t n, OIntersection* recs,
                        uint8_t* hit);

/* Streaming mode, PathTracer::path_trace (path_tracer.cu:413-471), iterations
 * [iter_begin, iter_begin+iter_count).  fb_* are vec3/vec3/float framebuffers (running means; must
 * hold the previous state when iter_begin > 0).  live_counts: [iter_count][max_bounces] uint32, the
 * number of live paths ENTERING each bounce (0 after the loop broke).  nthreads<=0 -> all cores.
 * Returns the total number of rays (closest-hit queries). */
uint64_t orc_render_streaming(const OScene* scene, const OCamera* cam, uint32_t w, uint32_t h,
                              uint32_t iter_begin, uint32_t iter_count, uint32_t max_bounces,
                              float* fb_color, float* fb_normal, float* fb_depth,
                              uint32_t* live_counts, int nthreads);

/* One ROW BAND [row0,row1) of one iteration of the streaming mode, for the multi-GPU partition (no
 * reference equivalent; the reference is single-GPU).  The reference seeds the material RNG from the GLOBAL
 * compacted slot index; a band reproduces the full-frame result iff, at every bounce, it offsets its local
 * slot indices by the number of live paths in all lower bands.  exchange(user, bounce, my_live) must return
 * that number (bounce 0: row0*w).  exchange == NULL: base = 0 from bounce 1 on ("band-local" numbering).
 * fb_* hold the band's rows only.  Returns the band's ray count. */
typedef uint32_t (*orc_exchange_fn)(void* user, uint32_t bounce, uint32_t my_live);
uint64_t orc_render_streaming_band(const OScene* scene, const OCamera* cam, uint32_t w, uint32_t h, uint32_t row0,
                                   uint32_t row1, uint32_t iteration, uint32_t max_bounces, float* fb_color,
                                   float* fb_normal, float* fb_depth, uint32_t* live_counts,
                                   orc_exchange_fn exchange, void* user, int nthreads);

/* The rows of ONE RANK under interleaved row blocks, the split bench.py --gpus N and hip_pt --gpus N run (no reference
 * equivalent): blocks of block_rows rows dealt round-robin to nranks ranks, paths numbered per rank, the material RNG
 * (path_tracer.cu:297-301) keyed on slot_offset + the rank's compacted slot index at every bounce; ray generation keyed
 * on the frame's pixel index as always.  Same semantics as ptc_set_interleave + "slot_offset".  fb_* hold the rank's
 * rows packed in frame order (orc_interleaved_rows() of them) and the running means over iterations
 * [iter_begin, iter_begin + iter_count); live_counts as in orc_render_streaming.  A different noise realisation from the
 * single-GPU frame by construction (rank 0 with nranks 1 and slot_offset 0 IS the single-GPU frame). */
uint32_t orc_interleaved_rows(uint32_t h, uint32_t rank, uint32_t nranks, uint32_t block_rows);
uint64_t orc_render_streaming_interleaved(const OScene* scene, const OCamera* cam, uint32_t w, uint32_t h, uint32_t rank,
                                          uint32_t nranks, uint32_t block_rows, uint32_t slot_offset, uint32_t iter_begin,
                                          uint32_t iter_count, uint32_t max_bounces, float* fb_color, float* fb_normal,
                                          float* fb_depth, uint32_t* live_counts, int nthreads);

/* Megakernel mode (path_tracer.cu:227-269). */
uint64_t orc_render_megakernel(const OScene* scene, const OCamera* cam, uint32_t w, uint32_t h,
                               uint32_t iter_begin, uint32_t iter_count, uint32_t max_bounces,
                               float* fb_color, float* fb_normal, float* fb_depth, int nthreads);

/* denoising/edge_avoiding_a_trous_denoiser.cu:24-115.  buf_a/buf_b are the two ping-pong
 * buffers (dev_denoised_buffer_, dev_denoised_buffer2_).  Returns 0 if the result is in buf_a,
 * 1 if in buf_b, -1 if no pass ran.  Taps whose clamped index falls outside the W*H array (the
 * reference reads out of bounds there) read index clamped to W*H-1; `touched_oob` (may be NULL,
 * W*H bytes) is set to 1 for every pixel whose value depends on such a tap in any pass. */
int orc_denoise(const OCamera* cam, uint32_t w, uint32_t h, const float* color, const float* normal,
                const float* depth, float* buf_a, float* buf_b, int filter_size, float c_phi,
                float n_phi, float p_phi, uint8_t* touched_oob, int nthreads);

/* preview_kernel / preview_depth_kernel (path_tracer.cu:334-385). mode: 0 none, 1 normal (n*.5+.5), 2 depth */
void orc_preview(const float* buffer, uint32_t w, uint32_t h, int mode, uint8_t* rgba);

int orc_hardware_threads(void);

#ifdef __cplusplus
}
#endif
#endif
