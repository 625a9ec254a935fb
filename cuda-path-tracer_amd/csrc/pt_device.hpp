// pt_device.hpp -- device-visible types of the render core and the launch interface between the
// C ABI (ptcore.cpp) and the HIP kernels (pt_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/ptcore.h"
#include "pt_math.hpp"

namespace pt {

constexpr int kWave = 64;            // gfx950 wavefront
constexpr int kChunk = 64;           // paths per compaction chunk = one wavefront
constexpr int kStackDepth = 64;      // traversal stack entries per ray (LDS, [depth][lane])
constexpr int kMaxBounces = 64;      // == PTC_MAX_BOUNCES_CAP
// Ray-fetch cursor sets of the persistent traversal launches.  k_raygen zeroes them at the start of a frame and the
// last wavefront of every traversal launch zeroes the set it used (launch_epilogue), so launch n of a slot's stream
// takes set n % kWorkSlots whatever the number of launches per frame (objects x bounces is unbounded).
constexpr int kWorkSlots = 4;
// Traversal stack entries per lane the four-wide walk keeps in LDS (6 KiB per wavefront at 24); deeper entries go to
// DScene::spill.  ONE constant for the kernels' LDS arrays and the host's DScene::lds_cap / spill sizing: a build with
// -DPT_T4_LDS=16 against a host that assumed 24 wrote past its stack (round-2 A/B fault).
#ifndef PT_T4_LDS
#define PT_T4_LDS 24
#endif
constexpr int kLds4 = PT_T4_LDS;
static_assert(kLds4 >= 4 && kLds4 <= kStackDepth, "PT_T4_LDS out of range");

// error bits in DeviceCounters::flags
constexpr uint32_t kFlagStackOverflow = 1u;
constexpr uint32_t kFlagDispatchOrder = 2u;  // k_shade_fused gave up waiting for a predecessor tile (bounded look-back wait)
constexpr uint32_t kFlagPersistStall = 4u;   // k_persist: a wavefront waited for work (or for the redo lock) beyond the bound and gave up

// Same 160-byte layout as ptc_object / the reference's GPUObject (scene.hpp:16-22)
struct DObject {
  uint32_t type;
  uint32_t index;
  m4 m;
  m4 inv_m;
  float bmin[3];
  float bmax[3];
};
static_assert(sizeof(DObject) == 160, "object layout");

struct DMaterial {
  int32_t type;
  float p[4];
};
static_assert(sizeof(DMaterial) == 20, "material layout");

// Read-only scene in HBM.
//   bvh: two float4 per node: {min.xyz, bits(first_child_or_primitive)}, {max.xyz, bits(primitive_count)}
//   wide: the same tree re-laid for the fast traversal (k_trace_wide): one 64-byte record per INNER node
//         holding both children's boxes,
//           w0 = {lmin.xyz, lmax.x}  w1 = {lmax.yz, rmin.xy}  w2 = {rmin.z, rmax.xyz}  w3 = {bits lref, bits rref, -, -}
//         ref = index of the child's record, or kLeafBit | rank of the child's triangle in depth-first
//         (left-first) leaf order -- the order in which the reference's traversal reaches the leaves.
//   tris: per MESH OBJECT (instance), world-space triangles in that depth-first order, kTriVec4 (= 4) float4 each,
//         three of them used:
//           {p0.xyz, e1.x} {e1.yz, e2.xy} {e2.z, n.xyz}   with p = transform_point(M, position),
//           e1 = p1 - p0, e2 = p2 - p0, n = normalize(cross(e1, e2))   (exactly what the reference
//           recomputes per test, path_tracer.cu:57-59, intersections.cuh:45-46,54-55)
//   bvh4q: the same tree collapsed to four children per node for the persistent traversal, 64 bytes per node
//         (half a cache line, four 16-byte loads): dwords 0-2 origin xyz (f32); dword 3 and dwords 10, 11 = the three
//         power-of-two grid steps x, y, z as floats; dwords 4-9 = child planes as 8-bit grid
//         coordinates, component-major (lo_x[4] lo_y[4] lo_z[4] hi_x[4] hi_y[4] hi_z[4], child c in byte c), rounded
//         outwards: a quantised child box contains the exact one, which is all a conservative walk needs;
//         dwords 12-15 = refs[4].  ref = node index or kLeafBit | depth-first triangle rank; an unused slot has an
//         inside-out box (lower planes 255, upper planes 0) and refers to the dummy triangle (DScene::dummy_ref).
//         Nodes in depth-first preorder.
//   leaf_parent: per triangle (depth-first rank) the box of the leaf's parent in the REFERENCE tree, 2 float4.
//         Why it suffices for exactness: boxes nest exactly (parent = componentwise min/max of children) and
//         IEEE subtraction/division are monotonic, so whenever a node passes the reference's box test all its
//         ancestors pass too; a triangle is reachable in the reference iff its parent's box passes.
// float4s per triangle record of DScene::tris: the 48 bytes described above padded to 64, so that a record never
// straddles two 64-byte segments of a cache line (measured +2 % rays/s against 48-byte records; 3 = unpadded)
#ifndef PT_TRI_VEC4
#define PT_TRI_VEC4 4
#endif
constexpr uint32_t kTriVec4 = PT_TRI_VEC4;
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kNoChild = 0xffffffffu;

// The acceleration data of ONE mesh (reference tree + the traversal layouts derived from it).  DScene::cur is the
// mesh of the object a traversal launch walks (the host sets it per launch); kernels that loop over all objects of
// the scene look the mesh up per object (DScene::mesh_views[DScene::object_mesh[i]]).
struct DMeshView {
  const float* positions;
  const uint32_t* indices;
  const float4* bvh;
  const float4* wide;              // 4 float4 per inner node
  const float4* leaf_parent;       // 2 float4 per triangle
  const uint4* bvh4q;              // the four-wide nodes in 64 bytes (Wide4Accel::nodes_q), 4 x uint4 per node
  uint32_t bvh4_root;
  uint32_t dummy_ref;              // reference in the unused child slots of a four-wide node (an inside-out box, a triangle no ray hits)
  float root_min[3];               // box of the root (tested before descending, like any inner node)
  float root_max[3];
  uint32_t root_ref;               // record 0, or kLeafBit for a single-triangle mesh
  uint32_t bvh_node_count;
};

// Which pixels of the frame this context renders.  Contiguous (nranks == 1): pixel = pix_begin + s.
// Interleaved (multi-GPU, load-balanced): the frame is cut into blocks of `block_rows` rows dealt round-robin to
// `nranks` contexts; local row lr is row ((lr / block_rows) * nranks + rank) * block_rows + lr % block_rows.
struct DBand {
  uint32_t pix_begin, width, rank, nranks, block_rows;
};
PT_HD uint32_t band_pixel(const DBand& b, uint32_t s)
{
  if (b.nranks <= 1u) return b.pix_begin + s;
  const uint32_t lr = s / b.width, x = s - lr * b.width;
  const uint32_t y = ((lr / b.block_rows) * b.nranks + b.rank) * b.block_rows + lr % b.block_rows;
  return y * b.width + x;
}
PT_HD uint32_t band_local(const DBand& b, uint32_t pixel)
{
  if (b.nranks <= 1u) return pixel - b.pix_begin;
  const uint32_t y = pixel / b.width, x = pixel - y * b.width;
  const uint32_t lr = ((y / b.block_rows) / b.nranks) * b.block_rows + y % b.block_rows;
  return lr * b.width + x;
}

// Entry points for primary rays ("beam", round 4): per distinct camera of a batch and per tile of kBeamTile x kBeamTile
// pixels, the (at most four) deepest nodes of the launch's four-wide tree that the tile's frustum overlaps, each with its
// box in the object's space: 2 float4 per entry {box min, reference bits} {box max, 0}, 4 entries per tile (unused:
// an inside-out box).  A primary ray tests the four boxes of its tile like the children of one node and starts its walk
// there instead of at the root (k_beam writes them, k_traverse4<.., kBeam> reads them).
constexpr uint32_t kBeamTile = 8u;
constexpr uint32_t kBeamEntries = 4u;
struct DBeam {
  const float4* entries;        // null: off
  uint32_t tiles_x, tiles;      // tiles per row, tiles per camera
  uint32_t width;               // of the frame, pixels
  DBand band;                   // which pixel a slot of bounce 0 holds (band_pixel)
  uint8_t beam_of[32];          // frame of the batch -> camera of the batch (frames with the same camera share one)
};
constexpr uint32_t kSphereTab = 8u;  // float4 per object in DScene::sphere_ball
struct DScene {
  const DObject* objects;
  const uint32_t* object_material;
  const float4* spheres;  // center.xyz, radius
  const DMaterial* materials;
  DMeshView cur;                   // the mesh of this launch's object (mesh 0 unless the host says otherwise)
  const DMeshView* mesh_views;     // all meshes of the scene ...
  const uint32_t* object_mesh;     // ... and which one each object instantiates (0 for spheres)
  const float4* tris;              // kTriVec4 float4 per instance triangle
  const uint32_t* object_tri_base; // per object: first triangle of its instance in `tris` (meshes only)
  uint32_t* slow_stack;            // global traversal stack of the launch's exact redo (redo_slow_rays), [kStackDepth][kWave]
  uint2* spill;                    // traversal stack entries beyond the LDS part, [entry][persistent thread]
  uint32_t spill_stride;           // number of persistent threads
  uint32_t spill_cap;              // entries per thread in `spill`
  uint32_t lds_cap;                // entries per lane kept in LDS before `spill` (kLds4; fewer only in tests)
  uint32_t force_slow;             // test hook: hand EVERY ray to the exact redo (redo_slow_rays)
  uint32_t static_eighths;         // persistent traversal: share (x/8) of each image region dealt without atomics
  uint32_t refill_lanes;           // persistent traversal: fetch new rays once this many lanes of a wavefront are idle
  uint32_t split_idle;             // ... and once the launch has no rays left: split busy lanes' stacks among idle ones when
                                   // at least this many lanes are idle (0: never)
  uint32_t object_count;
  // per object kSphereTab float4, for spheres: [0] the world-space ball that contains the transformed sphere {centre.xyz,
  // radius (rounded up)} -- radius < 0: no ball (a projective matrix; a mesh object): never skipped; [1] {1 / largest
  // stretch of the object's matrix (rounded down), 1 if "simple" (both matrices pure translations spelled 1.0f / +0.0f),
  // radius of the INNER ball of a simple object (inside what the reference's float sequence can hit; sphere_ball_of), 0};
  // [2..6] what a lane needs of a simple object: {box min, inverse translation x}, {box max, .. y},
  // {sphere centre, .. z}, {translation, sphere radius}, {material bits, 0, 0, 0} (sphere_run_lanes).
  const float4* sphere_ball;
  uint32_t lanes_run;              // set per launch by the host: the launch's sphere run may take sphere_run_lanes
  uint32_t fold_run;               // ... every object of the launch's sphere run is a "simple" sphere: k_spheres may take sphere_fold
  DBeam beam;                      // set per launch by the host (bounce 0's first traversal launch, or off)
};

// Camera constants prepared on the host once per frame (GPUCamera, camera.hpp:10-15, plus the
// per-frame invariants of generate_ray, ray_gen.cu:37-47, hoisted out of the per-pixel code).
struct DCamera {
  m4 cam;            // translate(position) * mat4_cast(rotation)
  f3 origin;         // (cam * (0,0,0,1)).xyz
  float vw, vh;      // viewport width / height
  float llx, lly;    // lower-left corner x, y  ( = -(vw/2), -(vh/2) )
  uint32_t width, height;
};

// Live-path state, 40 B/path in three arrays (round 5; 48 until then: a float4 of its own for the throughput, two words unused --
// the kernels that end a bounce are HBM-bound and read and write every word of it):
//   o4 = origin.xyz, bits(pixel | tmin_flag<<31)   tmin_flag: t_min is 1e-5 (after a dielectric) instead of 1e-4
//   d4 = direction.xyz, throughput.r               (the closest-hit kernels load o4 and d4 and ignore d4.w; k_raygen leaves it 0:
//   t2 = throughput.gb                              bounce 0 starts from throughput 1 without reading either)
struct DPaths {
  float4* o4;
  float4* d4;
  float2* t2;
};

// Closest-hit record written by the trace kernel (32 B/path):
//   tp = t (or -1 on miss), point.xyz ;  nm = normal.xyz, bits(material | side<<31)
// (point is stored because for spheres the reference keeps the TRANSFORMED object-space hit point,
//  path_tracer.cu:93, which is not origin + direction * t)
struct DHits {
  float4* tp;
  float4* nm;
};

// Accumulated framebuffers (running means): color4 = rgb,-  ;  nd4 = normal.xyz, depth
struct DFrame {
  float4* color4;
  float4* nd4;
};

struct DeviceCounters {
  uint32_t live[kMaxBounces + 1];  // live paths entering bounce b of the current frame
  uint32_t listed_now[kMaxBounces + 1];  // rays on the work list of bounce b's (last) listed traversal launch; with live[]: what the host sizes later launches by
  uint32_t flags;
  uint32_t slow_count;             // rays set aside for the exact redo by the running traversal launch
  uint32_t waves_done;             // wavefronts of the running traversal launch that have signed off
  uint32_t shade_ticket;           // k_shade_fused: next tile of this frame to be taken (zero between launches)
  uint32_t list_count;             // rays k_raygen / k_spheres<.., kFilter> have put on the next traversal launch's work list (written by the list's last tile)
  unsigned long long rays_total;
  unsigned long long paths[kMaxBounces];      // sum of live[b] over frames since the last profile reset
  unsigned long long box_tests[kMaxBounces];  // instrumented runs only
  unsigned long long tri_tests[kMaxBounces];
  uint32_t max_box_tests[kMaxBounces];        // longest single traversal (box tests of one ray), instrumented runs
  unsigned long long listed_rays[kMaxBounces];  // rays the traversal launches fetched through a work list (frame 0's block: whole batch)
  unsigned long long slow_rays[kMaxBounces];  // rays redone exactly (redo_slow_rays)
  unsigned long long node_visits[kMaxBounces];  // BVH node records fetched by the closest-hit kernels, instrumented runs
  // ray-fetch cursors of the persistent traversal launches, one per image region, each on its own 128-byte
  // line (cursors sharing a line serialise in L2: measured ~30 atomics/us for the whole line)
  uint32_t work[kWorkSlots][8][32];
};

// Several iterations ("frames") traced by the same launches: every per-frame array of a slot is one allocation
// with frame f at element offset f * stride (chunk arrays: f * chunk_stride; counters: f), and the kernels of
// one bounce cover all frames of the batch (blockIdx.y = frame; the persistent traversal kernel feeds its lanes
// from all of them).  One launch then carries count times the rays, so the latency tail of a bounce (a few
// long rays) is paid once per batch instead of once per frame.
constexpr int kMaxBatch = 32;
struct DBatchInfo {
  uint32_t stride, chunk_stride, count, pad;
  uint32_t iteration[kMaxBatch];
};
struct DCameras {
  DCamera c[kMaxBatch];
};

// "prefold" (round 5, review item 4): the kernel that ends bounce b has a survivor's NEXT ray in registers; when the next bounce
// opens with a sphere run in front of a mesh (a room's walls: config 2) it tests that run there and then -- the hit record of
// bounce b + 1 goes straight to the survivor's new slot (a second set of hit records: other tiles still read this bounce's),
// one byte says whether the ray may reach the mesh launch's boxes -- and bounce b + 1 starts with k_list_flags (the work list
// from those bytes) instead of k_spheres' pass over every ray.
struct DNextRun {
  uint32_t begin, end;            // the sphere run in front of the next bounce's first traversal launch (objects)
  uint32_t filt_begin, filt_end;  // the mesh objects of that launch (may_hit_boxes)
  uint32_t fold_run;              // the run may take sphere_fold (every object a translated sphere)
  uint32_t pad;
  DHits hits;                     // the NEXT bounce's hit records
  uint8_t* flags;                 // per new slot: 1 = the ray goes on the traversal launch's work list
};

// ---- the bounce-spanning persistent launch (k_persist, round 5) ------------------------------------
// One launch per batch carries the traversal of bounces >= 1 and the shade passes of ALL bounces: of every five wavefronts
// four walk rays (traverse4_walk<.., kPersist>) and one shades tiles (shade_tile<.., 1, true>).  Frame f moves through
//   S(0) -> T(1) -> S(1) -> T(2) -> ... -> S(max_bounces - 1) -> done          (R(b): the exact redo, between T(b) and S(b))
// on its own; what a frame is in is one 64-bit word (phase code << 32 | count: rays of a T phase, tiles of an S phase)
// that the wavefront finishing the phase's last piece of work rewrites, and everybody else polls.  All of it through
// agent-scope atomics; the payload (hit records, path state) is stored write-through and loaded past the L1 (sc1).
constexpr uint32_t kPersistSlotBits = 26u;  // a lane's ray: position in the batch (frame * stride + slot) | frame << 26
constexpr uint32_t kPersistSlotMask = (1u << kPersistSlotBits) - 1u;
constexpr uint32_t kPersistDyn = 128u;      // rays per cursor add (fixed: a cursor then only ever stands on multiples of it)
constexpr uint32_t kPhaseT = 0u, kPhaseRedo = 1u, kPhaseRedoing = 2u, kPhaseS = 3u, kPhaseDone = 0xffffffffu;  // code = bounce << 3 | kind
constexpr uint32_t kPhaseKindBits = 3u, kPhaseKindMask = 7u;
struct DPersistFrame {       // every hot word on a 128-byte line of its own
  uint32_t cursor[8][32];    // [region][0]: bounce << 26 | rays of the frame's current T phase handed out from that region
  uint32_t t_done[32];       // [0]: bounce << 26 | rays of the current T phase whose results are in memory
  uint32_t s_ticket[32];     // [0]: bounce << 26 | next tile of the current S phase
  uint32_t s_done[32];       // [0]: tiles of the current S phase finished
};
struct DPersist {
  unsigned long long state[kMaxBatch];  // two lines, read by everybody
  uint32_t started, frames_done, redo_lock, error;
  uint32_t pad[28];
  DPersistFrame f[kMaxBatch];
  uint32_t dbg[kMaxBatch][16][8];   // PT_PERSIST_DEBUG builds: per frame and bounce {n, rays fetched, rays finalized, tiles, tiles done, live out, .., ..}
};
struct DPersistArgs {
  DPersist* st;
  DPaths paths[2];           // bounce b reads paths[b & 1] and writes paths[(b & 1) ^ 1]
  int max_bounces;
  uint32_t service_every;    // one wavefront in this many (by arrival) is a service wavefront
  uint32_t help_tiles;       // tiles a walking wavefront shades when it finds no rays to hand out, before it looks for rays again
  uint32_t tail_begin, tail_end;  // the sphere run that ends the object list (k_shade_fused's obj_begin / obj_end)
  int staged;
  const uint32_t* slot_base;
  unsigned long long* tile_desc;
  uint32_t tile_stride, epoch0;   // bounce b's shade pass carries epoch0 + b
  DFrame stage;
  DBand band;
  const uint32_t* list0;     // bounce 0: the work list k_raygen left when it finished the other primaries itself, or null
  uint32_t* slow_list;       // per frame at f * stride (DeviceCounters::slow_count of that frame)
};

struct DDenoise {
  float c_phi, n_phi, p_phi;
  int variant;  // 0: taps staged in LDS per sub-lattice (k_denoise_lds, default); 1: taps through L1 / L2 (k_denoise)
};

// The reference BVH built on the device (pt_bvh_gpu.hip): the same tree as build_bvh (pt_bvh.cpp), node for node.
// d_packed: 2 * (2T-1) float4 ({min, first}, {max, count}: DScene::bvh); d_nodes (may be null): the same nodes in the
// reference's 32-byte layout; level_base (may be null): first node of every depth, plus the node count at the end.
// Synchronises the stream (one readback per tree level).
int build_bvh_device(hipStream_t stream, const float* d_positions, const uint32_t* d_indices, uint32_t index_count,
                     float4* d_packed, ptc_bvh_node* d_nodes, uint32_t* node_count, uint32_t* max_depth,
                     std::vector<uint32_t>* level_base);

// The traversal layouts of pt_scene_host.cpp (WideAccel, Wide4Accel, the leaf order) built on the device from a
// reference tree whose nodes are stored depth by depth (level_base as above).  The arrays are hipMalloc'ed and belong
// to the caller.
struct DeviceLayouts {
  uint32_t* nodes_q = nullptr;     // DScene::bvh4q
  float4* leaf_parent = nullptr;   // DScene::leaf_parent (+ the dummy's entry)
  uint32_t* tri_order = nullptr;   // depth-first rank -> triangle of the mesh
  float4* wide = nullptr;          // DScene::wide
  uint32_t wide4_nodes = 0, wide4_depth = 0, triangles = 0, inner_nodes = 0;
  uint32_t root_ref4 = 0, root_ref2 = 0, dummy_ref = 0;
  float root_min[3] = {0, 0, 0}, root_max[3] = {0, 0, 0};
};
int build_layouts_device(hipStream_t stream, const float4* d_bvh, uint32_t count, const std::vector<uint32_t>& level_base,
                         DeviceLayouts* out);
// world-space triangle records of one instance in depth-first order (DScene::tris), kTriVec4 float4 per triangle
void launch_instance_triangles(hipStream_t stream, const m4& m, const float* d_positions, const uint32_t* d_indices,
                               const uint32_t* d_tri_order, uint32_t triangles, float4* d_out);

// ---- launch interface (implemented in pt_kernels.hip) ----
// filt_begin < filt_end and worklist != null: the bounce's first launch is a traversal launch over the mesh objects
// [filt_begin, filt_end); raygen lists the rays that may hit them (in slot order: a look-back scan on the descriptors
// tile_desc -- k_shade_fused's, tile_stride per frame -- under a launch epoch of its own) and writes the miss record of
// the others into `hits`.  finish_misses (that launch walks the scene's whole object list, so an unlisted ray hits
// nothing): the unlisted rays end in raygen -- sky into fb (the shade kernels' frame: staged, or the framebuffers) -- and
// bounce 0's launch_shade_fused gets list = worklist
void launch_raygen(hipStream_t s, const DCameras& cams, const DBatchInfo& bi, DBand band, uint32_t pix_count,
                   DPaths paths, DeviceCounters* counters, const DObject* objects = nullptr, uint32_t filt_begin = 0u,
                   uint32_t filt_end = 0u, uint32_t* worklist = nullptr, DHits hits = DHits{nullptr, nullptr},
                   unsigned long long* tile_desc = nullptr, uint32_t tile_stride = 0u, uint32_t epoch = 0u,
                   bool finish_misses = false, DFrame fb = DFrame{nullptr, nullptr}, bool staged = false);
// variant 0: reference-order traversal (k_trace); 1: culled near-first traversal over the wide layout (k_trace_wide)
void launch_trace(hipStream_t s, const DScene& scene, DPaths paths, DHits hits, uint32_t max_paths, int bounce,
                  DeviceCounters* counters, bool count_tests, int variant);
// variant 3 (default): per bounce, the object list is walked in the reference's order; the closest hit so far
// travels from launch to launch in the hit record:
//   launch_traverse   one mesh object: persistent wavefronts, each lane fetches the next ray when its own is done;
//                     rays it sets aside (degenerate direction, winner grazing its parent box) are redone exactly by
//                     the last wavefront of the same launch
//   launch_spheres    a run of spheres in FRONT of a mesh (objects [obj_begin, obj_end))
//   launch_tail_count the end of the stage: the run of spheres that ends the object list (may be empty, may be the
//                     whole list) and the live count of every 64-slot chunk from the hit records; also after launch_trace
//   launch_scan       exclusive scan of those counts (compaction offsets, live[bounce + 1])
// first: nothing has written the hit records in this bounce yet
// filt_begin < filt_end and worklist != null: also builds the work list of the traversal launch that follows over the mesh
// objects [filt_begin, filt_end) (rays that surely miss all their world boxes are left out); that launch then gets
// order = worklist, listed = true.  (tile_desc, tile_stride, epoch: as for launch_raygen)
void launch_spheres(hipStream_t s, const DScene& scene, uint32_t obj_begin, uint32_t obj_end, bool first, DPaths paths, DHits hits,
                    uint32_t max_paths, int bounce, DeviceCounters* counters, const DBatchInfo& bi, uint32_t filt_begin = 0u,
                    uint32_t filt_end = 0u, uint32_t* worklist = nullptr, unsigned long long* tile_desc = nullptr,
                    uint32_t tile_stride = 0u, uint32_t epoch = 0u);
// a run [obj_begin, obj_end) of mesh objects of the SAME mesh in one launch (k_traverse4m): variant 3 only
void launch_traverse_run(hipStream_t s, const DScene& scene, uint32_t obj_begin, uint32_t obj_end, bool first, DPaths paths,
                         DHits hits, int bounce, int work_slot, DeviceCounters* counters, bool count_tests, uint32_t waves,
                         uint32_t* slow_list, const uint32_t* order, const DBatchInfo& bi, bool listed = false);
// the entry points of `nbeam` cameras (cams.c[cam_of[b]]) for the mesh object obj_index, into `out` (DBeam::entries)
void launch_beam(hipStream_t s, const DScene& scene, uint32_t obj_index, const DCameras& cams, const uint8_t* cam_of, uint32_t nbeam,
                 uint32_t tiles_x, uint32_t tiles_y, uint32_t node_count4, float4* out);
void launch_traverse(hipStream_t s, const DScene& scene, uint32_t obj_index, bool first, DPaths paths, DHits hits,
                     int bounce, int work_slot, DeviceCounters* counters, bool count_tests, uint32_t waves,
                     uint32_t* slow_list, const uint32_t* order, int variant, const DBatchInfo& bi, bool listed = false);
// coherence sort of the pick-up order (never of the slots): octs = direction octant per slot (written by launch_shade
// when given), order = per block of 4096 slots the slots grouped by octant; launch_traverse reads its rays through it
void launch_sort_octant(hipStream_t s, const uint8_t* octs, uint32_t* order, uint32_t max_paths, int bounce,
                        DeviceCounters* counters, const DBatchInfo& bi);
void launch_tail_count(hipStream_t s, const DScene& scene, uint32_t obj_begin, uint32_t obj_end, bool first, DPaths paths,
                       DHits hits, uint32_t max_paths, int bounce, uint32_t* chunk_counts, DeviceCounters* counters,
                       const DBatchInfo& bi);
void launch_scan(hipStream_t s, int bounce, bool last_bounce, const uint32_t* chunk_counts, uint32_t* chunk_offsets,
                 DeviceCounters* counters, const DBatchInfo& bi);
void launch_shade(hipStream_t s, const DScene& scene, DPaths in, DPaths out, DHits hits, uint32_t max_paths,
                  bool staged, int bounce, bool last_bounce, const uint32_t* slot_base,
                  const uint32_t* chunk_offsets, DFrame fb, DBand band, DeviceCounters* counters, uint8_t* octs,
                  const DBatchInfo& bi);
// the end of a bounce in one pass (k_shade_fused): trailing sphere run [obj_begin, obj_end) + material + stable compaction +
// final gather.  tile_desc: shade_tiles_per_frame(max_paths) descriptors per frame of the batch (tile_stride apart), zero
// at allocation and never cleared; epoch: a number no earlier launch on these descriptors has used (1 .. 2^30 - 1).
// list: bounce 0 after launch_raygen(finish_misses): the work list of the bounce's one traversal launch -- the kernel
// walks it (DeviceCounters::list_count entries per frame) instead of all slots
void launch_shade_fused(hipStream_t s, const DScene& scene, uint32_t obj_begin, uint32_t obj_end, bool first, DPaths in, DPaths out,
                        DHits hits, uint32_t max_paths, bool staged, int bounce, bool last_bounce, const uint32_t* slot_base,
                        unsigned long long* tile_desc, uint32_t tile_stride, uint32_t epoch, DFrame fb, DBand band,
                        DeviceCounters* counters, uint8_t* octs, const DBatchInfo& bi, const uint32_t* list = nullptr,
                        const DNextRun* next = nullptr);
// "prefold" at bounce 0: ray generation that also walks the sphere run in front of bounce 0's first mesh launch (next.hits: the
// CURRENT bounce's records here) and leaves the bytes for launch_list_flags
void launch_raygen_next(hipStream_t s, const DScene& scene, const DCameras& cams, const DBatchInfo& bi, DBand band, uint32_t pix_count, DPaths paths,
                        DeviceCounters* counters, const DNextRun& next);
// "prefold": the work list of a bounce whose leading sphere run the previous bounce's shade kernel has already walked, from the
// bytes it left (k_list_flags; tile_desc / epoch as for launch_spheres)
void launch_list_flags(hipStream_t s, const uint8_t* flags, uint32_t max_paths, int bounce, DeviceCounters* counters, const DBatchInfo& bi,
                       uint32_t* worklist, unsigned long long* tile_desc, uint32_t tile_stride, uint32_t epoch);
uint32_t shade_tiles_per_frame(uint32_t max_paths);
void launch_accumulate(hipStream_t s, DFrame stage, DFrame fb, uint32_t pix_count, const DBatchInfo& bi);
// the bounce-spanning persistent launch (pt_kernels.hip, k_persist): state set-up (k_persist_init) + the launch, `waves` wavefronts;
// spheres: the object list ends in a sphere run (pa.tail_begin / tail_end); listed0: bounce 0's shade pass walks pa.list0
void launch_persist(hipStream_t s, const DScene& scene, uint32_t obj_index, DHits hits, DeviceCounters* counters, const DBatchInfo& bi,
                    const DPersistArgs& pa, uint32_t waves, bool spheres, bool listed0);
uint32_t persist_tiles_per_frame(uint32_t max_paths);
void launch_megakernel(hipStream_t s, const DScene& scene, const DCamera& cam, uint32_t iteration, DBand band,
                       uint32_t pix_count, int max_bounces, DFrame fb, DeviceCounters* counters);
void launch_preview(hipStream_t s, const float4* buf, uint32_t pix_count, int mode, uint32_t* rgba);
void launch_pack(hipStream_t s, const float4* buf, uint32_t pix_count, int which, float* dst);
// the bands of one multi-GPU gather launch (k_gather_bands): up to kGatherBands sources per launch
constexpr int kGatherBands = 16;
struct DGatherBands {
  struct Src {
    const float* src;
    DBand band;
    uint32_t pix_count;
  } src[kGatherBands];
};
void launch_gather_bands(hipStream_t s, const DGatherBands& bands, uint32_t count, uint32_t max_pix, int channels,
                         uint32_t frame_pixels, float* frame);
void launch_preview_packed(hipStream_t s, const float* buf, uint32_t pix_count, int channels, int mode, uint32_t* rgba);
void launch_denoise_positions(hipStream_t s, const DCamera& cam, uint32_t pix_count, const float4* nd, float4* pos);
void launch_denoise_pass(hipStream_t s, const DCamera& cam, uint32_t pix_count, const float4* color, const float4* nd,
                         const float4* pos, float4* out, int step_width, DDenoise params);
void launch_intersect(hipStream_t s, const DScene& scene, const float4* rays_o, const float4* rays_d, uint32_t n,
                      DHits hits, DeviceCounters* counters, int variant);
void launch_selftest(hipStream_t s, const float* a, const float* b, uint32_t n, float* out_div, float* out_sqrt,
                     float* out_sin, float* out_cos);

}  // namespace pt
