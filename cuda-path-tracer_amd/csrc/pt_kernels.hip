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



// This unit: the closest-hit stage (k_trace, k_trace_wide, k_beam, k_traverse4, k_traverse4m, k_megakernel, k_intersect).
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
#include "pt_shade_tile.inc"  // (the persistent launch's service wavefronts shade tiles: k_persist below)

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

// ------------------------------------------------------------------------------------------------
// persistent traversal: ray hand-out
// ------------------------------------------------------------------------------------------------
// Traversal lengths differ by an order of magnitude between neighbouring rays (a ray that grazes the terrain
// tests hundreds of boxes, its neighbour a few dozen), so "one wavefront = 64 fixed rays" leaves most lanes idle
// most of the time (measured: 21 % of lanes active per VALU instruction).  A traversal wavefront therefore lives
// for the whole launch and every lane that finishes its ray takes the next unprocessed one.  Results are written
// per slot, so the order in which rays are processed is irrelevant to the output.  One launch handles ONE mesh
// object (its matrices stay in scalar registers); the closest hit so far travels in the hit record between the
// segments of a bounce.  The live paths [0, n) of a frame are cut into eight image regions (one per XCD:
// blockIdx % 8) and each region into 64-ray batches; a share of the batches is dealt statically (no atomics),
// the rest is taken from one cursor per region (own 128-byte line).
// RayFeed for a batch of frames (DBatchInfo): count x 8 regions, region (f, r) = eighth r of frame f's live
// rays, its cursor on frame f's counters.  A wavefront's home keeps the XCD <-> image-region pairing of
// RayFeed (r = blockIdx & 7) and deals the frames round-robin over the wavefronts of that XCD.  Ranges are
// returned as batch-global slots (f * stride + slot).  When its static share is done a wavefront looks at
// all regions at once -- lane p probes the p-th region in its preference order (same eighth of the other
// frames first: same part of the tree in this XCD's L2) -- and takes from the first that has rays left.
struct BatchFeed {
  DeviceCounters* ctr;
  uint32_t stride, count, bounce, work_slot, static_eighths, dyn_batch;
  bool listed;  // the launch walks a work list (DeviceCounters::list_count entries per frame, read through `order`), not all live rays
  __device__ __forceinline__ uint32_t rays_of(uint32_t f) const { return listed ? ctr[f].list_count : ctr[f].live[bounce]; }
  uint32_t home_f, home_r, home_base, home_rs, stat_next, stat_step, stat_count;
  bool in_static, done;

  // (the regions' geometry: pt_feed_rules.hpp, shared with the host's check, ptc_check_feed)
  static __device__ __forceinline__ uint32_t region_size_of(uint32_t n) { return feed_rules::region_size_of(n); }
  static __device__ __forceinline__ uint32_t region_len_of(uint32_t n, uint32_t rs, uint32_t r) { return feed_rules::region_len_of(n, rs, r); }
  static __device__ __forceinline__ uint32_t pos_of(uint32_t rs, uint32_t r, uint32_t local) { return feed_rules::pos_of(rs, r, local); }
  __device__ __forceinline__ uint32_t static_batches_of(uint32_t len) const
  {
    return feed_rules::static_batches_of(len, static_eighths);
  }
  __device__ __forceinline__ void init(DeviceCounters* ctr_, const DBatchInfo& bi, int bounce_, int work_slot_,
                                       uint32_t static_eighths_, bool listed_ = false)
  {
    ctr = ctr_;
    listed = listed_;
    stride = bi.stride;
    count = bi.count;
    bounce = (uint32_t)bounce_;
    work_slot = (uint32_t)work_slot_;
    // every home needs at least one wavefront for its static share
    static_eighths = gridDim.x >= 8u * count ? static_eighths_ : 0u;
    const uint32_t j = blockIdx.x >> 3;
    home_r = blockIdx.x & 7u;
    home_f = j % count;
    const uint32_t with_r = (gridDim.x - home_r + 7u) / 8u;  // wavefronts of this r
    stat_step = (with_r - home_f + count - 1u) / count;      // ... of which this many share the home
    stat_next = j / count;
    const uint32_t n = rays_of(home_f);
    home_rs = region_size_of(n);
    stat_count = static_batches_of(region_len_of(n, home_rs, home_r));
    home_base = home_f * stride;
    in_static = static_eighths != 0u;
    done = false;
    // dynamic batches: one atomic hands out this many rays (a cursor line sustains ~30 atomics/us)
    // (re-measured on round 3's final build: 64 everywhere -6 %, 256 or other thresholds within noise)
    // (the same for every wavefront of the launch -- frame 0's count decides: the cursors then only ever stand on multiples
    // of it, and a batch of two never straddles two blocks of a region)
    dyn_batch = (uint64_t)rays_of(0u) * count / gridDim.x >= 256u ? 128u : (uint32_t)kWave;
  }
  __device__ __forceinline__ bool exhausted() const { return !in_static && done; }
  // wave-uniform: next batch [begin, end) of batch-global slots, or false
  __device__ __forceinline__ bool acquire(uint32_t& begin, uint32_t& end)
  {
    if (in_static) {
      if (stat_next < stat_count) {
        begin = home_base + pos_of(home_rs, home_r, stat_next * kWave);
        end = begin + kWave;  // static batches are full batches inside the region
        stat_next += stat_step;
        return true;
      }
      in_static = false;
    }
    while (!done) {
      bool any = false;
      for (uint32_t first_p = 0u; first_p < 8u * count; first_p += (uint32_t)kWave) {
        const uint32_t p = first_p + threadIdx.x;
        bool has = false;
        uint32_t len = 0u, first = 0u, gbase = 0u, grs = 0u, gr = 0u;
        uint32_t* cursor = nullptr;
        if (p < 8u * count) {
          const uint32_t df = p % count, dr = p / count;
          uint32_t f = home_f + df;
          if (f >= count) f -= count;
          // (own region first, then the next ones in turn.  Every wavefront walking the regions in the same order -- so that
          // the launch ends on a chosen one -- was measured: the end of a primary-ray launch 300 instead of 570 us after its
          // feed, the launch as a whole 2 % LONGER, the later bounces 5-20 %: twenty cursors for 5120 wavefronts.)
          const uint32_t r = (home_r + dr) & 7u;
          const uint32_t n = rays_of(f);
          const uint32_t rs = region_size_of(n);
          len = region_len_of(n, rs, r);
          first = static_batches_of(len) * kWave;
          gbase = f * stride;
          grs = rs;
          gr = r;
          cursor = &ctr[f].work[work_slot][r][0];
          has = first < len && first + __hip_atomic_load(cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < len;
        }
        const uint64_t mask = __ballot(has);
        if (mask == 0ull) continue;
        any = true;
        const int sel = __ffsll((unsigned long long)mask) - 1;
        uint32_t base = len;
        if ((int)threadIdx.x == sel) base = first + atomicAdd(cursor, dyn_batch);
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)base, sel);
        const uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)len, sel);
        const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)gbase, sel);
        if (b < l) {
          const uint32_t rs_sel = (uint32_t)__builtin_amdgcn_readlane((int)grs, sel), r_sel = (uint32_t)__builtin_amdgcn_readlane((int)gr, sel);
          begin = g + pos_of(rs_sel, r_sel, b);
          end = begin + (min(l, b + dyn_batch) - b);
          return true;
        }
        break;  // another wavefront took the rest of that region: look again
      }
      if (!any) done = true;
    }
    return false;
  }
};

// The feed of the bounce-spanning persistent launch (k_persist; DPersist in pt_device.hpp): the rays of frame f's current
// traversal phase T(b) -- live[b] rays, cut into the same eight interleaved regions as a launch's (pt_feed_rules.hpp) -- are
// handed out kPersistDyn at a time from one cursor per region; frames are dealt IN ORDER (the lowest frame that has rays
// left in this wavefront's region first, other regions after that), so that frame f's bounce is complete -- and its shade
// pass, and then its next bounce, can start -- while later frames are still being dealt.
// A cursor carries its phase (bounce << 26): a wavefront that decided on stale knowledge of a frame still gets, from its
// add, a range of the frame's CURRENT phase and the tag to tell -- it then re-reads the frame's state and goes on with that
// (the add cannot be undone, and it need not be: the rays it stands for exist and nobody else will be given them).
struct PersistFeed {
  DPersist* st;
  DeviceCounters* ctr;
  uint32_t stride, count, home_r;
  bool have;                              // (cf, cr) below is a frame / region this wavefront last found rays in
  uint32_t cf, cr, cbounce, cn, crs, clen, cdyn;
  __device__ __forceinline__ void init(DPersist* st_, DeviceCounters* counters, const DBatchInfo& bi, uint32_t region)
  {
    st = st_;
    ctr = counters;
    stride = bi.stride;
    count = bi.count;
    home_r = region & 7u;
    have = false;
    cf = cr = cbounce = cn = crs = clen = 0u;
  }
  // Frame f = this lane (f < count): v more of its rays are done (their hit records in memory).  `old` is what the done
  // counter held before this wavefront's add, `now` the frame's state word: the add that completes the phase's ray count
  // opens the phase behind it -- the exact redo if rays were set aside, else the shade pass -- in the order done counter,
  // state word, ticket tag (a wavefront that draws a ticket of the pass then finds the state word there, and its sign-off
  // cannot be wiped by the counter's reset).
  __device__ __forceinline__ void count_done(const uint32_t v, const uint32_t old, const unsigned long long now)
  {
    // (the counter carries its phase, bounce << 26, like the cursors: a count that belongs to a phase the state word no
    // longer shows -- it was not the last one, and the frame has moved on while this wavefront's load was on its way --
    // decides nothing)
    const uint32_t f = threadIdx.x, n = (uint32_t)now, b = (uint32_t)(now >> 32) >> kPhaseKindBits;
    if (v != 0u && (old >> kPersistSlotBits) == b && ((uint32_t)(now >> 32) & kPhaseKindMask) == kPhaseT && (old & kPersistSlotMask) + v == n) {
      const uint32_t slow = __hip_atomic_load(&ctr[f].slow_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t tiles = (n + kServiceTile - 1u) / kServiceTile;
#ifdef PT_PERSIST_DEBUG
      st->dbg[f][b & 15u][0] = n;
      st->dbg[f][b & 15u][2] = (uint32_t)wall_clock64();
#endif
      __hip_atomic_store(&st->f[f].s_done[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&st->state[f], ((unsigned long long)((b << kPhaseKindBits) | (slow ? kPhaseRedo : kPhaseS)) << 32) | tiles, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&st->f[f].s_ticket[0], b << kPersistSlotBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // the same, on its own (two dependent round trips: only where the wavefront has nothing else to do)
  __device__ __forceinline__ void flush(uint32_t& v)
  {
    if (v != 0u) {
      const uint32_t old = __hip_atomic_fetch_add(&st->f[threadIdx.x].t_done[0], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long now = __hip_atomic_load(&st->state[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      count_done(v, old, now);
    }
    v = 0u;
  }
  __device__ __forceinline__ void set_geometry(uint32_t n)
  {
    cn = n;
    crs = feed_rules::region_size_of(n);
    clen = feed_rules::region_len_of(n, crs, cr);
    // rays per draw: all wavefronts of an XCD draw from ONE cursor (frames are dealt in order), and a cursor's line sustains
    // ~30 adds per microsecond -- with 128 rays per draw the walk of a 770,000-ray phase spent more time queueing for its
    // cursor than walking (measured: 2.3 x slower than the per-bounce launch, whose feed has 160 cursors and a static share)
    cdyn = n >= 262144u ? 512u : (n >= 65536u ? 256u : kPersistDyn);
  }
  // batch position (frame * stride + slot) of the ray at region-local offset `local` of the range acquire handed out last
  __device__ __forceinline__ uint32_t position_of(uint32_t local) const { return cf * stride + feed_rules::pos_of(crs, cr, local); }
  // wave-uniform.  1: [begin, end) are REGION-LOCAL offsets (position_of) of rays of `frame` entering `bounce`;
  // 0: nothing to hand out right now (frames are between phases, or all rays of the running phases are out);
  // -1: every frame of the batch is done
  // v (lane f < count: finished rays of frame f not yet on its done counter, else 0) rides along: its adds are issued WITH
  // the cursor's, so that a wavefront pays one round trip for both (the first build's separate flush in front of every
  // refill -- a returning add, then a load of the state word -- made the walk 2.5 x slower); v comes back 0.
  __device__ __forceinline__ int acquire(uint32_t& begin, uint32_t& end, uint32_t& frame, uint32_t& bounce, uint32_t& v)
  {
    for (int round = 0; round < 6; ++round) {
      if (have) {
        uint32_t old = 0u, dold = 0u;
        unsigned long long dnow = 0ull;
        const uint32_t drawn = cdyn;
        if (threadIdx.x == 0u) old = __hip_atomic_fetch_add(&st->f[cf].cursor[cr][0], cdyn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != 0u) {
          dold = __hip_atomic_fetch_add(&st->f[threadIdx.x].t_done[0], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          dnow = __hip_atomic_load(&st->state[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
        count_done(v, dold, dnow);
        v = 0u;
        const uint32_t tag = old >> kPersistSlotBits, base = old & kPersistSlotMask;
        if (tag != cbounce) {  // the frame has moved on since this wavefront looked: the range is one of its current phase
          // (the phase's state word is stored BEFORE its cursors are re-tagged, so it is there; the wait is for a late store)
          // (... unless the draw came after the phase's last ray was handed out: the frame may then be anywhere BEHIND T(tag),
          // and the draw is simply a miss)
          unsigned long long now = 0ull;
          bool ok = false, past = false;
          for (uint32_t w = 0u; w < (1u << 20) && !ok && !past; ++w) {
            now = __hip_atomic_load(&st->state[cf], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t c = (uint32_t)(now >> 32);
            ok = c == ((tag << kPhaseKindBits) | kPhaseT);
            past = c == kPhaseDone || (c >> kPhaseKindBits) > tag || ((c >> kPhaseKindBits) == tag && (c & kPhaseKindMask) != kPhaseT);
            if (!ok && !past) __builtin_amdgcn_s_sleep(2);
          }
          if (past) {
            have = false;
            continue;
          }
          if (!ok) {  // cannot be: a T phase does not end before the rays it handed out are done
            if (threadIdx.x == 0u) __hip_atomic_fetch_or(&st->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            have = false;
            return -2;
          }
          cbounce = tag;
          set_geometry((uint32_t)now);
        }
        if (base < clen) {  // (region-local offsets: position_of maps a lane's own offset, so a draw may span blocks)
          begin = base;
          end = min(clen, base + drawn);
          frame = cf;
          bounce = cbounce;
          return 1;
        }
        have = false;
      }
      flush(v);
      // every frame's state at once (lanes 0..31 and 32..63 both hold frame lane & 31: two regions are probed per step)
      const uint32_t f = threadIdx.x & 31u, half = threadIdx.x >> 5;
      unsigned long long s = (unsigned long long)kPhaseDone << 32;
      if (f < count) s = __hip_atomic_load(&st->state[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t code = (uint32_t)(s >> 32), n = (uint32_t)s;
      if (__ballot(code != kPhaseDone) == 0ull) return -1;
      const bool is_t = code != kPhaseDone && (code & kPhaseKindMask) == kPhaseT;
      if (__ballot(is_t) == 0ull) return 0;
      const uint32_t rs = feed_rules::region_size_of(n);
      bool found = false;
      for (uint32_t dr = 0u; dr < 8u && !found; dr += 2u) {
        const uint32_t r = (home_r + dr + half) & 7u;
        const uint32_t len = is_t ? feed_rules::region_len_of(n, rs, r) : 0u;
        bool has = false;
        if (len != 0u) {
          const uint32_t c = __hip_atomic_load(&st->f[f].cursor[r][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          has = (c >> kPersistSlotBits) == (code >> kPhaseKindBits) && (c & kPersistSlotMask) < len;
        }
        const uint64_t m = __ballot(has);
        if (m != 0ull) {
          // the lowest frame; of its two probed regions the nearer one
          const uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
          const int flo = lo ? __ffs((int)lo) - 1 : 64, fhi = hi ? __ffs((int)hi) - 1 : 64;
          const int sel = flo <= fhi ? flo : fhi + 32;
          cf = (uint32_t)sel & 31u;
          cr = (home_r + dr + ((uint32_t)sel >> 5)) & 7u;
          cbounce = (uint32_t)__builtin_amdgcn_readlane((int)(code >> kPhaseKindBits), sel);
          set_geometry((uint32_t)__builtin_amdgcn_readlane((int)n, sel));
          have = true;
          found = true;
        }
      }
      if (!found) return 0;
    }
    return 0;
  }
};

// ------------------------------------------------------------------------------------------------
// variant 3 (default): persistent wavefronts over the four-wide collapse, 64-byte quantised nodes
// ------------------------------------------------------------------------------------------------
// Conservative FMA slabs on four children per step, children visited nearest first, optimistic acceptance,
// one exact test of the winner (finalize below), results written in batches just before a refill.
#ifndef PT_T4_WAVES
#define PT_T4_WAVES 5
#endif
// iterations between two rounds of work splitting at the end of a launch (4 until round 3: every iteration is 3.6 % faster
// on a 1/64 share of the frame, 1.2 % on an eighth, +-0 on the whole; every eighth is slower everywhere)
#ifndef PT_SPLIT_EVERY
#define PT_SPLIT_EVERY 1u
#endif
// end of a launch: finished lanes are retired when this many wait, or every so many iterations
#ifndef PT_RETIRE_LANES
#define PT_RETIRE_LANES 4u
#endif
#ifndef PT_RETIRE_EVERY
#define PT_RETIRE_EVERY 3u
#endif
#ifndef PT_FULL_SORT
#define PT_FULL_SORT 1
#endif
#ifndef PT_FLAT_TRI
#define PT_FLAT_TRI 1
#endif


// -DPT_TAILPROF (diagnostic builds only, tools/tailprof.py): per traversal launch and wavefront the times (100 MHz
// wall clock) at which it started, found the feed exhausted and left the loop, and its loop iterations / split rounds
// taken -- where does the end of a launch go?
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

// End of a persistent traversal launch, run by its LAST wavefront (after the exact redo): the bookkeeping of the next
// launch on this stream starts from zero -- the redo list, the sign-off counter and the fetch cursors of this launch's
// set for every frame of the batch (plain stores: the next launch starts after this one has completed).
__device__ __forceinline__ void launch_epilogue(DeviceCounters* counters, int bounce, int work_slot, uint32_t redone,
                                                const DBatchInfo& bi, bool was_listed)
{
  for (uint32_t i = threadIdx.x; i < bi.count * 8u; i += (uint32_t)kWave) counters[i >> 3].work[work_slot][i & 7u][0] = 0u;
  // what was on the work lists goes into the profile (list_count itself stays: the kernel that builds a list always
  // writes it, and bounce 0's k_shade_fused may walk the same list after this launch)
  uint32_t listed = 0u;
  if (was_listed)
    for (uint32_t f = threadIdx.x; f < bi.count; f += (uint32_t)kWave) listed += counters[f].list_count;
  listed = wave_sum(listed);
  if (threadIdx.x == 0u) {
    if (was_listed) counters->listed_now[bounce] = counters->list_count;
    counters->listed_rays[bounce] += listed;
    counters->slow_rays[bounce] += redone;
    counters->slow_count = 0u;
    counters->waves_done = 0u;
  }
}

// Set a ray aside for the exact redo at the end of the launch (redo_slow_rays).  The entry is written with an
// agent-scope atomic store: the wavefront that drains the list may run on another XCD (its own L2).
__device__ __forceinline__ void set_aside(DeviceCounters* counters, uint32_t* slow_list, uint32_t slot)
{
  const uint32_t at = atomicAdd(&counters->slow_count, 1u);
  __hip_atomic_store(&slow_list[at], slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sc1 forms of the hand-over stores and loads of the persistent launch (kPersist below): write-through / past the L1, so
// that a wavefront on another XCD sees them (MI355X_MICROARCH.md, inter-workgroup visibility).  Hand-issued: the compiler
// does not count them, which can only make one of its own waits longer; the loads carry their wait.
// The store ends in s_nop 1: a store of more than 8 bytes reads its data registers for two more cycles, and the compiler, which
// inserts that wait behind its own stores, does not look into a hand-written one (round 5's first build stored whatever the
// next instruction had put into those registers: hit records with a pointer's bits in them).
__device__ __forceinline__ void st_sc1(float4* p, const float4 v)
{
  v4f w;
  w.x = v.x;
  w.y = v.y;
  w.z = v.z;
  w.w = v.w;
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ void ld2_sc1(const float4* pa, const float4* pb, float4& a, float4& b)
{
  v4f x, y;
  asm volatile(
      "global_load_dwordx4 %0, %2, off sc1\n\t"
      "global_load_dwordx4 %1, %3, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(x), "=&v"(y)
      : "v"(pa), "v"(pb)
      : "memory");
  a = make_float4(x.x, x.y, x.z, x.w);
  b = make_float4(y.x, y.y, y.z, y.w);
}

// kPersist (k_persist, round 5): the wavefront does not belong to one bounce's launch but lives for the whole batch -- its
// rays come from PersistFeed (frame f's bounce b as soon as the shade pass of (f, b - 1) is done), a lane's `slot` carries
// its frame in the top bits, finished rays are counted per frame (in LDS, then on the frame's counter once their hit
// records are in memory), and when nothing can be handed out the wavefront finishes what it holds (the second loop),
// waits, and starts over.  `paths` / `bounce` / `work_slot` / `order` / `listed` are unused then (pa->paths[bounce & 1]).
template <bool kCount, bool kFirst, bool kBeam, bool kPersist = false>
__device__ __forceinline__ int traverse4_walk(const DScene& sc, uint32_t obj_index, const DPaths& paths, const DHits& hits,
                                               int bounce, int work_slot, DeviceCounters* counters, uint32_t* slow_list,
                                               const uint32_t* order, const DBatchInfo& bi, const bool listed,
                                               const DPersistArgs* pa = nullptr)
{
  static_assert(!kPersist || (kFirst && !kBeam && !kCount), "the persistent launch: one mesh object per bounce, bounces >= 1, not instrumented");
  constexpr uint32_t kSlotMask = kPersist ? kPersistSlotMask : 0xffffffffu;
  __shared__ uint32_t s_pend[kPersist ? kMaxBatch : 1];  // kPersist: finished rays per frame, not yet on the frame's counter
  __shared__ uint32_t s_stack[kLds4 * kWave];
  // work splitting at the end of a launch (see `split` below): per lane = per ray group led by that lane
  __shared__ unsigned long long s_grp_best[kWave];  // best candidate of the group so far: t bits << 32 | ~triangle
  __shared__ uint32_t s_grp_count[kWave];           // lanes still walking for the group
  __shared__ uint32_t s_leader[kWave];              // per lane: the lane that leads the ray it is walking for
  __shared__ uint32_t s_pair[kWave];                // scratch: r-th donor of a split round
  // explicitly an LDS pointer: as a generic pointer the pop below compiles to a flat load
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  lds_u32* stack = (lds_u32*)s_stack + threadIdx.x;
  const uint32_t gid = blockIdx.x * kWave + threadIdx.x;
  // `slot` below is batch-global (frame * bi.stride + slot in the frame); flags, the slow-ray list and the
  // test tallies of the whole batch go to frame 0's counters
  uint32_t n_max = 0u;
  if (!kPersist) {
    for (uint32_t f = 0; f < bi.count; ++f) n_max = max(n_max, listed ? counters[f].list_count : counters[f].live[bounce]);
    if (n_max == 0u) return 0;
  }
  const DObject* obj = sc.objects + obj_index;
  const uint32_t mat = sc.object_material[obj_index];
  const float4* tris = sc.tris + kTriVec4 * (size_t)sc.object_tri_base[obj_index];
  // more wavefronts than batches (the margin keeps every wavefront that owns a static batch, see BatchFeed)
  if (!kPersist && blockIdx.x >= ((n_max + kWave - 1u) / kWave + 8u) * bi.count) return 0;
  // the object's world box: the same for every ray of the launch (scalar registers)
  auto uni = [](float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); };
  const f3 obj_bmin = mk3(uni(obj->bmin[0]), uni(obj->bmin[1]), uni(obj->bmin[2]));
  const f3 obj_bmax = mk3(uni(obj->bmax[0]), uni(obj->bmax[1]), uni(obj->bmax[2]));
#ifdef PT_TAILPROF
  const unsigned long long tp_start = wall_clock64();
  unsigned long long tp_exhausted = 0ull;
  uint32_t tp_iters = 0u, tp_splits = 0u, tp_iters_exh = 0u, tp_lanes_exh = 0u;
  unsigned long long tp_c_split = 0ull, tp_c_retire = 0ull, tp_c_step = 0ull, tp_lanes_tail = 0ull;
#endif
  BatchFeed feed;
  PersistFeed pfeed;
  if (kPersist) {
    pfeed.init(pa->st, counters, bi, blockIdx.x);
    if (threadIdx.x < (uint32_t)kMaxBatch) s_pend[threadIdx.x] = 0u;
  } else {
    feed.init(counters, bi, bounce, work_slot, sc.static_eighths, listed);
  }
  uint32_t priv_next = 0u, priv_end = 0u;
  uint32_t cur_frame = 0u, cur_bounce = 0u;  // kPersist: whose rays [priv_next, priv_end) are
  bool dry = false;                           // kPersist: the feed had nothing to hand out when last asked
  int walk_status = 0;                        // kPersist: what the feed said last: 0 nothing right now, < 0 every frame is done

  bool active = false;
  bool pending = false;
  uint32_t slot = 0u, cur = 0u, flags = 0u;
  int sp = 0, sbase = 0, best_k = -1;  // the lane's stack is entries [sbase, sp) of its column
  bool split_mode = false;             // wave-uniform: some ray of this wavefront is walked by several lanes
  uint32_t since_split = 0u, since_retire = 0u;
  f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 0), inv = mk3(0, 0, 0);
  f3 oin = mk3(0, 0, 0), oif = mk3(0, 0, 0);
  bool neg_x = false, neg_y = false, neg_z = false;  // sign of 1/d per axis: which plane of a box is the near one
  float tmin = 0.0f, best_t = 0.0f, scale = 0.0f, limit = 0.0f;
  Tally tally;
  uint32_t ray_boxes = 0u;

  // entries [0, lds_cap) of a lane's stack live in LDS, the rest in the launch's global overflow area (lds_cap is
  // kLds4 except in tests that want the overflow path exercised by small scenes)
  const int lds_cap = min((int)sc.lds_cap, kLds4);  // (the host sets lds_cap <= kLds4: same header)
  auto push = [&](uint32_t ref) {
    if (sp < lds_cap) stack[sp * kWave] = ref;
    else if (sp < lds_cap + (int)sc.spill_cap) sc.spill[(size_t)(sp - lds_cap) * sc.spill_stride + gid].x = ref;
    else {
      flags |= kFlagStackOverflow;
      return;
    }
    ++sp;
  };
  // the conservative slab pair of one box for this lane's ray, tolerance folded INWARDS: if even that interval is
  // non-empty, the reference's exact test of the box passes
  auto surely_inside = [&](const f3 lo, const f3 hi) -> bool {
    const bool nx = neg_x, ny = neg_y, nz = neg_z;
    const float tn = fmaxf(fmaxf(__builtin_fmaf(nx ? hi.x : lo.x, inv.x, oif.x), __builtin_fmaf(ny ? hi.y : lo.y, inv.y, oif.y)),
                           __builtin_fmaf(nz ? hi.z : lo.z, inv.z, oif.z));
    const float tf = fminf(fminf(__builtin_fmaf(nx ? lo.x : hi.x, inv.x, oin.x), __builtin_fmaf(ny ? lo.y : hi.y, inv.y, oin.y)),
                           __builtin_fmaf(nz ? lo.z : hi.z, inv.z, oin.z));
    return tf >= tn;
  };
  // Result of a finished ray.  The winner is the closest of ALL candidates; it is the reference's answer iff the
  // reference reaches it: the object's world box passes (path_tracer.cu:84, tested here instead of before the
  // walk) and the box of the winner's parent passes the reference's own test (nesting).  Both tests have a
  // cheap sufficient form (approximate arithmetic with the error bound held against the ray); the exact
  // divisions run only for rays that graze a box.
  auto finalize = [&]() {
    // everything a winner needs from memory -- its parent's box, its normal -- requested in one go (three
    // dependent round trips otherwise: this code runs every few loop iterations)
    const size_t win = (size_t)max(best_k, 0);
    const float4 pb0 = sc.cur.leaf_parent[2u * win], pb1 = sc.cur.leaf_parent[2u * win + 1u];
    const float4 tc = tris[kTriVec4 * win + 2u];
    if (best_k >= 0) {
      // world box (ray_aabb, intersections.cuh:87-103): quotients by reciprocal, each within 3 ulp of the quotient
      const f3 bmin = obj_bmin, bmax = obj_bmax;
      const f3 winv = mk3(__builtin_amdgcn_rcpf(rd.x), __builtin_amdgcn_rcpf(rd.y), __builtin_amdgcn_rcpf(rd.z));
      const f3 a0 = (bmin - ro) * winv, a1 = (bmax - ro) * winv;
      const float wn = fmaxf(fmaxf(fminf(a0.x, a1.x), fminf(a0.y, a1.y)), fminf(a0.z, a1.z));
      const float wf = fminf(fminf(fmaxf(a0.x, a1.x), fmaxf(a0.y, a1.y)), fmaxf(a0.z, a1.z));
      const bool box_ok = !(bmin.x > bmax.x || bmin.y > bmax.y || bmin.z > bmax.z);
      bool world_sure = box_ok && finite_f(winv.x + winv.y + winv.z) && (wf - wn) > 2e-6f * (fabsf(wf) + fabsf(wn));
      const bool parent_sure = surely_inside(xyz(pb0), xyz(pb1));
      asm volatile("" ::"v"(tc.y), "v"(tc.z), "v"(tc.w));  // keeps the normal's load up there with the box's
      if (__builtin_expect(!(world_sure && parent_sure) || sc.force_slow == 2u, 0)) {
        if (!ray_aabb(ro, rd, bmin, bmax)) {
          best_k = -1;  // the reference skips the object: the carried hit (or the miss) stands
        } else {
          const f3 od = normalize(xform_vector(obj->inv_m, rd));  // inverse_transform_ray, transform.hpp:51-58
          const f3 oo = xform_point(obj->inv_m, ro);
          float en, ef;
          if (!slab_exact(xyz(pb0), xyz(pb1), oo, od, en, ef) || sc.force_slow == 2u) {
            // a ray grazing the parent's box within rounding: redone with exact box decisions by the launch's epilogue (redo_slow_rays)
            if (kPersist) set_aside(counters + (slot >> kPersistSlotBits), pa->slow_list + (size_t)(slot >> kPersistSlotBits) * bi.stride, slot & kSlotMask);
            else set_aside(counters, slow_list, slot);
            best_k = -2;
          }
        }
      }
    }
    if (best_k >= 0) {
      const f3 outward = mk3(tc.y, tc.z, tc.w);
      const f3 p = ro + rd * best_t;
      const uint32_t side = dot(rd, outward) < 0.0f ? 0u : 1u;
      const f3 nn = side == 0u ? outward : -outward;
      if (kPersist) {
        st_sc1(&hits.tp[slot & kSlotMask], make_float4(best_t, p.x, p.y, p.z));
        st_sc1(&hits.nm[slot & kSlotMask], make_float4(nn.x, nn.y, nn.z, __uint_as_float(mat | (side << 31))));
      } else {
        stnt(&hits.tp[slot], make_float4(best_t, p.x, p.y, p.z));
        stnt(&hits.nm[slot], make_float4(nn.x, nn.y, nn.z, __uint_as_float(mat | (side << 31))));
      }
    } else if (kFirst && best_k == -1) {
      if (kPersist) st_sc1(&hits.tp[slot & kSlotMask], make_float4(-1.0f, 0.f, 0.f, 0.f));
      else stnt(&hits.tp[slot], make_float4(-1.0f, 0.f, 0.f, 0.f));
    }
    if (kCount) atomicMax(&counters->max_box_tests[bounce], ray_boxes);
    // (kPersist: one more ray of its frame is done -- counted here, added to the frame's counter by flush_done once the
    // stores above have landed)
    if (kPersist) __hip_atomic_fetch_add(&s_pend[slot >> kPersistSlotBits], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  // kPersist: what this wavefront has finished since the last call goes onto the frames' counters; the lane whose add
  // completes a frame's traversal phase opens the phase behind it (the exact redo if rays were set aside, else the shade pass)
  // kPersist: finished rays per frame (lane f < count) whose hit records are in memory, taken off the wavefront's LDS counts
  auto take_done = [&]() -> uint32_t {
    uint32_t v = 0u;
    if (kPersist) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the hit records of what is counted are in memory (sc1 stores)
      if (threadIdx.x < bi.count) {
        v = s_pend[threadIdx.x];
        if (v != 0u) __hip_atomic_fetch_sub(&s_pend[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    return v;
  };

  // ---- work splitting: the tail of a launch -------------------------------------------------------------------
  // When the launch has no rays left to hand out, a wavefront's lanes fall idle one by one while a few long rays
  // (a ray grazing the terrain visits hundreds of nodes) keep the launch -- and the whole bounce behind it -- alive:
  // 100-150 us per launch, whatever its size, which is most of the time of a small launch (a single frame, a rank's
  // share of a multi-GPU frame, the late bounces).  An idle lane then takes the BOTTOM entry of a busy lane's stack
  // (the farthest, usually largest pending subtree) together with a copy of its ray and walks it as a member of
  // that ray's group.  Candidates are merged in LDS with a 64-bit minimum (t, then the reference's tie rule: the
  // later triangle); the last member to finish writes the ray's result like an unsplit lane would.  The closest
  // hit is the same whoever walks which subtree, so results do not change (tests: every schedule bit-identical).
  auto candidate_key = [&]() -> unsigned long long {
    return best_k >= 0 ? ((unsigned long long)__float_as_uint(best_t) << 32) | (unsigned long long)(~(uint32_t)best_k) : ~0ull;
  };
  auto adopt = [&](unsigned long long key) {
    if (key != ~0ull) {
      const float t = __uint_as_float((uint32_t)(key >> 32));
      const int k = (int)~(uint32_t)key;
      if (t < best_t || (t == best_t && k > best_k)) {
        best_t = t;
        best_k = k;
        limit = scale * t * 1.001f;
      }
    }
  };
  // a lane whose walk has ended: alone -> finalize; in a group -> hand in the candidate, and finalize only as the last
  auto retire = [&]() {
    if (split_mode) {
      const uint32_t leader = s_leader[threadIdx.x];
      __hip_atomic_fetch_min(&s_grp_best[leader], candidate_key(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t left = __hip_atomic_fetch_sub(&s_grp_count[leader], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (left != 1u) return;  // another member is still walking: it will finish the ray
      best_k = -1;
      best_t = FLT_MAX;
      adopt(__hip_atomic_load(&s_grp_best[leader], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    }
    finalize();
  };
  auto split = [&]() {
    // share what the members of a group know (tightens every member's culling limit)
    if (split_mode && active) {
      const uint32_t leader = s_leader[threadIdx.x];
      const unsigned long long mine = candidate_key();
      const unsigned long long seen = __hip_atomic_fetch_min(&s_grp_best[leader], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      adopt(seen);
    }
    const bool can_give = active && sp - sbase >= 1 && sbase < lds_cap;
    const uint64_t donors = __ballot(can_give), takers = __ballot(!active && !pending);
    const uint32_t pairs = min((uint32_t)__popcll(donors), (uint32_t)__popcll(takers));
    if (pairs == 0u) return;
    if (!split_mode) {  // first split of this wavefront: every lane leads its own ray
      s_leader[threadIdx.x] = threadIdx.x;
      s_grp_count[threadIdx.x] = active || pending ? 1u : 0u;  // (a finished lane that has not retired yet still owes its ray's result)
      s_grp_best[threadIdx.x] = ~0ull;
      split_mode = true;
    }
    const uint32_t drank = rank_below(donors), trank = rank_below(takers);
    if (can_give && drank < pairs) s_pair[drank] = threadIdx.x;
    const bool take = !active && !pending && trank < pairs;
    const uint32_t d = take ? s_pair[trank] : threadIdx.x;  // my donor (myself: no change)
    // the donor's ray and walk state (every lane reads its partner's registers)
    auto from = [&](float v) { return __shfl(v, (int)d, kWave); };
    const f3 d_ro = mk3(from(ro.x), from(ro.y), from(ro.z)), d_rd = mk3(from(rd.x), from(rd.y), from(rd.z));
    const f3 d_inv = mk3(from(inv.x), from(inv.y), from(inv.z));
    const f3 d_oin = mk3(from(oin.x), from(oin.y), from(oin.z)), d_oif = mk3(from(oif.x), from(oif.y), from(oif.z));
    const float d_tmin = from(tmin), d_best_t = from(best_t), d_scale = from(scale), d_limit = from(limit);
    const int d_best_k = __shfl(best_k, (int)d, kWave), d_sbase = __shfl(sbase, (int)d, kWave);
    const uint32_t d_slot = (uint32_t)__shfl((int)slot, (int)d, kWave);
    if (take) {
      ro = d_ro;
      rd = d_rd;
      inv = d_inv;
      oin = d_oin;
      oif = d_oif;
      neg_x = inv.x < 0.0f;
      neg_y = inv.y < 0.0f;
      neg_z = inv.z < 0.0f;
      tmin = d_tmin;
      best_t = d_best_t;
      best_k = d_best_k;
      scale = d_scale;
      limit = d_limit;
      slot = d_slot;
      cur = ((lds_u32*)s_stack)[d_sbase * kWave + (int)d];  // the bottom of the donor's stack
      sp = sbase = 0;
      ray_boxes = 0u;
      const uint32_t leader = s_leader[d];
      s_leader[threadIdx.x] = leader;
      __hip_atomic_fetch_add(&s_grp_count[leader], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      active = true;
    }
    if (can_give && drank < pairs) ++sbase;
  };

  // One step of every active lane (the body of both loops below)
  auto step = [&]() {
    // One loop iteration = one step per lane, and ONE memory round trip: the lane's current reference is either a
    // node or a leaf, both records are fetched by the SAME four 16-byte loads from a per-lane address (a 64-byte
    // node, or a 48-byte triangle record whose last 16 bytes are simply requested twice), and the stack entry the
    // lane falls back to is read from LDS meanwhile.  Vector-memory instructions and dependent round trips are
    // what this loop is bound by (lesson x): four loads and one wait per iteration, where
    // separate node and triangle phases needed seven loads and two waits.
    if (active) {
      const bool is_leaf = (cur & kLeafBit) != 0u;
      const uint32_t index = cur & ~kLeafBit;
      // The four loads are written as instructions: left to the compiler they are split by use (the leaf branch
      // needs 36 of the 64 bytes), narrowed and partly sunk into the branches -- five to seven loads again.
      // Loads, the LDS read of the stack entry the lane falls back to, and the ONE wait for all of them are a single asm
      // statement (round 4).  Until then the loads and the wait were separate statements around compiler-issued code:
      // the compiler believes a hand-issued load complete the moment it is issued and is free to spill or move a
      // destination register in between (nothing it does shows up in its wait bookkeeping either: a build of round 2
      // waited between the loads, -17 %).  As one statement there is no "in between"; early-clobber outputs keep the
      // addresses alive until the last load is out.
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      u32x4 w0, w1, w2, w3;
      // (the "fourth quarter" of a 48-byte triangle record would be the head of the next record: its last 16 bytes
      // are requested twice instead; a 64-byte record is simply read whole)
      constexpr bool kLeaf48 = kTriVec4 == 3u;
      const char* rec = is_leaf ? reinterpret_cast<const char*>(tris) + (16u * kTriVec4) * (size_t)index
                                : reinterpret_cast<const char*>(sc.cur.bvh4q) + 64u * (size_t)index;
      const char* rec3 = rec + (kLeaf48 && is_leaf ? 32 : 48);
      const int top = sp - 1;  // (peek(): the top entry without removing it)
      const uint32_t below_addr = (uint32_t)(uintptr_t)(stack + min(max(top, 0), lds_cap - 1) * kWave);
      uint32_t below;
      asm volatile(
          "global_load_dwordx4 %0, %5, off\n\t"
          "global_load_dwordx4 %1, %5, off offset:16\n\t"
          "global_load_dwordx4 %2, %5, off offset:32\n\t"
          "global_load_dwordx4 %3, %6, off\n\t"
          "ds_read_b32 %4, %7\n\t"
          "s_waitcnt vmcnt(0) lgkmcnt(0)"
          : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(below)
          : "v"(rec), "v"(rec3), "v"(below_addr)
          : "memory");
      if (__builtin_expect(top >= lds_cap, 0)) below = sc.spill[(size_t)(top - lds_cap) * sc.spill_stride + gid].x;
      below = top >= sbase ? below : kNoChild;
      const uint4 q0 = make_uint4(w0.x, w0.y, w0.z, w0.w), q1 = make_uint4(w1.x, w1.y, w1.z, w1.w);
      const uint4 q2 = make_uint4(w2.x, w2.y, w2.z, w2.w), q3 = make_uint4(w3.x, w3.y, w3.z, w3.w);
      if (!is_leaf) {
        if (kCount) ++tally.nodes;
        // 64-byte node: origin + power-of-two grid steps + 8-bit plane coordinates (Wide4Accel::nodes_q).  A plane
        // is origin + q * step, so its slab term is fma(q, step / d, fma(origin, 1/d, -o/d -+ tol)): two terms per
        // axis and node, one fma per plane.  The quantised boxes contain the exact ones, so the walk stays
        // conservative; the exact tests of the winner use the exact parent box (leaf_parent) as before.
        const f3 org = mk3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
        const f3 ax = mk3(__uint_as_float(q0.w) * inv.x, __uint_as_float(q2.z) * inv.y, __uint_as_float(q2.w) * inv.z);
        const f3 bn = mk3(__builtin_fmaf(org.x, inv.x, oin.x), __builtin_fmaf(org.y, inv.y, oin.y), __builtin_fmaf(org.z, inv.z, oin.z));
        const f3 bf = mk3(__builtin_fmaf(org.x, inv.x, oif.x), __builtin_fmaf(org.y, inv.y, oif.y), __builtin_fmaf(org.z, inv.z, oif.z));
        const uint32_t nqx = neg_x ? q1.w : q1.x, fqx = neg_x ? q1.x : q1.w;
        const uint32_t nqy = neg_y ? q2.x : q1.y, fqy = neg_y ? q1.y : q2.x;
        const uint32_t nqz = neg_z ? q2.y : q1.z, fqz = neg_z ? q1.z : q2.y;
        float key[4];
        uint32_t ref[4] = {q3.x, q3.y, q3.z, q3.w};
        // tn is a lower bound of the true entry distance and tf an upper bound of the true exit distance (the
        // tolerance is inside oin / oif), so the child can be skipped when the interval [max(tn, 0), min(tf,
        // limit)] is empty: box missed, box behind the origin (every t in it < 0 < t_min), or box beyond the
        // closest hit so far (limit carries a 0.1 % margin; ties at equal t start no farther than the hit).
        // (An unused slot carries an inside-out box and needs no test of its own, see Collapse::quantise.)
        // (Measured dead end: the 24 plane FMAs as 12 v_pk_fma_f32 -- clean code, no pair-forming moves -- run 1 %
        // SLOWER; packed f32 does not issue faster than two plain FMAs on gfx950, MI355X_MICROARCH.md constants table.)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float cnx = (float)((nqx >> (8 * c)) & 0xffu), cny = (float)((nqy >> (8 * c)) & 0xffu), cnz = (float)((nqz >> (8 * c)) & 0xffu);
          const float cfx = (float)((fqx >> (8 * c)) & 0xffu), cfy = (float)((fqy >> (8 * c)) & 0xffu), cfz = (float)((fqz >> (8 * c)) & 0xffu);
          const float tn = fmaxf(fmaxf(fmaxf(__builtin_fmaf(cnx, ax.x, bn.x), __builtin_fmaf(cny, ax.y, bn.y)),
                                       __builtin_fmaf(cnz, ax.z, bn.z)), 0.0f);
          const float tf = fminf(fminf(fminf(__builtin_fmaf(cfx, ax.x, bf.x), __builtin_fmaf(cfy, ax.y, bf.y)),
                                       __builtin_fmaf(cfz, ax.z, bf.z)), limit);
          if (kCount && ref[c] != sc.cur.dummy_ref) { ++tally.boxes; ++ray_boxes; }
          key[c] = tn <= tf ? tn : __builtin_inff();
        }
        // children nearest first: sort the four (key, ref) pairs, 5 compare-exchanges
        auto cx = [&](int a, int b) {
          const bool sw = key[b] < key[a];
          const float ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b];
          const uint32_t ra = sw ? ref[b] : ref[a], rb = sw ? ref[a] : ref[b];
          key[a] = ka;
          key[b] = kb;
          ref[a] = ra;
          ref[b] = rb;
        };
        cx(0, 1);
        cx(2, 3);
        cx(0, 2);
        cx(1, 3);
#if PT_FULL_SORT
        cx(1, 2);
#endif
        // the others go on the stack, farthest first
        if (__builtin_expect(sp + 3 <= lds_cap, 1)) {
          // room for all three in LDS: write unconditionally, advance only past the ones that count (a slot that
          // does not count is overwritten by the next write or stays above the top)
          stack[sp * kWave] = ref[3];
          sp += key[3] < __builtin_inff() ? 1 : 0;
          stack[sp * kWave] = ref[2];
          sp += key[2] < __builtin_inff() ? 1 : 0;
          stack[sp * kWave] = ref[1];
          sp += key[1] < __builtin_inff() ? 1 : 0;
        } else {
          if (key[3] < __builtin_inff()) push(ref[3]);
          if (key[2] < __builtin_inff()) push(ref[2]);
          if (key[1] < __builtin_inff()) push(ref[1]);
        }
        if (key[0] < __builtin_inff()) {
          cur = ref[0];
        } else {  // nothing was pushed: `below` is still the top
          cur = below;
          sp = max(sp - 1, sbase);
        }
      } else {
        // ray_triangle_intersection_test (intersections.cuh:49-85) on the precomputed world-space edges
        if (kCount) ++tally.tris;
        const f3 p0 = mk3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
        const f3 e1 = mk3(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
        const f3 e2 = mk3(__uint_as_float(q1.z), __uint_as_float(q1.w), __uint_as_float(q2.x));
#if PT_FLAT_TRI
        // the same operations in the same order as the nested form below, evaluated unconditionally (a lane whose
        // test fails early computes garbage that is never looked at: with a dozen lanes per wavefront in this
        // branch some lane reaches every stage anyway, so the early outs only cost their branches)
        const f3 h = cross(rd, e2);
        const float a = dot(e1, h);
        const f3 sv = ro - p0;
        const float f = 1.0f / a;
        const float u = f * dot(sv, h);
        const f3 qv = cross(sv, e1);
        const float w = f * dot(rd, qv);
        const float t = f * dot(e2, qv);
        const bool hit = !(a > -0.0000001f && a < 0.0000001f) & !(u < 0.0f || u > 1.0f) & !(w < 0.0f || u + w > 1.0f) &
                         !(t < tmin) & (t < best_t || (t == best_t && (int)index > best_k));
        best_t = hit ? t : best_t;
        best_k = hit ? (int)index : best_k;
        limit = hit ? scale * t * 1.001f : limit;
#else
        const f3 h = cross(rd, e2);
        const float a = dot(e1, h);
        const f3 sv = ro - p0;
        if (!(a > -0.0000001f && a < 0.0000001f)) {
          const float f = 1.0f / a;
          const float u = f * dot(sv, h);
          if (!(u < 0.0f || u > 1.0f)) {
            const f3 qv = cross(sv, e1);
            const float w = f * dot(rd, qv);
            if (!(w < 0.0f || u + w > 1.0f)) {
              const float t = f * dot(e2, qv);
              if (!(t < tmin) && (t < best_t || (t == best_t && (int)index > best_k))) {
                best_t = t;
                best_k = (int)index;
                limit = scale * t * 1.001f;
              }
            }
          }
        }
#endif
        cur = below;
        sp = max(sp - 1, sbase);
      }
      if (cur == kNoChild) {
        active = false;
        pending = true;
      }
    }
  };

  for (;;) {  // kPersist: (first loop, second loop, wait) until every frame of the batch is done; else one pass
  for (;;) {
    const uint64_t idle_mask = __ballot(!active);
    const uint32_t idle = (uint32_t)__popcll(idle_mask);
    const bool more = priv_next < priv_end || (kPersist ? !dry : !feed.exhausted());
#ifdef PT_TAILPROF
    ++tp_iters;
#endif
    if (!more) break;  // nothing left to fetch: the lanes still walking finish in the second loop
    if (more && (idle == (uint32_t)kWave || idle >= sc.refill_lanes)) {
      // (kPersist: when the feed has to be asked anyway, what earlier refills finished is reported with the same round trip;
      // taken BEFORE this refill's finalize -- those stores have not landed yet)
      uint32_t done_v = 0u;
      if (kPersist && priv_next >= priv_end) done_v = take_done();
      if (pending) {
        finalize();  // (no ray is shared between lanes before the second loop)
        pending = false;
      }
      if (kPersist) {
        if (priv_next >= priv_end && pfeed.acquire(priv_next, priv_end, cur_frame, cur_bounce, done_v) <= 0) {
          priv_next = priv_end = 0u;
          dry = true;
        }
      } else if (priv_next >= priv_end && !feed.acquire(priv_next, priv_end)) priv_next = priv_end = 0u;
      const uint32_t mine = priv_next + rank_below(idle_mask);
      const uint32_t range_end = priv_end;
      priv_next = min(priv_end, priv_next + idle);
      if (!active && mine < range_end) {
        float4 o4, d4;
        if (kPersist) {
          const uint32_t at = pfeed.position_of(mine);
          slot = at | (cur_frame << kPersistSlotBits);
          const DPaths& pin = pa->paths[cur_bounce & 1u];
          ld2_sc1(&pin.o4[at], &pin.d4[at], o4, d4);
        } else {
          slot = order ? order[mine] : mine;  // (k_sort_octant: the same rays, picked up in a more coherent order)
          o4 = ldnt(&paths.o4[slot]);
          d4 = ldnt(&paths.d4[slot]);
        }
        // entry points (DBeam): the tile's four boxes, requested together with the ray -- at bounce 0 the slot says which
        // pixel the ray belongs to, so the address does not wait for the ray (one round trip for both; the first version
        // took the pixel from the loaded ray and paid a second one, in front of every lane of the wavefront)
        float4 eb[kBeam ? 8 : 1];
        if (kBeam) {
          const uint32_t frame = slot / bi.stride;
          const uint32_t pixel = band_pixel(sc.beam.band, slot - frame * bi.stride);
          const uint32_t py = pixel / sc.beam.width, px = pixel - py * sc.beam.width;
          const float4* e = sc.beam.entries + ((size_t)sc.beam.beam_of[frame] * sc.beam.tiles + (size_t)(py / kBeamTile) * sc.beam.tiles_x + px / kBeamTile) * (2u * kBeamEntries);
#pragma unroll
          for (int k = 0; k < 8; ++k) eb[k] = e[k];
        }
        ro = xyz(o4);
        rd = xyz(d4);
        tmin = (__float_as_uint(o4.w) >> 31) ? 1e-5f : 1e-4f;
        float t_in = FLT_MAX;
        if (!kFirst) {
          const float carried = ldnt(&hits.tp[slot]).x;
          if (carried >= 0.0f) t_in = carried;
        }
        bool go = sc.cur.bvh_node_count != 0u;
        bool wrote = false;
        if (go) {
          // inverse_transform_ray (transform.hpp:51-58) for the walk only: the walk has to be conservative, not
          // exact, so the normalisation and the reciprocals are the hardware approximations (1 ulp) and the
          // error bound below covers them; everything that decides the result is recomputed exactly in finalize
          const f3 v = xform_vector(obj->inv_m, rd);
          const float len2 = dot(v, v);
          const float rlen = __builtin_amdgcn_rsqf(len2);
          scale = len2 * rlen;
          const f3 od = v * rlen;
          // the origin in object space for the walk: the reference's (M^-1 (o,1)).xyz / w with the division by w
          // (1 for an affine transform) as a reciprocal -- finalize recomputes the exact one where it decides
          const f4 ow = mul(obj->inv_m, ro.x, ro.y, ro.z, 1.0f);
          const f3 oo_walk = mk3(ow.x, ow.y, ow.z) * __builtin_amdgcn_rcpf(ow.w);
          inv = mk3(__builtin_amdgcn_rcpf(od.x), __builtin_amdgcn_rcpf(od.y), __builtin_amdgcn_rcpf(od.z));
          // Slab form t = fma(b, 1/d, -o/d).  Against the reference's (b - o)/d (d normalised with IEEE sqrt and
          // divide) it is off by at most ~1e-6 of |b/d| + |o/d| per axis (rsq, rcp: 1 ulp each, three roundings);
          // four times that bound (|b| <= the root box) is folded into the two origin terms so the near side
          // can only move nearer and the far side farther.
          const f3 oi = mk3(-(oo_walk.x * inv.x), -(oo_walk.y * inv.y), -(oo_walk.z * inv.z));
          const float bx = fmaxf(fabsf(sc.cur.root_min[0]), fabsf(sc.cur.root_max[0]));
          const float by = fmaxf(fabsf(sc.cur.root_min[1]), fabsf(sc.cur.root_max[1]));
          const float bz = fmaxf(fabsf(sc.cur.root_min[2]), fabsf(sc.cur.root_max[2]));
          const f3 tol = mk3(4e-6f * (fabsf(oi.x) + bx * fabsf(inv.x)) + 1e-30f,
                             4e-6f * (fabsf(oi.y) + by * fabsf(inv.y)) + 1e-30f,
                             4e-6f * (fabsf(oi.z) + bz * fabsf(inv.z)) + 1e-30f);
          if (__builtin_expect(!(finite_f(inv.x) && finite_f(inv.y) && finite_f(inv.z) && finite_f(tol.x + tol.y + tol.z)) ||
                               sc.force_slow == 1u, 0)) {
            // degenerate direction (0/0 or overflow in the slab terms voids the error bound): set aside for
            // the launch's epilogue (redo_slow_rays), which takes every box decision with the reference's own test
            if (kPersist) set_aside(counters + cur_frame, pa->slow_list + (size_t)cur_frame * bi.stride, slot & kSlotMask);
            else set_aside(counters, slow_list, slot);
            wrote = true;
            go = false;
          } else {
            oin = oi - tol;
            oif = oi + tol;
            neg_x = inv.x < 0.0f;
            neg_y = inv.y < 0.0f;
            neg_z = inv.z < 0.0f;
            best_t = t_in;
            best_k = -1;
            limit = scale * best_t * 1.001f;
            cur = sc.cur.bvh4_root;
            sp = sbase = 0;
            ray_boxes = 0u;
            if (kBeam) {
              // entry points (DBeam): the four boxes of the ray's tile, tested like the children of one node -- same
              // conservative slab arithmetic, boxes in the object's space -- and entered nearest first.  Everything the
              // tile's frustum cannot reach was left out by k_beam; what the walk finds is verified exactly as always.
              float key[4];
              uint32_t ref[4];
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const float4 lo = eb[2 * c], hi = eb[2 * c + 1];
                const float tn = fmaxf(fmaxf(fmaxf(__builtin_fmaf(neg_x ? hi.x : lo.x, inv.x, oin.x), __builtin_fmaf(neg_y ? hi.y : lo.y, inv.y, oin.y)),
                                             __builtin_fmaf(neg_z ? hi.z : lo.z, inv.z, oin.z)), 0.0f);
                const float tf = fminf(fminf(fminf(__builtin_fmaf(neg_x ? lo.x : hi.x, inv.x, oif.x), __builtin_fmaf(neg_y ? lo.y : hi.y, inv.y, oif.y)),
                                             __builtin_fmaf(neg_z ? lo.z : hi.z, inv.z, oif.z)), limit);
                key[c] = tn <= tf ? tn : __builtin_inff();
                ref[c] = __float_as_uint(lo.w);
                if (kCount && key[c] < __builtin_inff()) { ++tally.boxes; ++ray_boxes; }
              }
              auto cx = [&](int a, int b) {
                const bool sw = key[b] < key[a];
                const float ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b];
                const uint32_t ra = sw ? ref[b] : ref[a], rb = sw ? ref[a] : ref[b];
                key[a] = ka;
                key[b] = kb;
                ref[a] = ra;
                ref[b] = rb;
              };
              cx(0, 1);
              cx(2, 3);
              cx(0, 2);
              cx(1, 3);
              cx(1, 2);
              if (key[3] < __builtin_inff()) push(ref[3]);
              if (key[2] < __builtin_inff()) push(ref[2]);
              if (key[1] < __builtin_inff()) push(ref[1]);
              cur = ref[0];
              go = key[0] < __builtin_inff();  // (no entry hit: the ray passes this object)
            }
          }
        }
        if (go) active = true;
        else if (kPersist) {
          if (!wrote) st_sc1(&hits.tp[slot & kSlotMask], make_float4(-1.0f, 0.f, 0.f, 0.f));
          __hip_atomic_fetch_add(&s_pend[cur_frame], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // (done without a walk)
        } else if (kFirst && !wrote) stnt(&hits.tp[slot], make_float4(-1.0f, 0.f, 0.f, 0.f));
      }
    }
    if (__ballot(active) == 0ull) continue;
    step();
  }

  // ---- the end of the launch: no rays left to fetch.  Lanes fall idle one by one; idle lanes take over parts of the
  // busy lanes' walks (split).  A wavefront runs with few others here, so an iteration costs its dependent round trips:
  // a retire is two of its own (the winner's parent box and normal, then the stores the next wait sits out) and a split
  // is a chain of LDS round trips -- measured 1600 + 1600 cycles beside a 2400-cycle step when both ran every iteration
  // (profiles/r04_tailprof_before_*.txt) -- so finished lanes are retired a few at a time and until then are no takers.
#ifdef PT_TAILPROF
  tp_exhausted = wall_clock64();
  tp_iters_exh = tp_iters;
  tp_lanes_exh = (uint32_t)__popcll(__ballot(active));
#endif
  for (;;) {
    const uint32_t idle = (uint32_t)__popcll(__ballot(!active));
#ifdef PT_TAILPROF
    ++tp_iters;
#endif
    if (sc.split_idle != 0u && idle >= sc.split_idle && ++since_split >= PT_SPLIT_EVERY) {
      since_split = 0u;
#ifdef PT_TAILPROF
      const unsigned long long c0 = clock64();
#endif
      const uint32_t waiting = (uint32_t)__popcll(__ballot(pending));
      if (waiting != 0u && (waiting >= PT_RETIRE_LANES || ++since_retire >= PT_RETIRE_EVERY || idle == (uint32_t)kWave)) {
        since_retire = 0u;
        if (pending) {
          retire();
          pending = false;
        }
      }
#ifdef PT_TAILPROF
      const unsigned long long c1 = clock64();
#endif
      split();
#ifdef PT_TAILPROF
      ++tp_splits;
      tp_c_retire += c1 - c0;
      tp_c_split += clock64() - c1;
#endif
    }
    if (__ballot(active) == 0ull) {
      if (pending) {
        retire();
        pending = false;
      }
      break;
    }
#ifdef PT_TAILPROF
    const unsigned long long c2 = clock64();
    tp_lanes_tail += (unsigned long long)__popcll(__ballot(active));
#endif
    step();
#ifdef PT_TAILPROF
    tp_c_step += clock64() - c2;
#endif
  }
  if (!kPersist) break;
  // kPersist: this wavefront holds nothing any more.  Its counts go out; if there are rays to hand out it starts over, else
  // it goes back to its caller (k_persist), which lets it shade tiles meanwhile and calls again.
  split_mode = false;
  since_split = since_retire = 0u;
  uint32_t done_v = take_done();
  pfeed.flush(done_v);
  walk_status = pfeed.acquire(priv_next, priv_end, cur_frame, cur_bounce, done_v);
  if (walk_status <= 0) break;
  dry = false;
  }
  if (flags) atomicOr(&counters->flags, flags);
  if (kCount) flush_tally(tally, counters, bounce, false);
  return walk_status;
}

// Rays a persistent traversal launch set aside (a direction with a zero / subnormal component, or a winner whose
// parent box the ray only grazes): redone with EXACT box decisions -- the culled near-first walk
// (mesh_closest_wide: every inner box decided like the reference's ray_aabb, also for NaN / infinite slab terms),
// which returns the reference's hit in ~50 box tests instead of the ~125 (worst case thousands) of the reference's
// own order.  Run by the LAST wavefront of the traversal launch to finish (k_traverse4's epilogue): the list is
// almost always empty, and a separate one-wavefront launch for it was a bubble on the stream every bounce (it
// waited for a wavefront slot behind the other stream's persistent wavefronts: 10 % of the kernel time of round 1).
// Its traversal stack lives in global memory (DScene::slow_stack, [depth][lane]): this code is off the fast path.
template <bool kFirst>
__device__ __forceinline__ void redo_slow_rays(const DScene& sc, uint32_t obj_index, const DPaths& paths, const DHits& hits,
                                            const uint32_t* slow_list, uint32_t count, DeviceCounters* counters)
{
  const DObject* obj = sc.objects + obj_index;
  const uint32_t mat = sc.object_material[obj_index];
  const uint32_t tri_base = sc.object_tri_base[obj_index];
  uint32_t flags = 0u;
  for (uint32_t i = threadIdx.x; i < count; i += kWave) {
    const uint32_t slot = __hip_atomic_load(&slow_list[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Ray ray = load_ray(paths, slot);
    if (!kFirst) {
      const float carried = ldnt(&hits.tp[slot]).x;
      if (carried >= 0.0f) ray.tmax = carried;
    }
    float best_t = ray.tmax;
    int best_k = -1;
    Tally unused;
    // the object's world box first (path_tracer.cu:84): the persistent kernel tests it only for its winners
    if (ray_aabb(ray.o, ray.d, ld3(obj->bmin), ld3(obj->bmax)))
      mesh_closest_wide<false>(ray, sc, sc.cur, obj, tri_base, best_t, best_k, sc.slow_stack + threadIdx.x, flags, unused);
    if (best_k >= 0) {
      const float4 tc = sc.tris[kTriVec4 * ((size_t)tri_base + (uint32_t)best_k) + 2u];
      const f3 outward = mk3(tc.y, tc.z, tc.w);
      const f3 p = ray.o + ray.d * best_t;
      const uint32_t side = dot(ray.d, outward) < 0.0f ? 0u : 1u;
      const f3 nn = side == 0u ? outward : -outward;
      stnt(&hits.tp[slot], make_float4(best_t, p.x, p.y, p.z));
      stnt(&hits.nm[slot], make_float4(nn.x, nn.y, nn.z, __uint_as_float(mat | (side << 31))));
    } else if (kFirst) {
      stnt(&hits.tp[slot], make_float4(-1.0f, 0.f, 0.f, 0.f));
    }
  }
  if (flags) atomicOr(&counters->flags, flags);
}

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

// ------------------------------------------------------------------------------------------------
// the bounce-spanning persistent launch (round 5; DESIGN section 4d, DPersist in pt_device.hpp)
// ------------------------------------------------------------------------------------------------
// What the per-bounce launches lose (review of round 4): every traversal launch drains the chip on its longest rays, and the
// streaming kernels between two launches run with no traversal beside them.  Here ONE launch per batch carries the
// traversal of bounces >= 1 and the shade passes of all bounces (bounce 0's traversal keeps its own launch: entry points,
// work list).  Of every `service_every` wavefronts (by arrival: whoever runs takes the next role, nothing is assigned to a
// wavefront that may not be resident) one is a SERVICE wavefront -- it shades tiles of whichever frame's traversal phase is
// complete, lowest frame first (shade_tile<.., 1, true>: k_shade_fused's tile for one wavefront) and runs the exact redo of
// set-aside rays -- and the others WALK: traverse4_walk<.., kPersist> over PersistFeed.
// Forward progress: a walking wavefront waits for nothing but rays to hand out; a shading wavefront waits (in the look-back
// of its tile) only for tiles with lower tickets, which running wavefronts hold; T(f, b) needs S(f, b - 1), which needs
// T(f, b - 1): a chain that starts at S(f, 0), ready when the launch starts.  Any five running wavefronts contain both
// roles, so the launch ends however few of its wavefronts the chip admits at a time.  Every wait is bounded all the same.
// Tiles a service wavefront draws with one ticket.  ONE: with four consecutive tiles per draw (measured, 7 x slower) the first
// tile of a draw waits, in its look-back, for the LAST tile of the draw before it, which its wavefront has not even begun
// while it works through the three in front -- the pass turns into a chain of draws.
#ifndef PT_SERVICE_TILES
#define PT_SERVICE_TILES 1u
#endif
template <bool kSpheres>
// dedicated: a service wavefront proper (stays until every frame is done, sleeps when there is nothing to shade); else a walking
// wavefront that found no rays to hand out: it shades up to `budget` tiles and goes back to look for rays.  Returns < 0 when
// every frame is done (or the launch has given up), else the tiles it shaded.
__device__ __forceinline__ int persist_service(const DScene& sc, const uint32_t obj_index, const DPersistArgs& pa, const DHits& hits,
                                               DeviceCounters* counters, const DBatchInfo& bi, const uint32_t arrival, const bool dedicated,
                                               const uint32_t budget)
{
  uint32_t shaded = 0u;
  __shared__ uint32_t s_cnt[kFuseK];
  __shared__ uint32_t s_excl;
  DPersist* st = pa.st;
  const uint32_t lane = threadIdx.x, f = lane & 31u;
  // Frames are SPREAD over the service wavefronts: each has a home frame and takes the first frame at or after it that has
  // tiles left -- one ticket word sustains ~30 draws per microsecond, and with every wavefront on the lowest frame's word
  // (the first build) a bounce-0 pass of 7200 tiles took 500 us however many wavefronts shaded (profiles/r05_persist_*.txt).
  const uint32_t home = (dedicated ? arrival / pa.service_every : arrival) % bi.count;
  bool have = false;       // (frame, bounce, tiles, n_all, n) below: the pass this wavefront last drew a ticket of
  uint32_t frame = 0u, bounce = 0u, tiles = 0u, n_all = 0u, n = 0u;
  uint32_t spins = 0u;
  for (;;) {
    uint32_t code = 0u, cnt = 0u;
    bool is_r = false;
    if (!have) {
      unsigned long long s = (unsigned long long)kPhaseDone << 32;
      if (lane < 32u && f < bi.count) s = __hip_atomic_load(&st->state[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      code = (uint32_t)(s >> 32);
      cnt = (uint32_t)s;
      if (__ballot(code != kPhaseDone) == 0ull) return -1;
      const bool is_s = code != kPhaseDone && (code & kPhaseKindMask) == kPhaseS;
      is_r = code != kPhaseDone && (code & kPhaseKindMask) == kPhaseRedo;
      bool has = false;
      if (is_s) {
        const uint32_t t = __hip_atomic_load(&st->f[f].s_ticket[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        has = (t >> kPersistSlotBits) == (code >> kPhaseKindBits) && (t & kPersistSlotMask) < cnt;
      }
      const uint32_t m = (uint32_t)__ballot(has);  // (lanes 0..31)
      if (m != 0u) {
        const uint32_t at_or_after = m & ~((1u << home) - 1u);
        const int sel = __ffs((int)(at_or_after ? at_or_after : m)) - 1;
        frame = (uint32_t)sel;
        bounce = (uint32_t)__builtin_amdgcn_readlane((int)(code >> kPhaseKindBits), sel);
        tiles = (uint32_t)__builtin_amdgcn_readlane((int)cnt, sel);
        n_all = __hip_atomic_load(&counters[frame].live[bounce], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        n = (bounce == 0u && pa.list0) ? __hip_atomic_load(&counters[frame].list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : n_all;
        have = true;
      }
    }
    if (have) {
      uint32_t old = 0u;
      if (lane == 0u) old = __hip_atomic_fetch_add(&st->f[frame].s_ticket[0], PT_SERVICE_TILES, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
      if ((old >> kPersistSlotBits) != bounce) {  // the frame moved on meanwhile: the ticket is one of its current shade pass
        bounce = old >> kPersistSlotBits;
        // (... unless the draw came after the pass's last tile was handed out: the frame may then be anywhere BEHIND S(bounce),
        // and the draw is simply a miss.  Before S(bounce) -- the exact redo of the bounce -- the ticket is good: wait.)
        unsigned long long now = 0ull;
        bool ok = false, past = false;
        for (uint32_t w = 0u; w < (1u << 20) && !ok && !past; ++w) {
          now = __hip_atomic_load(&st->state[frame], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t c = (uint32_t)(now >> 32);
          ok = c == ((bounce << kPhaseKindBits) | kPhaseS);
          past = c == kPhaseDone || (c >> kPhaseKindBits) > bounce;
          if (!ok && !past) __builtin_amdgcn_s_sleep(2);
        }
        if (past) {
          have = false;
          continue;
        }
        if (!ok) {  // cannot be: a pass does not end before the tiles it handed out are done
          if (lane == 0u) {
            __hip_atomic_fetch_or(&st->error, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(&counters->flags, kFlagPersistStall);
          }
          break;
        }
        tiles = (uint32_t)now;
        n_all = __hip_atomic_load(&counters[frame].live[bounce], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        n = (bounce == 0u && pa.list0) ? __hip_atomic_load(&counters[frame].list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : n_all;
      }
      const uint32_t first = old & kPersistSlotMask;
      if (first >= tiles) {  // the pass has no tiles left: look again
        have = false;
        continue;
      }
      spins = 0u;
      ++shaded;
      const uint32_t end_tile = min(tiles, first + PT_SERVICE_TILES);
      // ---- tiles [first, end_tile) of (frame, bounce): k_shade_fused's arguments for this frame ----
      const size_t fo = (size_t)frame * bi.stride;
      DeviceCounters* ctr = counters + frame;
      const uint32_t* list = bounce == 0u ? pa.list0 : nullptr;
      DPaths in = pa.paths[bounce & 1u], out = pa.paths[(bounce & 1u) ^ 1u];
      in.o4 += fo;
      in.d4 += fo;
      in.t2 += fo;
      out.o4 += fo;
      out.d4 += fo;
      out.t2 += fo;
      DHits h = hits;
      h.tp += fo;
      h.nm += fo;
      DFrame fb = pa.stage;
      if (pa.staged) {
        fb.color4 += fo;
        fb.nd4 += fo;
      }
      const int last = (int)bounce == pa.max_bounces - 1 ? 1 : 0;
#pragma unroll 1
      for (uint32_t tile = first; tile < end_tile; ++tile)
        shade_tile<kSpheres, false, 1, true>(sc, pa.tail_begin, pa.tail_end, in, out, h, pa.staged, (int)bounce, last, pa.slot_base,
                                             pa.tile_desc + (size_t)frame * pa.tile_stride, pa.epoch0 + bounce, fb, pa.band, ctr, nullptr, bi.iteration[frame],
                                             list ? list + fo : nullptr, fo, tile, tiles, n, n_all, s_cnt, &s_excl);
      // ---- sign the tiles off; the last sign-off of a pass opens the frame's next traversal phase ----
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // survivors, samples and the live count are in memory
      const uint32_t mine = end_tile - first;
      uint32_t done = 0u;
      if (lane == 0u) done = __hip_atomic_fetch_add(&st->f[frame].s_done[0], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      done = (uint32_t)__builtin_amdgcn_readfirstlane((int)done);
      if (done + mine == tiles) {
        have = false;
        const uint32_t next = bounce + 1u;
        const uint32_t live = last ? 0u : __hip_atomic_load(&ctr->live[next], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef PT_PERSIST_DEBUG
        if (lane == 0u) {
          st->dbg[frame][bounce & 15u][3] = tiles;
          st->dbg[frame][bounce & 15u][4] = (uint32_t)wall_clock64();   // S(bounce) complete = T(bounce + 1) opens
          st->dbg[frame][bounce & 15u][5] = live;
        }
#endif
        if (live == 0u) {
          if (lane == 0u) {
            __hip_atomic_store(&st->state[frame], (unsigned long long)kPhaseDone << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&st->frames_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        } else {
          // order: the done counter, then the state word, then the cursors' tags (see the hand-over to a shade pass)
          if (lane == 0u) __hip_atomic_store(&st->f[frame].t_done[0], next << kPersistSlotBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0u)
            __hip_atomic_store(&st->state[frame], ((unsigned long long)((next << kPhaseKindBits) | kPhaseT) << 32) | live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane < 8u) __hip_atomic_store(&st->f[frame].cursor[lane][0], next << kPersistSlotBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (!dedicated && shaded >= budget) return (int)shaded;
      continue;
    }
    const uint64_t mr = __ballot(is_r);
    if (mr != 0ull) {
      // the exact redo of the rays a frame's traversal phase set aside (rare: a handful per batch), one frame at a time
      // (redo_slow_rays' stack is one per launch): claim the frame, take the lock, walk, publish, open the shade pass
      const int sel = __ffsll((unsigned long long)mr) - 1;
      const uint32_t rframe = (uint32_t)sel;
      const uint32_t rcode = (uint32_t)__builtin_amdgcn_readlane((int)code, sel), rtiles = (uint32_t)__builtin_amdgcn_readlane((int)cnt, sel);
      const uint32_t rbounce = rcode >> kPhaseKindBits;
      unsigned long long expect = ((unsigned long long)rcode << 32) | rtiles;
      bool mine = false;
      if (lane == 0u)
        mine = __hip_atomic_compare_exchange_strong(&st->state[rframe], &expect, ((unsigned long long)((rbounce << kPhaseKindBits) | kPhaseRedoing) << 32) | rtiles,
                                                    __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__ballot(mine) == 0ull) continue;
      bool locked = false;
      for (uint32_t w = 0u; w < (1u << 22) && !locked; ++w) {
        uint32_t zero = 0u;
        bool got = false;
        if (lane == 0u) got = __hip_atomic_compare_exchange_strong(&st->redo_lock, &zero, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        locked = __ballot(got) != 0ull;
        if (!locked) __builtin_amdgcn_s_sleep(20);
      }
      if (!locked) {
        if (lane == 0u) {
          __hip_atomic_fetch_or(&st->error, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicOr(&counters->flags, kFlagPersistStall);
        }
        break;
      }
      DeviceCounters* ctr = counters + rframe;
      const uint32_t count = __hip_atomic_load(&ctr->slow_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      redo_slow_rays<true>(sc, obj_index, pa.paths[rbounce & 1u], hits, pa.slow_list + (size_t)rframe * bi.stride, count, counters);
      // (its hit records are plain non-temporal stores: an agent-scope release writes them back before anybody is told)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0u) {
        __hip_atomic_store(&ctr->slow_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(&counters->slow_rays[rbounce], (unsigned long long)count);
        __hip_atomic_store(&st->redo_lock, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0u)
        __hip_atomic_store(&st->state[rframe], ((unsigned long long)((rbounce << kPhaseKindBits) | kPhaseS) << 32) | rtiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      spins = 0u;
      continue;
    }
    // nothing to do right now
    if (!dedicated) return (int)shaded;
    __builtin_amdgcn_s_sleep(20);
    ++spins;
    if ((spins & 255u) == 255u && __hip_atomic_load(&st->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
    if (spins > (1u << 22)) {
      if (lane == 0u) {
        __hip_atomic_fetch_or(&st->error, 16u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicOr(&counters->flags, kFlagPersistStall);
      }
      break;
    }
  }
  return -1;  // (gave up: the launch's error word is set)
}

template <bool kSpheres>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(PT_T4_WAVES, PT_T4_WAVES)))
void k_persist(DScene sc, uint32_t obj_index, DHits hits, DeviceCounters* counters, DBatchInfo bi, DPersistArgs pa)
{
  uint32_t arrival = 0u;
  if (threadIdx.x == 0u) arrival = __hip_atomic_fetch_add(&pa.st->started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  arrival = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrival);
  if (arrival % pa.service_every == pa.service_every - 1u) {
    (void)persist_service<kSpheres>(sc, obj_index, pa, hits, counters, bi, arrival, true, 0u);
    return;
  }
  // A walking wavefront: walk while there are rays to hand out; when there are none, shade a few tiles (the service
  // wavefronts proper guarantee that the shade passes move while everybody walks -- these make them wide when the walk has
  // nothing to do: at the launch's start, all of bounce 0's passes; later whatever keeps a frame from its next bounce);
  // when there is neither, sleep.  Bounded like every wait of the launch.
  const DPaths unused{nullptr, nullptr, nullptr};
  for (uint32_t spins = 0u;;) {
    const int walked = traverse4_walk<false, true, false, true>(sc, obj_index, unused, hits, 0, 0, counters, nullptr, nullptr, bi, false, &pa);
    if (walked < 0) break;
    const int shaded = pa.help_tiles ? persist_service<kSpheres>(sc, obj_index, pa, hits, counters, bi, arrival, false, pa.help_tiles) : 0;
    if (shaded < 0) break;
    if (shaded > 0) {
      spins = 0u;
      continue;
    }
    __builtin_amdgcn_s_sleep(20);
    if ((++spins & 255u) == 255u && __hip_atomic_load(&pa.st->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
      if (threadIdx.x == 0u) atomicOr(&counters->flags, kFlagPersistStall);  // (somebody gave up: the host is told)
      break;
    }
    if (spins > (1u << 22)) {
      if (threadIdx.x == 0u) {
        __hip_atomic_fetch_or(&pa.st->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicOr(&counters->flags, kFlagPersistStall);
      }
      break;
    }
  }
}

// The state of a batch's persistent launch, set up on the device (one wavefront, in stream order behind bounce 0's
// traversal launch): every frame starts in S(0) with the tiles of what bounce 0's shade pass walks.
__global__ __launch_bounds__(kWave) void k_persist_init(DPersist* st, DeviceCounters* counters, DBatchInfo bi, int listed0)
{
  const uint32_t f = threadIdx.x;
#ifdef PT_PERSIST_DEBUG
  for (uint32_t i = f; i < (uint32_t)kMaxBatch * 16u * 8u; i += (uint32_t)kWave) (&st->dbg[0][0][0])[i] = 0u;
#endif
#ifdef PT_PERSIST_DEBUG
  if (f == 0u) st->dbg[0][15][0] = (uint32_t)wall_clock64();
#endif
  if (f == 0u) {
    st->started = 0u;
    st->frames_done = 0u;
    st->redo_lock = 0u;
    st->error = 0u;
  }
  if (f >= (uint32_t)kMaxBatch) return;
  unsigned long long state = (unsigned long long)kPhaseDone << 32;
  if (f < bi.count) {
    DeviceCounters* ctr = counters + f;
    const uint32_t n_all = ctr->live[0];
    const uint32_t n = listed0 ? ctr->list_count : n_all;
    const uint32_t tiles = (n + kServiceTile - 1u) / kServiceTile;
    for (int r = 0; r < 8; ++r) st->f[f].cursor[r][0] = 0u;
    st->f[f].t_done[0] = 0u;
    st->f[f].s_ticket[0] = 0u;
    st->f[f].s_done[0] = 0u;
    ctr->slow_count = 0u;
    if (tiles == 0u) {  // nothing alive (k_shade_fused's own case): nothing follows
      ctr->live[1] = 0u;
      ctr->rays_total += n_all;
      ctr->paths[0] += n_all;
    } else {
      state = ((unsigned long long)((0u << kPhaseKindBits) | kPhaseS) << 32) | tiles;
    }
  }
  st->state[f] = state;
}

#include "pt_traverse4m.inc"

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
void launch_traverse_run(hipStream_t s, const DScene& scene, uint32_t obj_begin, uint32_t obj_end, bool first, DPaths paths,
                         DHits hits, int bounce, int work_slot, DeviceCounters* counters, bool count_tests, uint32_t waves,
                         uint32_t* slow_list, const uint32_t* order, const DBatchInfo& bi, bool listed)
{
  const dim3 grid(waves), block(kWave);
  if (count_tests) {
    if (first) hipLaunchKernelGGL((k_traverse4m<true, true>), grid, block, 0, s, scene, obj_begin, obj_end, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
    else hipLaunchKernelGGL((k_traverse4m<true, false>), grid, block, 0, s, scene, obj_begin, obj_end, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
  } else {
    if (first) hipLaunchKernelGGL((k_traverse4m<false, true>), grid, block, 0, s, scene, obj_begin, obj_end, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
    else hipLaunchKernelGGL((k_traverse4m<false, false>), grid, block, 0, s, scene, obj_begin, obj_end, paths, hits, bounce, work_slot, counters, slow_list, order, bi, listed ? 1 : 0);
  }
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
void launch_persist(hipStream_t s, const DScene& scene, uint32_t obj_index, DHits hits, DeviceCounters* counters, const DBatchInfo& bi,
                    const DPersistArgs& pa, uint32_t waves, bool spheres, bool listed0)
{
  hipLaunchKernelGGL(k_persist_init, dim3(1), dim3(kWave), 0, s, pa.st, counters, bi, listed0 ? 1 : 0);
  const dim3 grid(waves), block(kWave);
  if (spheres) hipLaunchKernelGGL((k_persist<true>), grid, block, 0, s, scene, obj_index, hits, counters, bi, pa);
  else hipLaunchKernelGGL((k_persist<false>), grid, block, 0, s, scene, obj_index, hits, counters, bi, pa);
}
uint32_t persist_tiles_per_frame(uint32_t max_paths) { return div_up(max_paths, kServiceTile); }
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
