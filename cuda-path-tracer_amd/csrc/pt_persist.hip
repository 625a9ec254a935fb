// pt_persist.hip -- the bounce-spanning persistent launch (round 5; DESIGN section 4d, DPersist in pt_device.hpp): k_persist,
// k_persist_init and their launcher.  Walks with pt_walk.inc's traverse4_walk<.., kPersist> over PersistFeed and shades with
// pt_shade_tile.inc's shade_tile<.., 1, true>.  Part of libptcore.so; "persist" is 0 by default.
#include "pt_device.hpp"
#include "pt_rng.hpp"
#include "pt_beam_rules.hpp"
#include "pt_feed_rules.hpp"
#include <float.h>

#ifdef PT_TAILPROF
#undef PT_TAILPROF  // (the per-wavefront timeline belongs to k_traverse4's unit, pt_kernels.hip)
#endif

namespace pt {

#include "pt_kernels_common.inc"
#include "pt_shade_tile.inc"
#include "pt_walk.inc"

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

static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }
void launch_persist(hipStream_t s, const DScene& scene, uint32_t obj_index, DHits hits, DeviceCounters* counters, const DBatchInfo& bi,
                    const DPersistArgs& pa, uint32_t waves, bool spheres, bool listed0)
{
  hipLaunchKernelGGL(k_persist_init, dim3(1), dim3(kWave), 0, s, pa.st, counters, bi, listed0 ? 1 : 0);
  const dim3 grid(waves), block(kWave);
  if (spheres) hipLaunchKernelGGL((k_persist<true>), grid, block, 0, s, scene, obj_index, hits, counters, bi, pa);
  else hipLaunchKernelGGL((k_persist<false>), grid, block, 0, s, scene, obj_index, hits, counters, bi, pa);
}
uint32_t persist_tiles_per_frame(uint32_t max_paths) { return div_up(max_paths, kServiceTile); }
}  // namespace pt
