// pt_traverse_run.hip -- the closest-hit launch over a run of instances of one mesh (k_traverse4m, pt_traverse4m.inc;
// "merge_instances", config 2's two instances in one launch per bounce) and its launcher.  The ray feed, the set-aside list and
// the launch's epilogue are pt_walk.inc's.  Part of libptcore.so.
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
#include "pt_walk.inc"
#include "pt_traverse4m.inc"

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
}  // namespace pt
