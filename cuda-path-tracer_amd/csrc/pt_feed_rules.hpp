// pt_feed_rules.hpp -- how a traversal launch's rays are dealt to its persistent wavefronts: the geometry of the feed's
// regions (BatchFeed, pt_kernels.hip).  Shared by the kernels and the host's check (ptc_check_feed, tests/test_abi_cpu.py):
// every ray of a frame must belong to exactly one region, a region's rays must be reachable in batches that are contiguous
// in the frame's order, and the static and the dynamic part of a region must not overlap.
#pragma once

#include <stdint.h>

#include "pt_math.hpp"

#ifndef PT_REGION_BLOCKS
#define PT_REGION_BLOCKS 1
#endif

namespace pt {
namespace feed_rules {

constexpr uint32_t kBatch = 64u;  // one wavefront's worth of rays (kWave)

#if PT_REGION_BLOCKS
// The eight regions of a frame's rays are INTERLEAVED (round 4): the rays, in slot / list order, are cut into blocks of
// B batches of 64, and block j belongs to region j mod 8; a region walks its blocks in order.  Every region then holds
// the same mix of the image, top to bottom -- with contiguous eighths the most expensive eighth of a primary-ray
// launch (the rows at the horizon) was the last to run dry, the last rays fetched were the launch's longest, and
// the launch ended 570 us after its feed (profiles/r04_tail_bounce0.txt); now the last rays of every region are
// the image's bottom rows.  B = 256 batches (eight image rows) when the frame has that many, fewer for a small frame
// so that every region still gets eight blocks; a power of two >= 2, so a dynamic batch of 128 rays that starts on an
// even batch never straddles two blocks.  A region-local ray offset is mapped to its position in the frame's order by
// pos_of.  `rs` ("region size") is B here.
PT_HD uint32_t region_size_of(uint32_t n)
{
  const uint32_t want = ((n + kBatch - 1u) / kBatch) / 64u;  // batches / (8 regions x 8 blocks)
  uint32_t b = 2u;
  while (b < 256u && 2u * b <= want) b *= 2u;
  return b;
}
PT_HD uint32_t region_len_of(uint32_t n, uint32_t bb, uint32_t r)
{
  const uint32_t nb = (n + kBatch - 1u) / kBatch;  // batches of the frame
  if (nb == 0u) return 0u;
  const uint32_t nblk = (nb + bb - 1u) / bb;
  if (r >= nblk) return 0u;
  uint32_t batches = ((nblk - r + 7u) / 8u) * bb;
  const uint32_t last_blk = nblk - 1u;
  if ((last_blk & 7u) == r) {
    batches -= bb - (nb - last_blk * bb);          // the frame's last block may be short
    return batches * kBatch - (nb * kBatch - n);   // ... and its last batch
  }
  return batches * kBatch;
}
PT_HD uint32_t pos_of(uint32_t bb, uint32_t r, uint32_t local)
{
  const uint32_t k = local / kBatch;
  return (((k / bb) * 8u + r) * bb + k % bb) * kBatch + local % kBatch;
}
#else
// contiguous eighths (until round 4); rs = rays per region
PT_HD uint32_t region_size_of(uint32_t n) { return ((n + 8u * kBatch - 1u) / (8u * kBatch)) * kBatch; }
PT_HD uint32_t region_len_of(uint32_t n, uint32_t rs, uint32_t r)
{
  const uint32_t b = r * rs;
  return b < n ? (n - b < rs ? n - b : rs) : 0u;
}
PT_HD uint32_t pos_of(uint32_t rs, uint32_t r, uint32_t local) { return r * rs + local; }
#endif

// batches of a region that are dealt statically (static_eighths / 8 of them; an even number: a dynamic batch of two then
// starts on an even batch)
PT_HD uint32_t static_batches_of(uint32_t len, uint32_t static_eighths)
{
  const uint32_t share = ((len + kBatch - 1u) / kBatch) * static_eighths / 8u, full = len / kBatch;  // (static batches are full batches)
  return (share < full ? share : full) & ~1u;
}

}  // namespace feed_rules
}  // namespace pt
