// pt_bvh_gpu.hip -- the reference's BVH (accelerators/bvh.cpp:60-253) built on the GPU, level by level.
//
// Same tree as pt_bvh.cpp / the reference, node for node and bit for bit: every decision of the builder (split
// axis, SAH bucket, least-cost split, median of a small node) is a function of the SET of triangles in a node, taken
// here from pt_bvh_rules.hpp -- the very functions the host builder calls, compiled without contraction and with
// IEEE division.  What differs is the schedule: the reference recurses (one shared_ptr node per triangle, serial,
// then a breadth-first renumbering); here all nodes of one depth are split at once, and since the nodes of a depth
// are kept in left-to-right order the level-order numbering IS the reference's breadth-first numbering
// (bvh.cpp:228-250): root 0, the children of the r-th inner node of a level at (next level) + 2r, + 2r + 1.
//
// Layout: the triangle ids live in one array of T positions; a node owns a contiguous range [start, start + count).
// Per level:
//   k_centroid_bounds   nodes of more than 32: bounds of the centroids (order-preserving integer min/max atomics,
//                       one set per workgroup of 1024 positions / per wavefront where these lie inside one node)
//   k_bins              nodes of more than 32: each triangle's SAH bucket; per bucket count and box (LDS bins when a
//                       workgroup lies inside one node, global atomics otherwise: contention only exists at the top)
//   k_split_nodes       one thread per node: leaf record, or axis / split and the two child ranges (rank among the
//                       inner nodes of the level from a scan -> child indices); a node of at most 32 triangles gets
//                       its centroid bounds and buckets from this thread, too (no atomics on the deep levels)
//   k_flags + scan      stable partition ranks of the nodes of more than 4
//   k_partition         every triangle moves to its side (nodes of <= 4: to its rank in the (centroid, index) order)
// -0.0 and +0.0 compare equal in the reference's min / max and the order of visits decides which survives; the integer
// atomics order them (-0 < +0).  A mesh that has both signs of zero as coordinates can differ in the sign of a zero
// box bound; none of the builder's decisions depends on it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "../../include/ptcore.h"
#include "pt_bvh_rules.hpp"
#include "pt_device.hpp"
#include "pt_layout_rules.hpp"

namespace pt {
using namespace bvh_rules;
namespace {

constexpr uint32_t kFinished = 0xffffffffu;  // owner of a position whose leaf exists
constexpr uint32_t kBinWords = kBuckets * 7u;  // per bucket: count, lo.xyz, hi.xyz (encoded)
// Nodes of at most this many triangles are split by ONE thread (k_split_nodes walks their triangles itself): below it
// the atomics of k_centroid_bounds / k_bins cost more than they parallelise (the deep levels, where every triangle
// would issue 13 of them: two thirds of the build time before this cut).
constexpr uint32_t kSerialNode = 32u;

// order-preserving map float -> uint32 (all non-NaN values): integer atomicMin / atomicMax are float min / max
__device__ __forceinline__ uint32_t enc(float f)
{
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e); }

struct Levels {  // device arrays of the build
  // per triangle
  float4* tri_lo;  // box min (w unused)
  float4* tri_hi;
  float4* center;
  // per position (ping-pong)
  uint32_t* prim[2];
  uint32_t* owner[2];  // node id of the CURRENT level the position belongs to, kFinished once it is a leaf
  uint8_t* bucket;
  uint32_t* flag;
  uint32_t* flag_scan;
  // per node (global, level-order id)
  uint32_t* start;
  uint32_t* count;
  uint32_t* inner;   // 1: count >= 2
  uint32_t* rank;    // exclusive scan of `inner` within the level
  uint32_t* info;    // axis | best bucket << 8  (inner nodes)
  uint32_t* left;    // index of the left child
  uint32_t* cb;      // 6 words per node: encoded centroid bounds
  uint32_t* bins;    // kBinWords per node of more than 4, at index start / 5
  float4* out;       // two float4 per node: {min, first}, {max, count}
  uint32_t* status;  // [0] error flag, [1] scratch total
};

__global__ __launch_bounds__(256) void k_tri_setup(const float* positions, const uint32_t* indices, uint32_t T, Levels lv)
{
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= T) return;
  const uint32_t i0 = indices[3u * t], i1 = indices[3u * t + 1u], i2 = indices[3u * t + 2u];
  const f3 p0 = mk3(positions[3u * (size_t)i0], positions[3u * (size_t)i0 + 1u], positions[3u * (size_t)i0 + 2u]);
  const f3 p1 = mk3(positions[3u * (size_t)i1], positions[3u * (size_t)i1 + 1u], positions[3u * (size_t)i1 + 2u]);
  const f3 p2 = mk3(positions[3u * (size_t)i2], positions[3u * (size_t)i2 + 1u], positions[3u * (size_t)i2 + 2u]);
  const Box b = triangle_box(p0, p1, p2);
  const f3 c = box_center(b);
  lv.tri_lo[t] = make_float4(b.lo.x, b.lo.y, b.lo.z, 0.f);
  lv.tri_hi[t] = make_float4(b.hi.x, b.hi.y, b.hi.z, 0.f);
  lv.center[t] = make_float4(c.x, c.y, c.z, 0.f);
  lv.prim[0][t] = t;
  lv.owner[0][t] = 0u;
}

__device__ __forceinline__ void init_node_accumulators(const Levels& lv, uint32_t g, uint32_t start, uint32_t count)
{
  if (count > kSerialNode) {
    const uint32_t lo = enc(FLT_MAX), hi = enc(-FLT_MAX);
    uint32_t* cb = lv.cb + 6u * (size_t)g;
    cb[0] = cb[1] = cb[2] = lo;
    cb[3] = cb[4] = cb[5] = hi;
  }
  if (count > kSerialNode) {
    uint32_t* bins = lv.bins + (size_t)(start / 5u) * kBinWords;
    const uint32_t lo = enc(FLT_MAX), hi = enc(-FLT_MAX);
    for (int b = 0; b < kBuckets; ++b) {
      bins[7 * b] = 0u;
      bins[7 * b + 1] = bins[7 * b + 2] = bins[7 * b + 3] = lo;
      bins[7 * b + 4] = bins[7 * b + 5] = bins[7 * b + 6] = hi;
    }
  }
}

__global__ void k_root(uint32_t T, Levels lv)
{
  lv.start[0] = 0u;
  lv.count[0] = T;
  lv.inner[0] = T >= 2u ? 1u : 0u;
  lv.status[0] = 0u;
  init_node_accumulators(lv, 0u, 0u, T);
}

__device__ __forceinline__ Box load_cb(const Levels& lv, uint32_t g)
{
  const uint32_t* cb = lv.cb + 6u * (size_t)g;
  return Box{mk3(dec(cb[0]), dec(cb[1]), dec(cb[2])), mk3(dec(cb[3]), dec(cb[4]), dec(cb[5]))};
}

// The centroid pass gives a workgroup 1024 consecutive positions (4 per thread).  Where all of them belong to one node
// -- the top levels, where thousands of wavefronts would otherwise queue on the same six addresses -- the workgroup
// reduces in registers and LDS first and issues one set of global atomics (5.5 -> 1.4 ms per 1M-triangle build).
// (The bucket pass keeps 256 positions per workgroup: with 1024 fewer workgroups lie inside one node on the middle
// levels and fall back to global atomics -- measured 60 % slower.)
constexpr uint32_t kPerBlock = 1024u;

__global__ __launch_bounds__(256) void k_centroid_bounds(uint32_t T, int src, Levels lv)
{
  __shared__ uint32_t s_cb[6];
  __shared__ uint32_t s_first, s_mixed;
  uint32_t g[4];
  f3 c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t i = blockIdx.x * kPerBlock + (uint32_t)j * 256u + threadIdx.x;
    g[j] = kFinished;
    c[j] = mk3(0.f, 0.f, 0.f);
    if (i < T) {
      g[j] = lv.owner[src][i];
      if (g[j] != kFinished && lv.count[g[j]] <= kSerialNode) g[j] = kFinished;
    }
    if (g[j] != kFinished) {
      const float4 c4 = lv.center[lv.prim[src][i]];
      c[j] = mk3(c4.x, c4.y, c4.z);
    }
  }
  if (threadIdx.x == 0u) {
    s_first = g[0];
    s_mixed = 0u;
  }
  if (threadIdx.x < 6u) s_cb[threadIdx.x] = threadIdx.x < 3u ? enc(FLT_MAX) : enc(-FLT_MAX);
  __syncthreads();
  if (g[0] != s_first || g[1] != s_first || g[2] != s_first || g[3] != s_first) s_mixed = 1u;
  __syncthreads();
  if (s_mixed == 0u) {
    if (s_first == kFinished) return;
    f3 lo = min3(min3(c[0], c[1]), min3(c[2], c[3])), hi = max3(max3(c[0], c[1]), max3(c[2], c[3]));
    for (int off = 32; off >= 1; off >>= 1) {
      lo = min3(lo, mk3(__shfl_xor(lo.x, off, 64), __shfl_xor(lo.y, off, 64), __shfl_xor(lo.z, off, 64)));
      hi = max3(hi, mk3(__shfl_xor(hi.x, off, 64), __shfl_xor(hi.y, off, 64), __shfl_xor(hi.z, off, 64)));
    }
    if ((threadIdx.x & 63u) == 0u) {
      atomicMin(&s_cb[0], enc(lo.x)); atomicMin(&s_cb[1], enc(lo.y)); atomicMin(&s_cb[2], enc(lo.z));
      atomicMax(&s_cb[3], enc(hi.x)); atomicMax(&s_cb[4], enc(hi.y)); atomicMax(&s_cb[5], enc(hi.z));
    }
    __syncthreads();
    if (threadIdx.x < 6u) {
      uint32_t* cb = lv.cb + 6u * (size_t)s_first + threadIdx.x;
      if (threadIdx.x < 3u) atomicMin(cb, s_cb[threadIdx.x]);
      else atomicMax(cb, s_cb[threadIdx.x]);
    }
    return;
  }
  // several nodes in this workgroup: per row of 256 positions, a wavefront inside one node reduces first
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)g[j]);
    if (__all(g[j] == g0)) {
      if (g0 == kFinished) continue;
      f3 lo = c[j], hi = c[j];
      for (int off = 32; off >= 1; off >>= 1) {
        lo = min3(lo, mk3(__shfl_xor(lo.x, off, 64), __shfl_xor(lo.y, off, 64), __shfl_xor(lo.z, off, 64)));
        hi = max3(hi, mk3(__shfl_xor(hi.x, off, 64), __shfl_xor(hi.y, off, 64), __shfl_xor(hi.z, off, 64)));
      }
      if ((threadIdx.x & 63u) == 0u) {
        uint32_t* cb = lv.cb + 6u * (size_t)g0;
        atomicMin(cb + 0, enc(lo.x)); atomicMin(cb + 1, enc(lo.y)); atomicMin(cb + 2, enc(lo.z));
        atomicMax(cb + 3, enc(hi.x)); atomicMax(cb + 4, enc(hi.y)); atomicMax(cb + 5, enc(hi.z));
      }
    } else if (g[j] != kFinished) {
      uint32_t* cb = lv.cb + 6u * (size_t)g[j];
      atomicMin(cb + 0, enc(c[j].x)); atomicMin(cb + 1, enc(c[j].y)); atomicMin(cb + 2, enc(c[j].z));
      atomicMax(cb + 3, enc(c[j].x)); atomicMax(cb + 4, enc(c[j].y)); atomicMax(cb + 5, enc(c[j].z));
    }
  }
}

__global__ __launch_bounds__(256) void k_bins(uint32_t T, int src, Levels lv)
{
  __shared__ uint32_t s_bins[kBinWords];
  __shared__ uint32_t s_first, s_mixed;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  uint32_t g = kFinished;
  if (i < T) {
    g = lv.owner[src][i];
    if (g != kFinished && lv.count[g] <= kSerialNode) g = kFinished;
  }
  if (threadIdx.x == 0u) {
    s_first = g;
    s_mixed = 0u;
  }
  if (threadIdx.x < kBinWords) {
    const uint32_t w = threadIdx.x % 7u;
    s_bins[threadIdx.x] = w == 0u ? 0u : (w <= 3u ? enc(FLT_MAX) : enc(-FLT_MAX));
  }
  __syncthreads();
  if (g != s_first) s_mixed = 1u;
  __syncthreads();
  const bool in_lds = s_mixed == 0u;  // the whole workgroup lies inside one node (or has nothing to do)
  if (in_lds && s_first == kFinished) return;
  int b = 0;
  float4 lo = make_float4(0, 0, 0, 0), hi = lo;
  if (g != kFinished) {
    const uint32_t t = lv.prim[src][i];
    const float4 c4 = lv.center[t];
    const Box cb = load_cb(lv, g);
    const int axis = widest_axis(cb);
    b = bucket_of(cb, mk3(c4.x, c4.y, c4.z), axis);
    const float o = offset_along(cb, mk3(c4.x, c4.y, c4.z), axis);
    if (b < 0 || b >= kBuckets || !(o == o)) {  // non-finite centroid: the host builder fails the same way
      lv.status[0] = 1u;
      b = 0;
    }
    lv.bucket[i] = (uint8_t)b;
    lo = lv.tri_lo[t];
    hi = lv.tri_hi[t];
    uint32_t* bins = in_lds ? s_bins + 7 * b : lv.bins + (size_t)(lv.start[g] / 5u) * kBinWords + 7 * b;
    atomicAdd(bins, 1u);
    atomicMin(bins + 1, enc(lo.x)); atomicMin(bins + 2, enc(lo.y)); atomicMin(bins + 3, enc(lo.z));
    atomicMax(bins + 4, enc(hi.x)); atomicMax(bins + 5, enc(hi.y)); atomicMax(bins + 6, enc(hi.z));
  }
  if (!in_lds) return;
  __syncthreads();
  if (threadIdx.x < kBinWords) {
    const uint32_t bucket = threadIdx.x / 7u, w = threadIdx.x % 7u;
    if (s_bins[7u * bucket] != 0u) {  // empty buckets of this workgroup change nothing
      uint32_t* dst = lv.bins + (size_t)(lv.start[s_first] / 5u) * kBinWords + threadIdx.x;
      const uint32_t v = s_bins[threadIdx.x];
      if (w == 0u) atomicAdd(dst, v);
      else if (w <= 3u) atomicMin(dst, v);
      else atomicMax(dst, v);
    }
  }
}

// One thread per node of the level [base, base + m).
__global__ __launch_bounds__(128) void k_split_nodes(uint32_t base, uint32_t m, uint32_t next_base, int src, Levels lv)
{
  const uint32_t k = blockIdx.x * 128u + threadIdx.x;
  if (k >= m) return;
  const uint32_t g = base + k;
  const uint32_t n = lv.count[g], s = lv.start[g];
  if (n == 1u) {  // leaf: one triangle, `first` = offset into the index array (bvh.hpp:17-28)
    const uint32_t t = lv.prim[src][s];
    const float4 lo = lv.tri_lo[t], hi = lv.tri_hi[t];
    lv.out[2u * (size_t)g] = make_float4(lo.x, lo.y, lo.z, __uint_as_float(t * 3u));
    lv.out[2u * (size_t)g + 1u] = make_float4(hi.x, hi.y, hi.z, __uint_as_float(1u));
    return;
  }
  // centroid bounds: accumulated by k_centroid_bounds for the big nodes, by this thread for the others
  Box cb = empty_box();
  if (n > kSerialNode) {
    cb = load_cb(lv, g);
  } else {
    for (uint32_t j = 0; j < n; ++j) {
      const float4 c4 = lv.center[lv.prim[src][s + j]];
      cb = grow(cb, mk3(c4.x, c4.y, c4.z));
    }
  }
  const int axis = widest_axis(cb);
  Box all = empty_box();
  uint32_t mid, best = 0u;
  if (n <= 4u) {
    for (uint32_t j = 0; j < n; ++j) {
      const uint32_t t = lv.prim[src][s + j];
      const float4 lo = lv.tri_lo[t], hi = lv.tri_hi[t];
      all = merge(all, Box{mk3(lo.x, lo.y, lo.z), mk3(hi.x, hi.y, hi.z)});
    }
    mid = n / 2u;
  } else {
    int count[kBuckets];
    Box bounds[kBuckets];
    if (n > kSerialNode) {
      const uint32_t* bins = lv.bins + (size_t)(s / 5u) * kBinWords;
      for (int b = 0; b < kBuckets; ++b) {
        count[b] = (int)bins[7 * b];
        bounds[b] = Box{mk3(dec(bins[7 * b + 1]), dec(bins[7 * b + 2]), dec(bins[7 * b + 3])),
                        mk3(dec(bins[7 * b + 4]), dec(bins[7 * b + 5]), dec(bins[7 * b + 6]))};
      }
    } else {
      for (int b = 0; b < kBuckets; ++b) {
        count[b] = 0;
        bounds[b] = empty_box();
      }
      for (uint32_t j = 0; j < n; ++j) {
        const uint32_t t = lv.prim[src][s + j];
        const float4 c4 = lv.center[t];
        const f3 c = mk3(c4.x, c4.y, c4.z);
        int b = bucket_of(cb, c, axis);
        const float o = offset_along(cb, c, axis);
        if (b < 0 || b >= kBuckets || !(o == o)) {
          lv.status[0] = 1u;
          b = 0;
        }
        lv.bucket[s + j] = (uint8_t)b;
        const float4 lo = lv.tri_lo[t], hi = lv.tri_hi[t];
        ++count[b];
        bounds[b] = merge(bounds[b], Box{mk3(lo.x, lo.y, lo.z), mk3(hi.x, hi.y, hi.z)});
      }
    }
    for (int b = 0; b < kBuckets; ++b) all = merge(all, bounds[b]);
    best = (uint32_t)sah_best_split(count, bounds, all);
    mid = 0u;
    for (uint32_t b = 0; b <= best; ++b) mid += (uint32_t)count[b];
    if (mid == 0u || mid == n) {  // reference: panic("Shouldn't happen!"), bvh.cpp:84-85
      lv.status[0] = 1u;
      mid = n / 2u;  // keeps the build finite; the result is discarded
    }
  }
  const uint32_t l = next_base + 2u * lv.rank[g];
  lv.info[g] = (uint32_t)axis | (best << 8);
  lv.left[g] = l;
  lv.start[l] = s;
  lv.count[l] = mid;
  lv.inner[l] = mid >= 2u ? 1u : 0u;
  lv.start[l + 1u] = s + mid;
  lv.count[l + 1u] = n - mid;
  lv.inner[l + 1u] = n - mid >= 2u ? 1u : 0u;
  init_node_accumulators(lv, l, s, mid);
  init_node_accumulators(lv, l + 1u, s + mid, n - mid);
  lv.out[2u * (size_t)g] = make_float4(all.lo.x, all.lo.y, all.lo.z, __uint_as_float(l));
  lv.out[2u * (size_t)g + 1u] = make_float4(all.hi.x, all.hi.y, all.hi.z, __uint_as_float(0u));
}

__global__ __launch_bounds__(256) void k_flags(uint32_t T, int src, Levels lv)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= T) return;
  const uint32_t g = lv.owner[src][i];
  uint32_t f = 0u;
  if (g != kFinished && lv.count[g] > 4u) f = (uint32_t)lv.bucket[i] <= (lv.info[g] >> 8) ? 1u : 0u;
  lv.flag[i] = f;
}

__global__ __launch_bounds__(256) void k_partition(uint32_t T, int src, Levels lv)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= T) return;
  const int dst = src ^ 1;
  const uint32_t g = lv.owner[src][i];
  if (g == kFinished) {
    lv.owner[dst][i] = kFinished;
    return;
  }
  const uint32_t n = lv.count[g], s = lv.start[g];
  if (n == 1u) {
    lv.owner[dst][i] = kFinished;
    return;
  }
  const uint32_t l = lv.left[g];
  const uint32_t t = lv.prim[src][i];
  uint32_t dest, child;
  if (n <= 4u) {
    const int axis = (int)(lv.info[g] & 0xffu);
    const float4 c4 = lv.center[t];
    const float key = comp(mk3(c4.x, c4.y, c4.z), axis);
    uint32_t r = 0u;
    for (uint32_t j = 0; j < n; ++j) {
      const uint32_t u = lv.prim[src][s + j];
      const float4 d4 = lv.center[u];
      r += small_before(comp(mk3(d4.x, d4.y, d4.z), axis), u, key, t) ? 1u : 0u;
    }
    dest = s + r;
    child = r < n / 2u ? l : l + 1u;
  } else {
    const uint32_t f = lv.flag[i];
    const uint32_t before = lv.flag_scan[i] - lv.flag_scan[s];  // left-side triangles of this node in front of i
    const uint32_t mid = lv.count[l];
    dest = f ? s + before : s + mid + (i - s - before);
    child = f ? l : l + 1u;
  }
  lv.prim[dst][dest] = t;
  lv.owner[dst][dest] = child;
}

// ---- exclusive scan of uint32 (1024 elements per workgroup, recursive over the workgroup sums) ------------------
__global__ __launch_bounds__(256) void k_scan_blocks(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* sums)
{
  __shared__ uint32_t s_wave[4];
  const uint32_t first = (blockIdx.x * 256u + threadIdx.x) * 4u;
  uint32_t v[4];
  for (int j = 0; j < 4; ++j) v[j] = first + j < n ? in[first + j] : 0u;
  const uint32_t mine = v[0] + v[1] + v[2] + v[3];
  uint32_t x = mine;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off, 64);
    if (lane >= (uint32_t)off) x += y;
  }
  if (lane == 63u) s_wave[wave] = x;
  __syncthreads();
  uint32_t prefix = x - mine;
  for (uint32_t w = 0; w < wave; ++w) prefix += s_wave[w];
  for (int j = 0; j < 4; ++j) {
    if (first + j < n) out[first + j] = prefix;
    prefix += v[j];
  }
  if (threadIdx.x == 255u) sums[blockIdx.x] = prefix;
}
__global__ __launch_bounds__(256) void k_scan_add(uint32_t* out, uint32_t n, const uint32_t* sums_scanned)
{
  const uint32_t first = (blockIdx.x * 256u + threadIdx.x) * 4u;
  const uint32_t add = sums_scanned[blockIdx.x];
  for (int j = 0; j < 4; ++j)
    if (first + j < n) out[first + j] += add;
}

// out[i] = sum of in[0..i); *total (device) = sum of all.  tmp: room for 2 * (n / 1024 + 2) words per recursion level
void scan_u32(hipStream_t s, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* tmp, uint32_t* total)
{
  const uint32_t blocks = (n + 1023u) / 1024u;
  if (blocks <= 1u) {
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, s, in, out, n, total);
    return;
  }
  uint32_t* sums = tmp;
  uint32_t* sums_scanned = tmp + blocks;
  hipLaunchKernelGGL(k_scan_blocks, dim3(blocks), dim3(256), 0, s, in, out, n, sums);
  scan_u32(s, sums, sums_scanned, blocks, tmp + 2u * (size_t)blocks, total);
  hipLaunchKernelGGL(k_scan_add, dim3(blocks), dim3(256), 0, s, out, n, sums_scanned);
}

__global__ __launch_bounds__(256) void k_unpack_nodes(const float4* packed, uint32_t count, ptc_bvh_node* out)
{
  const uint32_t g = blockIdx.x * 256u + threadIdx.x;
  if (g >= count) return;
  const float4 a = packed[2u * (size_t)g], b = packed[2u * (size_t)g + 1u];
  ptc_bvh_node n;
  n.aabb_min[0] = a.x; n.aabb_min[1] = a.y; n.aabb_min[2] = a.z;
  n.aabb_max[0] = b.x; n.aabb_max[1] = b.y; n.aabb_max[2] = b.z;
  n.first_child_or_primitive = __float_as_uint(a.w);
  n.primitive_count = __float_as_uint(b.w);
  out[g] = n;
}

struct Pool {
  std::vector<void*> allocs;
  bool failed = false;
  template <typename T>
  T* get(size_t count)
  {
    void* p = nullptr;
    if (failed || hipMalloc(&p, std::max<size_t>(count, 1u) * sizeof(T)) != hipSuccess) {
      failed = true;
      return nullptr;
    }
    allocs.push_back(p);
    return static_cast<T*>(p);
  }
  ~Pool()
  {
    for (void* p : allocs) (void)hipFree(p);
  }
};

}  // namespace

int build_bvh_device(hipStream_t stream, const float* d_positions, const uint32_t* d_indices, uint32_t index_count,
                     float4* d_packed, ptc_bvh_node* d_nodes, uint32_t* node_count, uint32_t* max_depth,
                     std::vector<uint32_t>* level_base)
{
  const uint32_t T = index_count / 3u;
  if (T == 0u) return PTC_ERR_BVH;
  if (T > 0x3fffffffu) return PTC_ERR_INVALID;
  const size_t N = 2u * (size_t)T;
  Pool pool;
  Levels lv{};
  lv.tri_lo = pool.get<float4>(T);
  lv.tri_hi = pool.get<float4>(T);
  lv.center = pool.get<float4>(T);
  for (int k = 0; k < 2; ++k) {
    lv.prim[k] = pool.get<uint32_t>(T);
    lv.owner[k] = pool.get<uint32_t>(T);
  }
  lv.bucket = pool.get<uint8_t>(T);
  lv.flag = pool.get<uint32_t>(T);
  lv.flag_scan = pool.get<uint32_t>(T);
  lv.start = pool.get<uint32_t>(N);
  lv.count = pool.get<uint32_t>(N);
  lv.inner = pool.get<uint32_t>(N);
  lv.rank = pool.get<uint32_t>(N);
  lv.info = pool.get<uint32_t>(N);
  lv.left = pool.get<uint32_t>(N);
  lv.cb = pool.get<uint32_t>(6u * N);
  lv.bins = pool.get<uint32_t>(((size_t)T / 5u + 1u) * kBinWords);
  lv.out = d_packed;
  lv.status = pool.get<uint32_t>(4);
  uint32_t* scan_tmp = pool.get<uint32_t>(2u * (N / 1024u + 2u) + 2u * (N / (1024u * 1024u) + 2u) + 16u);
  uint32_t* host_status = nullptr;
  if (pool.failed || hipHostMalloc(reinterpret_cast<void**>(&host_status), 4 * sizeof(uint32_t)) != hipSuccess) return PTC_ERR_OOM;

  const dim3 per_tri((T + 255u) / 256u), b256(256);
  hipLaunchKernelGGL(k_tri_setup, per_tri, b256, 0, stream, d_positions, d_indices, T, lv);
  hipLaunchKernelGGL(k_root, dim3(1), dim3(1), 0, stream, T, lv);

  int rc = PTC_OK;
  uint32_t base = 0u, m = 1u, depth = 0u;
  int src = 0;
  if (level_base) level_base->assign(1, 0u);
  for (;;) {
    // rank of every inner node among the inner nodes of its level, and their number
    scan_u32(stream, lv.inner + base, lv.rank + base, m, scan_tmp, lv.status + 1);
    if (hipMemcpyAsync(host_status, lv.status, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
      rc = PTC_ERR_HIP;
      break;
    }
    if (host_status[0] != 0u) {
      rc = PTC_ERR_BVH;
      break;
    }
    const uint32_t inner = host_status[1];
    const uint32_t next_base = base + m;
    if (level_base) level_base->push_back(next_base);
    if ((size_t)next_base + 2u * (size_t)inner > N) {  // cannot happen: a binary tree over T leaves has 2T-1 nodes
      rc = PTC_ERR_BVH;
      break;
    }
    if (inner != 0u) {
      const dim3 per_1024((T + kPerBlock - 1u) / kPerBlock);
      hipLaunchKernelGGL(k_centroid_bounds, per_1024, b256, 0, stream, T, src, lv);
      hipLaunchKernelGGL(k_bins, per_tri, b256, 0, stream, T, src, lv);
    }
    hipLaunchKernelGGL(k_split_nodes, dim3((m + 127u) / 128u), dim3(128), 0, stream, base, m, next_base, src, lv);
    if (inner == 0u) {
      base = next_base;
      break;
    }
    hipLaunchKernelGGL(k_flags, per_tri, b256, 0, stream, T, src, lv);
    scan_u32(stream, lv.flag, lv.flag_scan, T, scan_tmp, lv.status + 2);
    hipLaunchKernelGGL(k_partition, per_tri, b256, 0, stream, T, src, lv);
    src ^= 1;
    base = next_base;
    m = 2u * inner;
    ++depth;
  }
  if (rc == PTC_OK) {
    // the last level may have raised the error flag
    if (hipMemcpyAsync(host_status, lv.status, sizeof(uint32_t), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
      rc = PTC_ERR_HIP;
    else if (host_status[0] != 0u || base != 2u * T - 1u)
      rc = PTC_ERR_BVH;
  }
  if (rc == PTC_OK && d_nodes) {
    hipLaunchKernelGGL(k_unpack_nodes, dim3((base + 255u) / 256u), b256, 0, stream, d_packed, base, d_nodes);
    if (hipStreamSynchronize(stream) != hipSuccess) rc = PTC_ERR_HIP;
  }
  (void)hipHostFree(host_status);
  if (rc != PTC_OK) {
    (void)hipStreamSynchronize(stream);
    return rc;
  }
  if (node_count) *node_count = base;
  if (max_depth) *max_depth = depth;
  return PTC_OK;
}


// =====================================================================================================================
// Part 2: the traversal layouts derived from the reference tree, on the device (what pt_scene_host.cpp does on the
// host; the decisions -- collapse costs, choice of children, node encoding -- come from pt_layout_rules.hpp, so the
// bytes are the same).  The tree is swept level by level: up for the collapse costs and leaf counts, down for the
// depth-first leaf ranks and for which nodes become four-wide nodes, up for the number of those below every node,
// down for their depth-first preorder indices; then one pass writes all records.
// =====================================================================================================================
namespace {

struct DevTree {
  const float4* bvh;
  __host__ __device__ bool is_leaf(uint32_t x) const { return __builtin_bit_cast(uint32_t, bvh[2u * (size_t)x + 1u].w) != 0u; }
  __host__ __device__ uint32_t first(uint32_t x) const { return __builtin_bit_cast(uint32_t, bvh[2u * (size_t)x].w); }
};

struct LayoutWork {
  DevTree tree;
  uint32_t count;
  uint32_t* parent;
  uint32_t* leaves;      // triangles below the node
  uint32_t* leaf_begin;  // depth-first rank of the node's first triangle
  float* best;           // 4 per node (layout_rules::collapse_costs)
  uint32_t* kept;        // 0, or 1 | children << 8: the node becomes a four-wide node
  uint32_t* kids;        // 4 per node
  uint32_t* wsub;        // four-wide nodes in the node's subtree
  uint32_t* wbegin;      // depth-first preorder index of the first of them
  uint32_t* wlevel;      // level of a four-wide node in the four-wide tree
  uint32_t* inner;       // 1: inner node
  uint32_t* inner_rank;  // rank among the inner nodes in array order (WideAccel's record index)
  uint32_t* status;      // [0] deepest four-wide level + 1
  // outputs
  uint32_t* nodes_q;
  float4* leaf_parent;
  uint32_t* tri_order;
  float4* wide;
  uint32_t dummy_ref;
};

__device__ __forceinline__ float node_area(const float4* bvh, uint32_t x)
{
  const float4 a = bvh[2u * (size_t)x], b = bvh[2u * (size_t)x + 1u];
  const float dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
  return 2.0f * (dx * dy + dx * dz + dy * dz);
}

__global__ __launch_bounds__(256) void k_lay_parent(LayoutWork w)
{
  const uint32_t x = blockIdx.x * 256u + threadIdx.x;
  if (x >= w.count) return;
  const bool leaf = w.tree.is_leaf(x);
  w.inner[x] = leaf ? 0u : 1u;
  if (x == 0u) w.parent[0] = 0xffffffffu;
  if (!leaf) {
    const uint32_t l = w.tree.first(x);
    w.parent[l] = x;
    w.parent[l + 1u] = x;
  }
}

__global__ __launch_bounds__(256) void k_lay_up_cost(LayoutWork w, uint32_t base, uint32_t m)
{
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= m) return;
  const uint32_t x = base + k;
  float* b = w.best + 4u * (size_t)x;
  if (w.tree.is_leaf(x)) {
    w.leaves[x] = 1u;
    b[0] = b[1] = b[2] = b[3] = 0.0f;
    return;
  }
  const uint32_t l = w.tree.first(x);
  layout_rules::collapse_costs(node_area(w.tree.bvh, x), w.best + 4u * (size_t)l, w.best + 4u * (size_t)l + 4u, b);
  w.leaves[x] = w.leaves[l] + w.leaves[l + 1u];
}

__global__ __launch_bounds__(256) void k_lay_down_keep(LayoutWork w, uint32_t base, uint32_t m)
{
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= m) return;
  const uint32_t x = base + k;
  if (w.tree.is_leaf(x)) {
    if (x == 0u) w.leaf_begin[0] = 0u;
    return;
  }
  if (x == 0u) {
    w.leaf_begin[0] = 0u;
    w.wlevel[0] = 0u;
    w.kept[0] = 1u;
  }
  const uint32_t l = w.tree.first(x);
  const uint32_t begin = w.leaf_begin[x];
  w.leaf_begin[l] = begin;
  w.leaf_begin[l + 1u] = begin + w.leaves[l];
  if (w.kept[x] == 0u) return;
  uint32_t kids[4];
  const int nk = layout_rules::choose_children(w.tree, w.best, x, kids);
  const uint32_t level = w.wlevel[x];
  for (int c = 0; c < nk; ++c) {
    w.kids[4u * (size_t)x + c] = kids[c];
    if (!w.tree.is_leaf(kids[c])) {
      w.kept[kids[c]] = 1u;
      w.wlevel[kids[c]] = level + 1u;
    }
  }
  w.kept[x] = 1u | ((uint32_t)nk << 8);
  atomicMax(&w.status[0], level + 1u);
}

__global__ __launch_bounds__(256) void k_lay_up_wsub(LayoutWork w, uint32_t base, uint32_t m)
{
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= m) return;
  const uint32_t x = base + k;
  uint32_t n = w.kept[x] != 0u ? 1u : 0u;
  if (!w.tree.is_leaf(x)) {
    const uint32_t l = w.tree.first(x);
    n += w.wsub[l] + w.wsub[l + 1u];
  }
  w.wsub[x] = n;
}

__global__ __launch_bounds__(256) void k_lay_down_wbegin(LayoutWork w, uint32_t base, uint32_t m)
{
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= m) return;
  const uint32_t x = base + k;
  if (x == 0u) w.wbegin[0] = 0u;
  if (w.tree.is_leaf(x)) return;
  const uint32_t l = w.tree.first(x);
  const uint32_t first = w.wbegin[x] + (w.kept[x] != 0u ? 1u : 0u);
  w.wbegin[l] = first;
  w.wbegin[l + 1u] = first + w.wsub[l];
}

__global__ __launch_bounds__(128) void k_lay_emit(LayoutWork w)
{
  const uint32_t x = blockIdx.x * 128u + threadIdx.x;
  if (x >= w.count) return;
  const float4* bvh = w.tree.bvh;
  if (w.tree.is_leaf(x)) {
    const uint32_t rank = w.leaf_begin[x];
    const uint32_t p = w.parent[x];
    // a single-triangle mesh has no inner node at all: every box test of the reference is vacuous
    const float big = 3.402823466e+38f;
    float4 lo = make_float4(-big, -big, -big, 0.f), hi = make_float4(big, big, big, 0.f);
    if (p != 0xffffffffu) {
      const float4 a = bvh[2u * (size_t)p], b = bvh[2u * (size_t)p + 1u];
      lo = make_float4(a.x, a.y, a.z, 0.f);
      hi = make_float4(b.x, b.y, b.z, 0.f);
    }
    w.leaf_parent[2u * (size_t)rank] = lo;
    w.leaf_parent[2u * (size_t)rank + 1u] = hi;
    w.tri_order[rank] = w.tree.first(x) / 3u;
    return;
  }
  auto ref2 = [&](uint32_t c) { return w.tree.is_leaf(c) ? (kLeafBit | w.leaf_begin[c]) : w.inner_rank[c]; };
  {  // the two-child record of the exact near-first walk (WideAccel::wide)
    const uint32_t l = w.tree.first(x);
    const float4 la = bvh[2u * (size_t)l], lb = bvh[2u * (size_t)l + 1u];
    const float4 ra = bvh[2u * (size_t)l + 2u], rb = bvh[2u * (size_t)l + 3u];
    float4* rec = w.wide + 4u * (size_t)w.inner_rank[x];
    rec[0] = make_float4(la.x, la.y, la.z, lb.x);
    rec[1] = make_float4(lb.y, lb.z, ra.x, ra.y);
    rec[2] = make_float4(ra.z, rb.x, rb.y, rb.z);
    rec[3] = make_float4(__uint_as_float(ref2(l)), __uint_as_float(ref2(l + 1u)), 0.0f, 0.0f);
  }
  const uint32_t kept = w.kept[x];
  if (kept == 0u) return;
  const int nk = (int)(kept >> 8);
  float lo[3][4], hi[3][4];
  uint32_t refs[4];
  for (int c = 0; c < 4; ++c) {
    if (c < nk) {
      const uint32_t kid = w.kids[4u * (size_t)x + c];
      const float4 a = bvh[2u * (size_t)kid], b = bvh[2u * (size_t)kid + 1u];
      lo[0][c] = a.x; lo[1][c] = a.y; lo[2][c] = a.z;
      hi[0][c] = b.x; hi[1][c] = b.y; hi[2][c] = b.z;
      refs[c] = w.tree.is_leaf(kid) ? (kLeafBit | w.leaf_begin[kid]) : w.wbegin[kid];
    } else {
      for (int a = 0; a < 3; ++a) lo[a][c] = hi[a][c] = 0.0f;
      refs[c] = w.dummy_ref;
    }
  }
  uint32_t q[16];
  layout_rules::quantise_node(q, nk, lo, hi, refs);
  uint4* out = reinterpret_cast<uint4*>(w.nodes_q) + 4u * (size_t)w.wbegin[x];
  for (int j = 0; j < 4; ++j) out[j] = make_uint4(q[4 * j], q[4 * j + 1], q[4 * j + 2], q[4 * j + 3]);
}

__global__ __launch_bounds__(256) void k_lay_instance_tris(m4 m, const float* positions, const uint32_t* indices,
                                                           const uint32_t* tri_order, uint32_t T, float4* out)
{
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= T) return;
  const uint32_t* idx = indices + 3u * (size_t)tri_order[k];
  const float* q0 = positions + 3u * (size_t)idx[0];
  const float* q1 = positions + 3u * (size_t)idx[1];
  const float* q2 = positions + 3u * (size_t)idx[2];
  layout_rules::instance_triangle(m, mk3(q0[0], q0[1], q0[2]), mk3(q1[0], q1[1], q1[2]), mk3(q2[0], q2[1], q2[2]), out + kTriVec4 * (size_t)k);
}

}  // namespace

int build_layouts_device(hipStream_t stream, const float4* d_bvh, uint32_t count, const std::vector<uint32_t>& level_base,
                         DeviceLayouts* out)
{
  *out = DeviceLayouts{};
  if (count == 0u) return PTC_OK;
  if (level_base.size() < 2u || level_base.front() != 0u || level_base.back() != count) return PTC_ERR_INVALID;
  const uint32_t T = (count + 1u) / 2u;
  const size_t N = count;
  Pool pool;
  LayoutWork w{};
  w.tree.bvh = d_bvh;
  w.count = count;
  w.parent = pool.get<uint32_t>(N);
  w.leaves = pool.get<uint32_t>(N);
  w.leaf_begin = pool.get<uint32_t>(N);
  w.best = pool.get<float>(4u * N);
  w.kept = pool.get<uint32_t>(N);
  w.kids = pool.get<uint32_t>(4u * N);
  w.wsub = pool.get<uint32_t>(N);
  w.wbegin = pool.get<uint32_t>(N);
  w.wlevel = pool.get<uint32_t>(N);
  w.inner = pool.get<uint32_t>(N);
  w.inner_rank = pool.get<uint32_t>(N);
  w.status = pool.get<uint32_t>(4);
  uint32_t* scan_tmp = pool.get<uint32_t>(2u * (N / 1024u + 2u) + 2u * (N / (1024u * 1024u) + 2u) + 16u);
  // outputs: owned by the caller once returned
  Pool outputs;
  w.leaf_parent = outputs.get<float4>(2u * ((size_t)T + 1u));
  w.tri_order = outputs.get<uint32_t>(T);
  w.wide = outputs.get<float4>(4u * (size_t)std::max(T - 1u, 1u));
  w.dummy_ref = kLeafBit | T;
  if (pool.failed || outputs.failed) return PTC_ERR_OOM;
  const size_t levels = level_base.size() - 1u;
  auto grid = [](uint32_t m) { return dim3((m + 255u) / 256u); };
  const dim3 b256(256);
  bool ok = hipMemsetAsync(w.kept, 0, N * sizeof(uint32_t), stream) == hipSuccess &&
            hipMemsetAsync(w.status, 0, 4 * sizeof(uint32_t), stream) == hipSuccess &&
            hipMemsetAsync(w.leaf_parent + 2u * (size_t)T, 0, 2 * sizeof(float4), stream) == hipSuccess;
  hipLaunchKernelGGL(k_lay_parent, grid(count), b256, 0, stream, w);
  scan_u32(stream, w.inner, w.inner_rank, count, scan_tmp, w.status + 2);
  for (size_t l = levels; l-- > 0u;)
    hipLaunchKernelGGL(k_lay_up_cost, grid(level_base[l + 1] - level_base[l]), b256, 0, stream, w, level_base[l], level_base[l + 1] - level_base[l]);
  for (size_t l = 0; l < levels; ++l)
    hipLaunchKernelGGL(k_lay_down_keep, grid(level_base[l + 1] - level_base[l]), b256, 0, stream, w, level_base[l], level_base[l + 1] - level_base[l]);
  for (size_t l = levels; l-- > 0u;)
    hipLaunchKernelGGL(k_lay_up_wsub, grid(level_base[l + 1] - level_base[l]), b256, 0, stream, w, level_base[l], level_base[l + 1] - level_base[l]);
  for (size_t l = 0; l < levels; ++l)
    hipLaunchKernelGGL(k_lay_down_wbegin, grid(level_base[l + 1] - level_base[l]), b256, 0, stream, w, level_base[l], level_base[l + 1] - level_base[l]);
  uint32_t host[4] = {0u, 0u, 0u, 0u};  // deepest level + 1 | . | . | four-wide nodes
  float4 root[2];
  ok = ok && hipMemcpyAsync(&host[0], w.status, sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess &&
       hipMemcpyAsync(&host[3], w.wsub, sizeof(uint32_t), hipMemcpyDeviceToHost, stream) == hipSuccess &&
       hipMemcpyAsync(root, d_bvh, 2 * sizeof(float4), hipMemcpyDeviceToHost, stream) == hipSuccess &&
       hipStreamSynchronize(stream) == hipSuccess;
  if (!ok) return PTC_ERR_HIP;
  const uint32_t wide4_nodes = host[3];
  w.nodes_q = outputs.get<uint32_t>(16u * (size_t)std::max(wide4_nodes, 1u));
  if (outputs.failed) return PTC_ERR_OOM;
  hipLaunchKernelGGL(k_lay_emit, dim3((count + 127u) / 128u), dim3(128), 0, stream, w);
  if (hipStreamSynchronize(stream) != hipSuccess) return PTC_ERR_HIP;
  out->nodes_q = w.nodes_q;
  out->leaf_parent = w.leaf_parent;
  out->tri_order = w.tri_order;
  out->wide = w.wide;
  out->wide4_nodes = wide4_nodes;
  out->wide4_depth = host[0];
  out->triangles = T;
  out->inner_nodes = T - 1u;
  const bool root_is_leaf = __builtin_bit_cast(uint32_t, root[1].w) != 0u;
  out->root_ref4 = root_is_leaf ? kLeafBit : 0u;
  out->root_ref2 = root_is_leaf ? kLeafBit : 0u;
  out->dummy_ref = w.dummy_ref;
  out->root_min[0] = root[0].x; out->root_min[1] = root[0].y; out->root_min[2] = root[0].z;
  out->root_max[0] = root[1].x; out->root_max[1] = root[1].y; out->root_max[2] = root[1].z;
  outputs.allocs.clear();  // handed over
  return PTC_OK;
}

void launch_instance_triangles(hipStream_t stream, const m4& m, const float* d_positions, const uint32_t* d_indices,
                               const uint32_t* d_tri_order, uint32_t triangles, float4* d_out)
{
  if (triangles == 0u) return;
  hipLaunchKernelGGL(k_lay_instance_tris, dim3((triangles + 255u) / 256u), dim3(256), 0, stream, m, d_positions, d_indices,
                     d_tri_order, triangles, d_out);
}

}  // namespace pt
