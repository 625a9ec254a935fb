// Micro-benchmark (measurement tool, not product): how fast can a CU fetch 128-byte / 64-byte records at random
// addresses, one record per lane per step (the BVH traversal access pattern), by load shape.
//   mode 0: lane loads its own 128-B record with 7 x 16-B loads
//   mode 1: lane loads its own 64-B record with 4 x 16-B loads
//   mode 2: 8 lanes load one 128-B record per instruction (8 rounds cover the 64 records of the wave), data
//           handed to the owner through LDS
//   mode 3: lane loads its own 128-B record with 8 x 16-B loads, 50 % of lanes masked off
// build: hipcc --offload-arch=gfx950 -O3 gather_bench.hip -o gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ __launch_bounds__(64) void k_gather(const float4* table, uint32_t records, int steps, float* out, int active_mod)
{
  __shared__ float4 s_buf[MODE == 2 ? 64 * 8 : 1];
  uint32_t seed = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0.0f;
  uint32_t idx = lcg(seed) % records;
  const bool on = active_mod <= 1 || (threadIdx.x % active_mod) == 0;
  for (int s = 0; s < steps; ++s) {
    if (MODE == 0 || MODE == 3) {
      if (on) {
        const float4* q = table + 8u * (size_t)idx;
        float4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6];
        acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z;
        idx = (__float_as_uint(g.w) + lcg(seed)) % records;  // dependent chain like a tree walk
      }
    } else if (MODE == 1) {
      if (on) {
        const float4* q = table + 4u * (size_t)idx;
        float4 a = q[0], b = q[1], c = q[2], d = q[3];
        acc += a.x + b.y + c.z + d.w;
        idx = (__float_as_uint(d.w) + lcg(seed)) % records;
      }
    } else if (MODE == 2) {
      const uint32_t sub = threadIdx.x & 7u, grp = threadIdx.x >> 3;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t owner = grp * 8u + (uint32_t)j;
        const uint32_t oidx = __shfl(idx, (int)owner, 64);
        s_buf[owner * 8u + sub] = table[8u * (size_t)oidx + sub];
      }
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      const float4* q = s_buf + threadIdx.x * 8u;
      float4 a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6];
      acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z;
      idx = (__float_as_uint(g.w) + lcg(seed)) % records;
      __builtin_amdgcn_wave_barrier();
    }
  }
  out[blockIdx.x * 64u + threadIdx.x] = acc;
}

int main(int argc, char** argv)
{
  const int waves_per_cu = argc > 1 ? atoi(argv[1]) : 20;
  const int cus = 256;
  const int grid = cus * waves_per_cu;
  const int steps = 2000;
  float* out;
  CHECK(hipMalloc(&out, sizeof(float) * grid * 64));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const size_t sizes[] = {16u << 10, 1u << 20, 24u << 20, 96u << 20, 512u << 20};
  for (size_t bytes : sizes) {
    float4* table;
    CHECK(hipMalloc(&table, bytes));
    std::vector<uint32_t> host(bytes / 4);
    uint32_t s = 777;
    for (auto& v : host) { s = s * 1664525u + 1013904223u; v = s >> 9; }  // small positive floats/ints
    CHECK(hipMemcpy(table, host.data(), bytes, hipMemcpyHostToDevice));
    for (int mode = 0; mode < 4; ++mode) {
      const uint32_t rec_bytes = mode == 1 ? 64u : 128u;
      const uint32_t records = (uint32_t)(bytes / rec_bytes);
      const int amod = mode == 3 ? 3 : 1;
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        switch (mode) {
        case 0: hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(64), 0, 0, table, records, steps, out, amod); break;
        case 1: hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(64), 0, 0, table, records, steps, out, amod); break;
        case 2: hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(64), 0, 0, table, records, steps, out, amod); break;
        case 3: hipLaunchKernelGGL(k_gather<3>, dim3(grid), dim3(64), 0, 0, table, records, steps, out, amod); break;
        }
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      const double lanes = mode == 3 ? 64.0 / 3.0 : 64.0;
      const double fetches = (double)grid * lanes * steps;
      printf("table %6.1f MiB mode %d waves/CU %d: %8.3f ms  %7.2f Gfetch/s  %6.2f clk/CU/fetch (2.4 GHz)  %7.2f TB/s\n",
             bytes / 1048576.0, mode, waves_per_cu, best, fetches / best / 1e6, best * 1e-3 * 2.4e9 * cus / fetches,
             fetches * rec_bytes / best / 1e9);
      fflush(stdout);
    }
    CHECK(hipFree(table));
  }
  return 0;
}
