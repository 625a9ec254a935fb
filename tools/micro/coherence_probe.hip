// coherence_probe.hip -- which loads see another wavefront's write-through (sc1) stores INSIDE one launch on gfx950, when the
// reader has read the same line before (its L1 / its XCD's L2 may hold the old copy)?  One producer wavefront rewrites a
// 4 KB buffer round after round (16-byte sc1 stores, s_waitcnt vmcnt(0), then an agent-scope atomic store of the round
// number); reader wavefronts -- one on the producer's XCD, one on every other -- wait for the round with an atomic RMW
// (always coherent), then read the buffer with one of five load forms and count words that are not the round's value.
//   hipcc --offload-arch=gfx950 -O2 -o coherence_probe tools/micro/coherence_probe.hip && ./coherence_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kWords = 1024, kRounds = 2000;

__device__ inline uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

__global__ __launch_bounds__(64) void probe(uint32_t* buf, uint32_t* flag, uint32_t* ack, unsigned long long* stale, uint32_t* who, int mode, int store_mode)
{
  const uint32_t lane = threadIdx.x, wg = blockIdx.x;
  if (lane == 0) who[wg] = xcc_id();
  if (wg == 0) {  // producer
    for (int r = 1; r <= kRounds; ++r) {
      // wait until every reader has acknowledged the previous round
      if (r > 1) while (__hip_atomic_fetch_or(ack, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)(r - 1) * (gridDim.x - 1)) __builtin_amdgcn_s_sleep(1);
      for (int i = lane * 4; i < kWords; i += 256) {
        u32x4 v = {(uint32_t)r, (uint32_t)r, (uint32_t)r, (uint32_t)r};
        uint32_t* p = buf + i;
        if (store_mode == 0) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
        else if (store_mode == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (store_mode == 2) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      if (lane == 0) __hip_atomic_store(flag, (uint32_t)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  unsigned long long bad = 0;
  for (int r = 1; r <= kRounds; ++r) {
    while (__hip_atomic_fetch_or(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)r) __builtin_amdgcn_s_sleep(1);
    if (mode == 3) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    for (int i = lane * 4; i < kWords; i += 256) {
      const uint32_t* p = buf + i;
      u32x4 v;
      if (mode == 0 || mode == 3) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
      else if (mode == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
      else if (mode == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
      else asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
      bad += (v.x != (uint32_t)r) + (v.y != (uint32_t)r) + (v.z != (uint32_t)r) + (v.w != (uint32_t)r);
    }
    if (lane == 0) __hip_atomic_fetch_add(ack, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  atomicAdd(&stale[wg], bad);
}

// a previous launch in which every workgroup (every XCD) wrote the buffer with plain / nt stores: its lines then sit in every
// XCD's L2 when the probe starts -- does a reader still see the producer's write-through stores?
__global__ __launch_bounds__(64) void dirty_all(uint32_t* buf, int nt)
{
  for (int i = threadIdx.x * 4; i < kWords; i += 256) {
    u32x4 v = {0xdeadu, 0xdeadu, 0xdeadu, 0xdeadu};
    uint32_t* p = buf + i;
    if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  }
}

int main()
{
  const int wgs = 16;
  uint32_t *buf, *flag, *ack, *who;
  unsigned long long* stale;
  hipMalloc(&buf, kWords * 4); hipMalloc(&flag, 256); hipMalloc(&ack, 256); hipMalloc(&who, wgs * 4); hipMalloc(&stale, wgs * 8);
  const char* modes[] = {"plain load", "sc1 load", "sc0 sc1 load", "acquire fence + plain load", "nt load"};
  const char* smodes[] = {"sc1 store", "sc0 sc1 store", "plain store + release fence"};
  for (int pre = 0; pre < 3; ++pre)
  for (int sm = 0; sm < 3; ++sm)
    for (int m = 0; m < 5; ++m) {
      hipMemset(buf, 0, kWords * 4); hipMemset(flag, 0, 256); hipMemset(ack, 0, 256); hipMemset(stale, 0, wgs * 8);
      if (pre) hipLaunchKernelGGL(dirty_all, dim3(wgs), dim3(64), 0, 0, buf, pre - 1);
      hipLaunchKernelGGL(probe, dim3(wgs), dim3(64), 0, 0, buf, flag, ack, stale, who, m, sm);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      unsigned long long h[wgs]; uint32_t w[wgs];
      hipMemcpy(h, stale, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(w, who, sizeof w, hipMemcpyDeviceToHost);
      unsigned long long same = 0, other = 0; int ns = 0, no = 0;
      for (int i = 1; i < wgs; ++i) { if (w[i] == w[0]) { same += h[i]; ++ns; } else { other += h[i]; ++no; } }
      printf("%-34s | %-28s | %-26s | stale words: same XCD %llu (%d readers)  other XCDs %llu (%d readers)  of %d per reader\n", pre == 0 ? "buffer fresh from hipMemset" : pre == 1 ? "previous launch: plain stores" : "previous launch: nt stores", smodes[sm], modes[m], same, ns, other, no,
             kWords * kRounds);
    }
  return 0;
}
