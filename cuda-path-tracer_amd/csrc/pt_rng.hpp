// pt_rng.hpp -- the reference's per-path random numbers.
//
// seed  = hash(hash(index) ^ iteration)                      (reference hash.cuh:4-14; ray_gen.cu:18,
//                                                              path_tracer.cu:239,300)
// engine = thrust::default_random_engine = minstd_rand: x <- 48271 * x mod (2^31 - 1), seed s -> s mod m,
//          0 -> 1;  discard(z) multiplies by 48271^z mod m.
// uniform_real_distribution<float>(0,1): float(x - 1) / 2^31   (may round to exactly 1.0f).
#pragma once

#include "pt_math.hpp"

namespace pt {

PT_HD uint32_t hash32(uint32_t a)
{
  a = (a + 0x7ed55d16u) + (a << 12);
  a = (a ^ 0xc761c23cu) ^ (a >> 19);
  a = (a + 0x165667b1u) + (a << 5);
  a = (a + 0xd3a2646cu) ^ (a << 9);
  a = (a + 0xfd7046c5u) + (a << 3);
  a = (a ^ 0xb55a4f09u) ^ (a >> 16);
  return a;
}

// x mod (2^31 - 1) for x < 2^62, by folding (2^31 == 1 mod m)
PT_HD uint32_t mod_m31(uint64_t x)
{
  uint64_t f = (x & 0x7fffffffull) + (x >> 31);  // < 2^32
  f = (f & 0x7fffffffull) + (f >> 31);           // <= 2^31
  uint32_t r = (uint32_t)f;
  if (r >= 0x7fffffffu) r -= 0x7fffffffu;
  return r;
}

struct Minstd {
  uint32_t x;

  PT_HD void seed(uint32_t s)
  {
    uint32_t v = s >= 0x7fffffffu ? s - 0x7fffffffu : s;  // s mod m for s < 2^32
    if (v >= 0x7fffffffu) v -= 0x7fffffffu;               // s == 2^32-1 or 2^32-2 ... fold twice
    x = v == 0u ? 1u : v;
  }
  PT_HD uint32_t next()
  {
    x = mod_m31((uint64_t)x * 48271ull);
    return x;
  }
  // discard(z) for the small z the bounce loop uses
  PT_HD void discard(uint32_t z)
  {
    uint64_t mult = 48271ull, acc = 1ull;
    while (z > 0u) {
      if (z & 1u) acc = mod_m31(acc * mult);
      z >>= 1;
      mult = mod_m31(mult * mult);
    }
    x = mod_m31(acc * (uint64_t)x);
  }
  PT_HD float uniform()
  {
    const uint32_t v = next();
    return (float)(v - 1u) / 2147483648.0f;
  }
};

PT_HD uint32_t path_seed(uint32_t index, uint32_t iteration) { return hash32(hash32(index) ^ iteration); }

}  // namespace pt
