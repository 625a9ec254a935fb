// Generates tests/golden/rng_kat.json by running rocThrust 7.2's minstd_rand (= the algorithm of
// thrust::default_random_engine, which the reference uses at ray_gen.cu:18-22, path_tracer.cu:239-243,
// 300-301, distributions.cuh:9-12) ON THE HOST.  Test infrastructure only.
//   build: make -C oracle tools/thrust_rng_kat ; run: oracle/tools/thrust_rng_kat > tests/golden/rng_kat.json
#include <thrust/random.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

static unsigned hash_ref(unsigned a)  // restated integer hash, only used to pick realistic seeds
{
  a = (a + 0x7ed55d16) + (a << 12);
  a = (a ^ 0xc761c23c) ^ (a >> 19);
  a = (a + 0x165667b1) + (a << 5);
  a = (a + 0xd3a2646c) ^ (a << 9);
  a = (a + 0xfd7046c5) + (a << 3);
  a = (a ^ 0xb55a4f09) ^ (a >> 16);
  return a;
}
static unsigned fbits(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }

int main()
{
  const unsigned seeds[] = {0u, 1u, 12345u, 2147483647u, 2147483648u, 4294967295u, 0x2b4f8145u,
                            hash_ref(hash_ref(77u) ^ 3u), hash_ref(hash_ref(2073599u) ^ 31u)};
  const unsigned long long discards[] = {0, 1, 2, 7, 49, 1000003ull};
  std::printf("{\n \"source\": \"rocThrust 7.2 thrust::default_random_engine + uniform_real_distribution<float>(0,1), host run\",\n \"cases\": [\n");
  bool first = true;
  for (unsigned s : seeds) {
    for (unsigned long long d : discards) {
      thrust::default_random_engine rng(s);
      rng.discard(d);
      thrust::uniform_real_distribution<float> dist(0.0, 1.0);
      thrust::default_random_engine raw = rng;
      std::printf("%s  {\"seed\": %u, \"discard\": %llu, \"raw\": [", first ? "" : ",\n", s, d);
      for (int i = 0; i < 4; ++i) std::printf("%s%u", i ? ", " : "", (unsigned)raw());
      std::printf("], \"uniform_bits\": [");
      for (int i = 0; i < 4; ++i) std::printf("%s%u", i ? ", " : "", fbits(dist(rng)));
      std::printf("]}");
      first = false;
    }
  }
  std::printf("\n ]\n}\n");
  return 0;
}
