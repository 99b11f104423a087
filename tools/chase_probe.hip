// Microbenchmark: dependent chains of record fetches, one chain per lane (what a BVH walk is to the
// memory system): each step loads K x 16 bytes of a random 128-byte record and derives the next
// record from what it read.  Reports the step rate for W waves per SIMD and two table sizes.
// build: hipcc -O3 --offload-arch=gfx950 -o chase_probe chase_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIter = 512;
__device__ inline uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int K>
__global__ __launch_bounds__(256) void chase(const float4* __restrict__ table, uint32_t mask, uint32_t* out, int active_lanes) {
  const uint32_t gtid = blockIdx.x * 256 + threadIdx.x;
  uint32_t idx = mix(gtid) & mask;
  if ((int)(threadIdx.x & 63) >= active_lanes) return;
  float sum = 0.0f;
  for (int it = 0; it < kIter; ++it) {
    const float4* np = table + (size_t)idx * 8;
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < K; ++c) s += np[c].x;
    sum += s;
    idx = mix(__float_as_uint(s) + it) & mask;
  }
  out[gtid] = __float_as_uint(sum) ^ idx;
}

int main() {
  const uint32_t n_max = 1u << 18;
  std::vector<float4> h((size_t)n_max * 8);
  for (uint32_t i = 0; i < n_max; ++i)
    for (int c = 0; c < 8; ++c) h[(size_t)i * 8 + c] = make_float4((float)(i * 8 + c), 0.f, 0.f, 0.f);
  float4* table; uint32_t* out;
  CHECK(hipMalloc(&table, h.size() * sizeof(float4)));
  CHECK(hipMalloc(&out, (size_t)256 * 8 * 256 * 4));
  CHECK(hipMemcpy(table, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (uint32_t n : {1u << 13, 1u << 18}) {
    printf("table %.1f MB\n", n * 128.0 / 1e6);
    for (int lanes : {64, 32, 16}) {
      for (int w : {1, 2, 4, 5, 8}) {
        const int blocks = 256 * w;
        auto time = [&](auto launch) {
          launch(); CHECK(hipDeviceSynchronize());
          CHECK(hipEventRecord(e0));
          for (int r = 0; r < 3; ++r) launch();
          CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
          return ms / 3;
        };
        const float m1 = time([&] { chase<1><<<blocks, 256>>>(table, n - 1, out, lanes); });
        const float m4 = time([&] { chase<4><<<blocks, 256>>>(table, n - 1, out, lanes); });
        const float m7 = time([&] { chase<7><<<blocks, 256>>>(table, n - 1, out, lanes); });
        const double steps = (double)blocks * 4 * lanes * kIter;
        printf("  lanes %2d waves/SIMD %d: K=1 %6.0f clk/step %6.1f G steps/s | K=4 %6.0f clk %6.1f G | K=7 %6.0f clk %6.1f G\n", lanes, w,
               m1 * 1e-3 / kIter * 2.4e9, steps / m1 * 1e-6, m4 * 1e-3 / kIter * 2.4e9, steps / m4 * 1e-6, m7 * 1e-3 / kIter * 2.4e9,
               steps / m7 * 1e-6);
      }
    }
  }
  return 0;
}
