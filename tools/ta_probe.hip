// Microbenchmark: how fast does one CU fetch 128-byte records that every lane picks at random from a
// cache-resident table, (a) each lane fetching its own record with K 16-byte loads, (b) eight lanes
// fetching one record together (one 128-byte line per eight lanes and instruction) into registers,
// (c) the same straight into LDS, each lane then reading its own record back with ds_read_b128.
// build: hipcc -O3 --offload-arch=gfx950 -o ta_probe ta_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIter = 256;
__device__ inline uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int K>
__global__ __launch_bounds__(256) void scatter(const float4* __restrict__ table, uint32_t mask, float* out) {
  const uint32_t gtid = blockIdx.x * 256 + threadIdx.x;
  float sum = 0.0f;
#pragma unroll 2
  for (int it = 0; it < kIter; ++it) {
    const uint32_t idx = mix(gtid * kIter + it) & mask;
    const float4* np = table + (size_t)idx * 8;
#pragma unroll
    for (int c = 0; c < K; ++c) sum += np[c].x;
  }
  out[gtid] = sum;
}

__global__ __launch_bounds__(256) void coop_regs(const float4* __restrict__ table, uint32_t mask, float* out) {
  const uint32_t gtid = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  float sum = 0.0f;
#pragma unroll 2
  for (int it = 0; it < kIter; ++it) {
    const uint32_t idx = mix(gtid * kIter + it) & mask;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t idx_t = __shfl(idx, 8 * i + (lane >> 3));
      sum += table[(size_t)idx_t * 8 + (lane & 7)].x;
    }
  }
  out[gtid] = sum;
}

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* global_void_ptr;

// HALVES = 1: all 64 records per round (8 KiB of LDS per wave); 2: 32 records at a time (4 KiB)
template <int HALVES>
__global__ __launch_bounds__(256) void coop_lds(const float4* __restrict__ table, uint32_t mask, float* out, uint32_t* errors) {
  __shared__ float4 stage[4][512 / HALVES];
  const uint32_t gtid = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4* mine = stage[wave];
  float sum = 0.0f;
  uint32_t bad = 0;
  for (int it = 0; it < kIter; ++it) {
    const uint32_t idx = mix(gtid * kIter + it) & mask;
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
      constexpr int kPer = 8 / HALVES;
#pragma unroll
      for (int i = 0; i < kPer; ++i) {
        const int t = 8 * (h * kPer + i) + (lane >> 3);       // the lane whose record this one helps fetch
        const uint32_t idx_t = __shfl(idx, t);
        const int chunk = (lane & 7) ^ ((t >> 1) & 7);        // position lane&7 of the record's LDS image holds this chunk
        __builtin_amdgcn_global_load_lds((global_void_ptr)(table + (size_t)idx_t * 8 + chunk), (lds_void_ptr)(mine + 64 * i), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const bool reader = HALVES == 1 || (lane >> 5) == h;
      if (reader) {
        const int slot = HALVES == 1 ? lane : (lane & 31);
        const int sw = (lane >> 1) & 7;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
          const float4 v = mine[slot * 8 + (c ^ sw)];
          sum += v.x;
          bad += v.x != (float)(idx * 8 + c);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the next round overwrites the image
    }
  }
  out[gtid] = sum;
  if (bad) atomicAdd(errors, bad);
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 262144u;  // records of 128 B (a power of two)
  std::vector<float4> h((size_t)n * 8);
  for (uint32_t i = 0; i < n; ++i)
    for (int c = 0; c < 8; ++c) h[(size_t)i * 8 + c] = make_float4((float)(i * 8 + c), 0.f, 0.f, 0.f);
  float4* table; float* out; uint32_t* errors;
  const int blocks = 256 * 8;
  CHECK(hipMalloc(&table, h.size() * sizeof(float4)));
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
  CHECK(hipMalloc(&errors, 4));
  CHECK(hipMemset(errors, 0, 4));
  CHECK(hipMemcpy(table, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double fetches = (double)blocks * 256 * kIter;
  auto run = [&](const char* name, auto launch) {
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double per_s = fetches / (ms * 1e-3);
    printf("%-14s %8.3f ms  %7.1f G records/s  %.3f records/clk/CU (2.4 GHz, 256 CUs)\n", name, ms, per_s * 1e-9, per_s / 2.4e9 / 256);
  };
  printf("table: %u records of 128 B = %.1f MB\n", n, n * 128.0 / 1e6);
  run("scatter<1>", [&] { scatter<1><<<blocks, 256>>>(table, n - 1, out); });
  run("scatter<2>", [&] { scatter<2><<<blocks, 256>>>(table, n - 1, out); });
  run("scatter<4>", [&] { scatter<4><<<blocks, 256>>>(table, n - 1, out); });
  run("scatter<7>", [&] { scatter<7><<<blocks, 256>>>(table, n - 1, out); });
  run("coop_regs", [&] { coop_regs<<<blocks, 256>>>(table, n - 1, out); });
  run("coop_lds<1>", [&] { coop_lds<1><<<blocks, 256>>>(table, n - 1, out, errors); });
  run("coop_lds<2>", [&] { coop_lds<2><<<blocks, 256>>>(table, n - 1, out, errors); });
  uint32_t bad = 0;
  CHECK(hipMemcpy(&bad, errors, 4, hipMemcpyDeviceToHost));
  printf("coop_lds mismatches: %u\n", bad);
  return bad != 0;
}
