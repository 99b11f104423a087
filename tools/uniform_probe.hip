// Microbenchmark: what a wave pays when all its lanes fetch the SAME 64-byte record -- with per-lane
// vector loads (4 x 16 bytes, identical addresses) and with one scalar load broadcast to the lanes.
// build: hipcc -O3 --offload-arch=gfx950 -o uniform_probe uniform_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIter = 512;
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
__device__ inline uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// mode 0: every lane its own record; 1: all lanes of a wave the same record, vector loads; 2: scalar load
template <int MODE>
__global__ __launch_bounds__(256) void fetch(const uint4* __restrict__ table, uint32_t mask, uint32_t* out) {
  const uint32_t gtid = blockIdx.x * 256 + threadIdx.x;
  const uint32_t wave = gtid >> 6;
  uint32_t idx = mix(MODE == 0 ? gtid : wave) & mask;
  uint32_t sum = 0;
  for (int it = 0; it < kIter; ++it) {
    uint32_t s;
    if (MODE == 2) {
      const uint32_t u = __builtin_amdgcn_readfirstlane(idx);
      const uint4* np = table + (size_t)u * 4;
      u32x16 v;
      asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(np) : "memory");
      s = v[0] + v[4] + v[8] + v[12];
    } else {
      const uint4* np = table + (size_t)idx * 4;
      s = np[0].x + np[1].x + np[2].x + np[3].x;
    }
    sum += s;
    idx = mix(s + it) & mask;
  }
  out[gtid] = sum ^ idx;
}

int main() {
  const uint32_t n = 1u << 19;  // 64-byte records: 33.6 MB
  std::vector<uint4> h((size_t)n * 4);
  for (uint32_t i = 0; i < n; ++i)
    for (int c = 0; c < 4; ++c) h[(size_t)i * 4 + c] = make_uint4(i * 4 + c, 0, 0, 0);
  uint4* table; uint32_t* out;
  const int blocks = 256 * 5;
  CHECK(hipMalloc(&table, h.size() * sizeof(uint4)));
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  CHECK(hipMemcpy(table, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<uint32_t> r0((size_t)blocks * 256), r1(r0.size()), r2(r0.size());
  auto run = [&](const char* name, auto launch, std::vector<uint32_t>& res) {
    launch(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) launch();
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    CHECK(hipMemcpy(res.data(), out, res.size() * 4, hipMemcpyDeviceToHost));
    printf("%-28s %7.3f ms  %6.0f clk per step (5 waves per SIMD)  %6.1f G lane-steps/s\n", name, ms, ms * 1e-3 / kIter * 2.4e9,
           (double)blocks * 256 * kIter / ms * 1e-6);
  };
  for (uint32_t m : {(1u << 13) - 1, n - 1}) {
    printf("table %.1f MB\n", (m + 1) * 64.0 / 1e6);
    run("own record per lane", [&] { fetch<0><<<blocks, 256>>>(table, m, out); }, r0);
    run("same record, vector loads", [&] { fetch<1><<<blocks, 256>>>(table, m, out); }, r1);
    run("same record, scalar load", [&] { fetch<2><<<blocks, 256>>>(table, m, out); }, r2);
    printf("scalar and vector results %s\n", r1 == r2 ? "agree" : "DIFFER");
  }
  return 0;
}
