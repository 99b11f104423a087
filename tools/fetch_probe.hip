// Calibration of the FETCH_SIZE counter for the traversal kernels' access pattern: every lane reads 16-byte
// words of a 128-byte line it picks at random from a table far larger than the caches (4 GiB), so the number of
// lines fetched from memory is known -- one per lane and iteration.
//   mode 0: one load, at offset 0 of the line
//   mode 1: two loads, offsets 0 and 64 (both 64-byte halves of the line)
//   mode 2: two loads, offsets 0 and 16 (one half)
//   mode 3: three loads, offsets 0, 16, 32 (a 48-byte record inside one half)
//   mode 4: three loads, offsets 48, 64, 80 (a 48-byte record across the halves)
// With FETCH_SIZE read for the dispatch (rocprofv3 --pmc FETCH_SIZE -- ./fetch_probe <mode> [log2 lines]) this
// says how many bytes the counter tallies per line, and whether a miss brings in 64 or 128 bytes.
// build: hipcc -O3 --offload-arch=gfx950 -o fetch_probe fetch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIter = 64;
__device__ inline uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ table, uint32_t mask, float* out) {
  const uint32_t gtid = blockIdx.x * 256 + threadIdx.x;
  float sum = 0.0f;
#pragma unroll 4
  for (int it = 0; it < kIter; ++it) {
    const uint32_t idx = mix(gtid * kIter + it) & mask;
    const float4* line = table + (size_t)idx * 8;
    if (MODE == 0) sum += line[0].x;
    if (MODE == 1) sum += line[0].x + line[4].x;
    if (MODE == 2) sum += line[0].x + line[1].x;
    if (MODE == 3) sum += line[0].x + line[1].x + line[2].x;
    if (MODE == 4) sum += line[3].x + line[4].x + line[5].x;
  }
  out[gtid] = sum;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int log2_lines = argc > 2 ? atoi(argv[2]) : 25;  // 2^25 lines of 128 B = 4 GiB
  const uint32_t n = 1u << log2_lines;
  float4* table; float* out;
  const int blocks = 2048;
  CHECK(hipMalloc(&table, (size_t)n * 128));
  CHECK(hipMemset(table, 0, (size_t)n * 128));
  CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto launch = [&] {
    switch (mode) {
      case 0: gather<0><<<blocks, 256>>>(table, n - 1, out); break;
      case 1: gather<1><<<blocks, 256>>>(table, n - 1, out); break;
      case 2: gather<2><<<blocks, 256>>>(table, n - 1, out); break;
      case 3: gather<3><<<blocks, 256>>>(table, n - 1, out); break;
      default: gather<4><<<blocks, 256>>>(table, n - 1, out); break;
    }
  };
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  launch();
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double lines = (double)blocks * 256 * kIter;
  printf("mode %d, table %.0f MiB: %.0f lines per launch, %.3f ms, %.1f G lines/s (x128 B = %.2f TB/s, x64 B = %.2f TB/s)\n", mode,
         n * 128.0 / 1048576.0, lines, ms, lines / ms * 1e-6, lines * 128 / ms * 1e-9, lines * 64 / ms * 1e-9);
  return 0;
}
