// Effective shader clock under load.
//   spin:  every wave runs independent v_fma_f32 chains (the SIMD can issue one VALU instruction every 4 cycles),
//          timed with HIP events -> wave-instructions / s per SIMD;
//   both kernels also read s_memtime and s_memrealtime (100 MHz) around their work: the ratio of the two deltas
//   says what s_memtime counts, and -- if that is the shader clock -- the clock a kernel actually ran at;
//   chase: a memory-latency-bound kernel (dependent 16-byte loads over a large table, little arithmetic),
//          to see whether the clock differs from the arithmetic-bound one.
//   clock_probe <waves per SIMD>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ unsigned long long realtime() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ unsigned long long memtime() { return __builtin_amdgcn_s_memtime(); }
__global__ void __launch_bounds__(256) spin(float* out, unsigned long long* t, int iters, float a, float b) {
  const unsigned long long m0 = memtime(), r0 = realtime();
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x4) : "v"(a), "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x5) : "v"(a), "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x6) : "v"(a), "v"(b));
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x7) : "v"(a), "v"(b));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (blockIdx.x == 0 && threadIdx.x == 0) { t[0] = memtime() - m0; t[1] = realtime() - r0; }
}
__global__ void __launch_bounds__(256) chase(const uint4* table, unsigned mask, int steps, unsigned* out, unsigned long long* t) {
  const unsigned long long m0 = memtime(), r0 = realtime();
  unsigned at = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u & mask;
  for (int i = 0; i < steps; ++i) at = table[at].x & mask;
  out[blockIdx.x * blockDim.x + threadIdx.x] = at;
  if (blockIdx.x == 0 && threadIdx.x == 0) { t[0] = memtime() - m0; t[1] = realtime() - r0; }
}
int main(int argc, char** argv) {
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 1;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * waves_per_simd;  // a block of 256 threads = one wave per SIMD of a CU
  float* out;
  unsigned long long* t;
  CK(hipMalloc(&out, sizeof(float) * 256 * blocks));
  CK(hipMalloc(&t, 16));
  const int iters = 100000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, out, t, iters, 0.999f, 0.001f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2];
    CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
    const double per_simd = (double)iters * 128.0 * waves_per_simd;  // wave-instructions each SIMD issued
    printf("spin, %d wave(s) per SIMD on %d CUs: %.1f ms, %.3f G wave-instructions/s per SIMD = %.2f GHz at 4 cycles each; "
           "s_memtime / s_memrealtime = %.2f (%.2f GHz if the second counts 100 MHz)\n", waves_per_simd, cus, ms,
           per_simd / (ms * 1e-3) * 1e-9, per_simd * 4.0 / (ms * 1e-3) * 1e-9, (double)h[0] / (double)h[1], (double)h[0] / (double)h[1] * 0.1);
  }
  // 256 MiB table of 16-byte entries, each pointing at a pseudo-random other one
  const unsigned n = 1u << 24;
  std::vector<uint4> host(n);
  unsigned s = 12345u;
  for (unsigned i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; host[i] = make_uint4(s >> 8, 0, 0, 0); }
  uint4* table;
  CK(hipMalloc(&table, sizeof(uint4) * (size_t)n));
  CK(hipMemcpy(table, host.data(), sizeof(uint4) * (size_t)n, hipMemcpyHostToDevice));
  unsigned* out2;
  CK(hipMalloc(&out2, 4 * 256 * cus * 8));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(chase, dim3(cus * 8), dim3(256), 0, 0, table, n - 1, 20000, out2, t);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2];
    CK(hipMemcpy(h, t, 16, hipMemcpyDeviceToHost));
    printf("chase, 8 waves per SIMD: %.1f ms, %.1f G dependent loads/s; s_memtime / s_memrealtime = %.2f\n", ms,
           256.0 * cus * 8 * 20000 / (ms * 1e-3) * 1e-9, (double)h[0] / (double)h[1]);
  }
  return 0;
}
