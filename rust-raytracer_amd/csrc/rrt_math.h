// f32 vector math, the counter-based RNG and the camera for the device kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rrt {

#define RRT_DEV __device__ __forceinline__

constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.318309886183790671538f;
constexpr float kPiOver2 = 1.57079632679489661923f;
constexpr float kPiOver4 = 0.785398163397448309616f;

struct V3 {
  float x, y, z;
};
RRT_DEV V3 mk(float x, float y, float z) { return V3{x, y, z}; }
RRT_DEV V3 mk(const float* p) { return V3{p[0], p[1], p[2]}; }
RRT_DEV V3 mk(float4 v) { return V3{v.x, v.y, v.z}; }
RRT_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RRT_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RRT_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RRT_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
RRT_DEV V3 operator*(float s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
RRT_DEV V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RRT_DEV V3 operator/(V3 a, float s) {
  const float r = 1.0f / s;
  return V3{a.x * r, a.y * r, a.z * r};
}
// The library is compiled with -ffp-contract=off: the same source then rounds the same way in every
// kernel it is inlined into (a ray must find the same hit whether k_primary or k_extend traces it),
// and the fused operations that are wanted are written out.
RRT_DEV float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
RRT_DEV V3 cross(V3 a, V3 b) { return V3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))}; }
RRT_DEV float norm2(V3 a) { return dot(a, a); }
RRT_DEV float norm(V3 a) { return sqrtf(dot(a, a)); }
RRT_DEV V3 normalize(V3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
RRT_DEV bool is_zero(V3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
RRT_DEV float vmax(V3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
RRT_DEV float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
RRT_DEV float4 f4(V3 a, float w) { return make_float4(a.x, a.y, a.z, w); }

// helpers.rs:73-83
RRT_DEV void coordinate_system(V3 v1, V3* v2, V3* v3) {
  if (fabsf(v1.x) > fabsf(v1.y)) *v2 = mk(-v1.z, 0.0f, v1.x) / sqrtf(v1.x * v1.x + v1.z * v1.z);
  else *v2 = mk(0.0f, v1.z, -v1.y) / sqrtf(v1.y * v1.y + v1.z * v1.z);
  *v3 = cross(v1, *v2);
}

// helpers.rs:50-58
RRT_DEV V3 offset_ray_origin(V3 p, V3 n, V3 w) {
  V3 off = ((fabsf(n.x) + fabsf(n.y) + fabsf(n.z)) * 1e-4f) * n;
  if (dot(w, n) < 0.0f) off = -off;
  return p + off;
}

// helpers.rs:67-71 with nf = ng = 1
RRT_DEV float power_heuristic(float f, float g) { return (f * f) / (f * f + g * g); }

// helpers.rs:200-220 with the two RNG draws passed in
// Hardware sine and cosine (v_sin_f32 / v_cos_f32, absolute error about 1e-6): every angle in this
// library is a few turns at most, and the library versions are a hundred instructions apiece.
RRT_DEV void sincos_fast(float x, float* s, float* c) {
  *s = __sinf(x);
  *c = __cosf(x);
}

RRT_DEV void concentric_sample_disk(float u0, float u1, float* dx, float* dy) {
  const float ox = u0 * 2.0f - 1.0f, oy = u1 * 2.0f - 1.0f;
  if (ox == 0.0f && oy == 0.0f) { *dx = 0.0f; *dy = 0.0f; return; }
  float theta, r;
  if (fabsf(ox) > fabsf(oy)) { r = ox; theta = kPiOver4 * (oy / ox); }
  else { r = oy; theta = kPiOver2 - kPiOver4 * (ox / oy); }
  float s, c;
  sincos_fast(theta, &s, &c);
  *dx = r * c;
  *dy = r * s;
}

// ---------------------------------------------------------------- RNG
// Philox4x32-10, counter (pixel, sample, block, 0x52525431), key = seed.
// A draw is ((word >> 9) + 0.5) * 2^-23: 23 bits, never 0 or 1, exact in f32.
struct Rng4 {
  float u[4];
};

RRT_DEV Rng4 rng_block(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t block) {
  uint32_t c0 = pixel, c1 = sample, c2 = block, c3 = 0x52525431u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a high and a low one:
    // integer multiplies run at a quarter of the f32 rate, and a block has forty of them
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Rng4 r;
  r.u[0] = ((float)(c0 >> 9) + 0.5f) * (1.0f / 8388608.0f);
  r.u[1] = ((float)(c1 >> 9) + 0.5f) * (1.0f / 8388608.0f);
  r.u[2] = ((float)(c2 >> 9) + 0.5f) * (1.0f / 8388608.0f);
  r.u[3] = ((float)(c3 >> 9) + 0.5f) * (1.0f / 8388608.0f);
  return r;
}

}  // namespace rrt
