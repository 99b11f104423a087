// CPU oracle: a double-precision restatement of the reference integrator.
// TEST INFRASTRUCTURE ONLY -- see oracle.h for what pins it and what does not.
//
// The structure follows the reference file by file so that each function can
// be checked against its source: helpers.rs, bsdf/helpers.rs, fresnel.rs,
// microfacet_distribution.rs, the five BxDFs, bsdf.rs, materials/*.rs,
// objects/*.rs, surface_interaction.rs, lights/*.rs, textures/mip_map.rs,
// renderer.rs (scene queries, bucket scheduler), tracer.rs, camera.rs, film.rs.
// Deliberate differences, all forced by the missing crates:
//   * random numbers come from a counter-based generator keyed by
//     (seed, pixel, sample, block, index) instead of sobol 1.0.2 / rand 0.9.2
//     (src/sampler.rs, rand::rng()); the device draws the same numbers;
//   * the bvh 0.8.0 crate is replaced by an own BVH with the same contract:
//     f32 boxes, one shape per leaf, traverse_iterator = every shape whose
//     box the f32 ray hits, in no particular order and without t-pruning.
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

namespace {

const double PI = 3.14159265358979323846;
const double FRAC_1_PI = 0.318309886183790671537767526745028724;
const double FRAC_PI_2 = 1.57079632679489661923132169163975144;
const double FRAC_PI_4 = 0.785398163397448309615660845819875721;
const double F64_EPSILON = 2.220446049250313e-16;

struct V3 {
  double x = 0, y = 0, z = 0;
  V3() {}
  V3(double a, double b, double c) : x(a), y(b), z(c) {}
  double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, double s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(double s, V3 a) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, double s) { return V3(a.x / s, a.y / s, a.z / s); }
inline V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline V3 cmul(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }  // component_mul
inline double norm2(V3 a) { return dot(a, a); }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a / norm(a); }
inline V3 vabs(V3 a) { return V3(std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)); }
inline bool is_zero(V3 a) { return a.x == 0.0 && a.y == 0.0 && a.z == 0.0; }
inline double vmax(V3 a) { return std::max(a.x, std::max(a.y, a.z)); }
inline double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline V3 v3(const double* p) { return V3(p[0], p[1], p[2]); }

struct Ray {
  V3 point, direction;
};

// ------------------------------------------------------------------ RNG
// Philox4x32-10 (Salmon et al. 2011).  counter = (pixel, sample, block,
// 0x52525431), key = seed.  A draw is (word >> 9 + 0.5) * 2^-23: 23 bits, never
// 0 or 1, exactly representable in f32 so that the device sees the same value.
inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

struct RngBlock {
  double u[4];
};

inline RngBlock rng_block(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t block) {
  uint32_t c[4] = {pixel, sample, block, 0x52525431u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  RngBlock b;
  for (int i = 0; i < 4; ++i) b.u[i] = ((double)(c[i] >> 9) + 0.5) * (1.0 / 8388608.0);
  return b;
}

// Random number blocks of one camera path.  Block 0: film x, film y, lens u0,
// lens u1.  Bounce b: block 1+3b = light select, light u0, light u1, roulette;
// block 2+3b = lobe choice, u0, u1 of the MIS BSDF sample; block 3+3b = lobe
// choice, u0, u1 of the path-continuation BSDF sample.
struct PathRng {
  uint64_t seed;
  uint32_t pixel, sample;
  RngBlock block(uint32_t b) const { return rng_block(seed, pixel, sample, b); }
};

// ------------------------------------------------------------------ helpers.rs
inline V3 vector_reflect(V3 vec, V3 normal) { return -vec + 2.0 * dot(vec, normal) * normal; }  // :27-29

inline V3 offset_ray_origin(V3 p, V3 normal, V3 w) {  // :50-58
  V3 origin_offset = dot(vabs(normal), V3(1e-4, 1e-4, 1e-4)) * normal;
  if (dot(w, normal) < 0.0) origin_offset = -origin_offset;
  return p + origin_offset;
}

inline double power_heuristic(int nf, double f_pdf, int ng, double g_pdf) {  // :67-71
  const double f = nf * f_pdf, g = ng * g_pdf;
  return (f * f) / (f * f + g * g);
}

inline void coordinate_system(V3 v1, V3* v2, V3* v3_) {  // :73-83
  if (std::fabs(v1.x) > std::fabs(v1.y))
    *v2 = V3(-v1.z, 0.0, v1.x) / std::sqrt(v1.x * v1.x + v1.z * v1.z);
  else
    *v2 = V3(0.0, v1.z, -v1.y) / std::sqrt(v1.y * v1.y + v1.z * v1.z);
  *v3_ = cross(v1, *v2);
}

inline double gamma_err(double n) { return (n * F64_EPSILON) / (1.0 - n * F64_EPSILON); }  // :85-87

inline V3 face_forward(V3 n, V3 v) { return dot(n, v) < 0.0 ? -n : n; }  // :89-95

inline double get_fresnel_ratio(V3 normal, V3 angle_of_incidence, double ior) {  // :97-128
  const double c = std::fabs(dot(normal, angle_of_incidence));
  const double sqrg = ior * ior + c * c - 1.0;
  if (sqrg <= 0.0) return 1.0;
  const double g = std::sqrt(sqrg);
  const double quot1 = (g - c) / (g + c);
  const double quot2 = (c * (g + c) - 1.0) / (c * (g - c) + 1.0);
  double f = 0.5 * (quot1 * quot1) * (1.0 + (quot2 * quot2));
  if (f > 1.0) f = 1.0;
  if (f < 0.0) f = 0.0;
  return f;
}

inline bool refract(V3 wi, V3 normal, double ior, V3* out) {  // :130-147
  const double cos_theta_i = dot(normal, wi);
  const double sin_2_theta_i = std::max(0.0, 1.0 - cos_theta_i * cos_theta_i);
  const double sin_2_theta_t = ior * ior * sin_2_theta_i;
  if (sin_2_theta_t > 1.0) return false;
  const double cos_theta_t = std::sqrt(1.0 - sin_2_theta_t);
  *out = ior * -wi + (ior * cos_theta_i - cos_theta_t) * normal;
  return true;
}

inline int max_dimension_vec_3(V3 v) {  // :182-194
  if (v.x > v.y) return v.x > v.z ? 0 : 2;
  return v.y > v.z ? 1 : 2;
}

inline V3 permute(V3 v, int x, int y, int z) { return V3(v[x], v[y], v[z]); }  // :196-198

// :200-220, with the two thread-RNG draws replaced by (u0, u1)
inline void concentric_sample_disk(double u0, double u1, double* dx, double* dy) {
  const double ox = u0 * 2.0 - 1.0, oy = u1 * 2.0 - 1.0;
  if (ox == 0.0 && oy == 0.0) { *dx = 0.0; *dy = 0.0; return; }
  double theta, r;
  if (std::fabs(ox) > std::fabs(oy)) {
    r = ox;
    theta = FRAC_PI_4 * (oy / ox);
  } else {
    r = oy;
    theta = FRAC_PI_2 - FRAC_PI_4 * (ox / oy);
  }
  *dx = r * std::cos(theta);
  *dy = r * std::sin(theta);
}

inline V3 spherical_direction(double sin_theta, double cos_theta, double phi) {  // :222-224
  return V3(sin_theta * std::cos(phi), sin_theta * std::sin(phi), cos_theta);
}
inline double spherical_theta(V3 v) { return std::acos(clampd(v.y, -1.0, 1.0)); }  // :226-228
inline double spherical_phi(V3 v) {  // :230-238
  const double p = std::atan2(v.x, v.z);
  return p < 0.0 ? p + 2.0 * PI : p;
}

// ------------------------------------------------------------------ bsdf/helpers.rs
inline double cos_theta(V3 a) { return a.z; }
inline double cos_2_theta(V3 a) { return a.z * a.z; }
inline double abs_cos_theta(V3 a) { return std::fabs(a.z); }
inline double sin_2_theta(V3 w) { return 1.0 - std::max(cos_2_theta(w), 0.0); }  // :18-20 (as written)
inline double sin_theta(V3 w) { return std::sqrt(sin_2_theta(w)); }
inline double tan_theta(V3 w) { return sin_theta(w) / cos_theta(w); }
inline double tan_2_theta(V3 w) { return sin_2_theta(w) / cos_2_theta(w); }
inline double cos_phi(V3 w) {  // :30-37
  const double s = sin_theta(w);
  return s == 0.0 ? 1.0 : clampd(w.x / s, -1.0, 1.0);
}
inline double sin_phi(V3 w) {  // :38-45 (returns 1.0, not 0.0, at the pole)
  const double s = sin_theta(w);
  return s == 0.0 ? 1.0 : clampd(w.y / s, -1.0, 1.0);
}
inline double cos_2_phi(V3 w) { return cos_phi(w) * cos_phi(w); }
inline double sin_2_phi(V3 w) { return sin_phi(w) * sin_phi(w); }
inline bool same_hemisphere(V3 a, V3 b) { return a.z * b.z > 0.0; }
inline V3 get_cosine_weighted_in_hemisphere(double u0, double u1) {  // :57-62
  double dx, dy;
  concentric_sample_disk(u0, u1, &dx, &dy);
  const double z = std::sqrt(std::max(0.0, 1.0 - dx * dx - dy * dy));
  return V3(dx, dy, z);
}

// ------------------------------------------------------------------ fresnel.rs
inline double fresnel_dielectric(double eta_i, double eta_t, double cos_theta_i) {  // :47-74
  cos_theta_i = clampd(cos_theta_i, -1.0, 1.0);
  if (cos_theta_i <= 0.0) {
    std::swap(eta_i, eta_t);
    cos_theta_i = std::fabs(cos_theta_i);
  }
  const double sin_theta_i = std::sqrt(std::max(1.0 - cos_theta_i * cos_theta_i, 0.0));
  const double sin_theta_t = eta_i / eta_t * sin_theta_i;
  if (sin_theta_t >= 1.0) return 1.0;
  const double cos_theta_t = std::sqrt(std::max(1.0 - sin_theta_t * sin_theta_t, 0.0));
  const double rpar_l = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) / ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
  const double rper_n = ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) / ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
  return (rpar_l * rpar_l + rper_n * rper_n) / 2.0;
}

struct Fresnel {
  bool dielectric = false;  // false = FresnelNoop (evaluates to 1)
  double eta_i = 1.0, eta_t = 1.0;
  double evaluate(double c) const { return dielectric ? fresnel_dielectric(eta_i, eta_t, c) : 1.0; }
};

// ------------------------------------------------------------------ microfacet_distribution.rs
struct TrowbridgeReitz {
  double alpha_x = 0, alpha_y = 0;
  bool sample_visible_area = true;

  static double roughness_to_alpha(double roughness) {  // :127-142
    roughness = std::max(roughness, 1.0e-3);
    return roughness * roughness;
  }

  double d(V3 wh) const {  // :144-156
    const double t2 = tan_2_theta(wh);
    if (std::isinf(t2)) return 0.0;
    const double cos_4_theta = cos_2_theta(wh) * cos_2_theta(wh);
    const double e = (cos_2_phi(wh) / (alpha_x * alpha_x) + sin_2_phi(wh) / (alpha_y * alpha_y)) * t2;
    return 1.0 / (PI * alpha_x * alpha_y * cos_4_theta * (1.0 + e) * (1.0 + e));
  }

  double lambda(V3 w) const {  // :158-171
    const double abs_tan_theta = std::fabs(tan_theta(w));
    if (std::isinf(abs_tan_theta)) return 0.0;
    const double alpha = std::sqrt(cos_2_phi(w) * alpha_x * alpha_x + sin_2_phi(w) * alpha_y * alpha_y);
    const double a2t2 = (alpha * abs_tan_theta) * (alpha * abs_tan_theta);
    return (-1.0 + std::sqrt(1.0 + a2t2)) / 2.0;
  }

  double g1(V3 w) const { return 1.0 / (1.0 + lambda(w)); }                         // :19-21
  double g(V3 wo, V3 wi) const { return 1.0 / (1.0 + lambda(wo) + lambda(wi)); }   // :23-25

  double pdf(V3 wo, V3 wh) const {  // :31-37
    if (sample_visible_area) return d(wh) * g1(wo) * std::fabs(dot(wh, wo)) / abs_cos_theta(wo);
    return d(wh) * abs_cos_theta(wh);
  }

  static void sample_11(double cos_theta_, double u1, double u2, double* sx, double* sy) {  // :56-103
    if (cos_theta_ > 0.9999) {
      const double r = std::sqrt(u1 / (1.0 - u1));
      const double phi = 2.0 * PI * u2;
      *sx = r * std::cos(phi);
      *sy = r * std::sin(phi);
      return;
    }
    const double sin_theta_ = std::sqrt(std::max(0.0, 1.0 - cos_theta_ * cos_theta_));
    const double tan_theta_ = sin_theta_ / cos_theta_;
    double a = 1.0 / tan_theta_;
    const double g1_ = 2.0 / (1.0 + std::sqrt(1.0 + 1.0 / (a * a)));
    a = 2.0 * u1 / g1_ - 1.0;
    double tmp = 1.0 / (a * a - 1.0);
    if (tmp > 1e10) tmp = 1e10;
    const double b = tan_theta_;
    const double dd = std::sqrt(std::max(b * b * tmp * tmp - (a * a - b * b) * tmp, 0.0));
    const double slope_x_1 = b * tmp - dd;
    const double slope_x_2 = b * tmp + dd;
    const double slope_x = (a < 0.0 || slope_x_2 > (1.0 / tan_theta_)) ? slope_x_1 : slope_x_2;
    double s;
    if (u2 > 0.5) { s = 1.0; u2 = 2.0 * (u2 - 0.5); }
    else { s = -1.0; u2 = 2.0 * (0.5 - u2); }
    const double z = (u2 * (u2 * (u2 * 0.27385 - 0.73369) + 0.46341)) /
                     (u2 * (u2 * (u2 * 0.093073 + 0.309420) - 1.0) + 0.597999);
    *sx = slope_x;
    *sy = s * z * std::sqrt(1.0 + slope_x * slope_x);
    // the reference asserts finiteness here (:97-100) and would panic otherwise
  }

  static V3 sample(V3 wi, double ax, double ay, double u1, double u2) {  // :105-123
    const V3 wi_stretched = normalize(V3(ax * wi.x, ay * wi.y, wi.z));
    double slope_x, slope_y;
    sample_11(cos_theta(wi_stretched), u1, u2, &slope_x, &slope_y);
    const double tmp = cos_phi(wi_stretched) * slope_x - sin_phi(wi_stretched) * slope_y;
    slope_y = sin_phi(wi_stretched) * slope_x + cos_phi(wi_stretched) * slope_y;
    slope_x = tmp;
    slope_x *= ax;
    slope_y *= ay;
    return normalize(V3(-slope_x, -slope_y, 1.0));
  }

  V3 sample_wh(V3 wo, double u0, double u1) const {  // :177-222
    if (!sample_visible_area) {
      double cos_theta_ = 0.0;
      double phi = 2.0 * PI * u1;
      if (alpha_x == alpha_y) {
        const double tan_theta_2 = alpha_x * alpha_x * u0 / (1.0 - u0);
        cos_theta_ = 1.0 / std::sqrt(1.0 + tan_theta_2);
      } else {
        phi = std::atan(alpha_x / alpha_y * std::tan(2.0 * PI * u1 + 0.5 * PI));
        if (u1 > 0.5) phi += PI;
        const double sp = std::sin(phi), cp = std::cos(phi);
        const double alpha2 = 1.0 / (cp * cp / (alpha_x * alpha_x) + sp * sp / (alpha_y * alpha_y));
        const double tan_theta_2 = alpha2 * u0 / (1.0 - u0);
        cos_theta_ = 1.0 / std::sqrt(1.0 + tan_theta_2);
      }
      const double sin_theta_ = std::sqrt(std::max(0.0, 1.0 - cos_theta_ * cos_theta_));
      const V3 wh = spherical_direction(sin_theta_, cos_theta_, phi);
      return same_hemisphere(wo, wh) ? wh : -wh;
    }
    const bool flip = wo.z < 0.0;
    const V3 wh = sample(flip ? -wo : wo, alpha_x, alpha_y, u0, u1);
    return flip ? -wh : wh;
  }
};

// ------------------------------------------------------------------ bsdf.rs: flags and BxDFs
enum : uint32_t {  // :286-298
  REFLECTION = 1, REFRACTION = 2, DIFFUSE = 4, SPECULAR = 8, TRANSMISSION = 16, GLOSSY = 32,
  ALL = REFLECTION | REFRACTION | DIFFUSE | SPECULAR | GLOSSY, NONE = 0
};

struct Bxdf {
  enum Kind { Lambertian, SpecularReflection, SpecularTransmission, OrenNayar, MicrofacetReflection } kind;
  V3 color;                   // reflectance / refraction colour
  double a = 0, b = 0;        // OrenNayar
  TrowbridgeReitz distribution;
  Fresnel fresnel;
  double eta_a = 1, eta_b = 1;  // SpecularTransmission

  uint32_t get_type_flags() const {
    switch (kind) {
      case Lambertian: return REFLECTION | DIFFUSE;             // lambertian.rs:19-21
      case OrenNayar: return REFLECTION | DIFFUSE;              // oren_nayar.rs:31-33
      case MicrofacetReflection: return REFLECTION | GLOSSY;    // microfacet_reflection.rs:37-39
      case SpecularReflection: return REFLECTION | SPECULAR;    // specular_reflection.rs:23-25
      case SpecularTransmission: return REFRACTION | SPECULAR;  // specular_transmission.rs:41-43
    }
    return NONE;
  }

  V3 f(V3 wo, V3 wi) const {
    switch (kind) {
      case Lambertian:  // lambertian.rs:23-25
        return color * FRAC_1_PI;
      case OrenNayar: {  // oren_nayar.rs:35-61
        const double sin_theta_i = sin_theta(wi), sin_theta_o = sin_theta(wo);
        double max_cos = 0.0;
        if (sin_theta_i > 1e-4 && sin_theta_o > 1e-4) {
          const double d_cos = cos_phi(wi) * cos_phi(wo) + sin_phi(wi) * sin_phi(wo);
          max_cos = std::max(d_cos, 0.0);
        }
        double sin_alpha, tan_beta;
        if (abs_cos_theta(wi) > abs_cos_theta(wo)) {
          sin_alpha = sin_theta_o;
          tan_beta = sin_theta_i / abs_cos_theta(wi);
        } else {
          sin_alpha = sin_theta_i;
          tan_beta = sin_theta_o / abs_cos_theta(wo);
        }
        return color * FRAC_1_PI * (a + b * max_cos * sin_alpha * tan_beta);
      }
      case MicrofacetReflection: {  // microfacet_reflection.rs:41-60
        const double cos_theta_o = abs_cos_theta(wo), cos_theta_i = abs_cos_theta(wi);
        if (cos_theta_o == 0.0 || cos_theta_i == 0.0) return V3();
        V3 wh = wi + wo;
        if (is_zero(wh)) return V3();
        wh = normalize(wh);
        const double fr = fresnel.evaluate(std::fabs(dot(wi, face_forward(wh, V3(0, 0, 1)))));
        return color * distribution.d(wh) * distribution.g(wo, wi) * fr / (4.0 * cos_theta_i * cos_theta_o);
      }
      case SpecularReflection:    // specular_reflection.rs:27-29
      case SpecularTransmission:  // specular_transmission.rs:45-47
        return V3();
    }
    return V3();
  }

  double pdf(V3 wo, V3 wi) const {
    switch (kind) {
      case Lambertian:
      case OrenNayar:  // trait default, bsdf.rs:312-318
        return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * FRAC_1_PI : 0.0;
      case MicrofacetReflection: {  // microfacet_reflection.rs:62-71
        if (!same_hemisphere(wo, wi)) return 0.0;
        const V3 wh = normalize(wo + wi);
        return distribution.pdf(wo, wh) / (4.0 * std::fabs(dot(wo, wh)));
      }
      case SpecularReflection:    // specular_reflection.rs:31-33
      case SpecularTransmission:  // specular_transmission.rs:49-51
        return 1.0;
    }
    return 0.0;
  }

  // returns (wi, pdf, f)
  void sample_f(double u0, double u1, V3 wo, V3* wi, double* pdf_, V3* f_) const {
    switch (kind) {
      case Lambertian:
      case OrenNayar: {  // trait default, bsdf.rs:319-326
        V3 w = get_cosine_weighted_in_hemisphere(u0, u1);
        if (wo.z < 0.0) w.z = -w.z;
        *wi = w;
        *pdf_ = pdf(wo, w);
        *f_ = f(wo, w);
        return;
      }
      case MicrofacetReflection: {  // microfacet_reflection.rs:73-105
        *wi = V3(); *pdf_ = 0.0; *f_ = V3();
        if (wo.z == 0.0) return;
        const V3 wh = distribution.sample_wh(wo, u0, u1);
        if (dot(wo, wh) < 0.0) return;
        const V3 w = vector_reflect(wo, wh);
        *wi = w;
        if (!same_hemisphere(wo, w)) return;
        const double wo_dot_wh = std::fabs(dot(wo, wh));
        if (wo_dot_wh < 1e-6) return;
        *pdf_ = distribution.pdf(wo, wh) / (4.0 * wo_dot_wh);
        *f_ = f(wo, w);
        return;
      }
      case SpecularReflection: {  // specular_reflection.rs:35-41
        const V3 w(-wo.x, -wo.y, wo.z);
        *wi = w;
        *pdf_ = pdf(wo, w);
        *f_ = fresnel.evaluate(cos_theta(w)) * color / abs_cos_theta(w);
        return;
      }
      case SpecularTransmission: {  // specular_transmission.rs:53-77
        double eta_i, eta_t;
        if (cos_theta(wo) > 0.0) { eta_i = eta_a; eta_t = eta_b; }
        else { eta_i = eta_b; eta_t = eta_a; }
        const V3 normal = face_forward(V3(0, 0, 1), wo);
        V3 w;
        if (!refract(wo, normal, eta_i / eta_t, &w)) { *wi = V3(); *pdf_ = 0.0; *f_ = V3(); return; }
        const double fresnel_eval = fresnel.evaluate(cos_theta(w));
        V3 ft = cmul(color, V3(1.0, 1.0, 1.0) - V3(fresnel_eval, fresnel_eval, fresnel_eval));
        ft = ft * ((eta_i * eta_i) / (eta_t * eta_t));  // TransportMode::Radiance
        *wi = w;
        *pdf_ = 1.0;
        *f_ = ft / abs_cos_theta(w);
        return;
      }
    }
  }
};

// ------------------------------------------------------------------ bsdf.rs: Bsdf
const int MAX_BXDF_COUNT = 5;

inline double shift_cos_in(double cos_in, double frequency_multiplier) {  // :280-284
  cos_in = std::min(cos_in, 1.0);
  const double angle = std::acos(cos_in);
  return std::max(std::cos(angle * frequency_multiplier), 0.0) / cos_in;
}

struct BsdfSampleResult {
  V3 wi;
  double pdf = 0.0;
  V3 f;
  uint32_t sampled_flags = NONE;
};

struct Bsdf {
  Bxdf bxdfs[MAX_BXDF_COUNT];
  int count = 0;
  V3 geometry_normal, shading_normal, ss, ts;

  void add(const Bxdf& b) {  // :55-61 (panics when full)
    if (count < MAX_BXDF_COUNT) bxdfs[count++] = b;
  }
  V3 world_to_local(V3 v) const { return V3(dot(v, ss), dot(v, ts), dot(v, shading_normal)); }  // :243-249
  V3 local_to_world(V3 v) const {                                                            // :251-257
    return V3(ss.x * v.x + ts.x * v.y + shading_normal.x * v.z, ss.y * v.x + ts.y * v.y + shading_normal.y * v.z,
              ss.z * v.x + ts.z * v.y + shading_normal.z * v.z);
  }

  // :74-183.  (u_lobe, u0, u1) stand in for the three thread-RNG draws; the
  // sample_u argument of the reference is ignored there too (:117-124).
  BsdfSampleResult sample_f(V3 wo_world, uint32_t flags, double u_lobe, double u0, double u1) const {
    BsdfSampleResult res;
    int matching[MAX_BXDF_COUNT];
    int matching_count = 0;
    for (int i = 0; i < count; ++i)
      if (bxdfs[i].get_type_flags() & flags) matching[matching_count++] = i;
    if (matching_count == 0) return res;
    const V3 wo = world_to_local(wo_world);
    if (wo.z == 0.0) return res;
    u0 = std::max(std::min(u0, 1.0 - F64_EPSILON), F64_EPSILON);
    u1 = std::max(std::min(u1, 1.0 - F64_EPSILON), F64_EPSILON);
    const int pick = std::min((int)(u_lobe * matching_count), matching_count - 1);
    const int chosen_index = matching[pick];
    const Bxdf& bxdf = bxdfs[chosen_index];
    V3 wi, f;
    double pdf = 0.0;
    bxdf.sample_f(u0, u1, wo, &wi, &pdf, &f);
    res.sampled_flags = bxdf.get_type_flags();
    if (pdf == 0.0) return res;
    const V3 wi_world = local_to_world(wi);
    if (!(bxdf.get_type_flags() & SPECULAR) || matching_count > 1) {
      for (int i = 0; i < count; ++i)
        if (i != chosen_index && (bxdfs[i].get_type_flags() & flags)) pdf += bxdf.pdf(wo, wi);  // sic: the chosen lobe's pdf
    }
    if (matching_count > 1) pdf /= (double)matching_count;
    if (!(bxdf.get_type_flags() & SPECULAR)) {
      const bool reflect = dot(wi_world, geometry_normal) * dot(wo_world, geometry_normal) > 0.0;
      f = V3();
      for (int i = 0; i < count; ++i) {
        const uint32_t t = bxdfs[i].get_type_flags();
        if ((t & flags) && ((reflect && (t & REFLECTION)) || (!reflect && (t & TRANSMISSION)))) f += bxdfs[i].f(wo, wi);
      }
    }
    res.wi = wi_world;
    res.pdf = pdf;
    res.f = f;
    return res;
  }

  V3 f(V3 wo_world, V3 wi_world, uint32_t flags) const {  // :185-213
    const V3 wi = world_to_local(wi_world), wo = world_to_local(wo_world);
    const bool reflect = dot(wi_world, geometry_normal) * dot(wo_world, geometry_normal) > 0.0;
    const uint32_t must_match = reflect ? REFLECTION : TRANSMISSION;
    V3 f;
    for (int i = 0; i < count; ++i) {
      const uint32_t t = bxdfs[i].get_type_flags();
      if ((t & flags) && (t & must_match) == must_match) f += bxdfs[i].f(wo, wi);
    }
    return f * shift_cos_in(dot(wi_world, shading_normal), 1.002);
  }

  double pdf(V3 wo_world, V3 wi_world, uint32_t flags) const {  // :215-241
    const V3 wi = world_to_local(wi_world), wo = world_to_local(wo_world);
    if (wo.z == 0.0) return 0.0;
    double p = 0.0;
    int matching = 0;
    for (int i = 0; i < count; ++i)
      if (bxdfs[i].get_type_flags() & flags) {
        ++matching;
        p += bxdfs[i].pdf(wo, wi);
      }
    return matching > 0 ? p / (double)matching : 0.0;
  }
};

// ------------------------------------------------------------------ surface_interaction.rs
struct SurfaceInteraction {
  bool has_bsdf = false;
  Bsdf bsdf;
  V3 point, geometry_normal, shading_normal, wo, ss, ts;
};

inline SurfaceInteraction make_si(V3 point, V3 geometry_normal, V3 wo, V3 ss, V3 ts) {  // :45-72
  SurfaceInteraction si;
  si.shading_normal = normalize(cross(ss, ts));
  si.geometry_normal = face_forward(geometry_normal, si.shading_normal);
  si.point = point;
  si.wo = wo;
  si.ss = ss;
  si.ts = ts;
  return si;
}

// ------------------------------------------------------------------ materials/*.rs
inline void ensure_bsdf(SurfaceInteraction* si) {  // si.bsdf.unwrap_or(Bsdf::new(*si, None))
  if (si->has_bsdf) return;
  si->bsdf = Bsdf();
  si->bsdf.geometry_normal = si->geometry_normal;
  si->bsdf.shading_normal = si->shading_normal;
  si->bsdf.ss = si->ss;
  si->bsdf.ts = si->ts;
  si->has_bsdf = true;
}

void compute_scattering_functions(const rrt_material& m, SurfaceInteraction* si) {
  ensure_bsdf(si);
  const V3 kd = v3(m.kd), ks = v3(m.ks);
  switch (m.kind) {
    case RRT_MAT_MATTE: {  // matte.rs:26-40
      const double sigma = clampd(m.roughness, 0.0, 90.0);
      if (!is_zero(kd)) {
        Bxdf b;
        b.color = kd;
        if (sigma == 0.0) {
          b.kind = Bxdf::Lambertian;
        } else {
          b.kind = Bxdf::OrenNayar;  // OrenNayar::new gets the raw roughness (oren_nayar.rs:17-27)
          const double sigma2 = m.roughness * m.roughness;
          b.a = 1.0 - (sigma2 / (2.0 * (sigma2 + 0.33)));
          b.b = 0.45 * sigma2 / (sigma2 + 0.09);
        }
        si->bsdf.add(b);
      }
      break;
    }
    case RRT_MAT_PLASTIC: {  // plastic.rs:36-65
      if (!is_zero(kd)) {
        Bxdf b;
        b.kind = Bxdf::Lambertian;
        b.color = kd;
        si->bsdf.add(b);
      }
      if (!is_zero(ks)) {
        Bxdf b;
        b.color = ks;
        b.fresnel.dielectric = true;
        b.fresnel.eta_i = 1.0;
        b.fresnel.eta_t = m.ior;
        if (m.roughness < 1.0e-3) {
          b.kind = Bxdf::SpecularReflection;
        } else {
          b.kind = Bxdf::MicrofacetReflection;
          const double alpha = TrowbridgeReitz::roughness_to_alpha(m.roughness);
          b.distribution.alpha_x = alpha;
          b.distribution.alpha_y = alpha;
          b.distribution.sample_visible_area = true;
        }
        si->bsdf.add(b);
      }
      break;
    }
    case RRT_MAT_MIRROR: {  // mirror.rs:21-30
      Bxdf b;
      b.kind = Bxdf::SpecularReflection;
      b.color = kd;
      si->bsdf.add(b);
      break;
    }
    case RRT_MAT_GLASS: {  // glass.rs:32-48
      Bxdf t;
      t.kind = Bxdf::SpecularTransmission;
      t.color = ks;
      t.eta_a = 1.0;
      t.eta_b = m.ior;
      t.fresnel.dielectric = true;
      t.fresnel.eta_i = 1.0;
      t.fresnel.eta_t = m.ior;
      si->bsdf.add(t);
      Bxdf r;
      r.kind = Bxdf::SpecularReflection;
      r.color = kd;
      r.fresnel.dielectric = true;
      r.fresnel.eta_i = 1.0;
      r.fresnel.eta_t = m.ior;
      si->bsdf.add(r);
      break;
    }
  }
}

// ------------------------------------------------------------------ objects/*.rs
struct Object {
  enum Kind { Triangle, Rectangle, Plane, Sphere } kind;
  V3 p0, p1, p2, n0, n1, n2;  // triangle
  V3 position, side_a, side_b;  // rectangle (position also: plane point, sphere centre)
  V3 normal;                  // plane
  double radius = 0;
  int material = -1;  // get_materials(): zero or one material
  int light = -1;     // get_light()
};

struct Aabb {
  float min[3], max[3];
};

// objects/triangle.rs:302-333
inline V3 project_on_plane(V3 p, V3 origin, V3 normal) { return p - dot(p - origin, normal) * normal; }
inline V3 compute_shading_position(V3 p_hit, V3 v0, V3 v1, V3 v2, V3 n0, V3 n1, V3 n2, double b0, double b1,
                                   double b2, V3 shading_normal) {
  const V3 q0 = project_on_plane(p_hit, v0, n0);
  const V3 q1 = project_on_plane(p_hit, v1, n1);
  const V3 q2 = project_on_plane(p_hit, v2, n2);
  const V3 shading_pos = b0 * q0 + b1 * q1 + b2 * q2;
  const bool convex = dot(shading_pos - p_hit, shading_normal) > 0.0;
  return convex ? shading_pos : p_hit;
}

struct TriDetail {  // what a rejected / accepted triangle test looked like (for orc_intersect_margin)
  double b0 = 0, b1 = 0, b2 = 0, t = 0;
  bool plane_hit = false;  // valid t, only the edge test decided
};

// objects/triangle.rs:119-257
bool triangle_intersect(const Object& o, const Ray& ray, double* t_out, SurfaceInteraction* si, TriDetail* detail) {
  const V3 p0 = o.p0, p1 = o.p1, p2 = o.p2;
  V3 p0t = p0 - ray.point, p1t = p1 - ray.point, p2t = p2 - ray.point;
  const int kz = max_dimension_vec_3(vabs(ray.direction));
  const int kx = (kz + 1) % 3;
  const int ky = (kx + 1) % 3;
  const V3 d = permute(ray.direction, kx, ky, kz);
  p0t = permute(p0t, kx, ky, kz);
  p1t = permute(p1t, kx, ky, kz);
  p2t = permute(p2t, kx, ky, kz);
  const double s_x = -d.x / d.z, s_y = -d.y / d.z, s_z = 1.0 / d.z;
  p0t.x += s_x * p0t.z; p0t.y += s_y * p0t.z;
  p1t.x += s_x * p1t.z; p1t.y += s_y * p1t.z;
  p2t.x += s_x * p2t.z; p2t.y += s_y * p2t.z;
  const double e0 = p1t.x * p2t.y - p1t.y * p2t.x;
  const double e1 = p2t.x * p0t.y - p2t.y * p0t.x;
  const double e2 = p0t.x * p1t.y - p0t.y * p1t.x;
  const bool outside = (e0 < 0.0 || e1 < 0.0 || e2 < 0.0) && (e0 > 0.0 || e1 > 0.0 || e2 > 0.0);
  const double det = e0 + e1 + e2;
  if (detail && det != 0.0) {
    const double ts = (e0 * p0t.z + e1 * p1t.z + e2 * p2t.z) * s_z / det;
    detail->b0 = e0 / det; detail->b1 = e1 / det; detail->b2 = e2 / det; detail->t = ts;
    detail->plane_hit = ts >= F64_EPSILON && ts <= 1000.0;
  }
  if (outside) return false;
  if (det == 0.0) return false;
  p0t.z *= s_z; p1t.z *= s_z; p2t.z *= s_z;
  const double t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
  if (det < 0.0 && (t_scaled >= 0.0 || t_scaled < 1000.0 * det)) return false;
  if (det > 0.0 && (t_scaled <= 0.0 || t_scaled > 1000.0 * det)) return false;
  const double inv_det = 1.0 / det;
  const double b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
  const double t = t_scaled * inv_det;
  if (t < F64_EPSILON) return false;

  // uv = (0,0), (1,0), (1,1)  (:183-204)
  const double duv02x = 0.0 - 1.0, duv02y = 0.0 - 1.0, duv12x = 1.0 - 1.0, duv12y = 0.0 - 1.0;
  const V3 dp02 = p0 - p2, dp12 = p1 - p2;
  const double determinant = duv02x * duv12y - duv02y * duv12x;
  V3 dpdu, dpdv;
  if (determinant == 0.0) {
    coordinate_system(normalize(cross(p2 - p0, p1 - p0)), &dpdu, &dpdv);
  } else {
    const double inv = 1.0 / determinant;
    dpdu = (duv12y * dp02 - duv02y * dp12) * inv;
    dpdv = (-duv12x * dp02 + duv02x * dp12) * inv;
  }
  const V3 shading_normal = normalize(b0 * o.n0 + b1 * o.n1 + b2 * o.n2);
  V3 ss = normalize(dpdu);
  V3 ts = cross(shading_normal, ss);
  if (norm2(ts) > 0.0) {
    ts = normalize(ts);
    ss = cross(ts, shading_normal);
  } else {
    coordinate_system(shading_normal, &ss, &ts);
  }
  V3 p_hit = b0 * p0 + b1 * p1 + b2 * p2;
  const V3 geometry_normal = normalize(cross(p2 - p0, p1 - p0));
  p_hit = compute_shading_position(p_hit, p0, p1, p2, o.n0, o.n1, o.n2, b0, b1, b2, geometry_normal);
  (void)dpdv;
  (void)gamma_err;
  *t_out = t;
  *si = make_si(p_hit, geometry_normal, -ray.direction, ss, ts);
  return true;
}

inline V3 rectangle_normal(const Object& o) { return normalize(cross(o.side_a, o.side_b)); }  // rectangle.rs:44-46

// objects/rectangle.rs:58-101
bool rectangle_intersect(const Object& o, const Ray& ray, double* t_out, SurfaceInteraction* si) {
  const V3 normal = rectangle_normal(o);
  const double denom = dot(normal, ray.direction);
  if (std::fabs(denom) < 1e-9) return false;
  const V3 v = o.position - ray.point;
  const double distance = dot(v, normal) / denom;
  if (distance < 1e-9) return false;
  const V3 p = ray.point + (ray.direction * distance);
  const V3 p0p = p - o.position;
  const double a = dot(p0p, o.side_a) / dot(o.side_a, o.side_a);
  const double b = dot(p0p, o.side_b) / dot(o.side_b, o.side_b);
  if (!(a >= 0.0 && a <= 1.0) || !(b >= 0.0 && b <= 1.0)) return false;
  V3 ss, ts;
  coordinate_system(normal, &ss, &ts);
  *t_out = distance;
  *si = make_si(p, normal, -ray.direction, ss, ts);
  return true;
}

inline double rectangle_area(const Object& o) { return norm(o.side_b) * norm(o.side_b); }  // rectangle.rs:131-133 (sic)

// objects/rectangle.rs:113-129
double rectangle_pdf(const Object& o, V3 point, V3 wi) {
  Ray ray{point + wi * 1e-9, wi};
  double t;
  SurfaceInteraction si;
  if (!rectangle_intersect(o, ray, &t, &si)) return 0.0;
  return norm2(point - si.point) / (std::fabs(dot(si.shading_normal, -wi)) * rectangle_area(o));
}

// objects/plane.rs:45-80
bool plane_intersect(const Object& o, const Ray& ray, double* t_out, SurfaceInteraction* si) {
  const double denom = dot(o.normal, ray.direction);
  if (std::fabs(denom) < 1e-9) return false;
  const V3 v = o.position - ray.point;
  const double distance = dot(v, o.normal) / denom;
  if (distance < 0.0000001) return false;
  const V3 p_hit = ray.point + ray.direction * distance + o.normal * 1e-9;
  *t_out = distance;
  *si = make_si(p_hit, o.normal, -ray.direction, V3(1.0, -0.0, 0.0), V3(0.0, 0.0, -1.0));
  return true;
}

// objects/sphere.rs:36-104
SurfaceInteraction sphere_interaction(const Object& o, V3 point, V3 normal, V3 wo) {
  const V3 object_point = point - o.position;
  double phi = std::atan2(object_point.y, object_point.x);
  if (phi < 0.0) phi += 2.0 * PI;
  const double cos_t = clampd(object_point.z / o.radius, -1.0, 1.0);
  const double theta = std::acos(cos_t);
  const double sin_p = std::sin(phi), cos_p = std::cos(phi), sin_t = std::sin(theta);
  const V3 dp_du(-2.0 * PI * o.radius * sin_t * sin_p, 2.0 * PI * o.radius * sin_t * cos_p, 0.0);
  V3 ss = normalize(dp_du);
  V3 ts = cross(normal, ss);
  if (norm2(ts) > 0.0) {  // false for NaN too, as in Rust
    ts = normalize(ts);
    ss = cross(ts, normal);
  } else {
    coordinate_system(normal, &ss, &ts);
  }
  return make_si(point + normal * 1e-9, normal, wo, ss, ts);
}

// objects/sphere.rs:116-154
bool sphere_intersect(const Object& o, const Ray& ray, double* t_out, SurfaceInteraction* si) {
  const V3 oc = ray.point - o.position;
  const double a = dot(ray.direction, ray.direction);
  const double b = dot(oc, ray.direction);
  const double c = dot(oc, oc) - o.radius * o.radius;
  const double discriminant = b * b - a * c;
  if (discriminant < 0.0) return false;
  double temp = (-b - std::sqrt(b * b - a * c)) / a;
  if (!(temp > 0.0001 && temp < std::numeric_limits<double>::max())) {
    temp = (-b + std::sqrt(b * b - a * c)) / a;
    if (!(temp > 0.0001 && temp < std::numeric_limits<double>::max())) return false;
  }
  const V3 contact = ray.point + ray.direction * temp;
  const V3 normal = normalize(contact - o.position);
  *t_out = temp;
  *si = sphere_interaction(o, contact, normal, -ray.direction);
  return true;
}

bool object_intersect(const Object& o, const Ray& ray, double* t, SurfaceInteraction* si, TriDetail* detail = nullptr) {
  switch (o.kind) {
    case Object::Triangle: return triangle_intersect(o, ray, t, si, detail);
    case Object::Rectangle: return rectangle_intersect(o, ray, t, si);
    case Object::Plane: return plane_intersect(o, ray, t, si);
    case Object::Sphere: return sphere_intersect(o, ray, t, si);
  }
  return false;
}

// Bounded::aabb of each shape: triangle.rs:335-353, rectangle.rs:136-147,
// plane.rs:110-126, sphere.rs:169-180
Aabb object_aabb(const Object& o) {
  Aabb b;
  switch (o.kind) {
    case Object::Triangle:
      for (int k = 0; k < 3; ++k) {
        const double mn = std::min(o.p0[k], std::min(o.p1[k], o.p2[k]));
        const double mx = std::max(o.p0[k], std::max(o.p1[k], o.p2[k]));
        b.min[k] = (float)mn;
        b.max[k] = (float)(mx + 0.001);
      }
      break;
    case Object::Rectangle: {
      const V3 opp = o.position + o.side_a + o.side_b;
      for (int k = 0; k < 3; ++k) {
        b.min[k] = (float)std::min(o.position[k], opp[k]);
        b.max[k] = (float)std::max(o.position[k], opp[k]);
      }
      break;
    }
    case Object::Plane:
      for (int k = 0; k < 3; ++k) {
        b.min[k] = (float)(o.position[k] - 1e14);
        b.max[k] = (float)(o.position[k] + 1e14);
      }
      break;
    case Object::Sphere:
      for (int k = 0; k < 3; ++k) {
        b.min[k] = (float)(o.position[k] - o.radius);
        b.max[k] = (float)(o.position[k] + o.radius);
      }
      break;
  }
  return b;
}

// ------------------------------------------------------------------ stand-in for the bvh crate
// Contract taken from the reference's use of bvh 0.8.0 (src/scene.rs:196,
// src/renderer.rs:217-226): Bvh::build over the shapes' f32 boxes with one
// shape per leaf; traverse_iterator(ray_f32) yields every shape whose box the
// ray intersects.  Build: top-down, 6 centroid buckets on the widest centroid
// axis, split at the cheapest surface-area cost, halves when the centroids
// coincide -- the published algorithm of that crate.  Which split is taken
// changes only the order candidates come out in, never the set.
struct BvhNode {
  Aabb box_l, box_r;
  int child_l, child_r;  // >= 0: node index; < 0: ~object index (leaf)
};

struct Bvh {
  std::vector<BvhNode> nodes;
  int root = 0;  // >= 0 node, < 0 leaf (single object), INT32_MIN = empty
};

inline void aabb_empty(Aabb* b) {
  for (int k = 0; k < 3; ++k) { b->min[k] = INFINITY; b->max[k] = -INFINITY; }
}
inline void aabb_join(Aabb* a, const Aabb& b) {
  for (int k = 0; k < 3; ++k) { a->min[k] = std::min(a->min[k], b.min[k]); a->max[k] = std::max(a->max[k], b.max[k]); }
}
inline float aabb_area(const Aabb& b) {
  const float dx = b.max[0] - b.min[0], dy = b.max[1] - b.min[1], dz = b.max[2] - b.min[2];
  return 2.0f * (dx * dy + dx * dz + dy * dz);
}

struct BvhBuilder {
  const std::vector<Aabb>& boxes;
  std::vector<BvhNode>& nodes;
  std::vector<float> cx, cy, cz;

  BvhBuilder(const std::vector<Aabb>& b, std::vector<BvhNode>& n) : boxes(b), nodes(n) {
    cx.resize(b.size()); cy.resize(b.size()); cz.resize(b.size());
    for (size_t i = 0; i < b.size(); ++i) {
      cx[i] = 0.5f * (b[i].min[0] + b[i].max[0]);
      cy[i] = 0.5f * (b[i].min[1] + b[i].max[1]);
      cz[i] = 0.5f * (b[i].min[2] + b[i].max[2]);
    }
  }
  float centroid(int i, int axis) const { return axis == 0 ? cx[i] : (axis == 1 ? cy[i] : cz[i]); }

  // returns child reference (node index or ~object) and the box of the subtree
  int build(int* idx, int n, Aabb* out_box) {
    if (n == 1) {
      *out_box = boxes[idx[0]];
      return ~idx[0];
    }
    Aabb bounds, cbounds;
    aabb_empty(&bounds);
    aabb_empty(&cbounds);
    for (int i = 0; i < n; ++i) {
      aabb_join(&bounds, boxes[idx[i]]);
      for (int k = 0; k < 3; ++k) {
        const float c = centroid(idx[i], k);
        cbounds.min[k] = std::min(cbounds.min[k], c);
        cbounds.max[k] = std::max(cbounds.max[k], c);
      }
    }
    int axis = 0;
    float extent = cbounds.max[0] - cbounds.min[0];
    for (int k = 1; k < 3; ++k)
      if (cbounds.max[k] - cbounds.min[k] > extent) { extent = cbounds.max[k] - cbounds.min[k]; axis = k; }
    int mid;
    if (!(extent > 1e-5f)) {
      mid = n / 2;
    } else {
      const int NB = 6;
      int count[NB] = {0};
      Aabb bbox[NB];
      for (int b = 0; b < NB; ++b) aabb_empty(&bbox[b]);
      auto bucket_of = [&](int i) {
        const float rel = (centroid(i, axis) - cbounds.min[axis]) / extent;
        int b = (int)(rel * (NB - 0.01f));
        return b < 0 ? 0 : (b >= NB ? NB - 1 : b);
      };
      for (int i = 0; i < n; ++i) {
        const int b = bucket_of(idx[i]);
        ++count[b];
        aabb_join(&bbox[b], boxes[idx[i]]);
      }
      float best = INFINITY;
      int best_split = 0;
      for (int s = 0; s < NB - 1; ++s) {
        Aabb l, r;
        aabb_empty(&l);
        aabb_empty(&r);
        int nl = 0, nr = 0;
        for (int b = 0; b <= s; ++b) if (count[b]) { aabb_join(&l, bbox[b]); nl += count[b]; }
        for (int b = s + 1; b < NB; ++b) if (count[b]) { aabb_join(&r, bbox[b]); nr += count[b]; }
        if (nl == 0 || nr == 0) continue;
        const float cost = (nl * aabb_area(l) + nr * aabb_area(r)) / aabb_area(bounds);
        if (cost < best) { best = cost; best_split = s; }
      }
      int* m = std::partition(idx, idx + n, [&](int i) { return bucket_of(i) <= best_split; });
      mid = (int)(m - idx);
      if (mid == 0 || mid == n) mid = n / 2;
    }
    const int me = (int)nodes.size();
    nodes.emplace_back();
    Aabb bl, br;
    const int cl = build(idx, mid, &bl);
    const int cr = build(idx + mid, n - mid, &br);
    nodes[me].box_l = bl;
    nodes[me].box_r = br;
    nodes[me].child_l = cl;
    nodes[me].child_r = cr;
    *out_box = bl;
    aabb_join(out_box, br);
    return me;
  }
};

// Ray / box test in f32 on the f32 copy of the ray (src/renderer.rs:217-224).
struct RayF32 {
  float o[3], inv[3];
};
inline RayF32 make_ray_f32(const Ray& r) {
  RayF32 q;
  for (int k = 0; k < 3; ++k) {
    q.o[k] = (float)r.point[k];
    q.inv[k] = 1.0f / (float)r.direction[k];
  }
  return q;
}
inline bool ray_hits_aabb(const RayF32& r, const Aabb& b, uint32_t mode) {
  float tmin = -INFINITY, tmax = INFINITY;
  for (int k = 0; k < 3; ++k) {
    const float t0 = (b.min[k] - r.o[k]) * r.inv[k];
    const float t1 = (b.max[k] - r.o[k]) * r.inv[k];
    tmin = std::fmax(tmin, std::fmin(t0, t1));
    tmax = std::fmin(tmax, std::fmax(t0, t1));
  }
  const float lo = std::fmax(tmin, 0.0f);
  return mode == RRT_RECT_AABB_INCLUSIVE ? tmax >= lo : tmax > lo;
}

// ------------------------------------------------------------------ lights/*.rs
struct Light {
  uint32_t kind;
  V3 intensity;
  V3 position;    // point
  V3 direction;   // distant (normalised, distant.rs:81)
  Object rect;    // area: the sampling rectangle (no material, no light)
};

struct LightIrradianceSample {
  V3 irradiance, point, wi;
  double pdf = 0;
};

struct alignas(64) Counters {  // (a cache line each: workers update them per ray)
  uint64_t rays_extend = 0, rays_shadow = 0, path_bounces = 0, nodes_visited = 0, prims_tested = 0, paths = 0;
};

struct Camera {
  double raster_to_camera[16], camera_to_world[16];
  double aperture = 0, focal_distance = 0;
  bool set = false;
};

}  // namespace

struct orc_ctx {
  std::vector<Object> objects;
  std::vector<Light> lights;
  std::vector<rrt_material> materials;
  std::vector<uint8_t> env;
  uint32_t env_w = 0, env_h = 0;
  uint32_t rect_aabb_mode = RRT_RECT_AABB_STRICT;
  std::vector<Aabb> boxes;
  Bvh bvh;
  Camera camera;
  // Film::new (src/film.rs:59-132): reconstruction filter
  uint32_t filter_method = RRT_FILTER_NONE;
  double filter_radius = 0.0;
  std::vector<double> filter_table;  // 16 x 16
};

namespace {

// textures/mip_map.rs:26-47
inline V3 mip_map_lookup(const orc_ctx& s, double px, double py) {
  const uint32_t w = s.env_w, h = s.env_h;
  const double u = px - std::floor(px);
  const double v = clampd(py, 0.0, 1.0);
  const uint32_t x = std::min((uint32_t)(u * (double)w), w - 1);
  const uint32_t y = std::min((uint32_t)(v * (double)h), h - 1);
  const uint8_t* t = &s.env[((size_t)y * w + x) * 3];
  return V3(t[0] / 255.0, t[1] / 255.0, t[2] / 255.0);
}

// lights/infinite_area.rs:96-106 (world_to_light is a translation: vectors unchanged, src/scene.rs:125-129)
inline V3 infinite_environment_emitting(const orc_ctx& s, V3 direction) {
  const V3 w = direction;
  return mip_map_lookup(s, 1.0 - spherical_phi(w) * 1.0 / (2.0 * PI), spherical_theta(w) * FRAC_1_PI);
}

// LightTrait::environment_emitting: lights.rs:45-47 default, distant.rs:62-64, infinite_area.rs:96
inline V3 light_environment_emitting(const orc_ctx& s, const Light& l, const Ray& ray) {
  switch (l.kind) {
    case RRT_LIGHT_DISTANT: return l.intensity;
    case RRT_LIGHT_INFINITE: return infinite_environment_emitting(s, ray.direction);
    default: return V3();
  }
}

// LightTrait::emitting: area.rs:23-25 (zero), infinite_area.rs:31-43; Distant / Point are unimplemented!()
inline V3 light_emitting(const orc_ctx& s, const Light& l, V3 wi) {
  if (l.kind == RRT_LIGHT_INFINITE) return infinite_environment_emitting(s, normalize(wi));
  return V3();
}

inline bool light_is_delta(const Light& l) { return l.kind == RRT_LIGHT_POINT || l.kind == RRT_LIGHT_DISTANT; }

// LightTrait::sample_irradiance
LightIrradianceSample light_sample_irradiance(const orc_ctx& s, const Light& l, const SurfaceInteraction& si, double u0,
                                              double u1) {
  LightIrradianceSample r;
  switch (l.kind) {
    case RRT_LIGHT_POINT: {  // point.rs:26-43
      r.wi = normalize(l.position - si.point);
      r.pdf = 1.0;
      r.irradiance = l.intensity / norm2(l.position - si.point);
      r.point = l.position;
      break;
    }
    case RRT_LIGHT_AREA: {  // area.rs:28-45 with Rectangle::sample_point (rectangle.rs:103-110)
      const V3 lp = l.rect.position + (l.rect.side_a * u0) + (l.rect.side_b * u1);
      const V3 ln = rectangle_normal(l.rect);
      r.wi = normalize(lp - si.point);
      r.pdf = rectangle_pdf(l.rect, si.point, r.wi);
      r.irradiance = dot(ln, -r.wi) > 0.0 ? l.intensity : V3();  // irradiance_at_point, area.rs:80-86
      r.point = lp;
      break;
    }
    case RRT_LIGHT_DISTANT: {  // distant.rs:29-45
      r.wi = l.direction;
      r.pdf = 1.0;
      r.point = si.point + l.direction * (2.0 * 1e20);
      r.irradiance = l.intensity;
      break;
    }
    case RRT_LIGHT_INFINITE: {  // infinite_area.rs:45-82
      const double theta = u1 * PI, phi = u0 * 2.0 * PI;
      const double cos_t = std::cos(theta), sin_t = std::sin(theta), cos_p = std::cos(phi), sin_p = std::sin(phi);
      const V3 wi(sin_t * sin_p, cos_t, sin_t * cos_p);
      const V3 le = infinite_environment_emitting(s, normalize(wi));
      r.pdf = sin_t != 0.0 ? vmax(le) / (2.0 * PI * PI * sin_t) : 0.0;
      r.point = si.point + wi * (2.0 * 1e20);
      r.wi = wi;
      r.irradiance = le;
      break;
    }
  }
  return r;
}

// LightTrait::pdf_incidence: area.rs:53-55, infinite_area.rs:88-90
inline double light_pdf_incidence(const Light& l, V3 point, V3 wi) {
  if (l.kind == RRT_LIGHT_AREA) return rectangle_pdf(l.rect, point, wi);
  return 1.0;  // InfiniteArea; the delta lights never get here (tracer.rs:204)
}

// ------------------------------------------------------------------ renderer.rs scene queries
// traverse_iterator: every object whose box the f32 ray hits
template <typename Visit>
inline void traverse(const orc_ctx& s, const Ray& ray, Counters* c, Visit&& visit) {
  const Bvh& bvh = s.bvh;
  if (bvh.root == INT32_MIN) return;
  const RayF32 r = make_ray_f32(ray);
  if (bvh.root < 0) {
    const int obj = ~bvh.root;
    c->nodes_visited++;
    if ((s.rect_aabb_mode == RRT_RECT_AABB_OFF && s.objects[obj].kind == Object::Rectangle) ||
        ray_hits_aabb(r, s.boxes[obj], s.rect_aabb_mode))
      visit(obj);
    return;
  }
  int stack[256];
  int sp = 0;
  stack[sp++] = bvh.root;
  while (sp > 0) {
    const BvhNode& n = bvh.nodes[stack[--sp]];
    c->nodes_visited += 2;
    for (int side = 0; side < 2; ++side) {
      const Aabb& box = side == 0 ? n.box_l : n.box_r;
      const int child = side == 0 ? n.child_l : n.child_r;
      bool hit = ray_hits_aabb(r, box, s.rect_aabb_mode);
      if (!hit && child < 0 && s.rect_aabb_mode == RRT_RECT_AABB_OFF && s.objects[~child].kind == Object::Rectangle)
        hit = true;
      if (!hit) continue;
      if (child < 0) {
        if (!visit(~child)) return;
      } else if (sp < 256) {
        stack[sp++] = child;
      }
    }
  }
}

// check_intersect_scene (src/renderer.rs:213-249)
bool check_intersect_scene(const orc_ctx& s, const Ray& ray, Counters* c, SurfaceInteraction* out_si, int* out_obj,
                           double* out_t = nullptr) {
  bool found = false;
  double closest_distance = std::numeric_limits<double>::max();
  c->rays_extend++;
  traverse(s, ray, c, [&](int obj) {
    double distance;
    SurfaceInteraction si;
    c->prims_tested++;
    if (object_intersect(s.objects[obj], ray, &distance, &si)) {
      if (!found || distance < closest_distance) {
        found = true;
        closest_distance = distance;
        *out_si = si;
        *out_obj = obj;
      }
    }
    return true;
  });
  if (found && out_t) *out_t = closest_distance;
  return found;
}

// check_intersect_scene_simple (src/renderer.rs:251-275)
bool check_intersect_scene_simple(const orc_ctx& s, const Ray& ray, double max_dist, Counters* c) {
  bool any = false;
  c->rays_shadow++;
  traverse(s, ray, c, [&](int obj) {
    double distance;
    SurfaceInteraction si;
    c->prims_tested++;
    if (object_intersect(s.objects[obj], ray, &distance, &si) && distance < max_dist) {
      any = true;
      return false;
    }
    return true;
  });
  return any;
}

// check_light_visible (src/renderer.rs:277-295)
bool check_light_visible(const orc_ctx& s, const SurfaceInteraction& si, const LightIrradianceSample& ls, Counters* c) {
  const V3 direction = normalize(ls.point - si.point);
  const Ray ray{si.point + (direction * 1e-9), direction};
  const double distance = norm(si.point - ls.point) - 1e-7;
  return !check_intersect_scene_simple(s, ray, distance, c);
}

// ------------------------------------------------------------------ tracer.rs
// estimate_direct (:157-283)
V3 estimate_direct(const orc_ctx& s, const SurfaceInteraction& si, const RngBlock& ra, const RngBlock& rb, int light_index,
                   Counters* c) {
  const Light& light = s.lights[light_index];
  const uint32_t bsdf_flags = ALL & ~SPECULAR;
  V3 direct;
  LightIrradianceSample ls = light_sample_irradiance(s, light, si, ra.u[1], ra.u[2]);
  const double light_pdf = ls.pdf;
  double scattering_pdf = 0.0;
  if (ls.pdf > 1e-6 && !is_zero(ls.irradiance)) {
    V3 f = si.has_bsdf ? si.bsdf.f(si.wo, ls.wi, bsdf_flags) : V3();
    f = f * std::fabs(dot(ls.wi, si.shading_normal));
    scattering_pdf = si.bsdf.pdf(si.wo, ls.wi, bsdf_flags);
    if (!is_zero(f)) {
      if (!check_light_visible(s, si, ls, c)) ls.irradiance = V3();
      if (light_is_delta(light)) {
        direct += cmul(f, ls.irradiance) / light_pdf;
      } else {
        const double weight = power_heuristic(1, light_pdf, 1, scattering_pdf);
        direct += cmul(f, ls.irradiance) * weight / light_pdf;
      }
    }
  }
  if (!light_is_delta(light)) {
    BsdfSampleResult bs;
    if (si.has_bsdf) bs = si.bsdf.sample_f(si.wo, bsdf_flags, rb.u[0], rb.u[1], rb.u[2]);
    const V3 f = bs.f * std::fabs(dot(bs.wi, si.shading_normal));
    scattering_pdf = bs.pdf;
    const bool sampled_specular = (bs.sampled_flags & SPECULAR) != 0;
    if (!is_zero(f) && scattering_pdf > 1e-6) {
      double weight = 1.0;
      if (!sampled_specular) {
        const double lp = light_pdf_incidence(light, si.point, bs.wi);
        if (lp < 1e-6) return direct;
        weight = power_heuristic(1, scattering_pdf, 1, lp);
      }
      const Ray ray{offset_ray_origin(si.point, si.geometry_normal, bs.wi), bs.wi};
      V3 light_irradiance;
      SurfaceInteraction hit_si;
      int hit_obj;
      if (check_intersect_scene(s, ray, c, &hit_si, &hit_obj)) {
        if (s.objects[hit_obj].light == light_index && light.kind == RRT_LIGHT_AREA)
          light_irradiance = light_emitting(s, light, -bs.wi);  // AreaLight::emitting: zero
      } else {
        light_irradiance = light_emitting(s, light, ray.direction);
      }
      if (!is_zero(light_irradiance)) direct += cmul(f, light_irradiance * weight) / bs.pdf;
    }
  }
  return direct;
}

// uniform_sample_light (:141-155)
V3 uniform_sample_light(const orc_ctx& s, const SurfaceInteraction& si, const RngBlock& ra, const RngBlock& rb,
                        Counters* c) {
  const size_t light_count = s.lights.size();
  if (light_count == 0) return V3();  // the reference would index an empty Vec and panic
  const double light_num = std::min(ra.u[0] * (double)light_count, (double)light_count - 1.0);
  const double light_pdf = 1.0 / (double)light_count;
  return estimate_direct(s, si, ra, rb, (int)light_num, c) / light_pdf;
}

// trace (:24-139)
V3 trace(const orc_ctx& s, Ray ray, const rrt_settings& settings, const PathRng& rng, Counters* c) {
  V3 l(0, 0, 0), contribution(1, 1, 1);
  bool specular_bounce = false;
  for (uint32_t bounce = 0; bounce < settings.depth_limit; ++bounce) {
    SurfaceInteraction si;
    int obj = -1;
    const bool hit = check_intersect_scene(s, ray, c, &si, &obj);
    if (bounce == 0 || specular_bounce) {
      if (hit) {
        const int li = s.objects[obj].light;
        if (li >= 0) l += cmul(contribution, light_emitting(s, s.lights[li], -ray.direction));
      } else {
        for (const Light& light : s.lights) l += cmul(contribution, light_environment_emitting(s, light, ray));
      }
    }
    if (!hit) break;
    c->path_bounces++;
    if (s.objects[obj].material >= 0) compute_scattering_functions(s.materials[s.objects[obj].material], &si);
    ensure_bsdf(&si);  // an object without materials would make the reference unwrap() None; none exists in Scene.objects
    const RngBlock ra = rng.block(1 + 3 * bounce), rb = rng.block(2 + 3 * bounce), rc = rng.block(3 + 3 * bounce);
    l += cmul(contribution, uniform_sample_light(s, si, ra, rb, c));
    const BsdfSampleResult bs = si.bsdf.sample_f(si.wo, ALL, rc.u[0], rc.u[1], rc.u[2]);
    if (bs.pdf == 0.0 || is_zero(bs.f)) break;
    contribution = cmul(contribution, (bs.f * std::fabs(dot(bs.wi, si.shading_normal))) / bs.pdf);
    specular_bounce = (bs.sampled_flags & SPECULAR) != 0;
    ray = Ray{offset_ray_origin(si.point, si.geometry_normal, bs.wi), bs.wi};
    if (settings.clamp > 0.0) {
      l.x = std::min(l.x, settings.clamp);
      l.y = std::min(l.y, settings.clamp);
      l.z = std::min(l.z, settings.clamp);
    }
    if (bounce > 3) {
      const double q = std::max(1.0 - vmax(contribution), 0.05);
      if (ra.u[3] < q) break;
      contribution = contribution / (1.0 - q);
    }
  }
  return l;
}

// ------------------------------------------------------------------ camera.rs
inline V3 transform_point(const double* m, V3 p) {
  const double x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
  const double y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
  const double z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
  const double w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
  return w != 0.0 ? V3(x / w, y / w, z / w) : V3(x, y, z);
}
inline V3 transform_vector(const double* m, V3 v) {
  return V3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
            m[8] * v.x + m[9] * v.y + m[10] * v.z);
}

// Camera::generate_ray (:103-125); the lens point comes from (u2, u3)
Ray generate_ray(const Camera& cam, double film_x, double film_y, double u2, double u3) {
  V3 origin(0, 0, 0);
  V3 direction = transform_point(cam.raster_to_camera, V3(film_x, film_y, 0.0));
  if (cam.aperture > 0.0) {
    double dx, dy;
    concentric_sample_disk(u2, u3, &dx, &dy);
    const double lx = cam.aperture * dx, ly = cam.aperture * dy;
    const double ft = cam.focal_distance / direction.z;
    const V3 p_focus = ft * direction;
    origin = V3(lx, ly, 0.0);
    direction = normalize(p_focus - origin);
  }
  Ray r;
  r.point = transform_point(cam.camera_to_world, origin);
  r.direction = normalize(transform_vector(cam.camera_to_world, direction));
  return r;
}

// One camera sample: get_camera_sample + generate_ray + trace (src/renderer.rs:200-203)
V3 sample_pixel(const orc_ctx& s, const rrt_settings& settings, uint32_t x, uint32_t y, uint32_t sample, Counters* c,
                double* film_x, double* film_y) {
  PathRng rng{settings.seed, y * settings.width + x, sample};
  const RngBlock cam = rng.block(0);
  *film_x = (double)x + cam.u[0];
  *film_y = (double)y + cam.u[1];
  const Ray ray = generate_ray(s.camera, *film_x, *film_y, cam.u[2], cam.u[3]);
  c->paths++;
  return trace(s, ray, settings, rng, c);
}

// film.rs:360-394
inline double gamma_correct_srgb(double v) {
  if (v <= 0.0) return 0.0;
  if (v < 0.0031308) return v * 12.92;
  if (v < 1.0) return std::pow(v, 1.0 / 2.4) * 1.055 - 0.055;
  return 1.0;
}

}  // namespace

// ====================================================================== C interface
extern "C" {

orc_ctx* orc_create(const rrt_scene_desc* d) {
  if (!d) return nullptr;
  orc_ctx* s = new orc_ctx();
  s->materials.assign(d->materials, d->materials + d->n_materials);
  s->rect_aabb_mode = d->rect_aabb_mode;
  // Scene.objects order: mesh triangles, light rectangles (src/scene.rs:64-106), then planes, spheres
  for (uint32_t i = 0; i < d->n_tris; ++i) {
    Object o{};
    o.kind = Object::Triangle;
    const float* p = d->tri_pos + 9 * (size_t)i;
    const float* n = d->tri_nrm + 9 * (size_t)i;
    o.p0 = V3(p[0], p[1], p[2]); o.p1 = V3(p[3], p[4], p[5]); o.p2 = V3(p[6], p[7], p[8]);
    o.n0 = V3(n[0], n[1], n[2]); o.n1 = V3(n[3], n[4], n[5]); o.n2 = V3(n[6], n[7], n[8]);
    o.material = (int)d->tri_mat[i];
    s->objects.push_back(o);
  }
  for (uint32_t i = 0; i < d->n_rects; ++i) {
    Object o{};
    o.kind = Object::Rectangle;
    o.position = v3(d->rects[i].pos);
    o.side_a = v3(d->rects[i].side_a);
    o.side_b = v3(d->rects[i].side_b);
    o.material = (int)d->rects[i].material;
    o.light = d->rects[i].light;
    s->objects.push_back(o);
  }
  for (uint32_t i = 0; i < d->n_planes; ++i) {
    Object o{};
    o.kind = Object::Plane;
    o.position = v3(d->planes[i].pos);
    o.normal = v3(d->planes[i].normal);
    o.material = (int)d->planes[i].material;
    s->objects.push_back(o);
  }
  for (uint32_t i = 0; i < d->n_spheres; ++i) {
    Object o{};
    o.kind = Object::Sphere;
    o.position = v3(d->spheres[i].center);
    o.radius = d->spheres[i].radius;
    o.material = (int)d->spheres[i].material;
    s->objects.push_back(o);
  }
  for (uint32_t i = 0; i < d->n_lights; ++i) {
    const rrt_light& in = d->lights[i];
    Light l{};
    l.kind = in.kind;
    l.intensity = v3(in.intensity);
    if (in.kind == RRT_LIGHT_POINT) l.position = v3(in.v0);
    if (in.kind == RRT_LIGHT_DISTANT) l.direction = normalize(v3(in.v0));
    if (in.kind == RRT_LIGHT_AREA) {
      l.rect.kind = Object::Rectangle;
      l.rect.position = v3(in.v0);
      l.rect.side_a = v3(in.v1);
      l.rect.side_b = v3(in.v2);
    }
    s->lights.push_back(l);
  }
  if (d->env_rgb8 && d->env_w && d->env_h) {
    s->env.assign(d->env_rgb8, d->env_rgb8 + (size_t)d->env_w * d->env_h * 3);
    s->env_w = d->env_w;
    s->env_h = d->env_h;
  }
  s->boxes.resize(s->objects.size());
  for (size_t i = 0; i < s->objects.size(); ++i) s->boxes[i] = object_aabb(s->objects[i]);
  if (s->objects.empty()) {
    s->bvh.root = INT32_MIN;
  } else {
    std::vector<int> idx(s->objects.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
    BvhBuilder builder(s->boxes, s->bvh.nodes);
    Aabb root_box;
    s->bvh.root = builder.build(idx.data(), (int)idx.size(), &root_box);
  }
  return s;
}

void orc_destroy(orc_ctx* ctx) { delete ctx; }

int orc_set_camera(orc_ctx* ctx, const double raster_to_camera[16], const double camera_to_world[16], double aperture,
                   double focal_distance) {
  if (!ctx || !raster_to_camera || !camera_to_world) return RRT_E_INVALID;
  std::memcpy(ctx->camera.raster_to_camera, raster_to_camera, sizeof(double) * 16);
  std::memcpy(ctx->camera.camera_to_world, camera_to_world, sizeof(double) * 16);
  ctx->camera.aperture = aperture;
  ctx->camera.focal_distance = focal_distance;
  ctx->camera.set = true;
  return RRT_OK;
}

namespace {
// src/film.rs:326-333
double evaluate_gaussian(double px, double py, double radius, double alpha) {
  const double expv = std::exp(-alpha * radius * radius);
  const double x = std::max(std::exp(-alpha * px * px) - expv, 0.0);
  const double y = std::max(std::exp(-alpha * py * py) - expv, 0.0);
  return x * y;
}
// src/film.rs:340-358
double evaluate_mitchell_1d(double input) {
  const double b = 0.33, c = 0.33;
  const double x = std::fabs(2.0 * input);
  if (x > 1.0)
    return ((-b - 6.0 * c) * x * x * x + (6.0 * b + 30.0 * c) * x * x + (-12.0 * b - 48.0 * c) * x + (8.0 * b + 24.0 * c)) *
           (1.0 / 6.0);
  return ((12.0 - 9.0 * b - 6.0 * c) * x * x * x + (-18.0 + 12.0 * b + 6.0 * c) * x * x + (6.0 - 2.0 * b)) * (1.0 / 6.0);
}
// src/film.rs:335-338
double evaluate_mitchell(double px, double py, double radius) {
  const double inv = 1.0 / radius;
  return evaluate_mitchell_1d(px * inv) * evaluate_mitchell_1d(py * inv);
}
}  // namespace

// Film::new's filter table (src/film.rs:87-112); GAUSSIAN_ALPHA = 1.5 (src/film.rs:17)
int orc_set_filter(orc_ctx* ctx, uint32_t method, double radius) {
  if (!ctx || method > RRT_FILTER_MITCHELL) return RRT_E_INVALID;
  if (method != RRT_FILTER_NONE && !(radius > 0.0)) return RRT_E_INVALID;
  ctx->filter_method = method;
  ctx->filter_table.clear();
  if (method == RRT_FILTER_NONE) {
    ctx->filter_radius = 0.0;
    return RRT_OK;
  }
  ctx->filter_radius = radius;
  const int n = 16;
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) {
      const double xp = (x + 0.5) * radius / n, yp = (y + 0.5) * radius / n;
      ctx->filter_table.push_back(method == RRT_FILTER_GAUSSIAN ? evaluate_gaussian(xp, yp, radius, 1.5)
                                                                : evaluate_mitchell(xp, yp, radius));
    }
  return RRT_OK;
}

int orc_render(orc_ctx* ctx, const rrt_settings* st, uint32_t sample_begin, uint32_t sample_end, uint32_t bucket_w,
               uint32_t bucket_h, uint32_t threads, double* film, orc_stats* stats) {
  if (!ctx || !st || !film || !ctx->camera.set || sample_end < sample_begin || !bucket_w || !bucket_h)
    return RRT_E_INVALID;
  if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
  // Film::init_buckets (src/film.rs:257-324): x-outer, y-inner, starting at the crop origin
  struct Bucket { uint32_t x0, y0, x1, y1; };
  std::vector<Bucket> buckets;
  const uint32_t rw = st->crop_x1 - st->crop_x0, rh = st->crop_y1 - st->crop_y0;
  for (uint32_t bx = 0; bx < (rw + bucket_w - 1) / bucket_w; ++bx)
    for (uint32_t by = 0; by < (rh + bucket_h - 1) / bucket_h; ++by) {
      Bucket b;
      b.x0 = st->crop_x0 + bx * bucket_w;
      b.y0 = st->crop_y0 + by * bucket_h;
      b.x1 = std::min(std::min(b.x0 + bucket_w, st->width), st->crop_x1);
      b.y1 = std::min(std::min(b.y0 + bucket_h, st->height), st->crop_y1);
      buckets.push_back(b);
    }
  std::mutex film_mutex;  // Arc<RwLock<Film>>
  size_t current_bucket = 0;
  std::vector<Counters> counters(threads);
  std::vector<uint32_t> buckets_taken(threads, 0u);
  const auto t0 = std::chrono::steady_clock::now();
  auto worker = [&](uint32_t tid) {
    struct SampleResult { double radiance[3]; double p_film[2]; double normal[3]; double albedo[3]; };  // 88 B
    struct BPixel { double sum_weight; double sum_radiance[3]; };
    Counters& c = counters[tid];
    for (;;) {
      Bucket b;
      {
        std::lock_guard<std::mutex> lock(film_mutex);  // film.write().get_bucket()
        if (current_bucket >= buckets.size()) break;
        b = buckets[current_bucket++];
        buckets_taken[tid]++;
      }
      // render_work (src/renderer.rs:184-211): every sample of the bucket is kept, then splatted
      std::vector<SampleResult> samples;
      samples.reserve((size_t)(b.x1 - b.x0) * (b.y1 - b.y0) * (sample_end - sample_begin));
      for (uint32_t y = b.y0; y < b.y1; ++y)
        for (uint32_t x = b.x0; x < b.x1; ++x)
          for (uint32_t sidx = sample_begin; sidx < sample_end; ++sidx) {
            SampleResult r{};
            const V3 l = sample_pixel(*ctx, *st, x, y, sidx, &c, &r.p_film[0], &r.p_film[1]);
            r.radiance[0] = l.x; r.radiance[1] = l.y; r.radiance[2] = l.z;
            samples.push_back(r);
          }
      // the bucket's pixel raster reaches a filter radius beyond its samples (src/film.rs:289-300;
      // a negative start saturates to 0 in the cast to u32)
      const double fr = ctx->filter_radius;
      const uint32_t px0 = (uint32_t)std::max(0.0, std::floor((double)b.x0 - 0.5 - fr));
      const uint32_t py0 = (uint32_t)std::max(0.0, std::floor((double)b.y0 - 0.5 - fr));
      const uint32_t px1 = std::min((uint32_t)std::ceil((double)b.x1 + 0.5 + fr), st->width);
      const uint32_t py1 = std::min((uint32_t)std::ceil((double)b.y1 + 0.5 + fr), st->height);
      const uint32_t bw = px1 - px0, bh = py1 - py0;
      std::vector<BPixel> px((size_t)bw * bh, BPixel{0, {0, 0, 0}});
      // write_bucket_pixels (src/film.rs:150-219)
      for (const SampleResult& r : samples) {
        if (ctx->filter_method == RRT_FILTER_NONE) {
          const uint32_t ix = (uint32_t)r.p_film[0] - px0, iy = (uint32_t)r.p_film[1] - py0;
          if (ix >= bw || iy >= bh) continue;
          BPixel& p = px[(size_t)iy * bw + ix];
          p.sum_weight += 1.0;
          for (int k = 0; k < 3; ++k) p.sum_radiance[k] += r.radiance[k];
          continue;
        }
        const int x_min = (int)std::ceil(r.p_film[0] - fr), y_min = (int)std::ceil(r.p_film[1] - fr);
        const int x_max = (int)std::floor(r.p_film[0] + fr), y_max = (int)std::floor(r.p_film[1] + fr);
        const double n = 16.0;
        for (int x = x_min; x <= x_max; ++x)
          for (int y = y_min; y <= y_max; ++y) {
            if (x < 0 || y < 0 || x >= (int)st->width || y >= (int)st->height) continue;
            const size_t fx = (size_t)std::min(std::floor(std::fabs(((double)x - r.p_film[0]) * (1.0 / fr) * n)), n - 1.0);
            const size_t fy = (size_t)std::min(std::floor(std::fabs(((double)y - r.p_film[1]) * (1.0 / fr) * n)), n - 1.0);
            const double w = ctx->filter_table[fy * 16 + fx];
            const uint32_t ix = (uint32_t)x - px0, iy = (uint32_t)y - py0;
            if (ix >= bw || iy >= bh) continue;
            BPixel& p = px[(size_t)iy * bw + ix];
            p.sum_weight += w;
            for (int k = 0; k < 3; ++k) p.sum_radiance[k] += r.radiance[k] * w;
          }
      }
      // merge_bucket_pixels_to_image_buffer (src/film.rs:221-231)
      std::lock_guard<std::mutex> lock(film_mutex);
      for (uint32_t iy = 0; iy < bh; ++iy)
        for (uint32_t ix = 0; ix < bw; ++ix) {
          const BPixel& p = px[(size_t)iy * bw + ix];
          double* out = film + 4 * ((size_t)(py0 + iy) * st->width + (px0 + ix));
          out[0] += p.sum_radiance[0];
          out[1] += p.sum_radiance[1];
          out[2] += p.sum_radiance[2];
          out[3] += p.sum_weight;
        }
    }
  };
  std::vector<std::thread> pool;
  for (uint32_t t = 0; t < threads; ++t) pool.emplace_back(worker, t);
  for (auto& t : pool) t.join();
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    for (const Counters& c : counters) {
      stats->paths += c.paths;
      stats->rays_extend += c.rays_extend;
      stats->rays_shadow += c.rays_shadow;
      stats->path_bounces += c.path_bounces;
      stats->nodes_visited += c.nodes_visited;
      stats->prims_tested += c.prims_tested;
    }
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stats->threads = threads;
    for (uint32_t n : buckets_taken) stats->threads_busy += n > 0 ? 1u : 0u;
    stats->buckets = (uint32_t)buckets.size();
  }
  return RRT_OK;
}

int orc_trace_one(orc_ctx* ctx, const rrt_settings* st, uint32_t px, uint32_t py, uint32_t sample, double out[3]) {
  if (!ctx || !st || !out || !ctx->camera.set) return RRT_E_INVALID;
  Counters c;
  double fx, fy;
  const V3 l = sample_pixel(*ctx, *st, px, py, sample, &c, &fx, &fy);
  out[0] = l.x; out[1] = l.y; out[2] = l.z;
  return RRT_OK;
}

// What the device's first-hit buffers hold for one camera sample: the shading normal of the surface
// the camera ray hits and its albedo (Kd, or Ks for a plastic without a diffuse lobe).  The reference
// has the two fields but leaves them zero (src/tracer.rs:36-37, 65-68).
int orc_first_hit(orc_ctx* ctx, const rrt_settings* st, uint32_t px, uint32_t py, uint32_t sample, double normal[3],
                  double albedo[3]) {
  if (!ctx || !st || !normal || !albedo || !ctx->camera.set) return RRT_E_INVALID;
  PathRng rng{st->seed, py * st->width + px, sample};
  const RngBlock cam = rng.block(0);
  const Ray ray = generate_ray(ctx->camera, (double)px + cam.u[0], (double)py + cam.u[1], cam.u[2], cam.u[3]);
  Counters c;
  SurfaceInteraction si;
  int obj = -1;
  double dist = 0;
  for (int k = 0; k < 3; ++k) normal[k] = albedo[k] = 0.0;
  if (!check_intersect_scene(*ctx, ray, &c, &si, &obj, &dist)) return 0;
  normal[0] = si.shading_normal.x; normal[1] = si.shading_normal.y; normal[2] = si.shading_normal.z;
  const rrt_material& m = ctx->materials[ctx->objects[obj].material];
  const bool kd_zero = m.kd[0] == 0.0 && m.kd[1] == 0.0 && m.kd[2] == 0.0;
  for (int k = 0; k < 3; ++k) albedo[k] = (m.kind == RRT_MAT_PLASTIC && kd_zero) ? m.ks[k] : m.kd[k];
  return 1;
}

int orc_intersect(orc_ctx* ctx, const double* od, uint32_t n, int32_t* prim_id, double* t) {
  if (!ctx || !od || !prim_id || !t) return RRT_E_INVALID;
  Counters c;
  for (uint32_t i = 0; i < n; ++i) {
    const Ray ray{v3(od + 6 * (size_t)i), v3(od + 6 * (size_t)i + 3)};
    SurfaceInteraction si;
    int obj = -1;
    double dist = 0;
    if (check_intersect_scene(*ctx, ray, &c, &si, &obj, &dist)) { prim_id[i] = obj; t[i] = dist; }
    else { prim_id[i] = -1; t[i] = 0.0; }
  }
  return RRT_OK;
}

int orc_intersect_margin(orc_ctx* ctx, const double* od, uint32_t n, int32_t* prim_id, double* t, double* margin) {
  if (!ctx || !od || !prim_id || !t || !margin) return RRT_E_INVALID;
  Counters c;
  struct Cand { int obj; double t; bool hit; double inside; };
  std::vector<Cand> cands;
  for (uint32_t i = 0; i < n; ++i) {
    const Ray ray{v3(od + 6 * (size_t)i), v3(od + 6 * (size_t)i + 3)};
    cands.clear();
    traverse(*ctx, ray, &c, [&](int obj) {
      double dist = 0;
      SurfaceInteraction si;
      TriDetail det;
      const bool hit = object_intersect(ctx->objects[obj], ray, &dist, &si, &det);
      Cand cd{obj, hit ? dist : det.t, hit, 1.0};
      if (ctx->objects[obj].kind == Object::Triangle) {
        if (!hit && !det.plane_hit) return true;
        cd.inside = std::min(det.b0, std::min(det.b1, det.b2));
        if (hit) {
          const V3 ng = normalize(cross(ctx->objects[obj].p2 - ctx->objects[obj].p0, ctx->objects[obj].p1 - ctx->objects[obj].p0));
          cd.inside = std::min(cd.inside, std::fabs(dot(ray.direction, ng)));
        }
      } else if (!hit) {
        return true;
      }
      cands.push_back(cd);
      return true;
    });
    int best = -1;
    for (size_t k = 0; k < cands.size(); ++k)
      if (cands[k].hit && (best < 0 || cands[k].t < cands[best].t)) best = (int)k;
    double m = 1.0;
    if (best >= 0) {
      prim_id[i] = cands[best].obj;
      t[i] = cands[best].t;
      m = std::min(m, cands[best].inside);
      for (size_t k = 0; k < cands.size(); ++k) {
        if ((int)k == best) continue;
        if (cands[k].hit) m = std::min(m, (cands[k].t - cands[best].t) / cands[best].t);
        else if (cands[k].t < cands[best].t * (1.0 + 1e-4)) m = std::min(m, -cands[k].inside);  // near miss in front
      }
    } else {
      prim_id[i] = -1;
      t[i] = 0.0;
      for (const Cand& cd : cands) m = std::min(m, -cd.inside);  // all near misses
    }
    margin[i] = m;
  }
  return RRT_OK;
}

int orc_occluded(orc_ctx* ctx, const double* odt, uint32_t n, uint8_t* hit) {
  if (!ctx || !odt || !hit) return RRT_E_INVALID;
  Counters c;
  for (uint32_t i = 0; i < n; ++i) {
    const Ray ray{v3(odt + 7 * (size_t)i), v3(odt + 7 * (size_t)i + 3)};
    hit[i] = check_intersect_scene_simple(*ctx, ray, odt[7 * (size_t)i + 6], &c) ? 1 : 0;
  }
  return RRT_OK;
}

void orc_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c[4] = {counter[0], counter[1], counter[2], counter[3]};
  philox4x32_10(c, key[0], key[1]);
  for (int i = 0; i < 4; ++i) out[i] = c[i];
}

double orc_rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t block, uint32_t index) {
  return rng_block(seed, pixel, sample, block).u[index & 3];
}

int orc_triangle_intersect(const double p[9], const double n[9], const double org[3], const double dir[3], double* t,
                           double point[3], double shading_normal[3], double geometry_normal[3]) {
  Object o{};
  o.kind = Object::Triangle;
  o.p0 = v3(p); o.p1 = v3(p + 3); o.p2 = v3(p + 6);
  o.n0 = v3(n); o.n1 = v3(n + 3); o.n2 = v3(n + 6);
  SurfaceInteraction si;
  double dist = 0;
  if (!triangle_intersect(o, Ray{v3(org), v3(dir)}, &dist, &si, nullptr)) return 0;
  if (t) *t = dist;
  if (point) { point[0] = si.point.x; point[1] = si.point.y; point[2] = si.point.z; }
  if (shading_normal) { shading_normal[0] = si.shading_normal.x; shading_normal[1] = si.shading_normal.y; shading_normal[2] = si.shading_normal.z; }
  if (geometry_normal) { geometry_normal[0] = si.geometry_normal.x; geometry_normal[1] = si.geometry_normal.y; geometry_normal[2] = si.geometry_normal.z; }
  return 1;
}

double orc_fresnel_dielectric(double eta_i, double eta_t, double c) { return fresnel_dielectric(eta_i, eta_t, c); }
double orc_get_fresnel_ratio(const double normal[3], const double incidence[3], double ior) {
  return get_fresnel_ratio(v3(normal), v3(incidence), ior);
}
int orc_max_dimension(const double v[3]) { return max_dimension_vec_3(v3(v)); }
void orc_permute(const double v[3], int x, int y, int z, double out[3]) {
  const V3 r = permute(v3(v), x, y, z);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
double orc_spherical_phi(const double v[3]) { return spherical_phi(v3(v)); }
double orc_spherical_theta(const double v[3]) { return spherical_theta(v3(v)); }
void orc_tr_sample_11(double c, double u1, double u2, double out[2]) { TrowbridgeReitz::sample_11(c, u1, u2, &out[0], &out[1]); }
void orc_tr_sample(const double wi[3], double ax, double ay, double u1, double u2, double out[3]) {
  const V3 r = TrowbridgeReitz::sample(v3(wi), ax, ay, u1, u2);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_tr_sample_wh(const double wo[3], double ax, double ay, int visible, double u1, double u2, double out[3]) {
  TrowbridgeReitz d;
  d.alpha_x = ax; d.alpha_y = ay; d.sample_visible_area = visible != 0;
  const V3 r = d.sample_wh(v3(wo), u1, u2);
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
double orc_tr_pdf(const double wo[3], const double wh[3], double ax, double ay, int visible) {
  TrowbridgeReitz d;
  d.alpha_x = ax; d.alpha_y = ay; d.sample_visible_area = visible != 0;
  return d.pdf(v3(wo), v3(wh));
}
void orc_microfacet_f(const double r[3], double ax, double ay, double eta_i, double eta_t, const double wo[3],
                      const double wi[3], double out[3]) {
  Bxdf b;
  b.kind = Bxdf::MicrofacetReflection;
  b.color = v3(r);
  b.distribution.alpha_x = ax; b.distribution.alpha_y = ay; b.distribution.sample_visible_area = true;
  b.fresnel.dielectric = true; b.fresnel.eta_i = eta_i; b.fresnel.eta_t = eta_t;
  const V3 f = b.f(v3(wo), v3(wi));
  out[0] = f.x; out[1] = f.y; out[2] = f.z;
}
void orc_specular_reflection_sample_f(const double r[3], int fresnel_kind, double eta_i, double eta_t, const double wo[3],
                                      double wi[3], double* pdf, double f[3]) {
  Bxdf b;
  b.kind = Bxdf::SpecularReflection;
  b.color = v3(r);
  b.fresnel.dielectric = fresnel_kind == 1; b.fresnel.eta_i = eta_i; b.fresnel.eta_t = eta_t;
  V3 w, ff;
  double p;
  b.sample_f(0.5, 0.5, v3(wo), &w, &p, &ff);
  wi[0] = w.x; wi[1] = w.y; wi[2] = w.z;
  *pdf = p;
  f[0] = ff.x; f[1] = ff.y; f[2] = ff.z;
}
void orc_specular_reflection_f_pdf(const double r[3], const double wo[3], const double wi[3], double f[3], double* pdf) {
  Bxdf b;
  b.kind = Bxdf::SpecularReflection;
  b.color = v3(r);
  const V3 ff = b.f(v3(wo), v3(wi));
  f[0] = ff.x; f[1] = ff.y; f[2] = ff.z;
  *pdf = b.pdf(v3(wo), v3(wi));
}
void orc_tonemap_pixel(const double radiance[3], uint8_t rgb[3]) {
  const double x = radiance[0], y = radiance[1], z = radiance[2];
  const double r = 3.240479 * x - 1.537150 * y - 0.498535 * z;
  const double g = -0.969256 * x + 1.875991 * y + 0.041556 * z;
  const double b = 0.055648 * x - 0.204043 * y + 1.057311 * z;
  rgb[0] = (uint8_t)(gamma_correct_srgb(r) * 255.0);
  rgb[1] = (uint8_t)(gamma_correct_srgb(g) * 255.0);
  rgb[2] = (uint8_t)(gamma_correct_srgb(b) * 255.0);
}

}  // extern "C"
