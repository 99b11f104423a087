// Shading on the device: surface interaction, the lobes of a material, light
// sampling.  Device forms of src/surface_interaction.rs, src/materials/*.rs,
// src/bsdf.rs + src/bsdf/*.rs and src/lights/*.rs, in f32.
//
// The reference keeps a generic array of up to five BxDFs per hit
// (src/bsdf.rs:23-33) filled by the object's materials.  Every object the
// loader produces has exactly one material (src/scene.rs:319-323, 91-102), and
// a material adds at most one transmission, one diffuse and one specular /
// glossy lobe, in that order.  DeviceMaterial stores that digest, and f / pdf /
// sample_f below evaluate the (at most three) lobes in one pass.
#pragma once
#include "rrt_device.h"
#include "rrt_math.h"

namespace rrt {

struct Surface {
  V3 p;            // hit point (objects/triangle.rs:231-238: barycentric point, lifted when convex)
  V3 ng;           // geometry normal, flipped to the shading-normal side (surface_interaction.rs:56-57)
  V3 ns, ss, ts;   // shading frame
  V3 wo;
  int material;
  int light;       // index of the area light this object belongs to, or -1
  float lift;      // triangles: signed height of p over the triangle's plane along ng; 0 when p lies in the plane
};

// ---------------------------------------------------------------- surface interaction
// Triangle::test_intersect after the hit test (objects/triangle.rs:183-256).
RRT_DEV Surface surface_triangle(const TriShade* __restrict__ shade, int prim, float u, float v, V3 dir) {
  const float4* ts4 = reinterpret_cast<const float4*>(shade + prim);
  const float4 a = ts4[0], b = ts4[1], c = ts4[2];
  const float4 na = ts4[3], nb = ts4[4], nc = ts4[5];
  const V3 p0 = mk(a), e1 = mk(b), e2 = mk(c);
  const V3 n0 = mk(na), n1 = mk(nb), n2 = mk(nc);
  const float b0 = 1.0f - u - v, b1 = u, b2 = v;
  Surface s;
  s.material = __float_as_int(a.w);
  s.light = -1;
  s.wo = -dir;
  s.ns = normalize(b0 * n0 + b1 * n1 + b2 * n2);
  // dpdu = p1 - p0 for the fixed uv (0,0), (1,0), (1,1)
  V3 ss = normalize(e1);
  V3 ts = cross(s.ns, ss);
  if (norm2(ts) > 0.0f) {
    ts = normalize(ts);
    ss = cross(ts, s.ns);
  } else {
    coordinate_system(s.ns, &ss, &ts);
  }
  s.ss = ss;
  s.ts = ts;
  const V3 ng_raw = normalize(cross(e2, e1));  // (p2 - p0) x (p1 - p0)
  const V3 p_hit = p0 + b1 * e1 + b2 * e2;
  // compute_shading_position (objects/triangle.rs:307-333)
  const V3 p1 = p0 + e1, p2 = p0 + e2;
  const V3 q0 = p_hit - dot(p_hit - p0, n0) * n0;
  const V3 q1 = p_hit - dot(p_hit - p1, n1) * n1;
  const V3 q2 = p_hit - dot(p_hit - p2, n2) * n2;
  const V3 sp = b0 * q0 + b1 * q1 + b2 * q2;
  const float height = dot(sp - p_hit, ng_raw);
  const bool flipped = dot(ng_raw, s.ns) < 0.0f;
  s.p = height > 0.0f ? sp : p_hit;
  s.ng = flipped ? -ng_raw : ng_raw;
  s.lift = height > 0.0f ? (flipped ? -height : height) : 0.0f;
  return s;
}

// The reference's visibility ray (src/renderer.rs:277-295) starts at the lifted point.  When the
// lift took the point to one side of its own triangle and the ray heads for the other, the ray
// crosses that triangle -- at a distance of the order of the lift, which f64 resolves and the
// reference therefore counts as an occluder (self-shadowing of a smoothed mesh).  The device's
// visibility ray leaves its own triangle out of the traversal (in f32 a ray that starts in the
// plane would hit it at random), so the crossing is decided here, from the lift itself.
// A lift below 1e-5 of the triangle's size is taken for zero: on a flat-shaded face the exact lift
// is 0, f64 computes ~1e-17 (less than the ray's 1e-9 head start, so no hit), f32 computes ~1e-7.
RRT_DEV bool own_triangle_occludes(const TriShade* __restrict__ shade, int prim, const Surface& s, V3 wi) {
  const float c = dot(wi, s.ng);
  if (s.lift == 0.0f || c * s.lift >= 0.0f) return false;
  const float4* ti = reinterpret_cast<const float4*>(shade + prim);
  const V3 p0 = mk(ti[0]), e1 = mk(ti[1]), e2 = mk(ti[2]);
  if (s.lift * s.lift <= 1e-10f * fmaxf(norm2(e1), norm2(e2))) return false;
  const V3 x = s.p + wi * (-s.lift / c);
  const V3 w = x - p0, n = cross(e1, e2);
  const float inv = 1.0f / dot(n, n);
  const float u = dot(cross(w, e2), n) * inv, v = dot(cross(e1, w), n) * inv;
  return u >= 0.0f && v >= 0.0f && u + v <= 1.0f;
}

// Rectangle / Plane / Sphere (objects/rectangle.rs:85-100, plane.rs:59-79, sphere.rs:36-104).
RRT_DEV Surface surface_other(const SceneTables* __restrict__ sc, int prim, float t, V3 o, V3 dir) {
  Surface s;
  s.wo = -dir;
  s.light = -1;
  s.lift = 0.0f;
  int k = prim - sc->n_tris;
  if (k < sc->n_rects) {
    const DeviceRect& q = sc->rects[k];
    s.p = o + dir * t;
    s.ns = mk(q.normal);
    s.ng = s.ns;
    s.ss = mk(q.ss);
    s.ts = mk(q.ts);
    s.material = q.material;
    s.light = q.light;
    return s;
  }
  k -= sc->n_rects;
  if (k < sc->n_planes) {
    const DevicePlane& pl = sc->planes[k];
    s.p = o + dir * t + mk(pl.normal) * 1e-9f;
    s.ss = mk(1.0f, -0.0f, 0.0f);
    s.ts = mk(0.0f, 0.0f, -1.0f);
    s.ns = mk(0.0f, 1.0f, 0.0f);
    s.ng = mk(pl.ng);
    s.material = pl.material;
    return s;
  }
  k -= sc->n_planes;
  const DeviceSphere& sp = sc->spheres[k];
  const V3 contact = o + dir * t;
  const V3 n = normalize(contact - mk(sp.center));
  const V3 op = contact - mk(sp.center);
  float phi = atan2f(op.y, op.x);
  if (phi < 0.0f) phi += 2.0f * kPi;
  const float cos_t = clampf(op.z / sp.radius, -1.0f, 1.0f);
  const float sin_t = sqrtf(fmaxf(1.0f - cos_t * cos_t, 0.0f));  // sin(acos(cos_t))
  float sin_p, cos_p;
  sincos_fast(phi, &sin_p, &cos_p);
  const V3 dp_du = mk(-2.0f * kPi * sp.radius * sin_t * sin_p, 2.0f * kPi * sp.radius * sin_t * cos_p, 0.0f);
  V3 ss = normalize(dp_du);
  V3 ts = cross(n, ss);
  if (norm2(ts) > 0.0f) {  // false for NaN: the pole falls through to coordinate_system
    ts = normalize(ts);
    ss = cross(ts, n);
  } else {
    coordinate_system(n, &ss, &ts);
  }
  s.p = contact + n * 1e-9f;
  s.ns = normalize(cross(ss, ts));
  s.ng = dot(n, s.ns) < 0.0f ? -n : n;
  s.ss = ss;
  s.ts = ts;
  s.material = sp.material;
  return s;
}

RRT_DEV V3 to_local(const Surface& s, V3 w) { return mk(dot(w, s.ss), dot(w, s.ts), dot(w, s.ns)); }
RRT_DEV V3 to_world(const Surface& s, V3 w) {
  return mk(s.ss.x * w.x + s.ts.x * w.y + s.ns.x * w.z, s.ss.y * w.x + s.ts.y * w.y + s.ns.y * w.z,
            s.ss.z * w.x + s.ts.z * w.y + s.ns.z * w.z);
}

// ---------------------------------------------------------------- Fresnel, GGX
// FresnelDielectric::evaluate (bsdf/helpers/fresnel.rs:47-74)
RRT_DEV float fresnel_dielectric(float eta_i, float eta_t, float cos_i) {
  cos_i = clampf(cos_i, -1.0f, 1.0f);
  if (cos_i <= 0.0f) {
    const float tmp = eta_i;
    eta_i = eta_t;
    eta_t = tmp;
    cos_i = fabsf(cos_i);
  }
  const float sin_i = sqrtf(fmaxf(1.0f - cos_i * cos_i, 0.0f));
  const float sin_t = eta_i / eta_t * sin_i;
  if (sin_t >= 1.0f) return 1.0f;
  const float cos_t = sqrtf(fmaxf(1.0f - sin_t * sin_t, 0.0f));
  const float rpar = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
  const float rper = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
  return (rpar * rpar + rper * rper) * 0.5f;
}

RRT_DEV float sin2_theta(V3 w) { return fmaxf(1.0f - w.z * w.z, 0.0f); }

// Isotropic Trowbridge-Reitz (bsdf/helpers/microfacet_distribution.rs:144-171
// with alpha_x == alpha_y, the only form plastic.rs:52-54 constructs).
RRT_DEV float ggx_d(V3 wh, float alpha) {
  const float c2 = wh.z * wh.z;
  const float t2 = sin2_theta(wh) / c2;
  if (isinf(t2) || isnan(t2)) return 0.0f;
  const float e = t2 / (alpha * alpha);
  return 1.0f / (kPi * alpha * alpha * c2 * c2 * (1.0f + e) * (1.0f + e));
}
RRT_DEV float ggx_lambda(V3 w, float alpha) {
  const float t2 = sin2_theta(w) / (w.z * w.z);
  if (isinf(t2) || isnan(t2)) return 0.0f;
  return (-1.0f + sqrtf(1.0f + alpha * alpha * t2)) * 0.5f;
}

// trowbridge_reitz_sample_11 (microfacet_distribution.rs:56-103); the asserts
// there become clamps.
RRT_DEV void ggx_sample_11(float cos_theta, float u1, float u2, float* sx, float* sy) {
  if (cos_theta > 0.9999f) {
    const float r = sqrtf(u1 / (1.0f - u1));
    float s, c;
    sincos_fast(2.0f * kPi * u2, &s, &c);
    *sx = r * c;
    *sy = r * s;
    return;
  }
  const float sin_theta = sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
  const float tan_theta = sin_theta / cos_theta;
  float a = 1.0f / tan_theta;
  const float g1 = 2.0f / (1.0f + sqrtf(1.0f + 1.0f / (a * a)));
  a = 2.0f * u1 / g1 - 1.0f;
  float tmp = 1.0f / (a * a - 1.0f);
  if (tmp > 1e10f) tmp = 1e10f;
  const float b = tan_theta;
  const float d = sqrtf(fmaxf(b * b * tmp * tmp - (a * a - b * b) * tmp, 0.0f));
  const float x1 = b * tmp - d, x2 = b * tmp + d;
  const float slope_x = (a < 0.0f || x2 > (1.0f / tan_theta)) ? x1 : x2;
  float s;
  if (u2 > 0.5f) { s = 1.0f; u2 = 2.0f * (u2 - 0.5f); }
  else { s = -1.0f; u2 = 2.0f * (0.5f - u2); }
  const float z = (u2 * (u2 * (u2 * 0.27385f - 0.73369f) + 0.46341f)) /
                  (u2 * (u2 * (u2 * 0.093073f + 0.309420f) - 1.0f) + 0.597999f);
  float slope_y = s * z * sqrtf(1.0f + slope_x * slope_x);
  *sx = isfinite(slope_x) ? slope_x : 0.0f;
  *sy = isfinite(slope_y) ? slope_y : 0.0f;
}

// sample_wh with sample_visible_area = true (microfacet_distribution.rs:105-123, 207-221)
RRT_DEV V3 ggx_sample_wh(V3 wo, float alpha, float u1, float u2) {
  const bool flip = wo.z < 0.0f;
  const V3 wi = flip ? -wo : wo;
  const V3 ws = normalize(mk(alpha * wi.x, alpha * wi.y, wi.z));
  float sx, sy;
  ggx_sample_11(ws.z, u1, u2, &sx, &sy);
  const float st = sqrtf(sin2_theta(ws));
  // cos_phi / sin_phi both return 1 at the pole (bsdf/helpers.rs:30-45)
  const float cp = st == 0.0f ? 1.0f : clampf(ws.x / st, -1.0f, 1.0f);
  const float sp = st == 0.0f ? 1.0f : clampf(ws.y / st, -1.0f, 1.0f);
  const float tmp = cp * sx - sp * sy;
  sy = sp * sx + cp * sy;
  sx = tmp;
  const V3 wh = normalize(mk(-sx * alpha, -sy * alpha, 1.0f));
  return flip ? -wh : wh;
}

// ---------------------------------------------------------------- lobes
// OrenNayar::f without the colour (bsdf/oren_nayar.rs:35-61)
RRT_DEV float oren_nayar_term(const DeviceMaterial& m, V3 wo, V3 wi) {
  const float sin_i = sqrtf(sin2_theta(wi)), sin_o = sqrtf(sin2_theta(wo));
  float max_cos = 0.0f;
  if (sin_i > 1e-4f && sin_o > 1e-4f) {
    const float cpi = clampf(wi.x / sin_i, -1.0f, 1.0f), spi = clampf(wi.y / sin_i, -1.0f, 1.0f);
    const float cpo = clampf(wo.x / sin_o, -1.0f, 1.0f), spo = clampf(wo.y / sin_o, -1.0f, 1.0f);
    max_cos = fmaxf(cpi * cpo + spi * spo, 0.0f);
  }
  float sin_alpha, tan_beta;
  if (fabsf(wi.z) > fabsf(wo.z)) { sin_alpha = sin_o; tan_beta = sin_i / fabsf(wi.z); }
  else { sin_alpha = sin_i; tan_beta = sin_o / fabsf(wo.z); }
  return m.oren_a + m.oren_b * max_cos * sin_alpha * tan_beta;
}

// Sum of f over the lobes that carry REFLECTION (every lobe but the
// transmission one; the two mirror kinds evaluate to zero), local directions.
RRT_DEV V3 lobes_f_reflect(const DeviceMaterial& m, V3 wo, V3 wi) {
  V3 f = mk(0.0f, 0.0f, 0.0f);
  if (m.diffuse_kind == DIFF_LAMBERT) f = mk(m.diffuse) * kInvPi;                                   // lambertian.rs:23-25
  else if (m.diffuse_kind == DIFF_OREN_NAYAR) f = mk(m.diffuse) * (kInvPi * oren_nayar_term(m, wo, wi));
  if (m.spec_kind == SPEC_MICROFACET) {                                                             // microfacet_reflection.rs:41-60
    const float co = fabsf(wo.z), ci = fabsf(wi.z);
    V3 wh = wi + wo;
    if (co != 0.0f && ci != 0.0f && !is_zero(wh)) {
      wh = normalize(wh);
      const V3 whf = wh.z < 0.0f ? -wh : wh;
      const float fr = fresnel_dielectric(1.0f, m.ior, fabsf(dot(wi, whf)));
      const float g = 1.0f / (1.0f + ggx_lambda(wo, m.alpha) + ggx_lambda(wi, m.alpha));
      f = f + mk(m.spec) * (ggx_d(wh, m.alpha) * g * fr / (4.0f * ci * co));
    }
  }
  return f;
}

// MicrofacetReflection::pdf (microfacet_reflection.rs:62-71)
RRT_DEV float microfacet_pdf(const DeviceMaterial& m, V3 wo, V3 wi) {
  if (!(wo.z * wi.z > 0.0f)) return 0.0f;
  const V3 wh = normalize(wo + wi);
  const float g1 = 1.0f / (1.0f + ggx_lambda(wo, m.alpha));
  return ggx_d(wh, m.alpha) * g1 * fabsf(dot(wh, wo)) / fabsf(wo.z) / (4.0f * fabsf(dot(wo, wh)));
}

// shift_cos_in (bsdf.rs:280-284)
RRT_DEV float shift_cos_in(float cos_in) {
  cos_in = fminf(cos_in, 1.0f);
  if (!(cos_in > 0.0f)) return 0.0f;  // the reference yields -0 (or NaN at exactly 0) here: no contribution
  return fmaxf(__cosf(acosf(cos_in) * 1.002f), 0.0f) / cos_in;
}

// Bsdf::f(wo, wi, ALL & !SPECULAR) and Bsdf::pdf(wo, wi, ALL & !SPECULAR)
// (bsdf.rs:185-241) in one pass.  Every lobe matches that flag set (matching is
// `intersects` and each lobe carries REFLECTION or REFRACTION), so pdf is the
// mean over all lobes with the two mirror kinds and transmission counting 1.
RRT_DEV void bsdf_eval(const Surface& s, const DeviceMaterial& m, V3 wi_world, V3* f_out, float* pdf_out) {
  const V3 wo = to_local(s, s.wo), wi = to_local(s, wi_world);
  const bool reflect = dot(wi_world, s.ng) * dot(s.wo, s.ng) > 0.0f;
  V3 f = mk(0.0f, 0.0f, 0.0f);
  if (reflect) f = lobes_f_reflect(m, wo, wi) * shift_cos_in(dot(wi_world, s.ns));  // no lobe carries TRANSMISSION
  *f_out = f;
  float pdf = 0.0f;
  if (wo.z != 0.0f && m.n_lobes > 0) {
    const bool same = wo.z * wi.z > 0.0f;
    if (m.has_trans) pdf += 1.0f;
    if (m.diffuse_kind != DIFF_NONE) pdf += same ? fabsf(wi.z) * kInvPi : 0.0f;
    if (m.spec_kind == SPEC_MICROFACET) pdf += microfacet_pdf(m, wo, wi);
    else if (m.spec_kind != SPEC_NONE) pdf += 1.0f;
    pdf /= (float)m.n_lobes;
  }
  *pdf_out = pdf;
}

struct BsdfSample {
  V3 wi;     // world
  V3 f;
  float pdf;
  bool specular;
};

// Bsdf::sample_f (bsdf.rs:74-183): uniform choice among the lobes, then the
// chosen lobe's own sample_f; the returned pdf is the chosen lobe's pdf (the
// accumulation at :147-160 adds that same pdf once per other lobe and divides
// by the count); f is the sum over reflective lobes unless the chosen lobe is
// specular.
RRT_DEV BsdfSample bsdf_sample(const Surface& s, const DeviceMaterial& m, float u_lobe, float u0, float u1) {
  BsdfSample r;
  r.wi = mk(0.0f, 0.0f, 0.0f);
  r.f = r.wi;
  r.pdf = 0.0f;
  r.specular = false;
  if (m.n_lobes == 0) return r;
  const V3 wo = to_local(s, s.wo);
  if (wo.z == 0.0f) return r;
  int pick = min((int)(u_lobe * (float)m.n_lobes), m.n_lobes - 1);
  // lobe order: transmission, diffuse, specular / glossy
  int kind;  // 0 transmission, 1 diffuse, 2 specular slot
  if (m.has_trans) {
    if (pick == 0) kind = 0;
    else if (m.diffuse_kind != DIFF_NONE && pick == 1) kind = 1;
    else kind = 2;
  } else if (m.diffuse_kind != DIFF_NONE) {
    kind = pick == 0 ? 1 : 2;
  } else {
    kind = 2;
  }
  V3 wi;
  if (kind == 0) {
    // SpecularTransmission::sample_f (bsdf/specular_transmission.rs:53-77), eta_a = 1, eta_b = ior
    r.specular = true;
    const float eta_i = wo.z > 0.0f ? 1.0f : m.ior, eta_t = wo.z > 0.0f ? m.ior : 1.0f;
    const V3 n = mk(0.0f, 0.0f, wo.z < 0.0f ? -1.0f : 1.0f);
    const float eta = eta_i / eta_t;
    const float cos_i = dot(n, wo);
    const float sin2_t = eta * eta * fmaxf(0.0f, 1.0f - cos_i * cos_i);
    if (sin2_t > 1.0f) return r;  // total internal reflection
    const float cos_t = sqrtf(1.0f - sin2_t);
    wi = eta * -wo + (eta * cos_i - cos_t) * n;
    const float fr = fresnel_dielectric(1.0f, m.ior, wi.z);
    r.f = mk(m.trans) * ((1.0f - fr) * (eta_i * eta_i) / (eta_t * eta_t) / fabsf(wi.z));
    r.pdf = 1.0f;
    r.wi = to_world(s, wi);
    return r;
  }
  if (kind == 2 && (m.spec_kind == SPEC_MIRROR_DIELECTRIC || m.spec_kind == SPEC_MIRROR_NOOP)) {
    // SpecularReflection::sample_f (bsdf/specular_reflection.rs:35-41)
    r.specular = true;
    wi = mk(-wo.x, -wo.y, wo.z);
    const float fr = m.spec_kind == SPEC_MIRROR_NOOP ? 1.0f : fresnel_dielectric(1.0f, m.ior, wi.z);
    r.f = mk(m.spec) * (fr / fabsf(wi.z));
    r.pdf = 1.0f;
    r.wi = to_world(s, wi);
    return r;
  }
  float pdf;
  if (kind == 1) {
    // trait default (bsdf.rs:319-326): cosine hemisphere on wo's side
    float dx, dy;
    concentric_sample_disk(u0, u1, &dx, &dy);
    wi = mk(dx, dy, sqrtf(fmaxf(0.0f, 1.0f - dx * dx - dy * dy)));
    if (wo.z < 0.0f) wi.z = -wi.z;
    pdf = wo.z * wi.z > 0.0f ? fabsf(wi.z) * kInvPi : 0.0f;
  } else {
    // MicrofacetReflection::sample_f (bsdf/microfacet_reflection.rs:73-105)
    const V3 wh = ggx_sample_wh(wo, m.alpha, u0, u1);
    const float odh = dot(wo, wh);
    if (odh < 0.0f) return r;
    wi = -wo + 2.0f * odh * wh;
    if (!(wo.z * wi.z > 0.0f) || fabsf(odh) < 1e-6f) return r;
    const float g1 = 1.0f / (1.0f + ggx_lambda(wo, m.alpha));
    pdf = ggx_d(wh, m.alpha) * g1 * fabsf(odh) / fabsf(wo.z) / (4.0f * fabsf(odh));
  }
  if (pdf == 0.0f) return r;
  r.wi = to_world(s, wi);
  r.pdf = pdf;
  const bool reflect = dot(r.wi, s.ng) * dot(s.wo, s.ng) > 0.0f;
  if (reflect) r.f = lobes_f_reflect(m, wo, wi);
  return r;
}

// ---------------------------------------------------------------- lights
// MipMap::lookup (textures/mip_map.rs:26-47) behind InfiniteAreaLight::environment_emitting
// (lights/infinite_area.rs:96-106); world_to_light is a pure translation.
RRT_DEV V3 env_lookup(const SceneTables* __restrict__ sc, const uint8_t* __restrict__ env, V3 w) {
  float phi = atan2f(w.x, w.z);
  if (phi < 0.0f) phi += 2.0f * kPi;
  const float theta = acosf(clampf(w.y, -1.0f, 1.0f));
  const float px = 1.0f - phi * (1.0f / (2.0f * kPi));
  const float py = theta * kInvPi;
  const float u = px - floorf(px);
  const float v = clampf(py, 0.0f, 1.0f);
  const int x = min((int)(u * (float)sc->env_w), sc->env_w - 1);
  const int y = min((int)(v * (float)sc->env_h), sc->env_h - 1);
  // texels are stored four bytes apiece (r, g, b, unused): one load
  const uint32_t t = reinterpret_cast<const uint32_t*>(env)[(size_t)y * sc->env_w + x];
  return mk((float)(t & 0xffu) * (1.0f / 255.0f), (float)((t >> 8) & 0xffu) * (1.0f / 255.0f),
            (float)((t >> 16) & 0xffu) * (1.0f / 255.0f));
}

// What an escaping ray collects when the path is allowed to see emitters
// (src/tracer.rs:50-53): every DistantLight's intensity plus the environment texel.
RRT_DEV V3 escape_radiance(const SceneTables* __restrict__ sc, const uint8_t* __restrict__ env, V3 dir) {
  V3 l = mk(sc->distant_env);
  if (sc->has_env) l = l + env_lookup(sc, env, dir);
  return l;
}

struct LightSample {
  V3 wi;       // direction towards the light (unit)
  V3 li;       // incident radiance
  float pdf;
  float dist;  // distance to the sampled point; INFINITY for lights at infinity
  bool delta;
  int skip;    // object code of the light's own quad (never occludes its own sample), or -1
};

// LightTrait::sample_irradiance for the four kinds (lights/point.rs:26-43,
// area.rs:28-45, distant.rs:29-45, infinite_area.rs:45-82).
RRT_DEV LightSample sample_light(const SceneTables* __restrict__ sc, const uint8_t* __restrict__ env, int li, V3 p,
                                 float u0, float u1) {
  const DeviceLight& l = sc->lights[li];
  LightSample r;
  r.skip = -1;
  r.delta = false;
  if (l.kind == RRT_LIGHT_AREA) {
    const V3 lp = mk(l.v0) + mk(l.v1) * u0 + mk(l.v2) * u1;
    const V3 d = lp - p;
    const float dist2 = norm2(d);
    r.dist = sqrtf(dist2);
    r.wi = d / r.dist;
    const V3 n = mk(l.normal);
    const float cos_l = dot(n, -r.wi);
    // Rectangle::pdf re-intersects the quad along wi (objects/rectangle.rs:113-129).  The sampled
    // point is that intersection unless the ray is parallel to the quad -- but the intersection
    // routine accepts a point by its projections onto the two sides (rectangle.rs:82-93), and for
    // sides that are not perpendicular the projections of pos + u a + v b are u + v (a.b)/|a|^2 and
    // v + u (a.b)/|b|^2: where either leaves [0, 1] the pdf is 0 and the sample is dropped.
    const float proj_a = fmaf(u1, l.skew_a, u0), proj_b = fmaf(u0, l.skew_b, u1);
    const bool on_quad = proj_a >= 0.0f && proj_a <= 1.0f && proj_b >= 0.0f && proj_b <= 1.0f;
    r.pdf = (fabsf(cos_l) < 1e-9f || !on_quad) ? 0.0f : dist2 / (fabsf(cos_l) * l.area);
    r.li = cos_l > 0.0f ? mk(l.intensity) : mk(0.0f, 0.0f, 0.0f);
    r.skip = l.rect >= 0 ? sc->n_tris + l.rect : -1;
  } else if (l.kind == RRT_LIGHT_DISTANT) {
    r.wi = mk(l.v0);  // normalised at upload
    r.pdf = 1.0f;
    r.li = mk(l.intensity);
    r.dist = INFINITY;
    r.delta = true;
  } else if (l.kind == RRT_LIGHT_INFINITE) {
    float st, ct, sp, cp;
    sincos_fast(u1 * kPi, &st, &ct);
    sincos_fast(u0 * 2.0f * kPi, &sp, &cp);
    r.wi = mk(st * sp, ct, st * cp);
    r.li = env_lookup(sc, env, normalize(r.wi));
    r.pdf = st != 0.0f ? vmax(r.li) / (2.0f * kPi * kPi * st) : 0.0f;
    r.dist = INFINITY;
  } else {  // point
    const V3 d = mk(l.v0) - p;
    const float dist2 = norm2(d);
    r.dist = sqrtf(dist2);
    r.wi = d / r.dist;
    r.pdf = 1.0f;
    r.li = mk(l.intensity) / dist2;
    r.delta = true;
  }
  return r;
}

}  // namespace rrt
