// Scene queries on the device: BVH traversal over the leaf-ordered triangle
// records plus the handful of analytic shapes.  These are the device forms of
// check_intersect_scene (closest hit, src/renderer.rs:213-249) and
// check_intersect_scene_simple (any hit below a distance, src/renderer.rs:251-275).
#pragma once
#include "rrt_device.h"
#include "rrt_math.h"

namespace rrt {

struct Hit {
  int prim;  // < 0: miss; < n_tris: leaf-order triangle; else n_tris + index into rects, planes, spheres
  float t, u, v;
};

constexpr float kTriTMax = 1000.0f;  // objects/triangle.rs:164-171

// Traversal stack: the first kLdsDepth entries of every lane live in LDS, laid out
// [entry][thread] so that the 64 lanes of a wave hit 64 consecutive words (no bank
// conflicts); deeper entries -- rare -- spill to a per-lane array in scratch memory.
constexpr int kTraceBlock = 256;   // threads per block of every kernel that traverses
#ifndef RRT_LDS_STACK_DEPTH
#define RRT_LDS_STACK_DEPTH 0  // measured: 0 (all in scratch) beats 12, 16 and 24 -- LDS costs a wave of occupancy
#endif
constexpr int kLdsDepth = RRT_LDS_STACK_DEPTH;
constexpr int kSpillDepth = 96 - kLdsDepth;  // 96 entries in all: deeper than any tree the builder makes (SAH depth <= 40 + median levels)
#define RRT_DECLARE_TRACE_STACK() __shared__ int rrt_trace_stack[(rrt::kLdsDepth > 0 ? rrt::kLdsDepth : 1) * rrt::kTraceBlock]

struct TraceStack {
  int* lds;  // this thread's column: entry e at lds[e * kTraceBlock]
  int spill[kSpillDepth];
  int sp;
  RRT_DEV void push(int v) {
    if (sp < kLdsDepth) lds[sp * kTraceBlock] = v;
    else if (sp < kLdsDepth + kSpillDepth) spill[sp - kLdsDepth] = v;
    ++sp;
  }
  RRT_DEV int pop() {
    --sp;
    return sp < kLdsDepth ? lds[sp * kTraceBlock] : spill[min(sp - kLdsDepth, kSpillDepth - 1)];
  }
};

struct RayPre {
  V3 o, d, inv, ood;  // ood = o * inv
};

RRT_DEV RayPre prepare_ray(V3 o, V3 d) {
  RayPre r;
  r.o = o;
  r.d = d;
  // a zero component would make 0 * inf = NaN in the slab test
  const float eps = 1e-18f;
  const float dx = fabsf(d.x) > eps ? d.x : copysignf(eps, d.x);
  const float dy = fabsf(d.y) > eps ? d.y : copysignf(eps, d.y);
  const float dz = fabsf(d.z) > eps ? d.z : copysignf(eps, d.z);
  r.inv = mk(1.0f / dx, 1.0f / dy, 1.0f / dz);
  r.ood = o * r.inv;
  return r;
}

// Entry distance of the ray into a box, or +inf when it misses [tmin, tmax].
RRT_DEV float slab(const RayPre& r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float tmin,
                   float tmax) {
  const float x0 = fmaf(mnx, r.inv.x, -r.ood.x), x1 = fmaf(mxx, r.inv.x, -r.ood.x);
  const float y0 = fmaf(mny, r.inv.y, -r.ood.y), y1 = fmaf(mxy, r.inv.y, -r.ood.y);
  const float z0 = fmaf(mnz, r.inv.z, -r.ood.z), z1 = fmaf(mxz, r.inv.z, -r.ood.z);
  const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
  const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
  // widen by a few ulps so that rounding never rejects a box the exact test would enter
  return tn <= tf * 1.0000005f ? tn : INFINITY;
}

// Moeller-Trumbore.  Accepts t in (tmin, tmax); barycentrics u (vertex 1), v (vertex 2).
RRT_DEV bool intersect_tri(const TriIsect* __restrict__ tris, int index, const RayPre& r, float tmin, float tmax,
                           float* t_out, float* u_out, float* v_out) {
  const float4* rec = reinterpret_cast<const float4*>(tris + index);
  const float4 a = rec[0], b = rec[1], c = rec[2];
  const V3 v0 = mk(a), e1 = mk(b), e2 = mk(c);
  const V3 pvec = cross(r.d, e2);
  const float det = dot(e1, pvec);
  if (det == 0.0f) return false;
  const float inv_det = 1.0f / det;
  const V3 tvec = r.o - v0;
  const float u = dot(tvec, pvec) * inv_det;
  const V3 qvec = cross(tvec, e1);
  const float v = dot(r.d, qvec) * inv_det;
  const float t = dot(e2, qvec) * inv_det;
  if (u < 0.0f || v < 0.0f || u + v > 1.0f || !(t > tmin) || !(t < tmax)) return false;
  *t_out = t;
  *u_out = u;
  *v_out = v;
  return true;
}

struct TraceCounters {
  uint32_t nodes, tris;
};

// Closest (ANY = false) or any (ANY = true) triangle hit with t in (tmin, tmax).
// `skip` is a leaf-order triangle that is never reported (the surface a shadow
// ray leaves), or -1.  Returns true when something was hit.
template <bool ANY, bool COUNT>
RRT_DEV bool traverse_tris(const BvhNode* __restrict__ nodes, const TriIsect* __restrict__ tris, const RayPre& r,
                           float tmin, float tmax, int skip, Hit* hit, TraceCounters* ctr, int* lds_stack) {
  TraceStack stack;
  stack.lds = lds_stack + threadIdx.x;
  stack.sp = 0;
  int cur = 0;
  bool found = false;
  const int kDone = 0x7fffffff;
  while (cur != kDone) {
    while (cur >= 0 && cur != kDone) {
      const float4* np = reinterpret_cast<const float4*>(nodes + cur);
      const float4 n0 = np[0], n1 = np[1], n2 = np[2];
      const int4 n3 = reinterpret_cast<const int4*>(np)[3];
      if (COUNT) ctr->nodes++;
      const float tl = slab(r, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, tmin, tmax);
      const float tr = slab(r, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, tmin, tmax);
      const bool hl = tl < INFINITY, hr = tr < INFINITY;
      if (hl && hr) {
        const bool left_first = tl <= tr;
        const int nearc = left_first ? n3.x : n3.y;
        const int farc = left_first ? n3.y : n3.x;
        stack.push(farc);
        cur = nearc;
      } else if (hl) {
        cur = n3.x;
      } else if (hr) {
        cur = n3.y;
      } else {
        cur = stack.sp > 0 ? stack.pop() : kDone;
      }
    }
    if (cur == kDone) break;
    // leaf: ~cur = first << 2 | (count - 1)
    const int code = ~cur;
    const int first = code >> 2, cnt = (code & 3) + 1;
    for (int i = 0; i < cnt; ++i) {
      const int ti = first + i;
      if (COUNT) ctr->tris++;
      float t, u, v;
      if (ti != skip && intersect_tri(tris, ti, r, tmin, tmax, &t, &u, &v)) {
        if (ANY) return true;
        tmax = t;
        hit->prim = ti;
        hit->t = t;
        hit->u = u;
        hit->v = v;
        found = true;
      }
    }
    cur = stack.sp > 0 ? stack.pop() : kDone;
  }
  return found;
}

// The reference finds a quad only if the bvh crate's box test lets its leaf
// through: f32 slab test on the box {pos, pos + a + b} (objects/rectangle.rs:136-146).
RRT_DEV bool rect_box_pass(const DeviceRect& q, const RayPre& r, int mode) {
  if (mode == RRT_RECT_AABB_OFF) return true;
  const float inv[3] = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
  const float o[3] = {r.o.x, r.o.y, r.o.z};
  float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float t0 = (q.bmin[k] - o[k]) * inv[k], t1 = (q.bmax[k] - o[k]) * inv[k];
    tmin = fmaxf(tmin, fminf(t0, t1));
    tmax = fminf(tmax, fmaxf(t0, t1));
  }
  const float lo = fmaxf(tmin, 0.0f);
  return mode == RRT_RECT_AABB_INCLUSIVE ? tmax >= lo : tmax > lo;
}

// objects/rectangle.rs:58-101
RRT_DEV bool intersect_rect(const DeviceRect& q, const RayPre& r, float* t_out) {
  const V3 n = mk(q.normal);
  const float denom = dot(n, r.d);
  if (fabsf(denom) < 1e-9f) return false;
  const float t = dot(mk(q.pos) - r.o, n) / denom;
  if (t < 1e-9f) return false;
  const V3 p0p = r.o + r.d * t - mk(q.pos);
  const float a = dot(p0p, mk(q.a)) * q.inv_aa;
  const float b = dot(p0p, mk(q.b)) * q.inv_bb;
  if (!(a >= 0.0f && a <= 1.0f) || !(b >= 0.0f && b <= 1.0f)) return false;
  *t_out = t;
  return true;
}

// objects/plane.rs:45-58
RRT_DEV bool intersect_plane(const DevicePlane& pl, const RayPre& r, float* t_out) {
  const V3 n = mk(pl.normal);
  const float denom = dot(n, r.d);
  if (fabsf(denom) < 1e-9f) return false;
  const float t = dot(mk(pl.pos) - r.o, n) / denom;
  if (t < 0.0000001f) return false;
  *t_out = t;
  return true;
}

// objects/sphere.rs:116-154
RRT_DEV bool intersect_sphere(const DeviceSphere& s, const RayPre& r, float* t_out) {
  const V3 oc = r.o - mk(s.center);
  const float a = dot(r.d, r.d), b = dot(oc, r.d), c = dot(oc, oc) - s.radius * s.radius;
  const float disc = b * b - a * c;
  if (disc < 0.0f) return false;
  const float sq = sqrtf(disc);
  float t = (-b - sq) / a;
  if (!(t > 0.0001f && t < 3.4e38f)) {
    t = (-b + sq) / a;
    if (!(t > 0.0001f && t < 3.4e38f)) return false;
  }
  *t_out = t;
  return true;
}

// Analytic shapes after the triangles: closest hit below hit->t (or any below tmax).
// skip_a / skip_b: object codes that are not reported (n_tris + k), or -1.
template <bool ANY>
RRT_DEV bool intersect_others(const SceneTables* __restrict__ sc, const RayPre& r, float tmax, int skip_a, int skip_b,
                              Hit* hit) {
  bool found = false;
  int code = sc->n_tris;
  for (int i = 0; i < sc->n_rects; ++i, ++code) {
    float t;
    if (code != skip_a && code != skip_b && rect_box_pass(sc->rects[i], r, sc->rect_aabb_mode) && intersect_rect(sc->rects[i], r, &t) &&
        t < tmax) {
      if (ANY) return true;
      tmax = t; hit->prim = code; hit->t = t; hit->u = 0.0f; hit->v = 0.0f; found = true;
    }
  }
  for (int i = 0; i < sc->n_planes; ++i, ++code) {
    float t;
    if (code != skip_a && code != skip_b && intersect_plane(sc->planes[i], r, &t) && t < tmax) {
      if (ANY) return true;
      tmax = t; hit->prim = code; hit->t = t; hit->u = 0.0f; hit->v = 0.0f; found = true;
    }
  }
  for (int i = 0; i < sc->n_spheres; ++i, ++code) {
    float t;
    if (code != skip_a && code != skip_b && intersect_sphere(sc->spheres[i], r, &t) && t < tmax) {
      if (ANY) return true;
      tmax = t; hit->prim = code; hit->t = t; hit->u = 0.0f; hit->v = 0.0f; found = true;
    }
  }
  return found;
}

// check_intersect_scene
template <bool COUNT>
RRT_DEV Hit closest_hit(const BvhNode* __restrict__ nodes, const TriIsect* __restrict__ tris,
                        const SceneTables* __restrict__ sc, V3 o, V3 d, TraceCounters* ctr, int* lds_stack) {
  Hit hit;
  hit.prim = -1;
  hit.t = INFINITY;
  hit.u = hit.v = 0.0f;
  const RayPre r = prepare_ray(o, d);
  if (sc->n_tris > 0) traverse_tris<false, COUNT>(nodes, tris, r, 0.0f, kTriTMax, -1, &hit, ctr, lds_stack);
  if (sc->n_rects + sc->n_planes + sc->n_spheres > 0) intersect_others<false>(sc, r, hit.t, -1, -1, &hit);
  return hit;
}

// check_intersect_scene_simple: anything closer than tmax?
template <bool COUNT>
RRT_DEV bool any_hit(const BvhNode* __restrict__ nodes, const TriIsect* __restrict__ tris,
                     const SceneTables* __restrict__ sc, V3 o, V3 d, float tmax, int skip_a, int skip_b,
                     TraceCounters* ctr, int* lds_stack) {
  const RayPre r = prepare_ray(o, d);
  Hit hit;
  // only skip_a can name a triangle (the surface a ray leaves); skip_b names a light's quad
  if (sc->n_tris > 0 &&
      traverse_tris<true, COUNT>(nodes, tris, r, 0.0f, fminf(tmax, kTriTMax), skip_a < sc->n_tris ? skip_a : -1, &hit,
                                 ctr, lds_stack))
    return true;
  if (sc->n_rects + sc->n_planes + sc->n_spheres > 0 && intersect_others<true>(sc, r, tmax, skip_a, skip_b, &hit))
    return true;
  return false;
}

}  // namespace rrt
