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
#define RRT_LDS_STACK_DEPTH 12  // measured: 12 beats 0 (all in scratch) by 15% and 24 by 3%
#endif
constexpr int kLdsDepth = RRT_LDS_STACK_DEPTH;
constexpr int kSpillDepth = 96 - kLdsDepth;  // 96 entries in all: deeper than any tree the builder makes (SAH depth <= 40 + median levels)
#define RRT_DECLARE_TRACE_STACK() __shared__ int rrt_trace_stack[(rrt::kLdsDepth > 0 ? rrt::kLdsDepth : 1) * rrt::kTraceBlock]

// The stack pointer is a plain local of the traversal loop, and the spill array a local of its
// own: as members of one struct next to a dynamically indexed array they would all be placed in
// scratch memory, and every push and pop would reload the stack pointer from there.
#define RRT_STACK_PUSH(v)                                                        \
  do {                                                                           \
    if (kLdsDepth > 0 && sp < kLdsDepth) lds[sp * kTraceBlock] = (v);            \
    else spill[min(sp - kLdsDepth, kSpillDepth - 1)] = (v);                      \
    ++sp;                                                                        \
  } while (0)
#define RRT_STACK_POP(dst)                                                       \
  do {                                                                           \
    --sp;                                                                        \
    if (kLdsDepth > 0 && sp < kLdsDepth) (dst) = lds[sp * kTraceBlock];          \
    else (dst) = spill[min(sp - kLdsDepth, kSpillDepth - 1)];                    \
  } while (0)

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

struct Ray {
  V3 o, d;
  float tmin, tmax;
  int skip;  // a leaf-order triangle that is never reported (the surface a shadow ray leaves), or -1
};

// ---------------------------------------------------------------- the traversal engine
// A wave streams through a queue of jobs instead of taking 64 of them and waiting for the slowest:
// rays differ in length by an order of magnitude, and with one ray per lane per launch only 1 lane
// in 5 (closest hit) or 1 in 10 (visibility) was doing work at any time.  Here every lane is a small
// state machine; a lane whose ray is finished hands in its result and takes the next job as soon as
// kRefillQuorum lanes are free, so the wave stays mostly full until the queue is empty.
//
// Job indices come from a global cursor in blocks of `grab` (one atomic per block per wave), which
// also balances the load between waves.  A job is a type with
//   bool begin(uint32_t index, Ray* ray)       first ray of job `index`; false: the job needs no ray
//   bool advance(bool found, const Hit& hit, const RayPre& r, Ray* ray)
//                                              a ray is done (triangles only: the job looks at the
//                                              analytic shapes itself); true: `ray` is its next one
//   void flush(bool done, uint32_t block)      wave-collective; `done` lanes completed a job taken
//                                              from index block `block` since the last flush
// Closest hit (ANY = false) or any hit (ANY = true) with t in (tmin, tmax).
//
// Inside the loop the lanes at inner nodes step together until kLeafQuorum of the travelling lanes
// have reached a leaf, then those test their triangles (the while-while scheme, without making
// every lane wait for the last straggler to arrive at a leaf).
#ifndef RRT_REFILL_QUORUM
#define RRT_REFILL_QUORUM 16
#endif
#ifndef RRT_LEAF_FRACTION
#define RRT_LEAF_FRACTION 4  // leaf phase when 1/4 of the travelling lanes are at a leaf
#endif

template <bool ANY, bool COUNT, class Job>
RRT_DEV void trace_stream(const BvhNode* __restrict__ nodes, const TriIsect* __restrict__ tris, bool has_tris,
                          uint32_t* work_cursor, uint32_t total, uint32_t grab, Job& job, TraceCounters* ctr,
                          int* lds_stack) {
  enum { kIdle = 0, kTravelling = 1, kArrived = 2, kJobDone = 3 };
  int* lds = lds_stack + threadIdx.x;  // this thread's column: entry e at lds[e * kTraceBlock]
  int spill[kSpillDepth];
  const int lane = threadIdx.x & 63;
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  uint32_t cursor = 0, end = 0, block = 0;  // wave-uniform: the index block being handed out
  bool exhausted = false;                   // wave-uniform: the global cursor has run past `total`
  int state = kIdle;
  uint32_t my_block = 0;
  RayPre r;
  r.o = r.d = r.inv = r.ood = mk(0.0f, 0.0f, 0.0f);
  float tmin = 0.0f, tmax = 0.0f;
  int skip = -1, cur = 0, sp = 0;
  Hit hit;
  hit.prim = -1;
  hit.t = INFINITY;
  hit.u = hit.v = 0.0f;
  bool found = false;

#define RRT_START_RAY(ray)                                   \
  do {                                                       \
    r = prepare_ray((ray).o, (ray).d);                       \
    tmin = (ray).tmin;                                       \
    tmax = (ray).tmax;                                       \
    skip = (ray).skip;                                       \
    hit.prim = -1;                                           \
    hit.t = INFINITY;                                        \
    found = false;                                           \
    sp = 0;                                                  \
    cur = 0;                                                 \
    state = has_tris ? kTravelling : kArrived;               \
  } while (0)
#define RRT_POP_OR_ARRIVE()                                  \
  do {                                                       \
    if (sp > 0) RRT_STACK_POP(cur);                          \
    else state = kArrived;                                   \
  } while (0)

  for (;;) {
    if (state == kArrived) {
      Ray ray;
      if (job.advance(found, hit, r, &ray)) RRT_START_RAY(ray);
      else state = kJobDone;
    }
    // hand in finished jobs, one index block at a time
    unsigned long long done_mask = __ballot(state == kJobDone);
    while (done_mask) {
      const uint32_t b = __shfl(my_block, __ffsll((long long)done_mask) - 1);
      const bool mine = state == kJobDone && my_block == b;
      job.flush(mine, b);
      if (mine) state = kIdle;
      done_mask &= ~__ballot(mine);
    }
    // refill idle lanes (a second pass when the first one ran into the end of an index block)
    for (int pass = 0; pass < 2; ++pass) {
      const unsigned long long idle_mask = __ballot(state == kIdle);
      if (idle_mask == 0 || exhausted) break;
      if (cursor >= end) {
        uint32_t g = 0;
        if (lane == 0) g = atomicAdd(work_cursor, grab);
        g = __shfl(g, 0);
        if (g >= total) {
          exhausted = true;
          break;
        }
        cursor = g;
        end = min(g + grab, total);
        block = g / grab;
      }
      const uint32_t index = cursor + (uint32_t)__popcll(idle_mask & lanes_below);
      if (state == kIdle && index < end) {
        my_block = block;
        Ray ray;
        if (job.begin(index, &ray)) RRT_START_RAY(ray);
        else state = kJobDone;
      }
      cursor = min(end, cursor + (uint32_t)__popcll(idle_mask));
    }
    if (__ballot(state == kTravelling) == 0) {
      if (__ballot(state != kIdle) == 0 && exhausted) break;
      continue;
    }

    // ---- traversal until enough lanes have arrived to make a refill worth its cost
    for (;;) {
      for (;;) {
        if (state == kTravelling && cur >= 0) {
          const float4* np = reinterpret_cast<const float4*>(nodes + cur);
          const float4 n0 = np[0], n1 = np[1], n2 = np[2];
          const int4 n3 = reinterpret_cast<const int4*>(np)[3];
          if (COUNT) ctr->nodes++;
          const float tl = slab(r, n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, tmin, tmax);
          const float tr = slab(r, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, tmin, tmax);
          const bool hl = tl < INFINITY, hr = tr < INFINITY;
          if (hl && hr) {
            const bool left_first = tl <= tr;
            const int farc = left_first ? n3.y : n3.x;
            RRT_STACK_PUSH(farc);
            cur = left_first ? n3.x : n3.y;
          } else if (hl) {
            cur = n3.x;
          } else if (hr) {
            cur = n3.y;
          } else {
            RRT_POP_OR_ARRIVE();
          }
        }
        const unsigned long long inner = __ballot(state == kTravelling && cur >= 0);
        const unsigned long long at_leaf = __ballot(state == kTravelling && cur < 0);
        if (inner == 0 || RRT_LEAF_FRACTION * __popcll(at_leaf) >= __popcll(inner | at_leaf)) break;
      }
      if (state == kTravelling && cur < 0) {
        // leaf: ~cur = first << 2 | (count - 1)
        const int code = ~cur;
        const int first = code >> 2, cnt = (code & 3) + 1;
        for (int i = 0; i < cnt; ++i) {
          const int ti = first + i;
          if (COUNT) ctr->tris++;
          float t, u, v;
          if (ti != skip && intersect_tri(tris, ti, r, tmin, tmax, &t, &u, &v)) {
            found = true;
            if (ANY) break;
            tmax = t;
            hit.prim = ti;
            hit.t = t;
            hit.u = u;
            hit.v = v;
          }
        }
        if (ANY && found) state = kArrived;
        else RRT_POP_OR_ARRIVE();
      }
      const unsigned long long arrived = __ballot(state == kArrived);
      if (__ballot(state == kTravelling) == 0 || __popcll(arrived) >= RRT_REFILL_QUORUM) break;
    }
  }
#undef RRT_START_RAY
#undef RRT_POP_OR_ARRIVE
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

}  // namespace rrt
