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

// Traversal stack of (node reference, entry distance) pairs: the first kLdsDepth entries of every
// lane live in LDS, laid out [entry][thread] so that the lanes of a wave touch consecutive words
// (no bank conflicts); deeper entries -- rare -- spill to a per-lane array in scratch memory.
constexpr int kTraceBlock = 256;   // threads per block of every kernel that traverses
#ifndef RRT_LDS_STACK_DEPTH
#define RRT_LDS_STACK_DEPTH 12
#endif
constexpr int kLdsDepth = RRT_LDS_STACK_DEPTH;
constexpr int kStackDepth = 192;   // a node pushes up to 3 entries: trees up to 63 inner levels (checked at upload)
constexpr int kSpillDepth = kStackDepth - kLdsDepth;
#define RRT_DECLARE_TRACE_STACK() __shared__ int2 rrt_trace_stack[(rrt::kLdsDepth > 0 ? rrt::kLdsDepth : 1) * rrt::kTraceBlock]
// The LDS part is addressed through a pointer that names its address space.  Through a generic
// pointer the compiler folds "LDS or scratch?" into a select between two generic addresses and
// issues flat loads, which take the slow path for both and wait on every counter.
// The accesses are volatile for the same reason: a volatile load is never merged with the scratch one.
typedef volatile __attribute__((address_space(3))) unsigned long long* LdsStackWordPtr;  // one entry: reference | distance bits << 32

// The stack pointer is a plain local of the traversal loop, and the spill array a local of its
// own: as members of one struct next to a dynamically indexed array they would all be placed in
// scratch memory, and every push and pop would reload the stack pointer from there.
#define RRT_STACK_PUSH(ref, t)                                                   \
  do {                                                                           \
    if (kLdsDepth > 0 && sp < kLdsDepth) {                                       \
      lds[sp * kTraceBlock] = (unsigned long long)(unsigned)(ref) |              \
                              ((unsigned long long)__float_as_uint(t) << 32);    \
    } else {                                                                     \
      spill[min(sp - kLdsDepth, kSpillDepth - 1)] = make_int2((ref), __float_as_int(t)); \
    }                                                                            \
    ++sp;                                                                        \
  } while (0)
#define RRT_STACK_POP(dst)                                                       \
  do {                                                                           \
    --sp;                                                                        \
    if (kLdsDepth > 0 && sp < kLdsDepth) {                                       \
      const unsigned long long w_ = lds[sp * kTraceBlock];                       \
      (dst).x = (int)(unsigned)w_;                                               \
      (dst).y = (int)(unsigned)(w_ >> 32);                                       \
    }                                                                            \
    else (dst) = spill[min(sp - kLdsDepth, kSpillDepth - 1)];                    \
  } while (0)

struct RayPre {
  V3 o, d, inv;  // inv: 1 / d, of a tiny number with d's sign where d is zero
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
  // hardware reciprocals (1 ulp): the slab test below has more slack than that
  r.inv = mk(__builtin_amdgcn_rcpf(dx), __builtin_amdgcn_rcpf(dy), __builtin_amdgcn_rcpf(dz));
  return r;
}

// Entry distance of the ray into a child's box from the distances to the three planes it meets
// first and the three it meets last, or +inf when the ray misses the box within [tmin, tmax].  A
// slot without a child holds an inverted box (every maximum below its minimum): it always misses.
RRT_DEV float child_entry(float nx, float ny, float nz, float fx, float fy, float fz, float tmin, float tmax) {
  const float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, tmin));
  const float tf = fminf(fminf(fx, fy), fminf(fz, tmax));
  // widen by a few ulps so that rounding never rejects a box the exact test would enter
  return tn <= tf * 1.0000005f ? tn : INFINITY;
}

// Moeller-Trumbore on an already loaded record.  Accepts t in (tmin, tmax); barycentrics u
// (vertex 1), v (vertex 2).  No early exits: a branch on the determinant would let the compiler
// sink the load of v0 below it, and a leaf would pay two dependent memory round trips per triangle.
struct TriRec {
  float4 v0, e1, e2;
};
RRT_DEV TriRec load_tri(const TriIsect* __restrict__ tris, int index) {
  const float4* rec = reinterpret_cast<const float4*>(tris + index);
  TriRec t;
  t.v0 = rec[0];
  t.e1 = rec[1];
  t.e2 = rec[2];
  return t;
}
RRT_DEV bool intersect_tri(const TriRec& rec, V3 o, V3 d, float tmin, float tmax, float* t_out, float* u_out,
                           float* v_out) {
  const V3 v0 = mk(rec.v0), e1 = mk(rec.e1), e2 = mk(rec.e2);
  const V3 pvec = cross(d, e2);
  const float det = dot(e1, pvec);
  const float inv_det = __builtin_amdgcn_rcpf(det);  // 1 ulp; a full division is ten instructions per triangle
  const V3 tvec = o - v0;
  const float u = dot(tvec, pvec) * inv_det;
  const V3 qvec = cross(tvec, e1);
  const float v = dot(d, qvec) * inv_det;
  const float t = dot(e2, qvec) * inv_det;
  *t_out = t;
  *u_out = u;
  *v_out = v;
  // det == 0 gives inf or NaN above; every comparison is written so that NaN fails it
  return det != 0.0f && u >= 0.0f && v >= 0.0f && u + v <= 1.0f && t > tmin && t < tmax;
}

struct TraceCounters {
  uint32_t nodes, tris;
  uint32_t entered;  // rays that passed the root-box test (all rays of a job type without that test)
};

#ifdef RRT_PHASE_CLOCKS
// Diagnostic build only: wave-cycles per phase of trace_stream, printed by rrt_get_stats.
//   0 refill cycles  1 node-phase cycles  2 leaf-phase cycles  3 node iterations  4 leaf iterations
//   5 lanes stepping (sum over node iterations)  6 lanes at a leaf (sum over leaf iterations)  7 refills
//   8 total wave cycles  9 triangle-loop trips
//   16..23 node iterations by lanes stepping (1-8, 9-16, ... 57-64)   24..31 leaf iterations by lanes at a leaf
__device__ unsigned long long rrt_phase_dbg[96];  // rows of 32: continuation / hook rays, visibility rays, camera rays (Job::kDebugRow)
#define RRT_CLK() __builtin_readcyclecounter()
#define RRT_DBG_ADD(i, v) do { if (lane == 0) dbg[i] += (v); } while (0)
#else
#define RRT_DBG_ADD(i, v) do { } while (0)
#endif

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
// Jobs come in numbered blocks; a wave takes the next block number from a global cursor (one atomic
// per block), which also balances the load between waves.  A job is a type with
//   uint32_t open_block(uint32_t j)            wave-uniform: makes block j current, returns its job count
//   bool begin(uint32_t i, Ray* ray)           first ray of job i of the current block; false: no ray needed
//   bool advance(bool found, const Hit& hit, const RayPre& r, Ray* ray)
//                                              a ray is done (triangles only: the job looks at the
//                                              analytic shapes itself); true: `ray` is its next one
//   void retire()                              the job is complete: set its result aside for flush (the
//                                              lane's begin() for its next job runs before the flush)
//   void flush(bool done, uint32_t block)      wave-collective; `done` lanes retired a job taken
//                                              from block `block` since the last flush
// Closest hit (ANY = false) or any hit (ANY = true) with t in (tmin, tmax).
//
// Two step schemes.  Closest hit: the lanes at inner nodes step together until 1 in
// RRT_LEAF_FRACTION of the travelling lanes have reached a leaf, then those test their triangles
// (the while-while scheme, without making every lane wait for the last straggler to arrive at a
// leaf; 2 measured best, 4 and 8 lose 3-5 %, 1 -- plain while-while -- 20 %).  Any hit, whose step has less arithmetic in it: every trip steps every lane once, node or
// triangle, with both kinds of load in flight together.  Measured: each scheme loses 15-20% on
// the other kind of ray.
// Lanes that must have arrived before a wave stops to hand in results and refill (`refill_quorum`):
// a refill costs a memory round trip or two for the whole wave, an idle lane costs its share of
// every step.  Scattered rays finish at very different times: 32 of 64.  Camera rays of neighbouring
// samples finish almost together: 56.  (Measured optima; 16 and 48 lose 2-3 % on scattered rays.)
#ifndef RRT_REFILL_QUORUM
#define RRT_REFILL_QUORUM 32
#endif
#ifndef RRT_ANY_UNIFIED
#define RRT_ANY_UNIFIED 1
#endif
#ifndef RRT_REFILL_QUORUM_COHERENT
#define RRT_REFILL_QUORUM_COHERENT 56
#endif
#ifndef RRT_LEAF_FRACTION
#define RRT_LEAF_FRACTION 2
#endif
// A job type with kRootTest has its rays tested against the box of the whole hierarchy before they set
// out: camera rays that pass the model by -- whole waves of them, the samples of one pixel -- then never
// enter the traversal loop.  (Rays that start on a surface are inside that box; for them the six
// registers the box takes cost more than the test can save.)
// Triangle records of a leaf in flight before the first is tested (closest hit): 1 or 2.
#ifndef RRT_LEAF_AHEAD
#define RRT_LEAF_AHEAD 2
#endif
#ifndef RRT_ROOT_TEST
#define RRT_ROOT_TEST 1
#endif

// One inner-node step of a lane: the four child boxes (decoded from the node's 8-bit grid) against
// the ray, nearest child next, the other children that were hit onto the stack (farthest first, so
// the nearest comes off first).
#define RRT_CSWAP(ta, ca, tb, cb)                   \
  do {                                              \
    const bool sw_ = tb < ta;                       \
    const float tt_ = sw_ ? ta : tb;                \
    const int cc_ = sw_ ? ca : cb;                  \
    ta = sw_ ? tb : ta;                             \
    ca = sw_ ? cb : ca;                             \
    tb = tt_;                                       \
    cb = cc_;                                       \
  } while (0)
// The node's three 16-byte words: children, {lo x, lo y, lo z, hi x}, {hi y, hi z, corner x | corner y << 16,
// corner z | shifts << 16}.
#define RRT_NODE_LOAD(np, w0, w1, w2)                                                    \
  do {                                                                                   \
    w0 = (np)[0]; w1 = (np)[1]; w2 = (np)[2];                                            \
  } while (0)
// Distance along the ray to plane q of an axis, with c the node's corner and s its shift, both in steps of the
// scene's grid: (base + (c + q * 2^s) * step - o) / d = q * 2^s * (step / d) + c * (step / d) + (base - o) / d.
// The ray carries ka = step / d and kb = (base - o) / d; a node costs a = 2^s * ka and b = c * ka + kb per axis.
#define RRT_Q(w, i) ((float)(((w) >> (8 * (i))) & 0xffu))
#define RRT_SHIFT(w, k) (((int)((w) << (11 - 5 * (k)))) >> 27)
#define RRT_CHILD_T(i)                                                                   \
  child_entry(fmaf(RRT_Q(nx_, i), ax, bx), fmaf(RRT_Q(ny_, i), ay, by), fmaf(RRT_Q(nz_, i), az, bz), \
              fmaf(RRT_Q(fx_, i), ax, bx), fmaf(RRT_Q(fy_, i), ay, by), fmaf(RRT_Q(fz_, i), az, bz), tmin, tmax)
#define RRT_NODE_STEP(w0, w1, w2)                                                        \
  do {                                                                                   \
    RRT_DBG_STEP();                                                                      \
    if (COUNT) ctr->nodes++;                                                             \
    const float ax = ldexpf(ka.x, RRT_SHIFT(w2.w, 0));                                   \
    const float ay = ldexpf(ka.y, RRT_SHIFT(w2.w, 1));                                   \
    const float az = ldexpf(ka.z, RRT_SHIFT(w2.w, 2));                                   \
    const float bx = fmaf((float)(w2.z & 0xffffu), ka.x, kb.x);                          \
    const float by = fmaf((float)(w2.z >> 16), ka.y, kb.y);                              \
    const float bz = fmaf((float)(w2.w & 0xffffu), ka.z, kb.z);                          \
    /* per axis the ray meets the minimum plane first if it runs upwards, else the maximum plane */ \
    const bool ux_ = ka.x >= 0.0f, uy_ = ka.y >= 0.0f, uz_ = ka.z >= 0.0f;               \
    const uint32_t nx_ = ux_ ? w1.x : w1.w, fx_ = ux_ ? w1.w : w1.x;                     \
    const uint32_t ny_ = uy_ ? w1.y : w2.x, fy_ = uy_ ? w2.x : w1.y;                     \
    const uint32_t nz_ = uz_ ? w1.z : w2.y, fz_ = uz_ ? w2.y : w1.z;                     \
    int c0 = (int)w0.x, c1 = (int)w0.y, c2 = (int)w0.z, c3 = (int)w0.w;                  \
    float t0 = RRT_CHILD_T(0);                                                           \
    float t1 = RRT_CHILD_T(1);                                                           \
    float t2 = RRT_CHILD_T(2);                                                           \
    float t3 = RRT_CHILD_T(3);                                                           \
    /* sort the four by entry distance (misses are +inf and end up last) */              \
    RRT_CSWAP(t0, c0, t1, c1);                                                           \
    RRT_CSWAP(t2, c2, t3, c3);                                                           \
    RRT_CSWAP(t0, c0, t2, c2);                                                           \
    RRT_CSWAP(t1, c1, t3, c3);                                                           \
    RRT_CSWAP(t1, c1, t2, c2);                                                           \
    if (t0 < INFINITY) {                                                                 \
      if (t3 < INFINITY) RRT_STACK_PUSH(c3, t3);                                         \
      if (t2 < INFINITY) RRT_STACK_PUSH(c2, t2);                                         \
      if (t1 < INFINITY) RRT_STACK_PUSH(c1, t1);                                         \
      cur = c0;                                                                          \
    } else {                                                                             \
      RRT_POP_OR_ARRIVE();                                                               \
    }                                                                                    \
  } while (0)

template <bool ANY, bool COUNT, class Job>
RRT_DEV void trace_stream(const BvhNode* __restrict__ nodes, const BvhGrid& grid, const TriIsect* __restrict__ tris,
                          bool has_tris, uint32_t* work_cursor, uint32_t n_blocks, int refill_quorum, Job& job, TraceCounters* ctr,
                          int2* lds_stack) {
  enum { kIdle = 0, kTravelling = 1, kArrived = 2, kJobDone = 3 };
  // this thread's column: entry e at lds[e * kTraceBlock]
  const LdsStackWordPtr lds = (LdsStackWordPtr)lds_stack + threadIdx.x;
  int2 spill[kSpillDepth];
  const int lane = threadIdx.x & 63;
  const unsigned long long lanes_below = (1ull << lane) - 1ull;
  uint32_t cursor = 0, end = 0, block = 0;  // wave-uniform: the index block being handed out
  const uint32_t n_waves = gridDim.x * (blockDim.x >> 6), wave_id = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  bool first_block = true;
  bool exhausted = false;                   // wave-uniform: the global cursor has run past `total`
  int state = kIdle;
  uint32_t my_block = 0;
  // the ray: origin, direction, and the two vectors a node step needs (see RRT_CHILD_T)
  V3 ro = mk(0.0f, 0.0f, 0.0f), rd = ro, ka = ro, kb = ro;
  const V3 g_base = mk(grid.base[0], grid.base[1], grid.base[2]), g_step = mk(grid.step[0], grid.step[1], grid.step[2]),
           g_inv_step = mk(grid.inv_step[0], grid.inv_step[1], grid.inv_step[2]);
  float tmin = 0.0f, tmax = 0.0f;
  int skip = -1, cur = 0, sp = 0;
  Hit hit;
  hit.prim = -1;
  hit.t = INFINITY;
  hit.u = hit.v = 0.0f;
  bool found = false;
  // The root's grid spans every triangle: corner + [0, 255] node steps per axis, here in steps of the scene's
  // grid and padded so that a ray any child-box test below would let in (those round by a few ulps) is let in
  // here as well.
  float root_lo[3] = {0.0f, 0.0f, 0.0f}, root_hi[3] = {0.0f, 0.0f, 0.0f};
  if (RRT_ROOT_TEST && Job::kRootTest && has_tris) {
    for (int k = 0; k < 3; ++k) {
      const float extent = ldexpf(255.0f, node_shift(nodes[0], k));
      const float pad = 1e-4f * extent + 1e-30f;
      root_lo[k] = (float)nodes[0].corner[k] - pad;
      root_hi[k] = (float)nodes[0].corner[k] + extent + pad;
    }
  }
#ifdef RRT_PHASE_CLOCKS
  unsigned long long dbg[32] = {};
  unsigned long long clk2 = 0;
  uint32_t lane_steps = 0;
  const unsigned long long clk_begin = RRT_CLK();
  unsigned long long clk = clk_begin;
#endif

#ifdef RRT_PHASE_CLOCKS
#define RRT_DBG_RAY_BEGIN() lane_steps = 0
#define RRT_DBG_STEP() ++lane_steps
#define RRT_DBG_RAY_END()                                                                              \
  do {                                                                                                 \
    atomicMax(&rrt_phase_dbg[(32 * Job::kDebugRow) + 13], (unsigned long long)lane_steps);                     \
    if (lane_steps > 256) atomicAdd(&rrt_phase_dbg[(32 * Job::kDebugRow) + 14], 1ull);                         \
    if (lane_steps > 1024) atomicAdd(&rrt_phase_dbg[(32 * Job::kDebugRow) + 15], 1ull);                        \
  } while (0)
#else
#define RRT_DBG_RAY_BEGIN() do { } while (0)
#define RRT_DBG_STEP() do { } while (0)
#define RRT_DBG_RAY_END() do { } while (0)
#endif
#define RRT_START_RAY(ray)                                   \
  do {                                                       \
    RRT_DBG_RAY_BEGIN();                                     \
    {                                                        \
      const RayPre pre_ = prepare_ray((ray).o, (ray).d);     \
      ro = pre_.o;                                           \
      rd = pre_.d;                                           \
      ka = pre_.inv * g_step;                                \
      kb = (g_base - pre_.o) * pre_.inv;                     \
    }                                                        \
    tmin = (ray).tmin;                                       \
    tmax = (ray).tmax;                                       \
    skip = (ray).skip;                                       \
    hit.prim = -1;                                           \
    hit.t = INFINITY;                                        \
    found = false;                                           \
    sp = 0;                                                  \
    cur = 0;                                                 \
    bool enter_ = has_tris;                                  \
    if (RRT_ROOT_TEST && Job::kRootTest && has_tris) {       \
      const float ax_ = fmaf(root_lo[0], ka.x, kb.x), bx_ = fmaf(root_hi[0], ka.x, kb.x); \
      const float ay_ = fmaf(root_lo[1], ka.y, kb.y), by_ = fmaf(root_hi[1], ka.y, kb.y); \
      const float az_ = fmaf(root_lo[2], ka.z, kb.z), bz_ = fmaf(root_hi[2], ka.z, kb.z); \
      const float tn_ = fmaxf(fmaxf(fminf(ax_, bx_), fminf(ay_, by_)), fmaxf(fminf(az_, bz_), tmin));   \
      const float tf_ = fminf(fminf(fmaxf(ax_, bx_), fmaxf(ay_, by_)), fminf(fmaxf(az_, bz_), tmax));   \
      enter_ = tn_ <= tf_ * 1.0000005f;                      \
    }                                                        \
    if (COUNT && enter_) ctr->entered++;                     \
    state = enter_ ? kTravelling : kArrived;                 \
  } while (0)
  // entries pushed before the ray found its current hit may lie beyond it: those are skipped unread
#define RRT_POP_OR_ARRIVE()                                  \
  do {                                                       \
    state = kArrived;                                        \
    while (sp > 0) {                                         \
      int2 e_;                                               \
      RRT_STACK_POP(e_);                                     \
      if (ANY || __int_as_float(e_.y) < tmax) {              \
        cur = e_.x;                                          \
        state = kTravelling;                                 \
        break;                                               \
      }                                                      \
    }                                                        \
  } while (0)

  for (;;) {
    // a finished ray: the job may have another one; otherwise its result waits for the flush below
    bool done = false;
    uint32_t done_block = 0;
    if (state == kArrived) {
      RRT_DBG_RAY_END();
      Ray ray;
      RayPre r;  // ka = inv * step with step a power of two: the product below is inv again, to the bit
      r.o = ro;
      r.d = rd;
      r.inv = ka * g_inv_step;
      if (job.advance(found, hit, r, &ray)) RRT_START_RAY(ray);
      else state = kJobDone;
    }
    if (state == kJobDone) {
      done = true;
      done_block = my_block;
      job.retire();
      state = kIdle;
    }
#ifdef RRT_PHASE_CLOCKS
    { clk2 = RRT_CLK(); RRT_DBG_ADD(10, clk2 - clk); }
#endif
    // refill idle lanes: only issues the loads of the new jobs -- their rays are looked at after
    // the flush, so that the flush's atomics and these loads are in flight together (further
    // passes when one ran into the end of a block or met an empty one)
    bool fresh = false, fresh_ray = false;
    Ray pending;
    pending.o = pending.d = mk(0.0f, 0.0f, 0.0f);
    pending.tmin = pending.tmax = 0.0f;
    pending.skip = -1;
    for (int pass = 0; pass < 4; ++pass) {
      const unsigned long long idle_mask = __ballot(state == kIdle && !fresh);
      if (idle_mask == 0 || exhausted) break;
      if (cursor >= end) {
        // A launch with no more blocks than waves hands them out by wave number and never touches
        // the cursor: 8192 waves finding it exhausted one atomic at a time was a 90 us floor under
        // every launch.  (Larger launches take every block from the cursor: waves that start late,
        // when others have exited, must not be left holding blocks of their own.)
        uint32_t g = n_blocks;
        if (n_blocks <= n_waves) {
          if (first_block) g = wave_id;
        } else {
          if (lane == 0) g = atomicAdd(work_cursor, 1u);
          g = __shfl(g, 0);
        }
        first_block = false;
        if (g >= n_blocks) {
          exhausted = true;
          break;
        }
        block = g;
        cursor = 0;
        end = job.open_block(g);
        if (end == 0) continue;
      }
      const uint32_t index = cursor + (uint32_t)__popcll(idle_mask & lanes_below);
      if (state == kIdle && !fresh && index < end) {
        my_block = block;
        fresh = true;
        fresh_ray = job.begin(index, &pending);
      }
      cursor = min(end, cursor + (uint32_t)__popcll(idle_mask));
    }
#ifdef RRT_PHASE_CLOCKS
    { const unsigned long long now = RRT_CLK(); RRT_DBG_ADD(11, now - clk2); clk2 = now; }
#endif
    // hand in finished jobs, one index block at a time
    unsigned long long done_mask = __ballot(done);
    while (done_mask) {
      const uint32_t b = __shfl(done_block, __ffsll((long long)done_mask) - 1);
      const bool mine = done && done_block == b;
      job.flush(mine, b);
      if (mine) done = false;
      done_mask &= ~__ballot(mine);
    }
#ifdef RRT_PHASE_CLOCKS
    { const unsigned long long now = RRT_CLK(); RRT_DBG_ADD(12, now - clk2); clk2 = now; }
#endif
    if (fresh) {
      if (fresh_ray) RRT_START_RAY(pending);
      else state = kJobDone;
    }
#ifdef RRT_PHASE_CLOCKS
    { const unsigned long long now = RRT_CLK(); RRT_DBG_ADD(0, now - clk); RRT_DBG_ADD(7, 1); clk = now; }
#endif
    if (__ballot(state == kTravelling) == 0) {
      if (__ballot(state != kIdle) == 0 && exhausted) break;
      continue;
    }

    // ---- traversal until enough lanes have arrived to make a refill worth its cost
    if (ANY && RRT_ANY_UNIFIED) {
      // One step per lane per trip: a lane at an inner node tests its four child boxes, a lane at
      // a leaf tests ONE of the leaf's triangles.  Both kinds of load are issued before either is
      // waited for, so a trip costs one memory round trip whatever the mix of lanes -- the round
      // trips, not the arithmetic, are what the kernel's time is made of.
      for (;;) {
        const bool at_node = state == kTravelling && cur >= 0;
        const bool at_leaf = state == kTravelling && cur < 0;
        uint4 w0, w1, w2;
        TriRec rec;
        int ti = 0;
        w0 = w1 = w2 = make_uint4(0u, 0u, 0u, 0u);
        rec.v0 = rec.e1 = rec.e2 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (at_node) {
          const uint4* np = reinterpret_cast<const uint4*>(nodes + cur);
          RRT_NODE_LOAD(np, w0, w1, w2);
        }
        if (at_leaf) {
          ti = (~cur) >> 2;  // leaf: ~cur = first << 2 | (count - 1)
          rec = load_tri(tris, ti);
        }
        // keeps the compiler from fusing each load with its use below, which would put the two
        // round trips one after the other
        asm volatile("" ::: "memory");
        RRT_DBG_ADD(3, 1);
        if (at_node) {
          RRT_NODE_STEP(w0, w1, w2);
        }
        if (at_leaf) {
          if (COUNT) ctr->tris++;
          float t, u, v;
          if (intersect_tri(rec, ro, rd, tmin, tmax, &t, &u, &v) && ti != skip) {
            found = true;
            tmax = t;
            hit.prim = ti;
            hit.t = t;
            hit.u = u;
            hit.v = v;
          }
          const int left = (~cur) & 3;  // triangles of this leaf still to test
          if (ANY && found) state = kArrived;
          else if (left > 0) cur = ~(((ti + 1) << 2) | (left - 1));
          else RRT_POP_OR_ARRIVE();
        }
        const unsigned long long arrived = __ballot(state == kArrived);
        if (__ballot(state == kTravelling) == 0 || __popcll(arrived) >= refill_quorum) break;
      }
#ifdef RRT_PHASE_CLOCKS
      { const unsigned long long now = RRT_CLK(); RRT_DBG_ADD(1, now - clk); clk = now; }
#endif
    } else {
      for (;;) {
        for (;;) {
          RRT_DBG_ADD(5, __popcll(__ballot(state == kTravelling && cur >= 0)));
#ifdef RRT_PHASE_CLOCKS
          { const int n_ = __popcll(__ballot(state == kTravelling && cur >= 0)); if (n_ > 0) RRT_DBG_ADD(16 + (n_ - 1) / 8, 1); }
#endif
          if (state == kTravelling && cur >= 0) {
            const uint4* np = reinterpret_cast<const uint4*>(nodes + cur);
            uint4 w0, w1, w2;
            RRT_NODE_LOAD(np, w0, w1, w2);
            RRT_NODE_STEP(w0, w1, w2);
          }
          const unsigned long long inner = __ballot(state == kTravelling && cur >= 0);
          const unsigned long long at_leaf = __ballot(state == kTravelling && cur < 0);
          RRT_DBG_ADD(3, 1);
          if (inner == 0 || RRT_LEAF_FRACTION * __popcll(at_leaf) >= __popcll(inner | at_leaf)) break;
        }
#ifdef RRT_PHASE_CLOCKS
        { const unsigned long long now = RRT_CLK(); RRT_DBG_ADD(1, now - clk); clk = now;
          RRT_DBG_ADD(4, 1); RRT_DBG_ADD(6, __popcll(__ballot(state == kTravelling && cur < 0)));
          { const int n_ = __popcll(__ballot(state == kTravelling && cur < 0)); if (n_ > 0) RRT_DBG_ADD(24 + (n_ - 1) / 8, 1); } }
#endif
        if (state == kTravelling && cur < 0) {
          // leaf: ~cur = first << 2 | (count - 1)
          const int code = ~cur;
          const int first = code >> 2, cnt = (code & 3) + 1;
#define RRT_LEAF_TEST(record, index)                                                   \
  do {                                                                                 \
    if (COUNT) ctr->tris++;                                                            \
    float t_, u_, v_;                                                                  \
    if (intersect_tri(record, ro, rd, tmin, tmax, &t_, &u_, &v_) && (index) != skip) {      \
      found = true;                                                                    \
      tmax = t_;                                                                       \
      hit.prim = (index);                                                              \
      hit.t = t_;                                                                      \
      hit.u = u_;                                                                      \
      hit.v = v_;                                                                      \
    }                                                                                  \
  } while (0)
#if RRT_LEAF_AHEAD == 2
          // Two records are requested before the first test and each test is followed by the request for the
          // record after next: a leaf waits for memory about once instead of once per triangle.  (Three ahead
          // keeps 36 registers in records, spills, and loses what two ahead gain: 84.7 against 83.3 ms per frame,
          // one ahead 84.6.)
          TriRec ra = load_tri(tris, first), rb = ra;
          if (cnt > 1) rb = load_tri(tris, first + 1);
          asm volatile("" ::: "memory");
          RRT_LEAF_TEST(ra, first);
          if (cnt > 2) ra = load_tri(tris, first + 2);
          if (cnt > 1) RRT_LEAF_TEST(rb, first + 1);
          if (cnt > 3) rb = load_tri(tris, first + 3);
          if (cnt > 2) RRT_LEAF_TEST(ra, first + 2);
          if (cnt > 3) RRT_LEAF_TEST(rb, first + 3);
#else
          // the next triangle's record is in flight while this one is tested
          TriRec rec = load_tri(tris, first);
          for (int i = 0; i < cnt; ++i) {
            const int ti = first + i;
            const TriRec cur_rec = rec;
            if (i + 1 < cnt) rec = load_tri(tris, ti + 1);
            RRT_LEAF_TEST(cur_rec, ti);
            if (ANY && found) break;
          }
#endif
#undef RRT_LEAF_TEST
          if (ANY && found) state = kArrived;
          else RRT_POP_OR_ARRIVE();
        }
#ifdef RRT_PHASE_CLOCKS
        { const unsigned long long now = RRT_CLK(); RRT_DBG_ADD(2, now - clk); clk = now; }
#endif
        const unsigned long long arrived = __ballot(state == kArrived);
        if (__ballot(state == kTravelling) == 0 || __popcll(arrived) >= refill_quorum) break;
      }
    }
  }
#ifdef RRT_PHASE_CLOCKS
  if (lane == 0) {
    dbg[8] = RRT_CLK() - clk_begin;
    for (int i = 0; i < 32; ++i)
      if (i < 13 || i >= 16) atomicAdd(&rrt_phase_dbg[(32 * Job::kDebugRow) + i], dbg[i]);
  }
#endif
#undef RRT_START_RAY
#undef RRT_POP_OR_ARRIVE
}

// The reference finds a quad only if the bvh crate's box test lets its leaf
// through: f32 slab test on the box {pos, pos + a + b} (objects/rectangle.rs:136-146).
RRT_DEV bool rect_box_pass(const DeviceRect& q, const RayPre& r, int mode) {
  if (mode == RRT_RECT_AABB_OFF) return true;
  // the ray's reciprocals; a component that is exactly zero divides to an infinity of its sign as in the original
  const float inv[3] = {r.d.x == 0.0f ? copysignf(INFINITY, r.d.x) : r.inv.x, r.d.y == 0.0f ? copysignf(INFINITY, r.d.y) : r.inv.y,
                        r.d.z == 0.0f ? copysignf(INFINITY, r.d.z) : r.inv.z};
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
// (tmax: hits at or beyond it are of no interest to the caller; most rays of a wave end here, before
// the sides are looked at)
RRT_DEV bool intersect_rect(const DeviceRect& q, const RayPre& r, float tmax, float* t_out) {
  const V3 n = mk(q.normal);
  const float denom = dot(n, r.d);
  if (fabsf(denom) < 1e-9f) return false;
  const float t = dot(mk(q.pos) - r.o, n) * __builtin_amdgcn_rcpf(denom);
  if (t < 1e-9f || !(t < tmax)) return false;
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
  const float t = dot(mk(pl.pos) - r.o, n) * __builtin_amdgcn_rcpf(denom);
  if (t < 0.0000001f) return false;
  *t_out = t;
  return true;
}

// objects/sphere.rs:116-154
RRT_DEV bool intersect_sphere(const DeviceSphere& s, const RayPre& r, float* t_out) {
  const V3 oc = r.o - mk(s.center);
  const float a = dot(r.d, r.d), b = dot(oc, r.d);
  // discriminant b^2 - a c = a (radius^2 - |l|^2), l = the part of oc across the ray: in f32 the textbook
  // form loses a grazing hit's digits in the difference of two large squares, this one does not
  const V3 l = oc - r.d * (b / a);
  const float disc = a * fmaf(s.radius, s.radius, -dot(l, l));
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

// ---------------------------------------------------------------- exact distances for the first-hit hook
// rrt_intersect reports distances to 1e-5 relative.  The f32 traversal decides WHICH object a ray hits;
// its t, from edges already rounded to f32 and a hardware reciprocal, is short of that on short rays
// (a few 1e-3 relative at t ~ 1e-3 of the scene).  The hook therefore re-evaluates the one object
// that won, once, in f64, from the data the reference itself computes from: the f32 vertices as
// uploaded (objects/triangle.rs:59-107 widens them to f64) and the f64 shape parameters.
struct ShapesF64 {
  double rect_pos[RRT_MAX_RECTS][3], rect_normal[RRT_MAX_RECTS][3];  // normal = normalize(a x b), rectangle.rs:44-46
  double plane_pos[RRT_MAX_PLANES][3], plane_normal[RRT_MAX_PLANES][3];
  double sphere_center[RRT_MAX_SPHERES][3], sphere_radius[RRT_MAX_SPHERES];
};

// Distance to a triangle by the reference's own test (objects/triangle.rs:119-181: translate to the
// ray origin, permute so that the direction's largest component is z, shear, edge functions) in f64.
// Returns a negative value when that test rejects the ray (the caller then keeps the f32 distance:
// such a ray is on an edge, where the two arithmetics may disagree about hit or miss).
RRT_DEV double exact_triangle_t(const float* __restrict__ p, V3 of, V3 df) {
  const double o[3] = {(double)of.x, (double)of.y, (double)of.z}, d[3] = {(double)df.x, (double)df.y, (double)df.z};
  int kz = 0;
  if (fabs(d[1]) > fabs(d[kz])) kz = 1;
  if (fabs(d[2]) > fabs(d[kz])) kz = 2;
  const int kx = kz == 2 ? 0 : kz + 1, ky = kx == 2 ? 0 : kx + 1;
  const double sx = -d[kx] / d[kz], sy = -d[ky] / d[kz], sz = 1.0 / d[kz];
  double x[3], y[3], z[3];
  for (int v = 0; v < 3; ++v) {
    const double tx = (double)p[3 * v + kx] - o[kx], ty = (double)p[3 * v + ky] - o[ky], tz = (double)p[3 * v + kz] - o[kz];
    x[v] = tx + sx * tz;
    y[v] = ty + sy * tz;
    z[v] = tz * sz;
  }
  const double e0 = x[1] * y[2] - y[1] * x[2], e1 = x[2] * y[0] - y[2] * x[0], e2 = x[0] * y[1] - y[0] * x[1];
  if ((e0 < 0.0 || e1 < 0.0 || e2 < 0.0) && (e0 > 0.0 || e1 > 0.0 || e2 > 0.0)) return -1.0;
  const double det = e0 + e1 + e2;
  if (det == 0.0) return -1.0;
  const double t_scaled = e0 * z[0] + e1 * z[1] + e2 * z[2];
  if (det < 0.0 && (t_scaled >= 0.0 || t_scaled < 1000.0 * det)) return -1.0;
  if (det > 0.0 && (t_scaled <= 0.0 || t_scaled > 1000.0 * det)) return -1.0;
  const double t = t_scaled * (1.0 / det);
  return t < 2.220446049250313e-16 ? -1.0 : t;
}

// Distance to the analytic shape `k` of its kind (objects/rectangle.rs:58-71, plane.rs:45-57,
// sphere.rs:116-151) in f64; negative: rejected, keep the f32 distance.
RRT_DEV double exact_plane_t(const double* pos, const double* n, V3 of, V3 df, double t_least) {
  const double denom = n[0] * (double)df.x + n[1] * (double)df.y + n[2] * (double)df.z;
  if (fabs(denom) < 1e-9) return -1.0;
  const double t = ((pos[0] - (double)of.x) * n[0] + (pos[1] - (double)of.y) * n[1] + (pos[2] - (double)of.z) * n[2]) / denom;
  return t < t_least ? -1.0 : t;
}
RRT_DEV double exact_sphere_t(const double* c, double radius, V3 of, V3 df) {
  const double oc[3] = {(double)of.x - c[0], (double)of.y - c[1], (double)of.z - c[2]};
  const double d[3] = {(double)df.x, (double)df.y, (double)df.z};
  const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], b = oc[0] * d[0] + oc[1] * d[1] + oc[2] * d[2];
  const double cc = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - radius * radius;
  const double disc = b * b - a * cc;
  if (disc < 0.0) return -1.0;
  const double sq = sqrt(disc);
  double t = (-b - sq) / a;
  if (t > 0.0001) return t;
  t = (-b + sq) / a;
  return t > 0.0001 ? t : -1.0;
}

// Analytic shapes after the triangles: closest hit below hit->t (or any below tmax).
// skip_a / skip_b: object codes that are not reported (n_tris + k), or -1.
template <bool ANY>
RRT_DEV bool intersect_others(const SceneTables* __restrict__ sc, const RayPre& r, float tmax, int skip_a, int skip_b,
                              Hit* hit) {
  // most rays end before they come near any of these shapes (a light quad above the scene): the part
  // of the ray that matters, [0, tmax], against the box around all of them first
  {
    const float t0x = (sc->others_bmin[0] - r.o.x) * r.inv.x, t1x = (sc->others_bmax[0] - r.o.x) * r.inv.x;
    const float t0y = (sc->others_bmin[1] - r.o.y) * r.inv.y, t1y = (sc->others_bmax[1] - r.o.y) * r.inv.y;
    const float t0z = (sc->others_bmin[2] - r.o.z) * r.inv.z, t1z = (sc->others_bmax[2] - r.o.z) * r.inv.z;
    const float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
    const float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    if (tn > tf * 1.00001f + 1e-6f) return false;  // (a NaN from 0 * inf compares false: the shapes are looked at)
  }
  bool found = false;
  int code = sc->n_tris;
  for (int i = 0; i < sc->n_rects; ++i, ++code) {
    float t;
    // the quad itself first: few rays pass it, and only those need the box test
    if (code != skip_a && code != skip_b && intersect_rect(sc->rects[i], r, tmax, &t) &&
        rect_box_pass(sc->rects[i], r, sc->rect_aabb_mode)) {
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
