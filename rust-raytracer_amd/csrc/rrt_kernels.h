// The wavefront path tracer: one kernel per stage of trace() (src/tracer.rs:24-139),
// run once per bounce over queues of live path slots.
//
//   primary  bounce 0 in one kernel: get_camera_sample + Camera::generate_ray (src/sampler.rs:76-83,
//            src/camera.rs:103-125), check_intersect_scene (src/renderer.rs:213-249) and, for rays that
//            escape, the environment pickup (src/tracer.rs:50-53).  Only paths that hit a surface get
//            their state written and enter the queues.
//   shade    the body of the bounce loop for paths that hit something: the material's lobes,
//            uniform_sample_light + estimate_direct up to (not including) their visibility rays, the
//            continuation sample, throughput, Russian roulette (src/tracer.rs:70-130, 141-283)
//   connect  the visibility rays estimate_direct asked for (src/renderer.rs:277-295 and the
//            closest-hit ray of src/tracer.rs:252) and the radiance clamp (src/tracer.rs:116-120)
//   extend   check_intersect_scene for the continuation rays of bounce >= 1; an escaping ray picks
//            up the environment if the previous bounce was specular (src/tracer.rs:44-63) and ends
//   film     Film::write_bucket_pixels with filter none (src/film.rs:157-167): per-pixel sums
//
// A path lives in one slot of the structure-of-arrays state for its whole life; queues hold slot
// numbers and are rebuilt by every kernel with a wave ballot + prefix count + one atomic per wave,
// so shade only ever sees surface hits and later bounces run with full waves.
//
// Queues are split into kSegments segments with a counter each.  One counter for a whole launch
// would take one returning atomic per wave on ONE address -- about 11 ns each, 2 million of them for
// a 33 M-path batch -- and that, not the ray tracing, set the kernel time.  Block b appends to
// segment b % kSegments; a segment can never receive more than its blocks have threads, so
// segments are fixed-size regions of one array and need no overflow handling.
#pragma once
#include "rrt_device.h"
#include "rrt_math.h"
#include "rrt_shade.h"
#include "rrt_trace.h"

namespace rrt {

struct DeviceStats {
  unsigned long long paths, rays_extend, rays_shadow, path_bounces, nodes_visited, tris_tested;
};

// Flags kept in the w lane of the shadow-contribution record.
constexpr uint32_t kHasLightRay = 1u, kHasMisRay = 2u, kClamp = 4u;

constexpr int kSegments = 64;        // queue segments = lanes of a wave (one lane scans one counter)
constexpr int kCounterStride = 32;   // words between counters: each sits in a 128-byte line of its own

struct PathArrays {
  float4* ray_o;      // xyz origin
  float4* ray_d;      // xyz direction, w = bit 0: previous bounce sampled a specular lobe
  float4* beta;       // path throughput (`contribution` in the reference)
  float4* radiance;   // accumulated L
  float4* hit;        // prim (int bits), t, u, v
  float4* sh_o;       // light-sample visibility ray: origin xyz, w = max distance
  float4* sh_d;       // direction xyz, w = object code of the sampled light's quad (int bits)
  float4* sh_c;       // contribution rgb, w = flag bits
  float4* mis_o;      // escape ray of the BSDF-sampling estimator (environment light only)
  float4* mis_d;
  float4* mis_c;
  uint32_t* q_shade;    // slots whose ray hit a surface this bounce
  uint32_t* q_extend;   // slots with a continuation ray for the next bounce
  uint32_t* q_shadow;   // slots with visibility work this bounce
  uint32_t* counters;   // rows of kSegments counters: see the row_* helpers
  uint32_t seg_cap;     // capacity of one queue segment
};

struct RenderParams {
  PathArrays a;
  const BvhNode* nodes;
  const TriIsect* tris;
  const TriShade* tri_shade;
  const DeviceMaterial* materials;
  const SceneTables* scene;
  const uint8_t* env;
  DeviceStats* stats;
  float* film;  // [height][width][4]
  uint64_t seed;
  uint32_t width, height;           // film size
  uint32_t win_x0, win_y0, win_w, win_h;  // render window
  uint32_t tiles_x;                 // 8x8 pixel tiles per window row
  uint32_t tile_begin, n_tiles;     // tiles of this batch
  uint32_t sample_begin, n_samples; // samples of this batch (K)
  uint32_t n_slots;                 // n_tiles * 64 * n_samples
  uint32_t depth_limit;
  float clamp;
};

// Counter rows of one batch: shade[b], extend[b] (b = 1..depth), shadow[b], and the number of
// environment-light escape rays mis[b].
RRT_DEV uint32_t row_shade(const RenderParams& p, uint32_t b) { return b; }
RRT_DEV uint32_t row_extend(const RenderParams& p, uint32_t b) { return p.depth_limit + b; }
RRT_DEV uint32_t row_shadow(const RenderParams& p, uint32_t b) { return 2 * p.depth_limit + 1 + b; }
RRT_DEV uint32_t row_mis(const RenderParams& p, uint32_t b) { return 3 * p.depth_limit + 1 + b; }
RRT_DEV uint32_t row_light(const RenderParams& p, uint32_t b) { return 4 * p.depth_limit + 1 + b; }
inline uint32_t counter_rows(uint32_t depth) { return 5 * depth + 1; }

RRT_DEV uint32_t* counter_at(uint32_t* counters, uint32_t row, uint32_t seg) {
  return counters + ((size_t)row * kSegments + seg) * kCounterStride;
}

// slot -> pixel, sample.  Consecutive slots are the samples of one pixel;
// consecutive pixels walk an 8x8 tile.
RRT_DEV bool slot_pixel(const RenderParams& p, uint32_t slot, uint32_t* x, uint32_t* y, uint32_t* sample) {
  const uint32_t pix = slot / p.n_samples;
  *sample = p.sample_begin + slot % p.n_samples;
  const uint32_t tile = p.tile_begin + pix / 64, in = pix % 64;
  const uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;
  const uint32_t lx = tx * 8 + (in & 7), ly = ty * 8 + (in >> 3);
  *x = p.win_x0 + lx;
  *y = p.win_y0 + ly;
  return lx < p.win_w && ly < p.win_h;
}

// Appends `slot` of every lane with `keep` to a queue: one atomic per wave, on the counter of
// segment chunk % kSegments, where `chunk` numbers the 256 work items this block is working on.
RRT_DEV void queue_push(const RenderParams& p, uint32_t* __restrict__ queue, uint32_t row, uint32_t chunk, bool keep,
                        uint32_t slot) {
  const unsigned long long mask = __ballot(keep);
  if (mask == 0) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)mask) - 1;
  const uint32_t seg = chunk % kSegments;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter_at(p.a.counters, row, seg), (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  if (keep) queue[(size_t)seg * p.a.seg_cap + base + __popcll(mask & ((1ull << lane) - 1ull))] = slot;
}

// Adds the number of lanes with `pred` to a counter row (no queue behind it).
RRT_DEV void counter_add(const RenderParams& p, uint32_t row, uint32_t chunk, bool pred) {
  const unsigned long long mask = __ballot(pred);
  if (mask == 0) return;
  if ((threadIdx.x & 63) == __ffsll((long long)mask) - 1)
    atomicAdd(counter_at(p.a.counters, row, chunk % kSegments), (uint32_t)__popcll(mask));
}

// Consumer side.  queue_scan: the block's first wave scans the row's kSegments counters into LDS
// (all threads of the block must call it: it synchronises) and returns the queue length.
// queue_item: the slot stored as item i, found by a search over the scanned counters.
RRT_DEV uint32_t queue_scan(const RenderParams& p, uint32_t row, uint32_t* prefix /* LDS, kSegments + 1 words */) {
  if (threadIdx.x < kSegments) {
    uint32_t incl = *counter_at(p.a.counters, row, threadIdx.x);
#pragma unroll
    for (int d = 1; d < kSegments; d <<= 1) {
      const uint32_t up = __shfl_up(incl, d);
      if ((int)threadIdx.x >= d) incl += up;
    }
    prefix[threadIdx.x + 1] = incl;
    if (threadIdx.x == 0) prefix[0] = 0;
  }
  __syncthreads();
  return prefix[kSegments];
}

RRT_DEV uint32_t queue_item(const RenderParams& p, const uint32_t* __restrict__ queue, const uint32_t* prefix, uint32_t i) {
  int lo = 0;  // largest s with prefix[s] <= i
#pragma unroll
  for (int step = kSegments / 2; step > 0; step >>= 1)
    if (prefix[lo + step] <= i) lo += step;
  return queue[(size_t)lo * p.a.seg_cap + (i - prefix[lo])];
}

template <bool COUNT>
RRT_DEV void flush_counters(DeviceStats* stats, const TraceCounters& ctr) {
  if (COUNT) {
    atomicAdd(&stats->nodes_visited, (unsigned long long)ctr.nodes);
    atomicAdd(&stats->tris_tested, (unsigned long long)ctr.tris);
  }
}

// ---------------------------------------------------------------- camera
// Camera::generate_ray.  Film position and the raster -> camera transform run in f64 (pixel
// coordinate + 23-bit jitter, then a difference of products ~ the image width); everything after
// is a direction of unit scale and runs in f32.
RRT_DEV void camera_ray(const CameraParams& cam, uint32_t x, uint32_t y, const Rng4& r, V3* org, V3* dir) {
  const double fx = (double)x + (double)r.u[0], fy = (double)y + (double)r.u[1];
  const double* m = cam.raster_to_camera;
  const double cx = m[0] * fx + m[1] * fy + m[3], cy = m[4] * fx + m[5] * fy + m[7], cz = m[8] * fx + m[9] * fy + m[11];
  const double cw = m[12] * fx + m[13] * fy + m[15];
  const double iw = cw != 0.0 ? 1.0 / cw : 1.0;
  V3 d = mk((float)(cx * iw), (float)(cy * iw), (float)(cz * iw));
  V3 o = mk(0.0f, 0.0f, 0.0f);
  if (cam.aperture > 0.0) {
    float lx, ly;
    concentric_sample_disk(r.u[2], r.u[3], &lx, &ly);
    lx *= (float)cam.aperture;
    ly *= (float)cam.aperture;
    const float ft = (float)cam.focal_distance / d.z;
    o = mk(lx, ly, 0.0f);
    d = normalize(d * ft - o);
  }
  const double* w = cam.camera_to_world;
  *org = mk((float)w[0] * o.x + (float)w[1] * o.y + (float)w[2] * o.z + (float)w[3],
            (float)w[4] * o.x + (float)w[5] * o.y + (float)w[6] * o.z + (float)w[7],
            (float)w[8] * o.x + (float)w[9] * o.y + (float)w[10] * o.z + (float)w[11]);
  *dir = normalize(mk((float)w[0] * d.x + (float)w[1] * d.y + (float)w[2] * d.z,
                      (float)w[4] * d.x + (float)w[5] * d.y + (float)w[6] * d.z,
                      (float)w[8] * d.x + (float)w[9] * d.y + (float)w[10] * d.z));
}

// ---------------------------------------------------------------- primary (bounce 0)
template <bool COUNT>
__global__ void __launch_bounds__(kTraceBlock) k_primary(RenderParams p, CameraParams cam) {
  RRT_DECLARE_TRACE_STACK();
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  bool hit_surface = false;
  TraceCounters ctr = {0, 0};
  if (slot < p.n_slots) {
    uint32_t x, y, sample;
    if (slot_pixel(p, slot, &x, &y, &sample)) {
      const Rng4 r = rng_block(p.seed, y * p.width + x, sample, 0);
      V3 org, dir;
      camera_ray(cam, x, y, r, &org, &dir);
      const Hit h = closest_hit<COUNT>(p.nodes, p.tris, p.scene, org, dir, &ctr, rrt_trace_stack);
      if (h.prim < 0) {
        // the camera ray escapes: it sees the lights' environment term and the path is over
        p.a.radiance[slot] = f4(escape_radiance(p.scene, p.env, dir), 0.0f);
      } else {
        hit_surface = true;
        p.a.radiance[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        p.a.ray_o[slot] = f4(org, 0.0f);
        p.a.ray_d[slot] = f4(dir, __uint_as_float(0u));
        p.a.beta[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        p.a.hit[slot] = make_float4(__int_as_float(h.prim), h.t, h.u, h.v);
      }
    }
  }
  queue_push(p, p.a.q_shade, row_shade(p, 0), blockIdx.x, hit_surface, slot);
  flush_counters<COUNT>(p.stats, ctr);
}

// ---------------------------------------------------------------- extend (bounce >= 1)
template <bool COUNT>
__global__ void __launch_bounds__(kTraceBlock) k_extend(RenderParams p, uint32_t bounce) {
  RRT_DECLARE_TRACE_STACK();
  __shared__ uint32_t prefix[kSegments + 1];
  const uint32_t n = queue_scan(p, row_extend(p, bounce), prefix);
  TraceCounters ctr = {0, 0};
  // a fixed grid walks the queue in chunks of one block's worth of rays
  for (uint32_t chunk = blockIdx.x; chunk * kTraceBlock < n; chunk += gridDim.x) {
  const uint32_t i = chunk * kTraceBlock + threadIdx.x;
  uint32_t slot = 0;
  bool hit_surface = false;
  if (i < n) {
    slot = queue_item(p, p.a.q_extend, prefix, i);
    const float4 o = p.a.ray_o[slot], d = p.a.ray_d[slot];
    const Hit h = closest_hit<COUNT>(p.nodes, p.tris, p.scene, mk(o), mk(d), &ctr, rrt_trace_stack);
    if (h.prim < 0) {
      // src/tracer.rs:44-63: an escaping ray sees the environment only after a specular bounce
      if (__float_as_uint(d.w) & 1u) {
        const V3 le = escape_radiance(p.scene, p.env, mk(d));
        const float4 b = p.a.beta[slot];
        float4 l4 = p.a.radiance[slot];
        l4.x += b.x * le.x; l4.y += b.y * le.y; l4.z += b.z * le.z;
        p.a.radiance[slot] = l4;
      }
    } else {
      hit_surface = true;
      p.a.hit[slot] = make_float4(__int_as_float(h.prim), h.t, h.u, h.v);
    }
  }
  queue_push(p, p.a.q_shade, row_shade(p, bounce), chunk, hit_surface, slot);
  }
  flush_counters<COUNT>(p.stats, ctr);
}

// ---------------------------------------------------------------- shade
__global__ void __launch_bounds__(256) k_shade(RenderParams p, uint32_t bounce) {
  __shared__ uint32_t prefix[kSegments + 1];
  const uint32_t n = queue_scan(p, row_shade(p, bounce), prefix);
  for (uint32_t chunk = blockIdx.x; chunk * 256 < n; chunk += gridDim.x) {
  const uint32_t i = chunk * 256 + threadIdx.x;
  uint32_t slot = 0;
  bool survive = false, has_work = false, has_mis = false, has_light = false;
  if (i < n) {
    slot = queue_item(p, p.a.q_shade, prefix, i);
    const float4 o4 = p.a.ray_o[slot], d4 = p.a.ray_d[slot], h4 = p.a.hit[slot];
    const V3 org = mk(o4), dir = mk(d4);
    V3 beta = mk(p.a.beta[slot]);
    const int prim = __float_as_int(h4.x);
    const SceneTables* sc = p.scene;
    // AreaLight::emitting is zero (lights/area.rs:23-25): hitting an emitter adds nothing
    const Surface s = prim < sc->n_tris ? surface_triangle(p.tris, p.tri_shade, prim, h4.z, h4.w, dir)
                                        : surface_other(sc, prim, h4.y, org, dir);
    const DeviceMaterial m = p.materials[s.material];
    uint32_t x, y, sample;
    slot_pixel(p, slot, &x, &y, &sample);
    const uint32_t pixel = y * p.width + x;
    const Rng4 ra = rng_block(p.seed, pixel, sample, 1 + 3 * bounce);

    // ---- uniform_sample_light + estimate_direct (src/tracer.rs:141-283)
    uint32_t flags = 0;
    V3 light_c = mk(0.0f, 0.0f, 0.0f);
    if (sc->n_lights > 0) {
      const int li = min((int)(ra.u[0] * (float)sc->n_lights), sc->n_lights - 1);
      const float n_lights = (float)sc->n_lights;  // division by light_pdf = 1 / n
      const LightSample ls = sample_light(sc, p.env, li, s.p, ra.u[1], ra.u[2]);
      if (ls.pdf > 1e-6f && !is_zero(ls.li)) {
        V3 f;
        float pdf_s;
        bsdf_eval(s, m, ls.wi, &f, &pdf_s);
        f = f * fabsf(dot(ls.wi, s.ns));
        if (!is_zero(f)) {
          const float w = ls.delta ? 1.0f : power_heuristic(ls.pdf, pdf_s);
          light_c = beta * f * ls.li * (w * n_lights / ls.pdf);
          // check_light_visible (src/renderer.rs:277-295): from the surface point towards the
          // sample, stopping just short of it.  The 1e-9 / 1e-7 margins of the f64 original do not
          // exist in f32: the surface being left and the light's own quad are excluded by identity.
          const float tmax = ls.dist < INFINITY ? ls.dist * (1.0f - 1e-5f) : INFINITY;
          p.a.sh_o[slot] = f4(s.p, tmax);
          p.a.sh_d[slot] = f4(ls.wi, __int_as_float(ls.skip));
          flags |= kHasLightRay;
          has_light = true;
        }
      }
      // BSDF-sampling half of the estimator (src/tracer.rs:204-280).  For an area light it can
      // only ever add AreaLight::emitting() = 0, so it is not evaluated; for the environment
      // light it adds the texel along the sampled direction if the ray escapes.
      if (sc->lights[li].kind == RRT_LIGHT_INFINITE) {
        const Rng4 rb = rng_block(p.seed, pixel, sample, 2 + 3 * bounce);
        const BsdfSample bs = bsdf_sample(s, m, rb.u[0], rb.u[1], rb.u[2]);
        const V3 f = bs.f * fabsf(dot(bs.wi, s.ns));
        if (!is_zero(f) && bs.pdf > 1e-6f) {
          const float w = bs.specular ? 1.0f : power_heuristic(bs.pdf, 1.0f);  // pdf_incidence == 1
          const V3 le = env_lookup(sc, p.env, normalize(bs.wi));
          if (!is_zero(le)) {
            const V3 c = beta * f * le * (w * n_lights / bs.pdf);
            p.a.mis_o[slot] = f4(offset_ray_origin(s.p, s.ng, bs.wi), 0.0f);
            p.a.mis_d[slot] = f4(bs.wi, 0.0f);
            p.a.mis_c[slot] = f4(c, 0.0f);
            flags |= kHasMisRay;
            has_mis = true;
          }
        }
      }
    }

    // ---- continuation (src/tracer.rs:77-130)
    const Rng4 rc = rng_block(p.seed, pixel, sample, 3 + 3 * bounce);
    const BsdfSample bs = bsdf_sample(s, m, rc.u[0], rc.u[1], rc.u[2]);
    if (!(bs.pdf == 0.0f || is_zero(bs.f))) {
      beta = beta * (bs.f * (fabsf(dot(bs.wi, s.ns)) / bs.pdf));
      if (p.clamp > 0.0f) flags |= kClamp;  // the clamp comes after this bounce's direct light is in L
      survive = bounce + 1 < p.depth_limit;
      if (bounce > 3) {
        const float q = fmaxf(1.0f - vmax(beta), 0.05f);
        if (ra.u[3] < q) survive = false;
        beta = beta / (1.0f - q);
      }
      if (survive) {
        p.a.ray_o[slot] = f4(offset_ray_origin(s.p, s.ng, bs.wi), 0.0f);
        p.a.ray_d[slot] = f4(bs.wi, __uint_as_float(bs.specular ? 1u : 0u));
        p.a.beta[slot] = f4(beta, 0.0f);
      }
    }
    has_work = (flags & (kHasLightRay | kHasMisRay)) != 0;
    // without pending light the running sum is unchanged since its last clamp: nothing to do
    if (has_work) p.a.sh_c[slot] = f4(light_c, __uint_as_float(flags));
  }
  queue_push(p, p.a.q_extend, row_extend(p, bounce + 1), chunk, survive, slot);
  queue_push(p, p.a.q_shadow, row_shadow(p, bounce), chunk, has_work, slot);
  counter_add(p, row_mis(p, bounce), chunk, has_mis);
  counter_add(p, row_light(p, bounce), chunk, has_light);
  }
}

// ---------------------------------------------------------------- connect
template <bool COUNT>
__global__ void __launch_bounds__(kTraceBlock) k_connect(RenderParams p, uint32_t bounce) {
  RRT_DECLARE_TRACE_STACK();
  __shared__ uint32_t prefix[kSegments + 1];
  const uint32_t n = queue_scan(p, row_shadow(p, bounce), prefix);
  TraceCounters ctr = {0, 0};
  for (uint32_t i = blockIdx.x * kTraceBlock + threadIdx.x; i < n; i += gridDim.x * kTraceBlock) {
  const uint32_t slot = queue_item(p, p.a.q_shadow, prefix, i);
  const float4 c4 = p.a.sh_c[slot];
  const uint32_t flags = __float_as_uint(c4.w);
  float4 l4 = p.a.radiance[slot];
  if (flags & kHasLightRay) {
    const float4 o = p.a.sh_o[slot], d = p.a.sh_d[slot];
    // two things may not occlude by identity: the surface the ray leaves (a triangle, quad or
    // plane cannot be hit twice by a ray that starts on it) and the sampled light's own quad
    int own = __float_as_int(p.a.hit[slot].x);
    if (own >= p.scene->n_tris + p.scene->n_rects + p.scene->n_planes) own = -1;  // spheres keep their t > 1e-4 guard
    if (!any_hit<COUNT>(p.nodes, p.tris, p.scene, mk(o), mk(d), o.w, own, __float_as_int(d.w), &ctr, rrt_trace_stack)) {
      l4.x += c4.x; l4.y += c4.y; l4.z += c4.z;
    }
  }
  if (flags & kHasMisRay) {
    const float4 o = p.a.mis_o[slot], d = p.a.mis_d[slot], c = p.a.mis_c[slot];
    if (!any_hit<COUNT>(p.nodes, p.tris, p.scene, mk(o), mk(d), INFINITY, -1, -1, &ctr, rrt_trace_stack)) {
      l4.x += c.x; l4.y += c.y; l4.z += c.z;
    }
  }
  if (flags & kClamp) {
    l4.x = fminf(l4.x, p.clamp); l4.y = fminf(l4.y, p.clamp); l4.z = fminf(l4.z, p.clamp);
  }
  p.a.radiance[slot] = l4;
  }
  flush_counters<COUNT>(p.stats, ctr);
}

// ---------------------------------------------------------------- film
// One thread per pixel of the batch: sums its n_samples radiances (they sit in
// consecutive slots) and adds them to the film.  A pixel belongs to one thread
// of one batch at a time, so no atomics are needed and the sum is reproducible.
__global__ void __launch_bounds__(256) k_film(RenderParams p) {
  const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= p.n_tiles * 64) return;
  uint32_t x, y, sample;
  if (!slot_pixel(p, pix * p.n_samples, &x, &y, &sample)) return;
  float r = 0.0f, g = 0.0f, b = 0.0f;
  const float4* l = p.a.radiance + (size_t)pix * p.n_samples;
  for (uint32_t s = 0; s < p.n_samples; ++s) {
    const float4 v = l[s];
    r += v.x; g += v.y; b += v.z;
  }
  float4* out = reinterpret_cast<float4*>(p.film) + ((size_t)y * p.width + x);
  float4 f = *out;
  f.x += r; f.y += g; f.z += b; f.w += (float)p.n_samples;
  *out = f;
}

// Folds this batch's queue counters into the running statistics (one wave, lane = segment).
__global__ void k_batch_stats(RenderParams p, unsigned long long paths) {
  const uint32_t seg = threadIdx.x;
  if (blockIdx.x != 0 || seg >= (uint32_t)kSegments) return;
  unsigned long long extend = 0, shadow = 0, bounces = 0, mis = 0;
  for (uint32_t b = 0; b < p.depth_limit; ++b) {
    if (b > 0) extend += *counter_at(p.a.counters, row_extend(p, b), seg);
    bounces += *counter_at(p.a.counters, row_shade(p, b), seg);
    // a shadow-queue entry carries a light-sample ray, an environment escape ray, or both
    shadow += *counter_at(p.a.counters, row_light(p, b), seg);
    mis += *counter_at(p.a.counters, row_mis(p, b), seg);
  }
  for (int d = kSegments / 2; d > 0; d >>= 1) {
    extend += __shfl_down(extend, d);
    shadow += __shfl_down(shadow, d);
    bounces += __shfl_down(bounces, d);
    mis += __shfl_down(mis, d);
  }
  if (seg == 0) {
    p.stats->paths += paths;
    p.stats->rays_extend += paths + extend;
    p.stats->path_bounces += bounces;
    p.stats->rays_shadow += shadow + mis;
  }
}

// ---------------------------------------------------------------- parity hooks
__global__ void __launch_bounds__(kTraceBlock) k_intersect(const BvhNode* nodes, const TriIsect* tris,
                                                           const SceneTables* sc, const float* __restrict__ org_dir,
                                                           uint32_t n, int32_t* prim_id, float* t_out) {
  RRT_DECLARE_TRACE_STACK();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = org_dir + (size_t)i * 6;
  TraceCounters ctr = {0, 0};
  const Hit h = closest_hit<false>(nodes, tris, sc, mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), &ctr, rrt_trace_stack);
  int id = h.prim;
  if (id >= 0 && id < sc->n_tris) id = __float_as_int(tris[id].v0.w);  // leaf order -> scene order
  prim_id[i] = id;
  t_out[i] = h.prim >= 0 ? h.t : 0.0f;
}

__global__ void __launch_bounds__(kTraceBlock) k_occluded(const BvhNode* nodes, const TriIsect* tris,
                                                          const SceneTables* sc, const float* __restrict__ odt,
                                                          uint32_t n, uint8_t* out) {
  RRT_DECLARE_TRACE_STACK();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = odt + (size_t)i * 7;
  TraceCounters ctr = {0, 0};
  out[i] = any_hit<false>(nodes, tris, sc, mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), r[6], -1, -1, &ctr,
                          rrt_trace_stack) ? 1 : 0;
}

}  // namespace rrt
