// The wavefront path tracer: one kernel per stage of trace() (src/tracer.rs:24-139),
// run once per bounce over queues of live path slots.
//
//   raygen   get_camera_sample + Camera::generate_ray (src/sampler.rs:76-83, src/camera.rs:103-125)
//   extend   check_intersect_scene for every live path (src/renderer.rs:213-249)
//   shade    the body of the bounce loop: emitter / environment pickup, the material's lobes,
//            uniform_sample_light + estimate_direct up to (not including) their visibility
//            rays, the continuation sample, throughput, Russian roulette (src/tracer.rs:44-130,
//            141-283)
//   connect  the visibility rays estimate_direct asked for (src/renderer.rs:277-295 and the
//            closest-hit ray of src/tracer.rs:252) and the radiance clamp (src/tracer.rs:116-120)
//   film     Film::write_bucket_pixels with filter none (src/film.rs:157-167): per-pixel sums
//
// A path lives in one slot of the structure-of-arrays state for its whole life;
// queues hold slot numbers and are rebuilt every bounce with a wave ballot +
// prefix count + one atomic per wave, so that later bounces run with full waves.
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

struct PathArrays {
  float4* ray_o;      // xyz origin
  float4* ray_d;      // xyz direction, w = bit 0: previous bounce sampled a specular lobe
  float4* beta;       // path throughput (`contribution` in the reference)
  float4* radiance;   // accumulated L
  float4* hit;        // prim (int bits), t, u, v
  float4* sh_o;       // light-sample visibility ray: origin xyz, w = max distance
  float4* sh_d;       // direction xyz, w = object code never reported (int bits)
  float4* sh_c;       // contribution rgb, w = flag bits
  float4* mis_o;      // escape ray of the BSDF-sampling estimator (environment light only)
  float4* mis_d;
  float4* mis_c;
  uint32_t* queue[2];   // live slots, ping-pong per bounce
  uint32_t* sh_queue;   // slots with visibility work this bounce
  uint32_t* live_count; // [depth + 1]
  uint32_t* sh_count;   // [depth]
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

// Appends `slot` of every lane with `keep` to a queue: one atomic per wave.
RRT_DEV void queue_push(uint32_t* __restrict__ queue, uint32_t* __restrict__ counter, bool keep, uint32_t slot) {
  const unsigned long long mask = __ballot(keep);
  if (mask == 0) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  if (keep) queue[base + __popcll(mask & ((1ull << lane) - 1ull))] = slot;
}

RRT_DEV void stat_add(unsigned long long* counter, bool pred) {
  const unsigned long long mask = __ballot(pred);
  if (mask == 0) return;
  if ((threadIdx.x & 63) == __ffsll((long long)mask) - 1) atomicAdd(counter, (unsigned long long)__popcll(mask));
}

// ---------------------------------------------------------------- raygen
RRT_DEV void xform_point(const double* m, double x, double y, double z, double* o) {
  const double px = m[0] * x + m[1] * y + m[2] * z + m[3];
  const double py = m[4] * x + m[5] * y + m[6] * z + m[7];
  const double pz = m[8] * x + m[9] * y + m[10] * z + m[11];
  const double w = m[12] * x + m[13] * y + m[14] * z + m[15];
  if (w != 0.0) { o[0] = px / w; o[1] = py / w; o[2] = pz / w; }
  else { o[0] = px; o[1] = py; o[2] = pz; }
}

__global__ void __launch_bounds__(256) k_raygen(RenderParams p, CameraParams cam) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  bool live = false;
  if (slot < p.n_slots) {
    uint32_t x, y, sample;
    live = slot_pixel(p, slot, &x, &y, &sample);
    p.a.radiance[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (live) {
      const Rng4 r = rng_block(p.seed, y * p.width + x, sample, 0);
      // the camera runs in f64 like the reference; one ray per path makes it free
      const double fx = (double)x + (double)r.u[0], fy = (double)y + (double)r.u[1];
      double dir[3], org[3] = {0.0, 0.0, 0.0};
      xform_point(cam.raster_to_camera, fx, fy, 0.0, dir);
      if (cam.aperture > 0.0) {
        const double ox = (double)r.u[2] * 2.0 - 1.0, oy = (double)r.u[3] * 2.0 - 1.0;
        double lx = 0.0, ly = 0.0;
        if (!(ox == 0.0 && oy == 0.0)) {
          double theta, rad;
          if (fabs(ox) > fabs(oy)) { rad = ox; theta = 0.78539816339744830962 * (oy / ox); }
          else { rad = oy; theta = 1.57079632679489661923 - 0.78539816339744830962 * (ox / oy); }
          lx = rad * cos(theta);
          ly = rad * sin(theta);
        }
        lx *= cam.aperture;
        ly *= cam.aperture;
        const double ft = cam.focal_distance / dir[2];
        double fd[3] = {ft * dir[0] - lx, ft * dir[1] - ly, ft * dir[2]};
        const double inv = 1.0 / sqrt(fd[0] * fd[0] + fd[1] * fd[1] + fd[2] * fd[2]);
        dir[0] = fd[0] * inv; dir[1] = fd[1] * inv; dir[2] = fd[2] * inv;
        org[0] = lx; org[1] = ly;
      }
      double wo[3];
      xform_point(cam.camera_to_world, org[0], org[1], org[2], wo);
      const double* m = cam.camera_to_world;
      double wd[3] = {m[0] * dir[0] + m[1] * dir[1] + m[2] * dir[2], m[4] * dir[0] + m[5] * dir[1] + m[6] * dir[2],
                      m[8] * dir[0] + m[9] * dir[1] + m[10] * dir[2]};
      const double inv = 1.0 / sqrt(wd[0] * wd[0] + wd[1] * wd[1] + wd[2] * wd[2]);
      p.a.ray_o[slot] = make_float4((float)wo[0], (float)wo[1], (float)wo[2], 0.0f);
      p.a.ray_d[slot] = make_float4((float)(wd[0] * inv), (float)(wd[1] * inv), (float)(wd[2] * inv), __uint_as_float(0u));
      p.a.beta[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
    }
  }
  queue_push(p.a.queue[0], &p.a.live_count[0], live, slot);
  stat_add(&p.stats->paths, live);
}

// ---------------------------------------------------------------- extend
template <bool COUNT>
__global__ void __launch_bounds__(256) k_extend(RenderParams p, uint32_t bounce) {
  const uint32_t n = p.a.live_count[bounce];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t slot = p.a.queue[bounce & 1][i];
  const float4 o = p.a.ray_o[slot], d = p.a.ray_d[slot];
  TraceCounters ctr = {0, 0};
  const Hit h = closest_hit<COUNT>(p.nodes, p.tris, p.scene, mk(o), mk(d), &ctr);
  p.a.hit[slot] = make_float4(__int_as_float(h.prim), h.t, h.u, h.v);
  if (COUNT) {
    atomicAdd(&p.stats->nodes_visited, (unsigned long long)ctr.nodes);
    atomicAdd(&p.stats->tris_tested, (unsigned long long)ctr.tris);
  }
}

// ---------------------------------------------------------------- shade
__global__ void __launch_bounds__(256) k_shade(RenderParams p, uint32_t bounce) {
  const uint32_t n = p.a.live_count[bounce];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < n;
  uint32_t slot = 0;
  bool survive = false, has_work = false, shaded = false;
  uint32_t n_shadow_rays = 0;
  if (valid) {
    slot = p.a.queue[bounce & 1][i];
    const float4 o4 = p.a.ray_o[slot], d4 = p.a.ray_d[slot], h4 = p.a.hit[slot];
    const V3 org = mk(o4), dir = mk(d4);
    const bool specular_bounce = (__float_as_uint(d4.w) & 1u) != 0;
    V3 beta = mk(p.a.beta[slot]);
    const int prim = __float_as_int(h4.x);
    const SceneTables* sc = p.scene;
    if (prim < 0) {
      // src/tracer.rs:44-63: an escaping ray sees the lights' environment term on the
      // camera ray and after a specular bounce, then the path ends
      if (bounce == 0 || specular_bounce) {
        const V3 le = escape_radiance(sc, p.env, dir);
        float4 l4 = p.a.radiance[slot];
        l4.x += beta.x * le.x; l4.y += beta.y * le.y; l4.z += beta.z * le.z;
        p.a.radiance[slot] = l4;
      }
    } else {
      shaded = true;
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
            const V3 c = beta * f * ls.li * (w * n_lights / ls.pdf);
            // check_light_visible (src/renderer.rs:277-295): from the surface point towards the
            // sample, stopping just short of it.  The 1e-9 / 1e-7 margins of the f64 original do not
            // exist in f32: the surface being left and the light's own quad are excluded by identity.
            const float tmax = ls.dist < INFINITY ? ls.dist * (1.0f - 1e-5f) : INFINITY;
            p.a.sh_o[slot] = f4(s.p, tmax);
            p.a.sh_d[slot] = f4(ls.wi, __int_as_float(ls.skip));
            p.a.sh_c[slot] = f4(c, 0.0f);
            flags |= kHasLightRay;
            ++n_shadow_rays;
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
              ++n_shadow_rays;
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
      if (has_work) {
        float4 c4 = p.a.sh_c[slot];
        if (!(flags & kHasLightRay)) c4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        c4.w = __uint_as_float(flags);
        p.a.sh_c[slot] = c4;
      }
    }
  }
  queue_push(p.a.queue[(bounce + 1) & 1], &p.a.live_count[bounce + 1], survive, slot);
  queue_push(p.a.sh_queue, &p.a.sh_count[bounce], has_work, slot);
  stat_add(&p.stats->path_bounces, shaded);
  stat_add(&p.stats->rays_shadow, n_shadow_rays >= 1);
  stat_add(&p.stats->rays_shadow, n_shadow_rays >= 2);
}

// ---------------------------------------------------------------- connect
template <bool COUNT>
__global__ void __launch_bounds__(256) k_connect(RenderParams p, uint32_t bounce) {
  const uint32_t n = p.a.sh_count[bounce];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t slot = p.a.sh_queue[i];
  const float4 c4 = p.a.sh_c[slot];
  const uint32_t flags = __float_as_uint(c4.w);
  float4 l4 = p.a.radiance[slot];
  TraceCounters ctr = {0, 0};
  if (flags & kHasLightRay) {
    const float4 o = p.a.sh_o[slot], d = p.a.sh_d[slot];
    // two things may not occlude by identity: the surface the ray leaves (a triangle, quad or
    // plane cannot be hit twice by a ray that starts on it) and the sampled light's own quad
    int own = __float_as_int(p.a.hit[slot].x);
    if (own >= p.scene->n_tris + p.scene->n_rects + p.scene->n_planes) own = -1;  // spheres keep their t > 1e-4 guard
    if (!any_hit<COUNT>(p.nodes, p.tris, p.scene, mk(o), mk(d), o.w, own, __float_as_int(d.w), &ctr)) {
      l4.x += c4.x; l4.y += c4.y; l4.z += c4.z;
    }
  }
  if (flags & kHasMisRay) {
    const float4 o = p.a.mis_o[slot], d = p.a.mis_d[slot], c = p.a.mis_c[slot];
    if (!any_hit<COUNT>(p.nodes, p.tris, p.scene, mk(o), mk(d), INFINITY, -1, -1, &ctr)) {
      l4.x += c.x; l4.y += c.y; l4.z += c.z;
    }
  }
  if (flags & kClamp) {
    l4.x = fminf(l4.x, p.clamp); l4.y = fminf(l4.y, p.clamp); l4.z = fminf(l4.z, p.clamp);
  }
  p.a.radiance[slot] = l4;
  if (COUNT) {
    atomicAdd(&p.stats->nodes_visited, (unsigned long long)ctr.nodes);
    atomicAdd(&p.stats->tris_tested, (unsigned long long)ctr.tris);
  }
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

// Adds this batch's queue lengths to the running statistics.
__global__ void k_batch_stats(RenderParams p) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned long long rays = 0;
  for (uint32_t b = 0; b < p.depth_limit; ++b) rays += p.a.live_count[b];
  p.stats->rays_extend += rays;
}

// ---------------------------------------------------------------- parity hooks
__global__ void __launch_bounds__(256) k_intersect(const BvhNode* nodes, const TriIsect* tris, const SceneTables* sc,
                                                   const float* __restrict__ org_dir, uint32_t n, int32_t* prim_id,
                                                   float* t_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = org_dir + (size_t)i * 6;
  TraceCounters ctr = {0, 0};
  const Hit h = closest_hit<false>(nodes, tris, sc, mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), &ctr);
  int id = h.prim;
  if (id >= 0 && id < sc->n_tris) id = __float_as_int(tris[id].v0.w);  // leaf order -> scene order
  prim_id[i] = id;
  t_out[i] = h.prim >= 0 ? h.t : 0.0f;
}

__global__ void __launch_bounds__(256) k_occluded(const BvhNode* nodes, const TriIsect* tris, const SceneTables* sc,
                                                  const float* __restrict__ odt, uint32_t n, uint8_t* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = odt + (size_t)i * 7;
  TraceCounters ctr = {0, 0};
  out[i] = any_hit<false>(nodes, tris, sc, mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), r[6], -1, -1, &ctr) ? 1 : 0;
}

}  // namespace rrt
