// The streaming wavefront path tracer: the stages of trace() (src/tracer.rs:24-139) as kernels
// that run once per iteration over queues of path slots.
//
//   primary  iteration 0: every slot's first camera ray, generated (src/sampler.rs:76-83,
//            src/camera.rs:103-125) and traced by one kernel
//   start    slots whose path has ended: a ray that escaped picks up the environment if it was a
//            camera ray or followed a specular bounce (src/tracer.rs:44-63); the finished radiance
//            is banked; the slot takes its next sample and generates the camera ray
//   extend   check_intersect_scene (src/renderer.rs:213-249) for camera and continuation rays.
//            Nothing else lives in this kernel: its register count decides how many rays a CU
//            keeps in flight.
//   shade    the body of the bounce loop for paths that hit something: the material's lobes,
//            uniform_sample_light + estimate_direct up to (not including) their visibility rays, the
//            continuation sample, throughput, Russian roulette (src/tracer.rs:70-130, 141-283)
//   connect  the visibility rays estimate_direct asked for (src/renderer.rs:277-295 and the
//            closest-hit ray of src/tracer.rs:252) and the radiance clamp (src/tracer.rs:116-120)
//   film     Film::write_bucket_pixels with filter none (src/film.rs:157-167): per-pixel sums
//
// A slot belongs to one pixel for the whole render and works through that pixel's samples
// k, k + K, k + 2K, ... one path after the other; when a path ends the slot starts the next one in
// the following iteration, so every launch stays as wide as the pool until the samples run out.
// The host makes the pool as large as memory allows, up to K = one slot per sample (then no slot
// ever starts a second path): the dozen last iterations of a pass hold few paths and each waits
// for its longest rays, a fixed cost that only a large pool makes small.
// The result does not depend on the schedule: a path is a function of (seed, pixel, sample), a
// slot adds its finished paths in sample order, and film adds a pixel's slots in slot order.
//
// Queues hold slot numbers and are rebuilt by every kernel with a wave ballot + lane count + one
// atomic per wave.  They are split into kSegments segments with a counter each: one counter for a
// whole launch would take one returning atomic per wave on ONE address -- about 11 ns each, 2
// million of them for a 33 M-slot launch -- and that, not the ray tracing, would set the kernel
// time.  Chunk c of a launch appends to segment c % kSegments.
#pragma once
#include <stddef.h>

#include "rrt_device.h"
#include "rrt_math.h"
#include "rrt_shade.h"
#include "rrt_trace.h"

// Launch bounds of the kernels that traverse: the second figure (waves per SIMD the register
// allocator has to make room for) trades registers per ray against rays in flight.
#ifndef RRT_EXTEND_WAVES
#define RRT_EXTEND_WAVES 5
#endif
#ifdef RRT_TRACE_WAVES
#define RRT_TRACE_BOUNDS __launch_bounds__(rrt::kTraceBlock, RRT_TRACE_WAVES)
#else
#define RRT_TRACE_BOUNDS __launch_bounds__(rrt::kTraceBlock)
#endif

// The shade kernel wants 137 registers, which leaves room for 3 waves per SIMD; asked to fit 4 it
// spills a little and runs 14 % faster (5 and 6 spill too much: measured 25, 21.5, 27, 33 ms).
#ifndef RRT_SHADE_WAVES
#define RRT_SHADE_WAVES 4
#endif
#define RRT_SHADE_BOUNDS __launch_bounds__(256, RRT_SHADE_WAVES)
// Lanes of a wave of the tail kernel that must have finished their paths before the wave takes new ones.
#ifndef RRT_TAIL_QUORUM
#define RRT_TAIL_QUORUM 32
#endif

// Diagnostic builds only (timing experiments; the results are wrong): the analytic shapes are never looked at.
#ifndef RRT_DIAG_NO_OTHERS
#define RRT_DIAG_NO_OTHERS 0
#endif

namespace rrt {

struct DeviceStats {
  unsigned long long paths, rays_extend, rays_shadow, path_bounces, rays_primary;
  unsigned long long nodes[3], tris[3];  // per traversal kernel: primary, extend, connect
  unsigned long long entered[3];         // rays that went into the hierarchy, likewise
};

// Flags kept in the w lane of the shadow-contribution record.
constexpr uint32_t kHasLightRay = 1u, kHasMisRay = 2u, kClamp = 4u;

constexpr int kSegments = 64;        // queue segments = lanes of a wave (one lane scans one counter)
constexpr int kCounterStride = 32;   // words between counters: each sits in a 128-byte line of its own
// Flag of a start-queue entry (slot numbers stay below 2^30).
constexpr uint32_t kEscaped = 0x40000000u;  // the path ended because its ray left the scene
constexpr uint32_t kSlotMask = 0x3fffffffu;

// Counter rows.  The queues written during iteration i for iteration i + 1 use parity (i + 1) & 1;
// rows [2 * parity, 2 * parity + 7) are exactly the ones an iteration starts with at zero: all but the
// two of the visibility stage are cleared before iteration i starts, those two (kRowShadow,
// kRowWorkConnect) just before its shade -- the visibility rays of iteration i - 1 may still be running
// beside the start and extend of iteration i.
// A work row holds one word: the cursor from which a traversal kernel's waves take their jobs.
enum : uint32_t {
  kRowStart0 = 0, kRowCont0 = 1, kRowCamera = 2, kRowShade = 3, kRowWorkExtend = 4, kRowShadow = 5, kRowWorkConnect = 6,
  kRowStart1 = 7, kRowCont1 = 8,
  kRowsPerIteration = 7,
  // running totals over the whole pass (never reset between iterations)
  kRowPaths = 9, kRowContRays = 10, kRowBounces = 11, kRowLightRays = 12, kRowMisRays = 13,
  kCounterRows = 14
};
RRT_DEV uint32_t row_start(uint32_t parity) { return parity ? kRowStart1 : kRowStart0; }
RRT_DEV uint32_t row_cont(uint32_t parity) { return parity ? kRowCont1 : kRowCont0; }

// Per-slot state that one kernel touches together sits in one 64-byte record: the slots a launch
// works on are scattered over the pool, and with one array per field every 16-byte field was a
// memory transaction of its own (shade moved ~1 KB per event for ~350 B of payload).
struct alignas(64) TraceRecord {
  float4 ray_o;  // xyz origin of the ray that produced `hit` (written by extend when the ray hits)
  float4 ray_d;  // xyz direction of the slot's last ray, w = bounce << 1 | previous bounce sampled a specular lobe
  float4 hit;    // prim (int bits), t, u, v
  float4 beta;   // path throughput (`contribution` in the reference); 1 at bounce 0 and not stored
};
struct alignas(64) ShadowRecord {
  float4 sh_o;   // light-sample visibility ray: origin xyz, w = max distance
  float4 sh_d;   // direction xyz, w = object code of the sampled light's quad (int bits)
  float4 sh_c;   // contribution rgb, w = flag bits | object the rays leave << 3
  float4 pad;
};
struct alignas(64) MisRecord {
  float4 mis_o, mis_d, mis_c, pad;
};

struct PathArrays {
  TraceRecord* trace;    // per slot: the ray being traced, what it hit, the path's throughput
  ShadowRecord* shadow;  // per slot: the light-sample visibility ray shade hands to connect
  MisRecord* mis;        // per slot: escape ray of the BSDF-sampling estimator (environment light only)
  uint32_t* sample;   // sample index of the slot's current path
  float4* radiance;   // L of the current path; written by shade at bounce 0
  float4* accum;      // sum of the slot's finished paths, w = their number
  uint32_t* q_start[2];  // slots whose path has ended (or not begun), with entry flags
  // Ray queues carry the rays themselves, two float4 per entry: (origin, slot bits), (direction,
  // state bits as in ray_d).  A traversing lane that takes a new job then needs one memory round
  // trip, over consecutive addresses, instead of entry -> slot -> ray.
  float4* rays_cont[2];  // continuation rays
  float4* rays_camera;   // new camera rays
  uint32_t ray_seg_cap;  // capacity of one segment of a ray queue
  uint32_t* q_shade;     // slots whose ray hit a surface this iteration
  uint32_t* q_shadow;    // slots with visibility work this iteration
  uint32_t* counters;    // kCounterRows rows of kSegments counters
  uint32_t seg_cap;      // capacity of one queue segment
};

// Division of x < 2^31 by a run-time constant d >= 1 without the 20-instruction division sequence
// (Granlund & Montgomery): with l = ceil(log2 d) and m = floor(2^32 (2^l - d) / d) + 1,
// floor(x / d) = (mulhi(m, x) + x) >> l.  Slot, pixel and tile numbers stay below 2^30.
struct FastDiv {
  uint32_t m, shift, d;
};
inline FastDiv make_fast_div(uint32_t d) {
  FastDiv f;
  f.d = d;
  f.shift = 0;
  while ((1ull << f.shift) < d) ++f.shift;
  f.m = (uint32_t)((((1ull << f.shift) - d) << 32) / d + 1ull);
  return f;
}
RRT_DEV uint32_t fast_div(uint32_t x, const FastDiv& f) { return (__umulhi(f.m, x) + x) >> f.shift; }

struct RenderParams {
  PathArrays a;
  const BvhNode* nodes;
  BvhGrid grid;
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
  FastDiv div_tiles_x, div_samples; // divisions by tiles_x and n_samples
  float* aov;                       // first-hit buffers [2][height][width][4] or null
  uint32_t tile_begin, n_tiles;     // tiles of this pass
  uint32_t sample_begin, sample_end;
  uint32_t n_samples;               // K: slots per pixel
  uint32_t n_slots;                 // n_tiles * 64 * K
  uint32_t depth_limit;
  float clamp;
  const float* filter_table;  // 16 x 16 weights, or null: every sample lands in its own pixel with weight 1
  double filter_radius;
  // One sample per slot and no reconstruction filter: a slot's radiance is its pixel's sample, so
  // nothing is banked -- film reads `radiance` directly, a path that ends just stops, and the
  // start queue only sees rays that escaped after a specular bounce (they still owe the environment).
  uint32_t single;
  // Something is to be had from a ray that leaves the scene (an environment map or a distant light,
  // src/tracer.rs:44-63): otherwise, with one sample per slot, such a ray just ends its path.
  uint32_t escape_lit;
};

RRT_DEV uint32_t* counter_at(uint32_t* counters, uint32_t row, uint32_t seg) {
  return counters + ((size_t)row * kSegments + seg) * kCounterStride;
}

// slot -> pixel and the slot's lane k among the pixel's K slots.  Consecutive slots share a pixel;
// consecutive pixels walk an 8x8 tile.
RRT_DEV bool slot_pixel(const RenderParams& p, uint32_t slot, uint32_t* x, uint32_t* y, uint32_t* k) {
  const uint32_t pix = fast_div(slot, p.div_samples);
  *k = slot - pix * p.n_samples;
  const uint32_t tile = p.tile_begin + pix / 64, in = pix % 64;
  const uint32_t ty = fast_div(tile, p.div_tiles_x), tx = tile - ty * p.tiles_x;
  const uint32_t lx = tx * 8 + (in & 7), ly = ty * 8 + (in >> 3);
  *x = p.win_x0 + lx;
  *y = p.win_y0 + ly;
  return lx < p.win_w && ly < p.win_h;
}

// Reserves a queue position for every lane with `keep`: one atomic per wave, on the counter of
// segment chunk % kSegments, where `chunk` numbers the block of work items the lanes are working
// on (the same for all lanes that push together).  Returns the position (segment * cap + index in
// the segment); lanes without `keep` get garbage.  A segment can overflow only if the chunks that
// map to it produce more than `cap` entries; callers size cap from their chunk sizes.
RRT_DEV uint32_t queue_reserve(const RenderParams& p, uint32_t row, uint32_t chunk, bool keep, uint32_t cap) {
  const unsigned long long mask = __ballot(keep);
  if (mask == 0) return 0;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)mask) - 1;
  const uint32_t seg = chunk % kSegments;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter_at(p.a.counters, row, seg), (uint32_t)__popcll(mask));
  base = __shfl(base, leader);
  return seg * cap + base + __popcll(mask & ((1ull << lane) - 1ull));
}
RRT_DEV void queue_push(const RenderParams& p, uint32_t* __restrict__ queue, uint32_t row, uint32_t chunk, bool keep,
                        uint32_t entry) {
  const uint32_t at = queue_reserve(p, row, chunk, keep, p.a.seg_cap);
  if (keep) queue[at] = entry;
}
RRT_DEV void ray_queue_push(const RenderParams& p, float4* __restrict__ queue, uint32_t row, uint32_t chunk, bool keep,
                            uint32_t slot, V3 o, V3 d, uint32_t state) {
  const uint32_t at = queue_reserve(p, row, chunk, keep, p.a.ray_seg_cap);
  if (keep) {
    queue[(size_t)at * 2] = f4(o, __uint_as_float(slot));
    queue[(size_t)at * 2 + 1] = f4(d, __uint_as_float(state));
  }
}

// Adds the number of lanes with `pred` to a counter row (no queue behind it).
RRT_DEV void counter_add(const RenderParams& p, uint32_t row, uint32_t chunk, bool pred) {
  const unsigned long long mask = __ballot(pred);
  if (mask == 0) return;
  if ((threadIdx.x & 63) == __ffsll((long long)mask) - 1)
    atomicAdd(counter_at(p.a.counters, row, chunk % kSegments), (uint32_t)__popcll(mask));
}

// Consumer side.  Work is handed out in blocks of G consecutive entries of ONE segment, block j
// covering entries [j / kSegments * G, +G) of segment j % kSegments.  Walking j upwards therefore
// visits all segments side by side, which is the order the producer filled them in -- and that
// order follows the image (slots are laid out tile by tile).  Rays that are in flight together
// then start close together and share the nodes they fetch; reading segment after segment
// instead would mix rays from all over the image and halve the L2 hit rate.
// queue_lengths: the block's first wave loads the row's kSegments counters into LDS (all threads
// of the block must call it: it synchronises) and returns the longest.
RRT_DEV uint32_t queue_lengths(const RenderParams& p, uint32_t row, uint32_t* lens /* LDS, kSegments + 1 words */) {
  if (threadIdx.x < kSegments) {
    const uint32_t n = *counter_at(p.a.counters, row, threadIdx.x);
    lens[threadIdx.x] = n;
    uint32_t m = n;
#pragma unroll
    for (int d = kSegments / 2; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d));
    if (threadIdx.x == 0) lens[kSegments] = m;
  }
  __syncthreads();
  return lens[kSegments];
}
// Number of blocks of G entries that cover a queue whose longest segment has `longest` entries.
RRT_DEV uint32_t queue_blocks(uint32_t longest, uint32_t g) { return kSegments * ((longest + g - 1) / g); }
// First entry and entry count of block j (count 0: the segment is shorter).
RRT_DEV const uint32_t* queue_block(const RenderParams& p, const uint32_t* __restrict__ queue, const uint32_t* lens, uint32_t j,
                                    uint32_t g, uint32_t* count) {
  const uint32_t seg = j % kSegments, first = j / kSegments * g, len = lens[seg];
  *count = first < len ? min(g, len - first) : 0u;
  return queue + (size_t)seg * p.a.seg_cap + first;
}

// The analytic shapes, copied to LDS once per block.  A traversing lane looks at them every time
// a ray arrives, and through the global pointer every field read there is a memory round trip of
// its own (a dozen per arrival).  The copy is a SceneTables cut off before `lights`: the
// traversal kernels read nothing beyond.
constexpr uint32_t kShapeWords = (uint32_t)(offsetof(SceneTables, lights) + 3) / 4;
#define RRT_DECLARE_SHAPES() __shared__ uint32_t rrt_shape_words[rrt::kShapeWords]
RRT_DEV const SceneTables* shapes_to_lds(const SceneTables* __restrict__ scene, uint32_t* words) {
  const uint32_t* src = reinterpret_cast<const uint32_t*>(scene);
  for (uint32_t i = threadIdx.x; i < kShapeWords; i += blockDim.x) words[i] = src[i];
  __syncthreads();
  return reinterpret_cast<const SceneTables*>(words);
}

template <bool COUNT>
RRT_DEV void flush_counters(DeviceStats* stats, int kernel, const TraceCounters& ctr) {
  if (COUNT) {
    atomicAdd(&stats->nodes[kernel], (unsigned long long)ctr.nodes);
    atomicAdd(&stats->tris[kernel], (unsigned long long)ctr.tris);
    atomicAdd(&stats->entered[kernel], (unsigned long long)ctr.entered);
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
  double iw = cam.const_inv_w;  // the same 1 / w for every pixel of a projective camera: no division per ray
  if (iw == 0.0) {
    const double cw = m[12] * fx + m[13] * fy + m[15];
    iw = cw != 0.0 ? 1.0 / cw : 1.0;
  }
  V3 d = mk((float)(cx * iw), (float)(cy * iw), (float)(cz * iw));
  V3 o = mk(0.0f, 0.0f, 0.0f);
  if (cam.aperture > 0.0) {
    float lx, ly;
    concentric_sample_disk(r.u[2], r.u[3], &lx, &ly);
    lx *= cam.aperture_f;
    ly *= cam.aperture_f;
    const float ft = cam.focal_distance_f / d.z;
    o = mk(lx, ly, 0.0f);
    d = normalize(d * ft - o);
  }
  const float* w = cam.camera_to_world_f;  // rows of the 3 x 4 part
  *org = mk(w[0] * o.x + w[1] * o.y + w[2] * o.z + w[3],
            w[4] * o.x + w[5] * o.y + w[6] * o.z + w[7],
            w[8] * o.x + w[9] * o.y + w[10] * o.z + w[11]);
  *dir = normalize(mk(w[0] * d.x + w[1] * d.y + w[2] * d.z,
                      w[4] * d.x + w[5] * d.y + w[6] * d.z,
                      w[8] * d.x + w[9] * d.y + w[10] * d.z));
}

RRT_DEV const float4* ray_queue_block(const RenderParams& p, const float4* __restrict__ queue, const uint32_t* lens, uint32_t j,
                                      uint32_t g, uint32_t* count) {
  const uint32_t seg = j % kSegments, first = j / kSegments * g, len = lens[seg];
  *count = first < len ? min(g, len - first) : 0u;
  return queue + ((size_t)seg * p.a.ray_seg_cap + first) * 2;
}

// Film::write_bucket_pixels with a reconstruction filter (src/film.rs:169-215): the sample at film
// position (fx, fy) adds weight * L to every pixel within the radius.  Pixels are shared between
// slots here, so the sums are atomic (their order, and with it the last bits of the film, varies
// from run to run).  The pixel range and table index are computed in f64 like the original: the
// film position has more significant bits than an f32 holds.
RRT_DEV void splat_filtered(const RenderParams& p, double fx, double fy, float lr, float lg, float lb) {
  const double r = p.filter_radius;
  const int x_min = (int)ceil(fx - r), y_min = (int)ceil(fy - r);
  const int x_max = (int)floor(fx + r), y_max = (int)floor(fy + r);
  const double scale = (1.0 / r) * 16.0;
  for (int y = y_min; y <= y_max; ++y) {
    if (y < 0 || y >= (int)p.height) continue;
    const int iy = (int)fmin(floor(fabs(((double)y - fy) * scale)), 15.0);
    for (int x = x_min; x <= x_max; ++x) {
      if (x < 0 || x >= (int)p.width) continue;
      const int ix = (int)fmin(floor(fabs(((double)x - fx) * scale)), 15.0);
      const float w = p.filter_table[iy * 16 + ix];
      float* out = p.film + 4 * ((size_t)y * p.width + x);
      unsafeAtomicAdd(out + 0, lr * w);
      unsafeAtomicAdd(out + 1, lg * w);
      unsafeAtomicAdd(out + 2, lb * w);
      unsafeAtomicAdd(out + 3, w);
    }
  }
}

// ---------------------------------------------------------------- start
__global__ void __launch_bounds__(256) k_start(RenderParams p, CameraParams cam, uint32_t parity) {
  __shared__ uint32_t lens[kSegments + 1];
  const uint32_t chunks = queue_blocks(queue_lengths(p, row_start(parity), lens), 256);
  for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
    uint32_t n;
    const uint32_t* entries = queue_block(p, p.a.q_start[parity], lens, chunk, 256, &n);
    const uint32_t i = threadIdx.x;
    uint32_t slot = 0;
    bool started = false;
    V3 org = mk(0.0f, 0.0f, 0.0f), dir = org;
    if (i < n) {
      const uint32_t entry = entries[i];
      slot = entry & kSlotMask;
      uint32_t x, y, k;
      slot_pixel(p, slot, &x, &y, &k);
      if (p.single) {
        // only rays that escaped after a specular bounce come here (src/tracer.rs:44-63): they
        // still owe the environment; nothing is banked and no slot starts a second path
        const float4 d = p.a.trace[slot].ray_d;
        const V3 le = escape_radiance(p.scene, p.env, mk(d));
        const float4 b = p.a.trace[slot].beta;
        float4 l4 = p.a.radiance[slot];
        l4.x += b.x * le.x; l4.y += b.y * le.y; l4.z += b.z * le.z;
        p.a.radiance[slot] = l4;
      } else {
        float4 l4;
        if (entry & kEscaped) {
          const float4 d = p.a.trace[slot].ray_d;
          const uint32_t state = __float_as_uint(d.w);
          if (state == 0u) {
            // a camera ray that escapes sees the lights' environment term (src/tracer.rs:50-53)
            l4 = f4(escape_radiance(p.scene, p.env, mk(d)), 0.0f);
          } else {
            l4 = p.a.radiance[slot];
            if (state & 1u) {  // later rays only after a specular bounce (src/tracer.rs:44-63)
              const V3 le = escape_radiance(p.scene, p.env, mk(d));
              const float4 b = p.a.trace[slot].beta;
              l4.x += b.x * le.x; l4.y += b.y * le.y; l4.z += b.z * le.z;
            }
          }
        } else {
          l4 = p.a.radiance[slot];
        }
        const uint32_t finished = p.a.sample[slot];
        if (p.filter_table) {
          // the film position is the camera sample's (src/sampler.rs:76-83): pixel + draws 0, 1 of block 0
          const Rng4 r = rng_block(p.seed, y * p.width + x, finished, 0);
          splat_filtered(p, (double)x + (double)r.u[0], (double)y + (double)r.u[1], l4.x, l4.y, l4.z);
        } else {
          // the slot's first finished path initialises its sum, so the array needs no clearing pass
          float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          if (finished != p.sample_begin + k) acc = p.a.accum[slot];
          acc.x += l4.x; acc.y += l4.y; acc.z += l4.z; acc.w += 1.0f;
          p.a.accum[slot] = acc;
        }
        const uint32_t sample = finished + p.n_samples;
        if (sample < p.sample_end) {
          const Rng4 r = rng_block(p.seed, y * p.width + x, sample, 0);
          camera_ray(cam, x, y, r, &org, &dir);
          p.a.sample[slot] = sample;
          started = true;
        }
      }
    }
    ray_queue_push(p, p.a.rays_camera, kRowCamera, chunk, started, slot, org, dir, 0u);
    counter_add(p, kRowPaths, chunk, started);
  }
}

// ---------------------------------------------------------------- extend
// Block size in which a traversal kernel's waves take jobs from the global cursor: large enough
// that the cursor sees few atomics, small enough that every wave comes back several times.
#ifndef RRT_MAX_GRAB
#define RRT_MAX_GRAB 1024
#endif
RRT_DEV uint32_t grab_size(uint32_t longest, uint32_t waves) {
  const uint32_t g = (longest * kSegments / (4u * waves) + 63u) & ~63u;
  return min(max(g, 64u), (uint32_t)RRT_MAX_GRAB);
}

// Continuation rays of `parity` first (they take longest), camera rays behind them.
template <bool COUNT>
struct ExtendJob {
  static constexpr int kDebugRow = 0;  // row of the diagnostic build's counters
  static constexpr bool kRootTest = false;  // its rays start on surfaces
  const RenderParams& p;
  const SceneTables* sc;  // LDS copy of the shapes
  const uint32_t* lens_c;
  const uint32_t* lens_s;
  uint32_t blocks_c, grab, parity;
  const float4* entries;  // of the open block
  uint32_t slot, state;
  bool hit_surface;
  uint32_t out_slot;   // result of the retired job, until the flush
  bool out_hit, out_owes;  // owes: the escaped ray still has something to collect in start

  RRT_DEV uint32_t open_block(uint32_t j) {
    uint32_t n;
    entries = j < blocks_c ? ray_queue_block(p, p.a.rays_cont[parity], lens_c, j, grab, &n)
                           : ray_queue_block(p, p.a.rays_camera, lens_s, j - blocks_c, grab, &n);
    return n;
  }
  RRT_DEV bool begin(uint32_t i, Ray* ray) {
    const float4 o = entries[2 * i], d = entries[2 * i + 1];
    slot = __float_as_uint(o.w);
    state = __float_as_uint(d.w);
    ray->o = mk(o);
    ray->d = mk(d);
    ray->tmin = 0.0f;
    ray->tmax = kTriTMax;
    ray->skip = -1;
    return true;
  }
  RRT_DEV bool advance(bool found, const Hit& tri_hit, const RayPre& r, Ray*) {
    Hit h = tri_hit;
    if (!RRT_DIAG_NO_OTHERS && sc->n_rects + sc->n_planes + sc->n_spheres > 0) intersect_others<false>(sc, r, h.t, -1, -1, &h);
    hit_surface = h.prim >= 0;
    // shade needs the whole ray, start (for a ray that escaped and still has something to collect) its
    // direction and state
    if (hit_surface || !p.single || ((state & 1u) && p.escape_lit)) p.a.trace[slot].ray_d = f4(r.d, __uint_as_float(state));
    if (hit_surface) {
      if (h.prim >= sc->n_tris) p.a.trace[slot].ray_o = f4(r.o, 0.0f);  // only the analytic shapes' shading reads it
      p.a.trace[slot].hit = make_float4(__int_as_float(h.prim), h.t, h.u, h.v);
    }
    return false;
  }
  RRT_DEV void retire() {
    out_slot = slot;
    out_hit = hit_surface;
    out_owes = !p.single || ((state & 1u) && p.escape_lit);
  }
  // Two queues, two atomics: both issued before either result is used.
  RRT_DEV void flush(bool done, uint32_t block) {
    const bool to_shade = done && out_hit, to_start = done && !out_hit && out_owes;
    const unsigned long long mask_h = __ballot(to_shade), mask_m = __ballot(to_start);
    const int lane = threadIdx.x & 63;
    const int leader_h = mask_h ? __ffsll((long long)mask_h) - 1 : 0, leader_m = mask_m ? __ffsll((long long)mask_m) - 1 : 0;
    const uint32_t seg = block % kSegments, other = parity ^ 1u;
    uint32_t base_h = 0, base_m = 0;
    if (mask_h && lane == leader_h) base_h = atomicAdd(counter_at(p.a.counters, kRowShade, seg), (uint32_t)__popcll(mask_h));
    if (mask_m && lane == leader_m) base_m = atomicAdd(counter_at(p.a.counters, row_start(other), seg), (uint32_t)__popcll(mask_m));
    base_h = __shfl(base_h, leader_h);
    base_m = __shfl(base_m, leader_m);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (to_shade) p.a.q_shade[seg * p.a.seg_cap + base_h + __popcll(mask_h & below)] = out_slot;
    if (to_start) p.a.q_start[other][seg * p.a.seg_cap + base_m + __popcll(mask_m & below)] = out_slot | kEscaped;
    counter_add(p, kRowContRays, block, done && block < blocks_c);
  }
};

// (Bounded to 5 waves per SIMD = 96 registers, which is what it uses: a guard against creeping over.)
template <bool COUNT>
__global__ void __launch_bounds__(kTraceBlock, RRT_EXTEND_WAVES) k_extend(RenderParams p, uint32_t parity) {
  RRT_DECLARE_TRACE_STACK();
  RRT_DECLARE_SHAPES();
  const SceneTables* sc = shapes_to_lds(p.scene, rrt_shape_words);
  __shared__ uint32_t lens_c[kSegments + 1], lens_s[kSegments + 1];
  const uint32_t longest_c = queue_lengths(p, row_cont(parity), lens_c);
  const uint32_t longest_s = queue_lengths(p, kRowCamera, lens_s);
  const uint32_t grab = grab_size(longest_c + longest_s, gridDim.x * (kTraceBlock / 64));
  const uint32_t blocks_c = queue_blocks(longest_c, grab);
  TraceCounters ctr = {0, 0, 0};
  ExtendJob<COUNT> job = {p, sc, lens_c, lens_s, blocks_c, grab, parity, nullptr, 0, 0, false, 0, false, false};
  trace_stream<false, COUNT>(p.nodes, p.grid, p.tris, sc->n_tris > 0, counter_at(p.a.counters, kRowWorkExtend, 0),
                             blocks_c + queue_blocks(longest_s, grab), RRT_REFILL_QUORUM, job, &ctr, rrt_trace_stack);
  flush_counters<COUNT>(p.stats, 1, ctr);
}

// ---------------------------------------------------------------- primary
// Iteration 0 in one kernel: every slot generates its first camera ray (as k_start would) and
// traces it at once (as k_extend would), without the ray going through a queue in memory.  When
// every slot has a single sample (`single`) a camera ray that escapes ends its slot right here:
// three quarters of the bench scene's paths then cost one 16-byte store.
template <bool COUNT>
struct PrimaryJob {
  static constexpr int kDebugRow = 2;  // row of the diagnostic build's counters
  static constexpr bool kRootTest = true;  // its rays may start anywhere
  const RenderParams& p;
  const CameraParams& cam;
  const SceneTables* sc;  // LDS copy of the shapes
  uint32_t grab;
  bool single;            // RenderParams::single
  uint32_t base, slot;
  int kind;               // 0 nothing to hand in, 1 to shade, 2 to start (escaped)
  uint32_t out_slot;      // result of the retired job, until the flush
  int out_kind;
  bool out_started;

  RRT_DEV uint32_t open_block(uint32_t j) {
    base = j * grab;
    return min(grab, p.n_slots - base);
  }
  RRT_DEV bool begin(uint32_t i, Ray* ray) {
    slot = base + i;
    kind = -1;  // not a path: a slot of a tile's padding
    uint32_t x, y, k;
    if (!slot_pixel(p, slot, &x, &y, &k)) return false;
    kind = 0;
    const uint32_t sample = p.sample_begin + k;
    const Rng4 r = rng_block(p.seed, y * p.width + x, sample, 0);
    camera_ray(cam, x, y, r, &ray->o, &ray->d);
    ray->tmin = 0.0f;
    ray->tmax = kTriTMax;
    ray->skip = -1;
    if (!single) p.a.sample[slot] = sample;  // (one sample per slot: shade derives it from the slot number)
    return true;
  }
  RRT_DEV bool advance(bool found, const Hit& tri_hit, const RayPre& r, Ray*) {
    Hit h = tri_hit;
    if (!RRT_DIAG_NO_OTHERS && sc->n_rects + sc->n_planes + sc->n_spheres > 0) intersect_others<false>(sc, r, h.t, -1, -1, &h);
    if (h.prim >= 0) {
      if (h.prim >= sc->n_tris) p.a.trace[slot].ray_o = f4(r.o, 0.0f);  // only the analytic shapes' shading reads it
      p.a.trace[slot].ray_d = f4(r.d, __uint_as_float(0u));
      p.a.trace[slot].hit = make_float4(__int_as_float(h.prim), h.t, h.u, h.v);
      kind = 1;
    } else if (single) {
      // a camera ray that escapes sees the lights' environment term (src/tracer.rs:50-53): the
      // slot's only path is complete
      p.a.radiance[slot] = f4(escape_radiance(sc, p.env, r.d), 0.0f);  // (the fields it reads lie in the LDS copy)
      kind = 0;
    } else {
      p.a.trace[slot].ray_d = f4(r.d, __uint_as_float(0u));
      kind = 2;
    }
    return false;
  }
  RRT_DEV void retire() {
    out_slot = slot;
    out_kind = kind < 0 ? 0 : kind;
    out_started = kind >= 0;
  }
  RRT_DEV void flush(bool done, uint32_t block) {
    const bool to_shade = done && out_kind == 1, to_start = done && out_kind == 2;
    const unsigned long long mask_h = __ballot(to_shade), mask_m = __ballot(to_start);
    const int lane = threadIdx.x & 63;
    const int leader_h = mask_h ? __ffsll((long long)mask_h) - 1 : 0, leader_m = mask_m ? __ffsll((long long)mask_m) - 1 : 0;
    const uint32_t seg = block % kSegments;
    uint32_t base_h = 0, base_m = 0;
    if (mask_h && lane == leader_h) base_h = atomicAdd(counter_at(p.a.counters, kRowShade, seg), (uint32_t)__popcll(mask_h));
    if (mask_m && lane == leader_m) base_m = atomicAdd(counter_at(p.a.counters, row_start(1), seg), (uint32_t)__popcll(mask_m));
    base_h = __shfl(base_h, leader_h);
    base_m = __shfl(base_m, leader_m);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (to_shade) p.a.q_shade[seg * p.a.seg_cap + base_h + __popcll(mask_h & below)] = out_slot;
    if (to_start) p.a.q_start[1][seg * p.a.seg_cap + base_m + __popcll(mask_m & below)] = out_slot | kEscaped;
    counter_add(p, kRowPaths, block, done && out_started);
  }
};

// (Bounded to 5 waves per SIMD = 96 registers.  Compiled without SLP vectorisation the kernel wants 100; held to
// 96 it spills a few scalar registers and gains a fifth wave: 16.0 -> 14.9 ms on the bench frame.)
#ifndef RRT_PRIMARY_WAVES
#define RRT_PRIMARY_WAVES 5
#endif
template <bool COUNT>
__global__ void __launch_bounds__(kTraceBlock, RRT_PRIMARY_WAVES) k_primary(RenderParams p, CameraParams cam) {
  RRT_DECLARE_TRACE_STACK();
  RRT_DECLARE_SHAPES();
  const SceneTables* sc = shapes_to_lds(p.scene, rrt_shape_words);
  const uint32_t grab = grab_size((p.n_slots + kSegments - 1) / kSegments, gridDim.x * (kTraceBlock / 64));
  TraceCounters ctr = {0, 0, 0};
  PrimaryJob<COUNT> job = {p, cam, sc, grab, p.single != 0, 0, 0, 0, 0, 0, false};
  trace_stream<false, COUNT>(p.nodes, p.grid, p.tris, sc->n_tris > 0, counter_at(p.a.counters, kRowWorkExtend, 0),
                             (p.n_slots + grab - 1) / grab, RRT_REFILL_QUORUM_COHERENT, job, &ctr, rrt_trace_stack);
  flush_counters<COUNT>(p.stats, 0, ctr);
}

// ---------------------------------------------------------------- shade
// What one bounce of trace() (src/tracer.rs:70-130) leaves to be done: the visibility rays of
// estimate_direct with the radiance each would add, and the continuation ray.
struct ShadeOut {
  uint32_t flags;        // kHasLightRay | kHasMisRay | kClamp
  V3 light_c;            // contribution of the light sample if its visibility ray gets through
  V3 sh_o, sh_d;         // that ray
  float sh_tmax;
  int sh_skip;           // object code of the sampled light's own quad, or -1
  V3 mis_o, mis_d, mis_c;  // escape ray of the BSDF-sampling estimator (environment light) and its contribution
  bool survive;          // the path goes on with (next_o, next_d, next_state) and throughput beta
  V3 next_o, next_d;
  uint32_t next_state;
  V3 beta;
};

// The body of the bounce loop for a path that hit surface `s` (object `prim`): uniform_sample_light +
// estimate_direct up to their rays, the continuation sample, throughput, Russian roulette
// (src/tracer.rs:70-130, 141-283).  Shared by k_shade and k_tail: one path gives the same numbers in both.
RRT_DEV ShadeOut shade_bounce(const RenderParams& p, const SceneTables* sc, const Surface& s, int prim, uint32_t pixel,
                              uint32_t sample, uint32_t bounce, V3 beta) {
  ShadeOut o;
  o.flags = 0;
  o.light_c = o.sh_o = o.sh_d = o.mis_o = o.mis_d = o.mis_c = o.next_o = o.next_d = mk(0.0f, 0.0f, 0.0f);
  o.sh_tmax = 0.0f;
  o.sh_skip = -1;
  o.survive = false;
  o.next_state = 0;
  const DeviceMaterial m = p.materials[s.material];
  if (p.aov && bounce == 0) {
    // what a denoiser wants beside the image: the shading normal and the albedo of what the camera sees
    float* an = p.aov + 4 * (size_t)pixel;
    float* aa = an + 4 * (size_t)p.width * p.height;
    const V3 albedo = m.diffuse_kind != DIFF_NONE ? mk(m.diffuse) : (m.spec_kind != SPEC_NONE ? mk(m.spec) : mk(m.trans));
    unsafeAtomicAdd(an + 0, s.ns.x); unsafeAtomicAdd(an + 1, s.ns.y); unsafeAtomicAdd(an + 2, s.ns.z);
    unsafeAtomicAdd(an + 3, 1.0f);
    unsafeAtomicAdd(aa + 0, albedo.x); unsafeAtomicAdd(aa + 1, albedo.y); unsafeAtomicAdd(aa + 2, albedo.z);
  }
  const Rng4 ra = rng_block(p.seed, pixel, sample, 1 + 3 * bounce);

  // ---- uniform_sample_light + estimate_direct (src/tracer.rs:141-283)
  if (sc->n_lights > 0) {
    const int li = min((int)(ra.u[0] * (float)sc->n_lights), sc->n_lights - 1);
    const float n_lights = (float)sc->n_lights;  // division by light_pdf = 1 / n
    const LightSample ls = sample_light(sc, p.env, li, s.p, ra.u[1], ra.u[2]);
    if (ls.pdf > 1e-6f && !is_zero(ls.li)) {
      V3 f;
      float pdf_s;
      bsdf_eval(s, m, ls.wi, &f, &pdf_s);
      f = f * fabsf(dot(ls.wi, s.ns));
      if (!is_zero(f) && !(prim < sc->n_tris && own_triangle_occludes(p.tri_shade, prim, s, ls.wi))) {
        const float w = ls.delta ? 1.0f : power_heuristic(ls.pdf, pdf_s);
        o.light_c = beta * f * ls.li * (w * n_lights / ls.pdf);
        // check_light_visible (src/renderer.rs:277-295): from the surface point towards the
        // sample, stopping just short of it.  The 1e-9 / 1e-7 margins of the f64 original do not
        // exist in f32: the surface being left and the light's own quad are excluded by identity.
        o.sh_tmax = ls.dist < INFINITY ? ls.dist * (1.0f - 1e-5f) : INFINITY;
        o.sh_o = s.p;
        o.sh_d = ls.wi;
        o.sh_skip = ls.skip;
        o.flags |= kHasLightRay;
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
          o.mis_c = beta * f * le * (w * n_lights / bs.pdf);
          o.mis_o = offset_ray_origin(s.p, s.ng, bs.wi);
          o.mis_d = bs.wi;
          o.flags |= kHasMisRay;
        }
      }
    }
  }

  // ---- continuation (src/tracer.rs:77-130)
  const Rng4 rc = rng_block(p.seed, pixel, sample, 3 + 3 * bounce);
  const BsdfSample bs = bsdf_sample(s, m, rc.u[0], rc.u[1], rc.u[2]);
  if (!(bs.pdf == 0.0f || is_zero(bs.f))) {
    beta = beta * (bs.f * (fabsf(dot(bs.wi, s.ns)) / bs.pdf));
    if (p.clamp > 0.0f) o.flags |= kClamp;  // the clamp comes after this bounce's direct light is in L
    o.survive = bounce + 1 < p.depth_limit;
    if (bounce > 3) {
      const float q = fmaxf(1.0f - vmax(beta), 0.05f);
      if (ra.u[3] < q) o.survive = false;
      beta = beta / (1.0f - q);
    }
    if (o.survive) {
      o.next_o = offset_ray_origin(s.p, s.ng, bs.wi);
      o.next_d = bs.wi;
      o.next_state = ((bounce + 1) << 1) | (bs.specular ? 1u : 0u);
    }
  }
  o.beta = beta;
  return o;
}

__global__ void RRT_SHADE_BOUNDS k_shade(RenderParams p, uint32_t parity) {
  __shared__ uint32_t lens[kSegments + 1];
  // the whole of the small tables in LDS: every lane reads a score of fields of the light it draws
  __shared__ uint32_t table_words[(sizeof(SceneTables) + 3) / 4];
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(p.scene);
    for (uint32_t i = threadIdx.x; i < (sizeof(SceneTables) + 3) / 4; i += blockDim.x) table_words[i] = src[i];
  }
  const SceneTables* sc = reinterpret_cast<const SceneTables*>(table_words);  // (queue_lengths below synchronises)
  const uint32_t chunks = queue_blocks(queue_lengths(p, kRowShade, lens), 256);
  const uint32_t next = parity ^ 1u;
  for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
  uint32_t n;
  const uint32_t* entries = queue_block(p, p.a.q_shade, lens, chunk, 256, &n);
  const uint32_t i = threadIdx.x;
  uint32_t slot = 0;
  bool active = false, survive = false, has_work = false, has_mis = false, has_light = false;
  V3 next_o = mk(0.0f, 0.0f, 0.0f), next_d = next_o;
  uint32_t next_state = 0;
  if (i < n) {
    active = true;
    slot = entries[i];
    const float4 d4 = p.a.trace[slot].ray_d, h4 = p.a.trace[slot].hit;
    const V3 dir = mk(d4);
    uint32_t x, y, k;
    slot_pixel(p, slot, &x, &y, &k);
    // with one sample per slot the sample index follows from the slot number
    const uint32_t sample = p.single ? p.sample_begin + k : p.a.sample[slot], bounce = __float_as_uint(d4.w) >> 1;
    V3 beta = mk(1.0f, 1.0f, 1.0f);
    if (bounce > 0) beta = mk(p.a.trace[slot].beta);
    else p.a.radiance[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int prim = __float_as_int(h4.x);
    // AreaLight::emitting is zero (lights/area.rs:23-25): hitting an emitter adds nothing
    const Surface s = prim < sc->n_tris ? surface_triangle(p.tri_shade, prim, h4.z, h4.w, dir)
                                        : surface_other(sc, prim, h4.y, mk(p.a.trace[slot].ray_o), dir);
    const ShadeOut o = shade_bounce(p, sc, s, prim, y * p.width + x, sample, bounce, beta);
    has_light = (o.flags & kHasLightRay) != 0;
    has_mis = (o.flags & kHasMisRay) != 0;
    if (has_light) {
      p.a.shadow[slot].sh_o = f4(o.sh_o, o.sh_tmax);
      p.a.shadow[slot].sh_d = f4(o.sh_d, __int_as_float(o.sh_skip));
    }
    if (has_mis) {
      p.a.mis[slot].mis_o = f4(o.mis_o, 0.0f);
      p.a.mis[slot].mis_d = f4(o.mis_d, 0.0f);
      p.a.mis[slot].mis_c = f4(o.mis_c, 0.0f);
    }
    survive = o.survive;
    if (survive) {
      next_o = o.next_o;
      next_d = o.next_d;
      next_state = o.next_state;
      p.a.trace[slot].beta = f4(o.beta, 0.0f);
    }
    has_work = has_light || has_mis;
    // without pending light the running sum is unchanged since its last clamp: nothing to do
    // the record carries the object the rays leave: connect must not read hit[], which the next
    // extend may already be overwriting for this slot
    if (has_work) p.a.shadow[slot].sh_c = f4(o.light_c, __uint_as_float(o.flags | ((uint32_t)prim << 3)));
  }
  ray_queue_push(p, p.a.rays_cont[next], row_cont(next), chunk, survive, slot, next_o, next_d, next_state);
  queue_push(p, p.a.q_start[next], row_start(next), chunk, active && !survive && !p.single, slot);
  queue_push(p, p.a.q_shadow, kRowShadow, chunk, has_work, slot);
  counter_add(p, kRowBounces, chunk, active);
  counter_add(p, kRowMisRays, chunk, has_mis);
  counter_add(p, kRowLightRays, chunk, has_light);
  }
}

// ---------------------------------------------------------------- connect
// One job = one shadow-queue entry: its light-sample ray, then its environment escape ray, then
// the clamp; all by one lane, so the sum keeps the order of the reference's loop body.
struct ConnectJob {
  static constexpr int kDebugRow = 1;  // row of the diagnostic build's counters
  static constexpr bool kRootTest = false;  // its rays start on surfaces
  const RenderParams& p;
  const SceneTables* sc;  // LDS copy of the shapes
  const uint32_t* lens;
  uint32_t grab;
  const uint32_t* entries;  // of the open block
  uint32_t slot, flags;
  int stage;  // the ray in flight: 1 light sample, 2 escape ray
  int own, light_quad;
  float far;
  float4 l4, c4;

  RRT_DEV bool next_ray(Ray* ray) {
    if (stage == 0 && (flags & kHasLightRay)) {
      // two things may not occlude by identity: the surface the ray leaves (a triangle, quad or
      // plane cannot be hit twice by a ray that starts on it) and the sampled light's own quad
      if (own >= sc->n_tris + sc->n_rects + sc->n_planes) own = -1;  // spheres keep their t > 1e-4 guard
      ray->tmin = 0.0f;
      ray->tmax = fminf(far, kTriTMax);
      ray->skip = own < sc->n_tris ? own : -1;
      stage = 1;
      return true;
    }
    if (stage <= 1 && (flags & kHasMisRay)) {
      const float4 o = p.a.mis[slot].mis_o, d = p.a.mis[slot].mis_d;
      own = light_quad = -1;
      far = INFINITY;
      ray->o = mk(o);
      ray->d = mk(d);
      ray->tmin = 0.0f;
      ray->tmax = kTriTMax;
      ray->skip = -1;
      stage = 2;
      return true;
    }
    if (flags & kClamp) {
      l4.x = fminf(l4.x, p.clamp); l4.y = fminf(l4.y, p.clamp); l4.z = fminf(l4.z, p.clamp);
    }
    p.a.radiance[slot] = l4;
    return false;
  }
  RRT_DEV uint32_t open_block(uint32_t j) {
    uint32_t n;
    entries = queue_block(p, p.a.q_shadow, lens, j, grab, &n);
    return n;
  }
  RRT_DEV bool begin(uint32_t i, Ray* ray) {
    slot = entries[i];
    // everything the light-sample ray needs in one round trip, whether or not the entry has one
    c4 = p.a.shadow[slot].sh_c;
    l4 = p.a.radiance[slot];
    const float4 o = p.a.shadow[slot].sh_o, d = p.a.shadow[slot].sh_d;
    flags = __float_as_uint(c4.w) & 7u;
    own = (int)(__float_as_uint(c4.w) >> 3);
    light_quad = __float_as_int(d.w);
    far = o.w;
    ray->o = mk(o);
    ray->d = mk(d);
    stage = 0;
    return next_ray(ray);
  }
  RRT_DEV bool advance(bool found, const Hit&, const RayPre& r, Ray* ray) {
    bool occluded = found;
    if (!RRT_DIAG_NO_OTHERS && !occluded && sc->n_rects + sc->n_planes + sc->n_spheres > 0) {
      Hit h;
      occluded = intersect_others<true>(sc, r, far, own, light_quad, &h);
    }
    if (!occluded) {
      // (not `stage == 1 ? c4 : mis_c[slot]`: a conditional between two lvalues selects between their
      // addresses, which takes the address of a member and sends the whole job to scratch memory)
      float4 c = c4;
      if (stage != 1) c = p.a.mis[slot].mis_c;
      l4.x += c.x; l4.y += c.y; l4.z += c.z;
    }
    return next_ray(ray);
  }
  RRT_DEV void retire() {}
  RRT_DEV void flush(bool, uint32_t) {}
};

template <bool COUNT>
__global__ void RRT_TRACE_BOUNDS k_connect(RenderParams p) {
  RRT_DECLARE_TRACE_STACK();
  RRT_DECLARE_SHAPES();
  const SceneTables* sc = shapes_to_lds(p.scene, rrt_shape_words);
  __shared__ uint32_t lens[kSegments + 1];
  const uint32_t longest = queue_lengths(p, kRowShadow, lens);
  const uint32_t grab = grab_size(longest, gridDim.x * (kTraceBlock / 64));
  TraceCounters ctr = {0, 0, 0};
  ConnectJob job = {p, sc, lens, grab, nullptr, 0, 0, 0, -1, -1, 0.0f, make_float4(0.0f, 0.0f, 0.0f, 0.0f), make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
  trace_stream<true, COUNT>(p.nodes, p.grid, p.tris, sc->n_tris > 0, counter_at(p.a.counters, kRowWorkConnect, 0),
                            queue_blocks(longest, grab), RRT_REFILL_QUORUM, job, &ctr, rrt_trace_stack);
  flush_counters<COUNT>(p.stats, 2, ctr);
}

// ---------------------------------------------------------------- tail
// The last iterations of a pass hold few paths, and each of their launches waits for its longest
// ray: a dozen launches of start / extend / shade / connect that each take a fraction of a millisecond
// whatever the number of rays.  Once few paths are left the host hands ALL remaining work to this
// one kernel instead: a job is a path, and a lane runs it to its end -- continuation ray, shading,
// the visibility rays of that bounce, clamp, next continuation ray -- through the same traversal
// engine and the same shading code, in the order the separate kernels would have used, so every
// path arrives at the same radiance.  (Closest-hit traversal for the visibility rays too: whether
// anything lies in (tmin, tmax) does not depend on the order in which the candidates are found.)
// One sample per slot only (RenderParams::single).
template <bool COUNT>
struct TailJob {
  static constexpr int kDebugRow = 0;  // row of the diagnostic build's counters
  static constexpr bool kRootTest = false;  // its rays start on surfaces
  const RenderParams& p;
  const SceneTables* sc;  // LDS copy of all the small tables
  const uint32_t* lens;
  uint32_t grab, parity;
  const float4* entries;  // of the open block
  uint32_t slot, state;   // state of the continuation ray in flight (bounce << 1 | after a specular lobe)
  int stage;              // the ray in flight: 0 continuation, 1 light sample, 2 environment escape ray
  uint32_t pending;       // kHasLightRay | kHasMisRay still to trace, kClamp to apply after them, kSurvive
  static constexpr uint32_t kSurvive = 8u;
  V3 beta;
  float4 l4;              // the path's radiance so far
  V3 light_c, sh_o, sh_d, next_o, next_d;
  float far;
  int own, light_quad;
  uint32_t next_state;
  uint32_t n_bounces, n_light, n_mis, n_cont;       // of the job in flight
  uint32_t out_bounces, out_light, out_mis, out_cont;  // of the retired job, until the flush

  RRT_DEV uint32_t open_block(uint32_t j) {
    uint32_t n;
    entries = ray_queue_block(p, p.a.rays_cont[parity], lens, j, grab, &n);
    return n;
  }
  RRT_DEV bool begin(uint32_t i, Ray* ray) {
    const float4 o = entries[2 * i], d = entries[2 * i + 1];
    slot = __float_as_uint(o.w);
    state = __float_as_uint(d.w);
    beta = mk(p.a.trace[slot].beta);  // continuation rays have bounce >= 1
    l4 = p.a.radiance[slot];
    stage = 0;
    pending = 0;
    n_bounces = n_light = n_mis = 0;
    n_cont = 1;
    ray->o = mk(o);
    ray->d = mk(d);
    ray->tmin = 0.0f;
    ray->tmax = kTriTMax;
    ray->skip = -1;
    return true;
  }
  // The next ray of this path, in the order connect and extend would trace them; false: the path is complete.
  RRT_DEV bool next_ray(Ray* ray) {
    if (pending & kHasLightRay) {
      pending &= ~kHasLightRay;
      if (own >= sc->n_tris + sc->n_rects + sc->n_planes) own = -1;  // spheres keep their t > 1e-4 guard
      ray->o = sh_o;
      ray->d = sh_d;
      ray->tmin = 0.0f;
      ray->tmax = fminf(far, kTriTMax);
      ray->skip = own < sc->n_tris ? own : -1;
      stage = 1;
      ++n_light;
      return true;
    }
    if (pending & kHasMisRay) {
      pending &= ~kHasMisRay;
      const float4 o = p.a.mis[slot].mis_o, d = p.a.mis[slot].mis_d;  // parked there by this lane
      own = light_quad = -1;
      far = INFINITY;
      ray->o = mk(o);
      ray->d = mk(d);
      ray->tmin = 0.0f;
      ray->tmax = kTriTMax;
      ray->skip = -1;
      stage = 2;
      ++n_mis;
      return true;
    }
    if (pending & kClamp) {
      pending &= ~kClamp;
      l4.x = fminf(l4.x, p.clamp); l4.y = fminf(l4.y, p.clamp); l4.z = fminf(l4.z, p.clamp);
    }
    if (pending & kSurvive) {
      pending = 0;
      state = next_state;
      ray->o = next_o;
      ray->d = next_d;
      ray->tmin = 0.0f;
      ray->tmax = kTriTMax;
      ray->skip = -1;
      stage = 0;
      ++n_cont;
      return true;
    }
    p.a.radiance[slot] = l4;
    return false;
  }
  RRT_DEV bool advance(bool found, const Hit& tri_hit, const RayPre& r, Ray* ray) {
    const bool others = !RRT_DIAG_NO_OTHERS && sc->n_rects + sc->n_planes + sc->n_spheres > 0;
    if (stage != 0) {
      bool occluded = found;
      if (!occluded && others) {
        Hit h;
        occluded = intersect_others<true>(sc, r, far, own, light_quad, &h);
      }
      if (!occluded) {
        float4 c = f4(light_c, 0.0f);
        if (stage != 1) c = p.a.mis[slot].mis_c;
        l4.x += c.x; l4.y += c.y; l4.z += c.z;
      }
      return next_ray(ray);
    }
    Hit h = tri_hit;
    if (others) intersect_others<false>(sc, r, h.t, -1, -1, &h);
    if (h.prim < 0) {
      // the ray escaped: after a specular bounce it sees the lights' environment term (src/tracer.rs:44-63)
      if (state & 1u) {
        const V3 le = escape_radiance(sc, p.env, r.d);
        l4.x += beta.x * le.x; l4.y += beta.y * le.y; l4.z += beta.z * le.z;
      }
      p.a.radiance[slot] = l4;
      return false;
    }
    uint32_t x, y, k;
    slot_pixel(p, slot, &x, &y, &k);
    const Surface s = h.prim < sc->n_tris ? surface_triangle(p.tri_shade, h.prim, h.u, h.v, r.d)
                                          : surface_other(sc, h.prim, h.t, r.o, r.d);
    const ShadeOut o = shade_bounce(p, sc, s, h.prim, y * p.width + x, p.sample_begin + k, state >> 1, beta);
    ++n_bounces;
    pending = o.flags & (kHasLightRay | kHasMisRay);
    if (pending && (o.flags & kClamp)) pending |= kClamp;  // as connect: only an entry with visibility work is clamped
    if (o.flags & kHasLightRay) {
      light_c = o.light_c;
      sh_o = o.sh_o;
      sh_d = o.sh_d;
      far = o.sh_tmax;
      light_quad = o.sh_skip;
      own = h.prim;
    }
    if (o.flags & kHasMisRay) {
      p.a.mis[slot].mis_o = f4(o.mis_o, 0.0f);
      p.a.mis[slot].mis_d = f4(o.mis_d, 0.0f);
      p.a.mis[slot].mis_c = f4(o.mis_c, 0.0f);
    }
    if (o.survive) {
      pending |= kSurvive;
      next_o = o.next_o;
      next_d = o.next_d;
      next_state = o.next_state;
      beta = o.beta;
    }
    return next_ray(ray);
  }
  RRT_DEV void retire() {
    out_bounces = n_bounces;
    out_light = n_light;
    out_mis = n_mis;
    out_cont = n_cont;
  }
  // The pass's running totals: one atomic per counter and flush.
  RRT_DEV void flush(bool done, uint32_t block) {
    uint32_t b = done ? out_bounces : 0u, l = done ? out_light : 0u, m = done ? out_mis : 0u, c = done ? out_cont : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      b += (uint32_t)__shfl_xor((int)b, d);
      l += (uint32_t)__shfl_xor((int)l, d);
      m += (uint32_t)__shfl_xor((int)m, d);
      c += (uint32_t)__shfl_xor((int)c, d);
    }
    if ((threadIdx.x & 63) == 0) {
      const uint32_t seg = block % kSegments;
      if (b) atomicAdd(counter_at(p.a.counters, kRowBounces, seg), b);
      if (l) atomicAdd(counter_at(p.a.counters, kRowLightRays, seg), l);
      if (m) atomicAdd(counter_at(p.a.counters, kRowMisRays, seg), m);
      if (c) atomicAdd(counter_at(p.a.counters, kRowContRays, seg), c);
    }
  }
};

#ifdef RRT_TAIL_WAVES
#define RRT_TAIL_BOUNDS __launch_bounds__(rrt::kTraceBlock, RRT_TAIL_WAVES)
#else
#define RRT_TAIL_BOUNDS __launch_bounds__(rrt::kTraceBlock)
#endif
template <bool COUNT>
__global__ void RRT_TAIL_BOUNDS k_tail(RenderParams p, uint32_t parity) {
  RRT_DECLARE_TRACE_STACK();
  __shared__ uint32_t table_words[(sizeof(SceneTables) + 3) / 4];
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(p.scene);
    for (uint32_t i = threadIdx.x; i < (sizeof(SceneTables) + 3) / 4; i += blockDim.x) table_words[i] = src[i];
  }
  const SceneTables* sc = reinterpret_cast<const SceneTables*>(table_words);  // (queue_lengths below synchronises)
  __shared__ uint32_t lens[kSegments + 1];
  const uint32_t longest = queue_lengths(p, row_cont(parity), lens);
  // small blocks: the jobs are long and few, and every wave should get some
  const uint32_t grab = 64;
  TraceCounters ctr = {0, 0, 0};
  TailJob<COUNT> job = {p, sc, lens, grab, parity};
  trace_stream<false, COUNT>(p.nodes, p.grid, p.tris, sc->n_tris > 0, counter_at(p.a.counters, kRowWorkExtend, 0),
                             queue_blocks(longest, grab), RRT_TAIL_QUORUM, job, &ctr, rrt_trace_stack);
  flush_counters<COUNT>(p.stats, 1, ctr);
}

// ---------------------------------------------------------------- film
// Adds the sums of a pixel's K slots to the film.  A group of G lanes (a power of two, at most a
// wave) shares a pixel: lane g adds slots g, g + G, ... in that order, then the lanes' sums are
// combined pairwise -- consecutive lanes read consecutive slots, and the order of the additions is
// fixed, so the result is reproducible.  A pixel belongs to one group of one pass at a time: no
// atomics.  Every live slot has banked at least one path (its first sample), so `accum` holds no
// stale values where it is read.
__global__ void __launch_bounds__(256) k_film(RenderParams p, uint32_t group) {
  const uint32_t thread = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t pix = thread / group, g = thread % group;
  uint32_t x = 0, y = 0, k;
  const bool live = pix < p.n_tiles * 64 && slot_pixel(p, pix * p.n_samples, &x, &y, &k);
  float r = 0.0f, gr = 0.0f, b = 0.0f, w = 0.0f;
  if (live) {
    // with one sample per slot the slot's radiance is the sample (weight 1): nothing was banked
    const float4* acc = (p.single ? p.a.radiance : p.a.accum) + (size_t)pix * p.n_samples;
    for (uint32_t s = g; s < p.n_samples; s += group) {
      const float4 v = acc[s];
      r += v.x; gr += v.y; b += v.z; w += p.single ? 1.0f : v.w;
    }
  }
  for (uint32_t d = group >> 1; d > 0; d >>= 1) {
    r += __shfl_xor(r, d);
    gr += __shfl_xor(gr, d);
    b += __shfl_xor(b, d);
    w += __shfl_xor(w, d);
  }
  if (live && g == 0) {
    float4* out = reinterpret_cast<float4*>(p.film) + ((size_t)y * p.width + x);
    float4 f = *out;
    f.x += r; f.y += gr; f.z += b; f.w += w;
    *out = f;
  }
}

// Film::merge_bucket_pixels_to_image_buffer's tone map (src/film.rs:233-249, 360-394).
RRT_DEV float gamma_correct_srgb(float v) {
  if (v <= 0.0f) return 0.0f;
  if (v < 0.0031308f) return v * 12.92f;
  if (v < 1.0f) return powf(v, 1.0f / 2.4f) * 1.055f - 0.055f;
  return 1.0f;
}
__global__ void __launch_bounds__(256) k_tonemap(const float4* __restrict__ film, uint32_t n_pixels, uint8_t* __restrict__ rgb8) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pixels) return;
  const float4 f = film[i];
  uint8_t r = 0, g = 0, b = 0;
  if (!(f.w < 2.220446049250313e-16f)) {
    const float x = f.x / f.w, y = f.y / f.w, z = f.z / f.w;  // radiance goes into the XYZ matrix as it is
    r = (uint8_t)(gamma_correct_srgb(3.240479f * x - 1.537150f * y - 0.498535f * z) * 255.0f);
    g = (uint8_t)(gamma_correct_srgb(-0.969256f * x + 1.875991f * y + 0.041556f * z) * 255.0f);
    b = (uint8_t)(gamma_correct_srgb(0.055648f * x - 0.204043f * y + 1.057311f * z) * 255.0f);
  }
  rgb8[3 * (size_t)i + 0] = r;
  rgb8[3 * (size_t)i + 1] = g;
  rgb8[3 * (size_t)i + 2] = b;
}

// Folds the pass's running counters into the statistics (one wave, lane = segment).
__global__ void k_pass_stats(RenderParams p) {
  const uint32_t seg = threadIdx.x;
  if (blockIdx.x != 0 || seg >= (uint32_t)kSegments) return;
  unsigned long long paths = *counter_at(p.a.counters, kRowPaths, seg);
  unsigned long long cont = *counter_at(p.a.counters, kRowContRays, seg);
  unsigned long long bounces = *counter_at(p.a.counters, kRowBounces, seg);
  // a shadow-queue entry carries a light-sample ray, an environment escape ray, or both
  unsigned long long shadow = (unsigned long long)*counter_at(p.a.counters, kRowLightRays, seg) +
                              *counter_at(p.a.counters, kRowMisRays, seg);
  for (int d = kSegments / 2; d > 0; d >>= 1) {
    paths += __shfl_down(paths, d);
    cont += __shfl_down(cont, d);
    bounces += __shfl_down(bounces, d);
    shadow += __shfl_down(shadow, d);
  }
  if (seg == 0) {
    p.stats->paths += paths;
    p.stats->rays_extend += paths + cont;
    p.stats->path_bounces += bounces;
    p.stats->rays_shadow += shadow;
  }
}

// ---------------------------------------------------------------- parity hooks
struct IntersectJob {
  static constexpr int kDebugRow = 0;  // row of the diagnostic build's counters
  static constexpr bool kRootTest = true;  // its rays may start anywhere
  const TriIsect* tris;
  const float* tri_pos;     // scene order, [n_tris][3][3]: the vertices as uploaded
  const ShapesF64* exact;   // the analytic shapes' parameters in f64
  const SceneTables* sc;
  const float* org_dir;
  int32_t* prim_id;
  float* t_out;
  uint32_t n, base, index;

  RRT_DEV uint32_t open_block(uint32_t j) {
    base = j * 256u;
    return min(256u, n - base);
  }
  RRT_DEV bool begin(uint32_t i, Ray* ray) {
    index = base + i;
    const float* r = org_dir + (size_t)index * 6;
    ray->o = mk(r[0], r[1], r[2]);
    ray->d = mk(r[3], r[4], r[5]);
    ray->tmin = 0.0f;
    ray->tmax = kTriTMax;
    ray->skip = -1;
    return true;
  }
  RRT_DEV bool advance(bool, const Hit& tri_hit, const RayPre& r, Ray*) {
    Hit h = tri_hit;
    if (!RRT_DIAG_NO_OTHERS && sc->n_rects + sc->n_planes + sc->n_spheres > 0) intersect_others<false>(sc, r, h.t, -1, -1, &h);
    int id = h.prim;
    float t = h.prim >= 0 ? h.t : 0.0f;
    if (id >= 0) {
      // which object: decided above, in f32.  How far: that one object once more, in f64 (exact_*_t).
      double te = -1.0;
      int k = id - sc->n_tris;
      if (k < 0) {
        id = __float_as_int(tris[id].v0.w);  // leaf order -> scene order
        te = exact_triangle_t(tri_pos + (size_t)id * 9, r.o, r.d);
      } else if (k < sc->n_rects) {
        te = exact_plane_t(exact->rect_pos[k], exact->rect_normal[k], r.o, r.d, 1e-9);
      } else if ((k -= sc->n_rects) < sc->n_planes) {
        te = exact_plane_t(exact->plane_pos[k], exact->plane_normal[k], r.o, r.d, 1e-7);
      } else {
        k -= sc->n_planes;
        te = exact_sphere_t(exact->sphere_center[k], exact->sphere_radius[k], r.o, r.d);
      }
      if (te > 0.0) t = (float)te;
    }
    prim_id[index] = id;
    t_out[index] = t;
    return false;
  }
  RRT_DEV void retire() {}
  RRT_DEV void flush(bool, uint32_t) {}
};

__global__ void RRT_TRACE_BOUNDS k_intersect(const BvhNode* nodes, BvhGrid grid, const TriIsect* tris, const float* __restrict__ tri_pos,
                                             const ShapesF64* __restrict__ exact, const SceneTables* sc,
                                             const float* __restrict__ org_dir, uint32_t n, int32_t* prim_id, float* t_out,
                                             uint32_t* work_cursor) {
  RRT_DECLARE_TRACE_STACK();
  RRT_DECLARE_SHAPES();
  sc = shapes_to_lds(sc, rrt_shape_words);
  TraceCounters ctr = {0, 0, 0};
  IntersectJob job = {tris, tri_pos, exact, sc, org_dir, prim_id, t_out, n, 0, 0};
  trace_stream<false, false>(nodes, grid, tris, sc->n_tris > 0, work_cursor, (n + 255u) / 256u, RRT_REFILL_QUORUM, job, &ctr,
                             rrt_trace_stack);
}

struct OccludedJob {
  static constexpr int kDebugRow = 1;  // row of the diagnostic build's counters
  static constexpr bool kRootTest = true;  // its rays may start anywhere
  const SceneTables* sc;
  const float* odt;
  uint8_t* out;
  uint32_t n, base, index;
  float far;

  RRT_DEV uint32_t open_block(uint32_t j) {
    base = j * 256u;
    return min(256u, n - base);
  }
  RRT_DEV bool begin(uint32_t i, Ray* ray) {
    index = base + i;
    const float* r = odt + (size_t)index * 7;
    far = r[6];
    ray->o = mk(r[0], r[1], r[2]);
    ray->d = mk(r[3], r[4], r[5]);
    ray->tmin = 0.0f;
    ray->tmax = fminf(far, kTriTMax);
    ray->skip = -1;
    return true;
  }
  RRT_DEV bool advance(bool found, const Hit&, const RayPre& r, Ray*) {
    bool occluded = found;
    if (!RRT_DIAG_NO_OTHERS && !occluded && sc->n_rects + sc->n_planes + sc->n_spheres > 0) {
      Hit h;
      occluded = intersect_others<true>(sc, r, far, -1, -1, &h);
    }
    out[index] = occluded ? 1 : 0;
    return false;
  }
  RRT_DEV void retire() {}
  RRT_DEV void flush(bool, uint32_t) {}
};

__global__ void RRT_TRACE_BOUNDS k_occluded(const BvhNode* nodes, BvhGrid grid, const TriIsect* tris,
                                                          const SceneTables* sc, const float* __restrict__ odt,
                                                          uint32_t n, uint8_t* out, uint32_t* work_cursor) {
  RRT_DECLARE_TRACE_STACK();
  RRT_DECLARE_SHAPES();
  sc = shapes_to_lds(sc, rrt_shape_words);
  TraceCounters ctr = {0, 0, 0};
  OccludedJob job = {sc, odt, out, n, 0, 0, 0.0f};
  trace_stream<true, false>(nodes, grid, tris, sc->n_tris > 0, work_cursor, (n + 255u) / 256u, RRT_REFILL_QUORUM, job, &ctr,
                            rrt_trace_stack);
}

}  // namespace rrt
