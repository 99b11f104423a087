// C ABI of librrt_hip.so (include/rrt.h): context, scene upload, camera, film,
// the launch sequence of a pass of the streaming wavefront path tracer, parity hooks,
// statistics and the single-process RCCL film reduce.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rrt.h"
#include "rrt_device.h"
#include "rrt_kernels.h"
#include "rrt_threads.h"

using namespace rrt;

namespace {

thread_local std::string g_create_error;
std::atomic<int> g_live_contexts{0};  // contexts alive in this process
void destroy_comm_sets();              // frees the RCCL communicators rrt_film_reduce keeps

constexpr uint32_t kSlotLimit = 1u << 30;          // slot numbers share a word with two flag bits
#ifndef RRT_AUTO_MEMORY_SHARE
#define RRT_AUTO_MEMORY_SHARE 0.6
#endif
constexpr uint32_t kMinRoundSamples = 16;   // fewest slots per pixel for rendering in rounds of one sample per slot
constexpr double kAutoMemoryShare = RRT_AUTO_MEMORY_SHARE;  // automatic pool size: this share of the free device memory
constexpr int kBlock = 256;
constexpr uint32_t kMaxSlotsPerPixel = 1024;
constexpr uint32_t kQueueGridBlocks = 8192;  // 32 blocks per CU for the kernels that walk a queue chunk by chunk
#ifndef RRT_OVERLAP_FROM
#define RRT_OVERLAP_FROM 1   // first iteration whose visibility rays run beside the next extend (-1: never)
#endif
#ifndef RRT_AUTO_DEVICE_BUILD_TRIS
#define RRT_AUTO_DEVICE_BUILD_TRIS (4u << 20)
#endif
constexpr uint32_t kAutoDeviceBuildTris = RRT_AUTO_DEVICE_BUILD_TRIS;  // from here on the hierarchy is built by kernels unless told otherwise
#ifndef RRT_TAIL_PATHS
#define RRT_TAIL_PATHS (1u << 21)
#endif
#ifndef RRT_TAIL_BURST
#define RRT_TAIL_BURST 2   // iterations between two looks at the queue lengths while the tail kernel may take over
#endif
#ifndef RRT_TRACE_GRID
#define RRT_TRACE_GRID 2048
#endif
constexpr uint32_t kTraceGridBlocks = RRT_TRACE_GRID;  // 8 blocks = 32 waves per CU: at least what a CU can hold of the traversal kernels

inline uint32_t grid_for(uint32_t n) { return (n + kBlock - 1) / kBlock; }

// ---- rrt_intersect_noalloc: device memory that exists without any allocation call.
// One copy per GPU the library runs on, held until the process ends: the job cursor of the traversal
// kernel, then staging for RRT_NOALLOC_MAX_RAYS rays (6 floats each), their object ids and distances.
constexpr size_t kScratchCursorBytes = 256;
constexpr size_t kScratchRaysAt = kScratchCursorBytes;
constexpr size_t kScratchPrimAt = kScratchRaysAt + sizeof(float) * 6 * (size_t)RRT_NOALLOC_MAX_RAYS;
constexpr size_t kScratchTAt = kScratchPrimAt + sizeof(int32_t) * (size_t)RRT_NOALLOC_MAX_RAYS;
constexpr size_t kScratchBytes = kScratchTAt + sizeof(float) * (size_t)RRT_NOALLOC_MAX_RAYS;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
// In host code `scratch` is only a handle for hipGetSymbolAddress; the address of each GPU's copy is
// looked up once per device (with that device current) and kept here.
constexpr int kMaxScratchDevices = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxScratchDevices] = {};

// Address of the current device's copy of `scratch` (the caller has made `device` current), or null.
unsigned char* scratch_for_device(int device, std::string* error) {
  if (device < 0 || device >= kMaxScratchDevices) {
    *error = "device index beyond the scratch table";
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
      *error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
      return nullptr;
    }
    g_scratch_ptr[device] = (unsigned char*)ptr;
  }
  return g_scratch_ptr[device];
}

struct EventPair {
  hipEvent_t begin, end;
  int stage;  // 0 start, 1 extend, 2 shade, 3 connect, 4 film, 5 primary, 6 tail
};

}  // namespace

struct rrt_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t side_stream = nullptr;          // visibility rays of late iterations run here, beside the next extend
  hipEvent_t ev_shade = nullptr, ev_connect = nullptr;
  std::string error;

  // scene
  bool have_scene = false;
  BvhNode* d_nodes = nullptr;
  BvhGrid grid = {};  // of the hierarchy in d_nodes
  TriIsect* d_tris = nullptr;
  TriShade* d_tri_shade = nullptr;
  DeviceMaterial* d_materials = nullptr;
  SceneTables* d_scene = nullptr;
  uint8_t* d_env = nullptr;
  float* d_tri_pos = nullptr;      // scene order, as uploaded: what rrt_intersect computes its distances from
  ShapesF64* d_exact = nullptr;    // f64 parameters of the analytic shapes, for the same purpose
  SceneTables h_scene;
  uint32_t n_nodes = 0;
  uint64_t bvh_bytes = 0;
  uint32_t n_tris = 0, bvh_max_depth = 0;
  bool scene_has_specular = false;   // some material has a lobe whose sample is flagged specular (mirror lobe, glass)
  bool bvh_built_on_device = false;
  double bvh_build_seconds = 0.0;   // device build only
  std::string bvh_fallback_reason;  // why the automatic mode built on the host after all (empty: it did not)
  bool debug_fail_device_bvh = false;        // RRT_DEBUG_FAIL_DEVICE_BVH in the environment at rrt_create: every device build fails
  bool debug_reduce_through_rccl = false;    // RRT_DEBUG_REDUCE_SINGLE_THROUGH_RCCL at rrt_create: see rrt_film_reduce
  int device_bvh = 2;               // RRT_OPT_DEVICE_BVH: 0 host build, 1 device build, 2 by triangle count

  // camera
  bool have_camera = false;
  CameraParams camera;

  // film
  float* d_film = nullptr;
  bool film_owned = false;
  uint32_t film_w = 0, film_h = 0;

  // path state
  PathArrays arrays{};
  std::vector<void*> state_allocs;
  uint32_t state_slots = 0, state_depth = 0;
  bool state_has_mis = false;
  bool state_regenerates = false;  // the state has what regenerating slots need (accum, camera-ray queue)

  // reconstruction filter
  uint32_t filter_method = RRT_FILTER_NONE;
  double filter_radius = 0.0;
  float* d_filter_table = nullptr;

  DeviceStats* d_stats = nullptr;
  uint32_t* d_work = nullptr;     // job cursor of the rrt_intersect / rrt_occluded launches
  unsigned char* scratch = nullptr;  // this device's copy of the static scratch array (rrt_intersect_noalloc)
  uint32_t* h_pending = nullptr;  // pinned: the two queue-length rows read back between iterations

  // options
  bool timing = false;
  bool count_traversal = false;
  bool side_stream_allowed = true;            // RRT_OPT_SIDE_STREAM
  bool aovs = false;                          // RRT_OPT_AOVS
  bool debug_pending = false;                 // RRT_DEBUG_PENDING in the environment: print the queue lengths read back
  float* d_aov = nullptr;                     // [2][film_h][film_w][4]: first-hit normal sums (+ hit count), albedo sums
  uint32_t aov_w = 0, aov_h = 0;
  uint32_t batch_samples = 0;
  uint32_t max_slots = 0;  // 0: sized from the free device memory
  uint32_t tail_paths = RRT_TAIL_PATHS;  // RRT_OPT_TAIL_PATHS: at most this many live paths go to the tail kernel (0: never)

  // timing
  std::vector<EventPair> events;
  size_t events_used = 0;
  double ms[7] = {0, 0, 0, 0, 0, 0, 0};  // start, extend, shade, connect, film, primary, tail
  uint64_t launches_extend = 0, launches_primary = 0, rays_primary = 0, launches_tail = 0;
};

namespace {

// Test switch, environment only (no exported function): RRT_DEBUG_FAIL_THREADS_AFTER=n refuses every host
// thread creation of the builder after the first n, as if the system had none left; the work then runs
// on the thread that asked.  Read where a build starts; absent (the normal case): never refuse.
void thread_switch_from_environment() {
  const char* v = std::getenv("RRT_DEBUG_FAIL_THREADS_AFTER");
  g_thread_creations_allowed.store(v ? std::max(-1, std::atoi(v)) : -1);
}

int fail(rrt_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->error = msg;
  else g_create_error = msg;
  return code;
}

// For handlers of std::bad_alloc: the message goes into storage the string already owns when it can (assign
// may reuse the capacity), and a second failure leaves the previous message rather than throwing again.
int fail_static(rrt_ctx* ctx, int code, const char* msg) noexcept {
  try {
    if (ctx) ctx->error.assign(msg);
    else g_create_error.assign(msg);
  } catch (...) {
    if (ctx) ctx->error.clear();
  }
  return code;
}

#define HIP_TRY(ctx, expr)                                                                       \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return fail(ctx, RRT_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
  } while (0)

int bind_device(rrt_ctx* ctx) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return RRT_OK;
}

void free_scene(rrt_ctx* ctx) {
  (void)hipFree(ctx->d_nodes);
  (void)hipFree(ctx->d_tris);
  (void)hipFree(ctx->d_tri_shade);
  (void)hipFree(ctx->d_materials);
  (void)hipFree(ctx->d_scene);
  (void)hipFree(ctx->d_env);
  (void)hipFree(ctx->d_tri_pos);
  (void)hipFree(ctx->d_exact);
  ctx->d_tri_pos = nullptr;
  ctx->d_exact = nullptr;
  ctx->d_nodes = nullptr;
  ctx->d_tris = nullptr;
  ctx->d_tri_shade = nullptr;
  ctx->d_materials = nullptr;
  ctx->d_scene = nullptr;
  ctx->d_env = nullptr;
  ctx->have_scene = false;
}

void free_state(rrt_ctx* ctx) {
  for (void* p : ctx->state_allocs) (void)hipFree(p);
  ctx->state_allocs.clear();
  ctx->arrays = PathArrays{};
  ctx->state_slots = 0;
  ctx->state_depth = 0;
}

template <typename T>
int dev_alloc(rrt_ctx* ctx, T** out, size_t count) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
  if (e != hipSuccess) return fail(ctx, RRT_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
  *out = (T*)p;
  return RRT_OK;
}

// fn(i) for i in [0, n), in contiguous ranges on up to 32 host threads (small n: on this one).  fn must
// not throw.  A range whose thread cannot be started runs on the calling thread.
template <class Fn>
void parallel_for(uint32_t n, Fn fn) {
  const uint32_t workers = n < (1u << 18) ? 1u : std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
  if (workers == 1) {
    for (uint32_t i = 0; i < n; ++i) fn(i);
    return;
  }
  std::thread pool[32];
  JoinGuard join{pool, (int)workers};
  const uint64_t per = ((uint64_t)n + workers - 1) / workers;
  for (uint32_t w = 0; w < workers; ++w) {
    const uint64_t lo = per * w, hi = std::min<uint64_t>(n, lo + per);
    if (lo >= hi) break;
    auto run = [=]() { for (uint64_t i = lo; i < hi; ++i) fn((uint32_t)i); };
    if (!try_spawn(&pool[w], run)) run();
  }
}

template <typename T>
int state_alloc(rrt_ctx* ctx, T** out, size_t count) {
  int rc = dev_alloc(ctx, out, count);
  if (rc == RRT_OK) ctx->state_allocs.push_back(*out);
  return rc;
}

// Device memory ensure_state takes for n_slots slots (0 for none).
// `regenerates`: slots work through several samples (or splat through a filter), which takes the sum
// of finished paths and the camera-ray queue besides.
uint64_t state_bytes(uint32_t n_slots, bool with_mis, bool regenerates) {
  if (n_slots == 0) return 0;
  const uint64_t n = n_slots;
  const uint64_t seg_cap = 2 * ((n + kSegments - 1) / kSegments) + n / 2048 + 4096;
  const uint64_t ray_seg_cap = (n + kSegments - 1) / kSegments + 1024;
  return n * (64ull * (with_mis ? 3 : 2) + (regenerates ? 2 : 1) * 16 + 4) + 4ull * seg_cap * kSegments * 4 +
         (regenerates ? 3ull : 2ull) * ray_seg_cap * kSegments * 32;
}

// Path state for n_slots paths and `depth` bounces; grows only.
int ensure_state(rrt_ctx* ctx, uint32_t n_slots, uint32_t depth, bool need_mis, bool regenerates) {
  if (ctx->state_slots >= n_slots && ctx->state_depth >= depth && (ctx->state_has_mis || !need_mis) &&
      (ctx->state_regenerates || !regenerates))
    return RRT_OK;
  free_state(ctx);
  PathArrays& a = ctx->arrays;
  int rc;
  if ((rc = state_alloc(ctx, &a.trace, n_slots)) != RRT_OK) return rc;
  if ((rc = state_alloc(ctx, &a.shadow, n_slots)) != RRT_OK) return rc;
  if (need_mis && (rc = state_alloc(ctx, &a.mis, n_slots)) != RRT_OK) return rc;
  if ((rc = state_alloc(ctx, &a.radiance, n_slots)) != RRT_OK) return rc;
  if (regenerates && (rc = state_alloc(ctx, &a.accum, n_slots)) != RRT_OK) return rc;
  // queues: kSegments fixed-size segments.  A segment takes what the chunks c = segment (mod
  // kSegments) of a launch produce; the start queue is fed by two launches (extend and shade).
  // (extend hands in results per index block of at most 1024 jobs, shade per chunk of 256.)
  a.seg_cap = 2 * ((n_slots + kSegments - 1) / kSegments) + n_slots / 2048 + 4096;
  const size_t queue_words = (size_t)a.seg_cap * kSegments;
  uint32_t** queues[] = {&a.q_start[0], &a.q_start[1], &a.q_shade, &a.q_shadow};
  for (uint32_t** q : queues)
    if ((rc = state_alloc(ctx, q, queue_words)) != RRT_OK) return rc;
  // ray queues have one producer each (shade, start), handing in per chunk of 256
  a.ray_seg_cap = (n_slots + kSegments - 1) / kSegments + 1024;
  float4** ray_queues[] = {&a.rays_cont[0], &a.rays_cont[1], &a.rays_camera};
  for (float4** q : ray_queues)
    if ((regenerates || q != &a.rays_camera) && (rc = state_alloc(ctx, q, (size_t)a.ray_seg_cap * kSegments * 2)) != RRT_OK) return rc;
  if ((rc = state_alloc(ctx, &a.sample, n_slots)) != RRT_OK) return rc;
  if ((rc = state_alloc(ctx, &a.counters, (size_t)kCounterRows * kSegments * kCounterStride)) != RRT_OK) return rc;
  if (!ctx->h_pending)
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_pending, sizeof(uint32_t) * 2 * kSegments * kCounterStride, hipHostMallocDefault));
  ctx->state_slots = n_slots;
  ctx->state_depth = depth;
  ctx->state_has_mis = need_mis;
  ctx->state_regenerates = regenerates;
  return RRT_OK;
}

void norm3(const double* v, float* out) {
  const double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  for (int k = 0; k < 3; ++k) out[k] = (float)(v[k] / l);
}
void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// helpers::coordinate_system (src/helpers.rs:73-83) in f64 on the host
void coordinate_system_host(const double* v1, double* v2, double* v3) {
  if (std::fabs(v1[0]) > std::fabs(v1[1])) {
    const double l = std::sqrt(v1[0] * v1[0] + v1[2] * v1[2]);
    v2[0] = -v1[2] / l; v2[1] = 0.0; v2[2] = v1[0] / l;
  } else {
    const double l = std::sqrt(v1[1] * v1[1] + v1[2] * v1[2]);
    v2[0] = 0.0; v2[1] = v1[2] / l; v2[2] = -v1[1] / l;
  }
  cross3(v1, v2, v3);
}

// The lobes compute_scattering_functions adds for one material (materials/*.rs).
DeviceMaterial digest_material(const rrt_material& m) {
  DeviceMaterial d;
  std::memset(&d, 0, sizeof(d));
  const bool kd_zero = m.kd[0] == 0.0 && m.kd[1] == 0.0 && m.kd[2] == 0.0;
  const bool ks_zero = m.ks[0] == 0.0 && m.ks[1] == 0.0 && m.ks[2] == 0.0;
  d.ior = (float)m.ior;
  switch (m.kind) {
    case RRT_MAT_MATTE: {  // matte.rs:26-40
      const double sigma = std::min(std::max(m.roughness, 0.0), 90.0);
      if (!kd_zero) {
        for (int k = 0; k < 3; ++k) d.diffuse[k] = (float)m.kd[k];
        if (sigma == 0.0) {
          d.diffuse_kind = DIFF_LAMBERT;
        } else {
          d.diffuse_kind = DIFF_OREN_NAYAR;  // OrenNayar::new takes the raw roughness (oren_nayar.rs:17-27)
          const double s2 = m.roughness * m.roughness;
          d.oren_a = (float)(1.0 - (s2 / (2.0 * (s2 + 0.33))));
          d.oren_b = (float)(0.45 * s2 / (s2 + 0.09));
        }
      }
      break;
    }
    case RRT_MAT_PLASTIC: {  // plastic.rs:36-65
      if (!kd_zero) {
        for (int k = 0; k < 3; ++k) d.diffuse[k] = (float)m.kd[k];
        d.diffuse_kind = DIFF_LAMBERT;
      }
      if (!ks_zero) {
        for (int k = 0; k < 3; ++k) d.spec[k] = (float)m.ks[k];
        if (m.roughness < 1.0e-3) {
          d.spec_kind = SPEC_MIRROR_DIELECTRIC;
        } else {
          d.spec_kind = SPEC_MICROFACET;
          const double r = std::max(m.roughness, 1.0e-3);  // roughness_to_alpha, microfacet_distribution.rs:127-142
          d.alpha = (float)(r * r);
        }
      }
      break;
    }
    case RRT_MAT_MIRROR:  // mirror.rs:21-30
      for (int k = 0; k < 3; ++k) d.spec[k] = (float)m.kd[k];
      d.spec_kind = SPEC_MIRROR_NOOP;
      break;
    case RRT_MAT_GLASS:  // glass.rs:32-48: transmission (refraction colour) first, then reflection
      for (int k = 0; k < 3; ++k) { d.trans[k] = (float)m.ks[k]; d.spec[k] = (float)m.kd[k]; }
      d.has_trans = 1;
      d.spec_kind = SPEC_MIRROR_DIELECTRIC;
      break;
    default: break;
  }
  d.n_lobes = (d.has_trans ? 1 : 0) + (d.diffuse_kind != DIFF_NONE ? 1 : 0) + (d.spec_kind != SPEC_NONE ? 1 : 0);
  return d;
}

int record_begin(rrt_ctx* ctx, int stage, hipStream_t stream) {
  if (!ctx->timing) return RRT_OK;
  if (ctx->events_used == ctx->events.size()) {
    EventPair ep;
    HIP_TRY(ctx, hipEventCreate(&ep.begin));
    HIP_TRY(ctx, hipEventCreate(&ep.end));
    ctx->events.push_back(ep);
  }
  ctx->events[ctx->events_used].stage = stage;
  HIP_TRY(ctx, hipEventRecord(ctx->events[ctx->events_used].begin, stream));
  return RRT_OK;
}
int record_begin(rrt_ctx* ctx, int stage) { return record_begin(ctx, stage, ctx->stream); }

int record_end(rrt_ctx* ctx, hipStream_t stream) {
  if (!ctx->timing) return RRT_OK;
  HIP_TRY(ctx, hipEventRecord(ctx->events[ctx->events_used].end, stream));
  ctx->events_used++;
  return RRT_OK;
}
int record_end(rrt_ctx* ctx) { return record_end(ctx, ctx->stream); }

int collect_timing(rrt_ctx* ctx) {
  if (!ctx->timing || ctx->events_used == 0) return RRT_OK;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // (work on the side stream has been joined by then)
  for (size_t i = 0; i < ctx->events_used; ++i) {
    float ms = 0.0f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->events[i].begin, ctx->events[i].end));
    ctx->ms[ctx->events[i].stage] += ms;
    if (ctx->events[i].stage == 1) ctx->launches_extend++;
    if (ctx->events[i].stage == 5) ctx->launches_primary++;
    if (ctx->events[i].stage == 6) ctx->launches_tail++;
  }
  ctx->events_used = 0;
  return RRT_OK;
}

// One pass: every sample of [sample_begin, sample_end) for the pixels of tiles
// [tile_begin, tile_begin + n_tiles).  Iterations of extend / shade / connect run until no slot has
// a path or a sample left; how many that takes depends on the paths, so the host reads the two
// queue lengths back every few iterations (a 16 KB copy and one stream synchronisation).
int launch_pass(rrt_ctx* ctx, RenderParams& p) {
  const size_t row_words = (size_t)kSegments * kCounterStride, row_bytes = sizeof(uint32_t) * row_words;
  HIP_TRY(ctx, hipMemsetAsync(p.a.counters, 0, row_bytes * kCounterRows, ctx->stream));
  const uint32_t grid_all = grid_for(p.n_slots);
  // queue-driven kernels run on a fixed grid that walks its queue: a launch over a short queue
  // then costs a few microseconds instead of one empty block per 256 slots
  const uint32_t grid = std::min(grid_all + 1, kQueueGridBlocks);
  // traversal kernels: enough blocks to fill every CU; their waves take jobs from a cursor
  const uint32_t trace_grid = std::min(grid_all + 1, kTraceGridBlocks);
  int rc;
  const uint32_t per_slot = (p.sample_end - p.sample_begin + p.n_samples - 1) / p.n_samples;
  p.single = per_slot == 1 && !p.filter_table ? 1u : 0u;
  // a slot needs at least one iteration per sample plus one to retire, at most depth_limit per sample
  const uint64_t max_iterations = (uint64_t)per_slot * p.depth_limit + 2;
  // Late iterations hold few rays and each launch waits for its longest one: from iteration
  // RRT_OVERLAP_FROM on, the visibility rays of an iteration run on a second stream beside the start
  // and extend of the next (shade joins them again: it rewrites the records connect reads).  Only
  // with one sample per slot: a regenerating slot's start banks the radiance connect is still adding to.
  const bool may_overlap = p.single != 0 && RRT_OVERLAP_FROM >= 0 && ctx->side_stream_allowed;
  if (may_overlap && !ctx->side_stream) {
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_shade, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_connect, hipEventDisableTiming));
  }
  bool connect_outstanding = false;  // a connect on the side stream that the main stream has not joined yet
  // With one sample per slot the start kernel only serves rays that left the scene after a specular bounce
  // (they still owe the environment term).  A scene without a specular lobe, or without anything to collect
  // out there, never queues one: the launch -- which would wait behind the visibility rays of the previous
  // iteration for a free CU just to find its queue empty -- is left out.
  const bool start_has_work = p.single == 0 || (ctx->scene_has_specular && p.escape_lit != 0);
  // the tail kernel: one sample per slot only; not while traversal steps are being counted per kernel
  const bool use_tail = p.single != 0 && ctx->tail_paths > 0 && !ctx->count_traversal;
  // Whichever way the pass is left: the caller's stream waits for a connect still running on the side
  // stream (the next call's kernels rewrite the records it reads), and a pass that failed leaves no
  // half-recorded timing events behind.
  struct PassGuard {
    rrt_ctx* ctx;
    bool* outstanding;
    bool completed = false;
    ~PassGuard() {
      if (*outstanding) (void)hipStreamWaitEvent(ctx->stream, ctx->ev_connect, 0);
      if (!completed) ctx->events_used = 0;
    }
  } pass_guard{ctx, &connect_outstanding};
  uint64_t it = 0;
  uint32_t burst = per_slot + 1;
  for (;;) {
    for (uint32_t j = 0; j < burst; ++j, ++it) {
      const uint32_t parity = (uint32_t)(it & 1), next = parity ^ 1u;
      // camera, shade and the two queues of the other parity start this iteration empty
      if (next == 0) {
        HIP_TRY(ctx, hipMemsetAsync(p.a.counters, 0, row_bytes * kRowShadow, ctx->stream));
      } else {
        HIP_TRY(ctx, hipMemsetAsync(p.a.counters + row_words * kRowCamera, 0, row_bytes * (kRowShadow - kRowCamera), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(p.a.counters + row_words * kRowStart1, 0, row_bytes * 2, ctx->stream));
      }
      if (it == 0) {
        // the first camera ray of every slot: generated and traced by one kernel
        if ((rc = record_begin(ctx, 5)) != RRT_OK) return rc;
        if (ctx->count_traversal) hipLaunchKernelGGL(k_primary<true>, dim3(trace_grid), dim3(kBlock), 0, ctx->stream, p, ctx->camera);
        else hipLaunchKernelGGL(k_primary<false>, dim3(trace_grid), dim3(kBlock), 0, ctx->stream, p, ctx->camera);
        if ((rc = record_end(ctx)) != RRT_OK) return rc;
      } else {
        if (start_has_work) {
          if ((rc = record_begin(ctx, 0)) != RRT_OK) return rc;
          hipLaunchKernelGGL(k_start, dim3(grid), dim3(kBlock), 0, ctx->stream, p, ctx->camera, parity);
          if ((rc = record_end(ctx)) != RRT_OK) return rc;
        }
        if ((rc = record_begin(ctx, 1)) != RRT_OK) return rc;
        if (ctx->count_traversal) hipLaunchKernelGGL(k_extend<true>, dim3(trace_grid), dim3(kBlock), 0, ctx->stream, p, parity);
        else hipLaunchKernelGGL(k_extend<false>, dim3(trace_grid), dim3(kBlock), 0, ctx->stream, p, parity);
        if ((rc = record_end(ctx)) != RRT_OK) return rc;
      }
      if (connect_outstanding) {
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_connect, 0));
        connect_outstanding = false;
      }
      // the visibility stage's queue and cursor
      HIP_TRY(ctx, hipMemsetAsync(p.a.counters + row_words * kRowShadow, 0, row_bytes * 2, ctx->stream));
      if ((rc = record_begin(ctx, 2)) != RRT_OK) return rc;
      hipLaunchKernelGGL(k_shade, dim3(grid), dim3(kBlock), 0, ctx->stream, p, parity);
      if ((rc = record_end(ctx)) != RRT_OK) return rc;
      hipStream_t connect_stream = ctx->stream;
      if (may_overlap && it >= (uint64_t)RRT_OVERLAP_FROM) {
        connect_stream = ctx->side_stream;
        HIP_TRY(ctx, hipEventRecord(ctx->ev_shade, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(connect_stream, ctx->ev_shade, 0));
      }
      if ((rc = record_begin(ctx, 3, connect_stream)) != RRT_OK) return rc;
      if (ctx->count_traversal) hipLaunchKernelGGL(k_connect<true>, dim3(trace_grid), dim3(kBlock), 0, connect_stream, p);
      else hipLaunchKernelGGL(k_connect<false>, dim3(trace_grid), dim3(kBlock), 0, connect_stream, p);
      if ((rc = record_end(ctx, connect_stream)) != RRT_OK) return rc;
      if (connect_stream != ctx->stream) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_connect, connect_stream));
        connect_outstanding = true;
      }
    }
    // rows start/cont of the parity the next iteration would read sit next to each other
    // (extend and shade fill them: a connect still running on the side stream does not touch them)
    const uint32_t parity = (uint32_t)(it & 1);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pending, p.a.counters + row_words * (parity ? kRowStart1 : kRowStart0), row_bytes * 2,
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t pending = 0, pending_start = 0;
    for (int i = 0; i < 2 * kSegments; ++i) pending += ctx->h_pending[(size_t)i * kCounterStride];
    for (int i = 0; i < kSegments; ++i) pending_start += ctx->h_pending[(size_t)i * kCounterStride];  // the start row comes first
    if (ctx->debug_pending) std::fprintf(stderr, "after iteration %llu: %llu pending (%llu start)\n", (unsigned long long)it, (unsigned long long)pending, (unsigned long long)pending_start);
    if (pending == 0) break;
    if (it >= max_iterations) return fail(ctx, RRT_E_DEVICE, "path queues did not drain");
    if (use_tail && pending <= ctx->tail_paths) {
      // Few paths left: one kernel runs each of them to its end instead of a dozen more iterations that
      // would each wait for their longest rays (k_tail).  First whatever the last connect still adds, then
      // the rays that escaped after a specular bounce (they only collect the environment).
      if (connect_outstanding) {
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_connect, 0));
        connect_outstanding = false;
      }
      HIP_TRY(ctx, hipMemsetAsync(p.a.counters + row_words * kRowWorkExtend, 0, row_bytes, ctx->stream));
      if (pending_start) {
        if ((rc = record_begin(ctx, 0)) != RRT_OK) return rc;
        hipLaunchKernelGGL(k_start, dim3(grid), dim3(kBlock), 0, ctx->stream, p, ctx->camera, parity);
        if ((rc = record_end(ctx)) != RRT_OK) return rc;
      }
      if (pending > pending_start) {
        if ((rc = record_begin(ctx, 6)) != RRT_OK) return rc;
        hipLaunchKernelGGL(k_tail<false>, dim3(trace_grid), dim3(kBlock), 0, ctx->stream, p, parity);
        if ((rc = record_end(ctx)) != RRT_OK) return rc;
      }
      break;
    }
    burst = (uint32_t)std::min<uint64_t>(use_tail ? RRT_TAIL_BURST : 4, max_iterations - it);
  }
  if (connect_outstanding) {
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_connect, 0));
    connect_outstanding = false;
  }
  if ((rc = record_begin(ctx, 4)) != RRT_OK) return rc;
  // with a reconstruction filter the samples went into the film as their paths ended
  if (!p.filter_table) {
    uint32_t group = 1;  // lanes per pixel: the largest power of two that is at most min(K, 64)
    while (group * 2 <= std::min(p.n_samples, 64u)) group *= 2;
    const uint64_t threads = (uint64_t)p.n_tiles * 64 * group;
    hipLaunchKernelGGL(k_film, dim3((uint32_t)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, p, group);
  }
  hipLaunchKernelGGL(k_pass_stats, dim3(1), dim3(64), 0, ctx->stream, p);
  if ((rc = record_end(ctx)) != RRT_OK) return rc;
  HIP_TRY(ctx, hipGetLastError());
  pass_guard.completed = true;
  return collect_timing(ctx);
}

}  // namespace

extern "C" {

int rrt_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
    return RRT_E_DEVICE;
  }
  return n;
}

rrt_ctx* rrt_create(int device) {
  int n = rrt_device_count();
  if (n <= 0) {
    if (n == 0) g_create_error = "no HIP device visible";
    return nullptr;
  }
  if (device < 0 || device >= n) {
    g_create_error = "device index out of range";
    return nullptr;
  }
  rrt_ctx* ctx = new (std::nothrow) rrt_ctx();
  if (!ctx) {
    g_create_error = "out of host memory";
    return nullptr;
  }
  ctx->device = device;
  ctx->debug_pending = std::getenv("RRT_DEBUG_PENDING") != nullptr;
  // Two more switches for tests, read once per context, never exported as functions: a context created with
  // them in the environment behaves differently, every other context of the process does not.
  ctx->debug_fail_device_bvh = std::getenv("RRT_DEBUG_FAIL_DEVICE_BVH") != nullptr;
  ctx->debug_reduce_through_rccl = std::getenv("RRT_DEBUG_REDUCE_SINGLE_THROUGH_RCCL") != nullptr;
  if (hipSetDevice(device) != hipSuccess || hipMalloc((void**)&ctx->d_stats, sizeof(DeviceStats)) != hipSuccess ||
      hipMalloc((void**)&ctx->d_work, sizeof(uint32_t)) != hipSuccess ||
      hipMemset(ctx->d_stats, 0, sizeof(DeviceStats)) != hipSuccess) {
    g_create_error = "cannot initialise device " + std::to_string(device);
    delete ctx;
    return nullptr;
  }
  // (device `device` is current here)
  ctx->scratch = scratch_for_device(device, &g_create_error);
  if (!ctx->scratch) {
    (void)hipFree(ctx->d_stats);
    (void)hipFree(ctx->d_work);
    delete ctx;
    return nullptr;
  }
  g_live_contexts.fetch_add(1);
  return ctx;
}

void rrt_destroy(rrt_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  free_scene(ctx);
  free_state(ctx);
  if (ctx->film_owned) (void)hipFree(ctx->d_film);
  (void)hipFree(ctx->d_stats);
  (void)hipFree(ctx->d_work);
  (void)hipFree(ctx->d_filter_table);
  (void)hipFree(ctx->d_aov);
  if (ctx->h_pending) (void)hipHostFree(ctx->h_pending);
  for (EventPair& e : ctx->events) {
    (void)hipEventDestroy(e.begin);
    (void)hipEventDestroy(e.end);
  }
  if (ctx->side_stream) {
    (void)hipStreamSynchronize(ctx->side_stream);
    (void)hipStreamDestroy(ctx->side_stream);
    (void)hipEventDestroy(ctx->ev_shade);
    (void)hipEventDestroy(ctx->ev_connect);
  }
  delete ctx;
  if (g_live_contexts.fetch_sub(1) == 1) destroy_comm_sets();  // the last context takes the cached communicators with it
}

int rrt_set_stream(rrt_ctx* ctx, void* hip_stream) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  ctx->stream = (hipStream_t)hip_stream;
  return RRT_OK;
}

namespace {
int upload_scene_impl(rrt_ctx* ctx, const rrt_scene_desc* sd) {
  if (!sd) return fail(ctx, RRT_E_INVALID, "null scene");
  if (sd->n_tris && (!sd->tri_pos || !sd->tri_nrm || !sd->tri_mat)) return fail(ctx, RRT_E_INVALID, "null triangle arrays");
  if ((sd->n_rects && !sd->rects) || (sd->n_planes && !sd->planes) || (sd->n_spheres && !sd->spheres) ||
      (sd->n_materials && !sd->materials) || (sd->n_lights && !sd->lights))
    return fail(ctx, RRT_E_INVALID, "null table");
  if (sd->n_rects > RRT_MAX_RECTS || sd->n_planes > RRT_MAX_PLANES || sd->n_spheres > RRT_MAX_SPHERES ||
      sd->n_lights > RRT_MAX_LIGHTS)
    return fail(ctx, RRT_E_UNSUPPORTED, "too many analytic shapes or lights for the device tables");
  // object numbers share a word with two or three flag bits (leaf references, shadow records)
  if (sd->n_tris >= (1u << 29) - 64) return fail(ctx, RRT_E_UNSUPPORTED, "more than 2^29 - 64 triangles");
  if (sd->rect_aabb_mode > RRT_RECT_AABB_OFF) return fail(ctx, RRT_E_INVALID, "bad rect_aabb_mode");
  for (uint32_t i = 0; i < sd->n_tris; ++i)
    if (sd->tri_mat[i] >= sd->n_materials) return fail(ctx, RRT_E_INVALID, "triangle material index out of range");
  for (uint32_t i = 0; i < sd->n_rects; ++i)
    if (sd->rects[i].material >= sd->n_materials || sd->rects[i].light >= (int32_t)sd->n_lights)
      return fail(ctx, RRT_E_INVALID, "rectangle material / light index out of range");
  for (uint32_t i = 0; i < sd->n_planes; ++i)
    if (sd->planes[i].material >= sd->n_materials) return fail(ctx, RRT_E_INVALID, "plane material index out of range");
  for (uint32_t i = 0; i < sd->n_spheres; ++i)
    if (sd->spheres[i].material >= sd->n_materials) return fail(ctx, RRT_E_INVALID, "sphere material index out of range");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  free_scene(ctx);
  thread_switch_from_environment();

  // ---- BVH over the triangles and the leaf-ordered records: by kernels, or on the host.  In the automatic
  // mode a device build that fails (out of device memory for its ~290 B per triangle of temporaries, a
  // clustering that stops making progress on degenerate input, a tree too deep for the traversal stack)
  // is not the end of the upload: the host's binned-SAH build is depth-bounded and needs no device memory
  // beyond the result, so it runs instead.  Only an explicit RRT_OPT_DEVICE_BVH = 1 reports the failure.
  bool device_build = sd->n_tris >= kDeviceBuildMinTris &&
                      (ctx->device_bvh == 1 || (ctx->device_bvh == 2 && sd->n_tris >= kAutoDeviceBuildTris));
  uint32_t device_depth = 0, device_nodes = 0;
  BvhGrid device_grid = {};
  ctx->bvh_build_seconds = 0.0;
  ctx->bvh_fallback_reason.clear();
  if (device_build) {
    DeviceBvhResult built;
    std::string why;
    rc = ctx->debug_fail_device_bvh ? RRT_E_NOMEM : build_bvh_device(sd->tri_pos, sd->tri_nrm, sd->tri_mat, sd->n_tris, ctx->stream, &built, &why);
    if (ctx->debug_fail_device_bvh) why = "device build refused by RRT_DEBUG_FAIL_DEVICE_BVH";
    if (rc == RRT_OK && 3ull * built.max_depth + 1 > (uint64_t)kStackDepth) {
      rc = RRT_E_UNSUPPORTED;
      why = "hierarchy too deep for the traversal stack";
    }
    ctx->d_nodes = built.d_nodes;  // (free_scene releases them)
    ctx->d_tris = built.d_tris;
    ctx->d_tri_shade = built.d_tri_shade;
    if (rc != RRT_OK) {
      free_scene(ctx);
      if (ctx->device_bvh == 1) return fail(ctx, rc, why);
      ctx->bvh_fallback_reason = why;
      device_build = false;
    } else {
      device_depth = built.max_depth;
      device_nodes = built.n_nodes;
      device_grid = built.grid;
      ctx->bvh_build_seconds = built.seconds;
      if (std::getenv("RRT_BVH_TIMING"))
        std::fprintf(stderr, "bvh build on the device: %u triangles, %u clustering rounds, %u 4-wide nodes, depth %u, %.3f s\n", sd->n_tris,
                     built.rounds, built.n_nodes, built.max_depth, built.seconds);
    }
  }
  BvhBuildResult bvh;
  struct BvhOwner {  // frees the hierarchy on every way out (free_bvh leaves it empty, so the explicit calls below are harmless)
    BvhBuildResult* b;
    ~BvhOwner() { free_bvh(b); }
  } bvh_owner{&bvh};
  if (!device_build && !build_bvh(sd->tri_pos, sd->n_tris, 0, &bvh)) { free_scene(ctx); return fail(ctx, RRT_E_NOMEM, "BVH build: out of host memory"); }
  // (malloc'ed, not vectors: 176 bytes per triangle that need no zeroing first)
  TriIsect* tris = device_build ? nullptr : (TriIsect*)std::malloc(sizeof(TriIsect) * std::max<size_t>(sd->n_tris, 1));
  TriShade* shade = device_build ? nullptr : (TriShade*)std::aligned_alloc(alignof(TriShade), sizeof(TriShade) * std::max<size_t>(sd->n_tris, 1));
  struct HostRecords {
    TriIsect* a; TriShade* b;
    ~HostRecords() { std::free(a); std::free(b); }
  } host_records{tris, shade};
  if (!device_build && (!tris || !shade)) {
    free_bvh(&bvh);
    return fail(ctx, RRT_E_NOMEM, "triangle records: out of host memory");
  }
  if (!device_build) parallel_for(sd->n_tris, [&](uint32_t slot) {
    const uint32_t id = bvh.order[slot];
    const float* p = sd->tri_pos + (size_t)id * 9;
    const float* n = sd->tri_nrm + (size_t)id * 9;
    TriIsect& t = tris[slot];
    uint32_t bits = id;
    float idf;
    std::memcpy(&idf, &bits, 4);
    t.v0 = float4{p[0], p[1], p[2], idf};
    t.e1 = float4{p[3] - p[0], p[4] - p[1], p[5] - p[2], 0.0f};
    t.e2 = float4{p[6] - p[0], p[7] - p[1], p[8] - p[2], 0.0f};
    bits = sd->tri_mat[id];
    float matf;
    std::memcpy(&matf, &bits, 4);
    TriShade& s = shade[slot];
    s.v0 = float4{p[0], p[1], p[2], matf};
    s.e1 = t.e1;
    s.e2 = t.e2;
    s.n0 = float4{n[0], n[1], n[2], 0.0f};
    s.n1 = float4{n[3], n[4], n[5], 0.0f};
    s.n2 = float4{n[6], n[7], n[8], 0.0f};
    s.unused[0] = s.unused[1] = float4{0.0f, 0.0f, 0.0f, 0.0f};
  });

  // ---- small tables
  SceneTables& st = ctx->h_scene;
  std::memset(&st, 0, sizeof(st));
  st.n_tris = (int)sd->n_tris;
  st.n_rects = (int)sd->n_rects;
  st.n_planes = (int)sd->n_planes;
  st.n_spheres = (int)sd->n_spheres;
  st.n_lights = (int)sd->n_lights;
  st.n_materials = (int)sd->n_materials;
  st.rect_aabb_mode = (int)sd->rect_aabb_mode;
  st.infinite_light = -1;
  for (uint32_t i = 0; i < sd->n_rects; ++i) {
    const rrt_rect& r = sd->rects[i];
    DeviceRect& q = st.rects[i];
    double n[3], nn[3], ss[3], ts[3];
    cross3(r.side_a, r.side_b, n);
    const double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (int k = 0; k < 3; ++k) nn[k] = n[k] / l;
    coordinate_system_host(nn, ss, ts);
    double aa = 0, bb = 0;
    for (int k = 0; k < 3; ++k) {
      q.pos[k] = (float)r.pos[k]; q.a[k] = (float)r.side_a[k]; q.b[k] = (float)r.side_b[k];
      q.normal[k] = (float)nn[k]; q.ss[k] = (float)ss[k]; q.ts[k] = (float)ts[k];
      aa += r.side_a[k] * r.side_a[k]; bb += r.side_b[k] * r.side_b[k];
      const double opp = r.pos[k] + r.side_a[k] + r.side_b[k];  // objects/rectangle.rs:138-145
      q.bmin[k] = (float)std::min(r.pos[k], opp);
      q.bmax[k] = (float)std::max(r.pos[k], opp);
    }
    q.inv_aa = (float)(1.0 / aa);
    q.inv_bb = (float)(1.0 / bb);
    q.material = (int)r.material;
    q.light = r.light;
  }
  for (uint32_t i = 0; i < sd->n_planes; ++i) {
    const rrt_plane& pl = sd->planes[i];
    DevicePlane& d = st.planes[i];
    for (int k = 0; k < 3; ++k) { d.pos[k] = (float)pl.pos[k]; d.normal[k] = (float)pl.normal[k]; }
    const double flip = pl.normal[1] < 0.0 ? -1.0 : 1.0;  // face_forward(normal, (0,1,0)), surface_interaction.rs:57
    for (int k = 0; k < 3; ++k) d.ng[k] = (float)(flip * pl.normal[k]);
    d.material = (int)pl.material;
  }
  for (uint32_t i = 0; i < sd->n_spheres; ++i) {
    const rrt_sphere& s = sd->spheres[i];
    for (int k = 0; k < 3; ++k) st.spheres[i].center[k] = (float)s.center[k];
    st.spheres[i].radius = (float)s.radius;
    st.spheres[i].material = (int)s.material;
  }
  for (uint32_t i = 0; i < sd->n_lights; ++i) {
    const rrt_light& l = sd->lights[i];
    DeviceLight& d = st.lights[i];
    d.kind = (int)l.kind;
    d.rect = -1;
    for (int k = 0; k < 3; ++k) {
      d.intensity[k] = (float)l.intensity[k];
      d.v0[k] = (float)l.v0[k]; d.v1[k] = (float)l.v1[k]; d.v2[k] = (float)l.v2[k];
    }
    if (l.kind == RRT_LIGHT_AREA) {
      double n[3];
      cross3(l.v1, l.v2, n);
      norm3(n, d.normal);
      const double aa = l.v1[0] * l.v1[0] + l.v1[1] * l.v1[1] + l.v1[2] * l.v1[2];
      const double bb = l.v2[0] * l.v2[0] + l.v2[1] * l.v2[1] + l.v2[2] * l.v2[2];
      const double ab = l.v1[0] * l.v2[0] + l.v1[1] * l.v2[1] + l.v1[2] * l.v2[2];
      d.area = (float)bb;  // |side_b|^2, rectangle.rs:131-133
      d.skew_a = (float)(ab / aa);
      d.skew_b = (float)(ab / bb);
      for (uint32_t r = 0; r < sd->n_rects; ++r)
        if (sd->rects[r].light == (int32_t)i) d.rect = (int)r;
    } else if (l.kind == RRT_LIGHT_DISTANT) {
      norm3(l.v0, d.v0);  // DistantLight::new normalises (lights/distant.rs:81)
      for (int k = 0; k < 3; ++k) st.distant_env[k] += (float)l.intensity[k];
    } else if (l.kind == RRT_LIGHT_INFINITE) {
      if (!sd->env_rgb8 || !sd->env_w || !sd->env_h) {
        free_bvh(&bvh);
        free_scene(ctx);
        return fail(ctx, RRT_E_INVALID, "infinite light without environment texels");
      }
      st.infinite_light = (int)i;
    } else if (l.kind != RRT_LIGHT_POINT) {
      free_bvh(&bvh);
      free_scene(ctx);
      return fail(ctx, RRT_E_INVALID, "unknown light kind");
    }
  }
  st.has_env = st.infinite_light >= 0 ? 1 : 0;
  {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < sd->n_rects; ++i)
      for (int corner = 0; corner < 4; ++corner)
        for (int k = 0; k < 3; ++k) {
          const double v = sd->rects[i].pos[k] + ((corner & 1) ? sd->rects[i].side_a[k] : 0.0) + ((corner & 2) ? sd->rects[i].side_b[k] : 0.0);
          lo[k] = std::min(lo[k], v);
          hi[k] = std::max(hi[k], v);
        }
    for (uint32_t i = 0; i < sd->n_spheres; ++i)
      for (int k = 0; k < 3; ++k) {
        lo[k] = std::min(lo[k], sd->spheres[i].center[k] - std::fabs(sd->spheres[i].radius));
        hi[k] = std::max(hi[k], sd->spheres[i].center[k] + std::fabs(sd->spheres[i].radius));
      }
    for (int k = 0; k < 3; ++k) {
      // padded by far more than f32 rounding of a hit point on the shapes
      const double pad = 1e-3 * std::max(1.0, std::max(std::fabs(lo[k]), std::fabs(hi[k])));
      const bool bounded = sd->n_planes == 0 && lo[k] <= hi[k];
      st.others_bmin[k] = sd->n_planes ? -INFINITY : (bounded ? (float)(lo[k] - pad) : INFINITY);
      st.others_bmax[k] = sd->n_planes ? INFINITY : (bounded ? (float)(hi[k] + pad) : -INFINITY);
    }
  }
  st.env_w = (int)sd->env_w;
  st.env_h = (int)sd->env_h;
  std::vector<DeviceMaterial> mats(sd->n_materials);
  ctx->scene_has_specular = false;
  for (uint32_t i = 0; i < sd->n_materials; ++i) {
    mats[i] = digest_material(sd->materials[i]);
    if (mats[i].has_trans || mats[i].spec_kind == SPEC_MIRROR_DIELECTRIC || mats[i].spec_kind == SPEC_MIRROR_NOOP) ctx->scene_has_specular = true;
  }

  const uint32_t max_depth = device_build ? device_depth : bvh.max_depth;
  ctx->n_nodes = device_build ? device_nodes : bvh.n_nodes;
  ctx->grid = device_build ? device_grid : bvh.grid;
  ctx->bvh_built_on_device = device_build;
  // every 4-wide node on a ray's way down leaves at most 3 entries on the traversal stack
  if (3ull * max_depth + 1 > (uint64_t)kStackDepth) {
    free_bvh(&bvh);
    free_scene(ctx);
    return fail(ctx, RRT_E_UNSUPPORTED, "hierarchy too deep for the traversal stack");
  }
  ctx->bvh_max_depth = max_depth;

  // ---- upload
  rc = RRT_OK;
  if (!device_build) {
    rc = dev_alloc(ctx, &ctx->d_nodes, bvh.n_nodes);
    if (rc == RRT_OK) rc = dev_alloc(ctx, &ctx->d_tris, sd->n_tris);
    if (rc == RRT_OK) rc = dev_alloc(ctx, &ctx->d_tri_shade, sd->n_tris);
  }
  if (rc == RRT_OK) rc = dev_alloc(ctx, &ctx->d_materials, sd->n_materials);
  if (rc == RRT_OK) rc = dev_alloc(ctx, &ctx->d_scene, 1);
  if (rc == RRT_OK) rc = dev_alloc(ctx, &ctx->d_tri_pos, (size_t)sd->n_tris * 9);
  if (rc == RRT_OK) rc = dev_alloc(ctx, &ctx->d_exact, 1);
  if (rc == RRT_OK && st.has_env) rc = dev_alloc(ctx, &ctx->d_env, (size_t)sd->env_w * sd->env_h * 4);
  if (rc != RRT_OK) { free_bvh(&bvh); free_scene(ctx); return rc; }
  hipError_t e = hipSuccess;
  if (!device_build) e = hipMemcpy(ctx->d_nodes, bvh.nodes, sizeof(BvhNode) * (size_t)bvh.n_nodes, hipMemcpyHostToDevice);
  if (!device_build && e == hipSuccess && sd->n_tris) e = hipMemcpy(ctx->d_tris, tris, sizeof(TriIsect) * (size_t)sd->n_tris, hipMemcpyHostToDevice);
  if (!device_build && e == hipSuccess && sd->n_tris) e = hipMemcpy(ctx->d_tri_shade, shade, sizeof(TriShade) * (size_t)sd->n_tris, hipMemcpyHostToDevice);
  if (e == hipSuccess && sd->n_materials) e = hipMemcpy(ctx->d_materials, mats.data(), sizeof(DeviceMaterial) * mats.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ctx->d_scene, &st, sizeof(st), hipMemcpyHostToDevice);
  if (e == hipSuccess && sd->n_tris) e = hipMemcpy(ctx->d_tri_pos, sd->tri_pos, sizeof(float) * 9 * (size_t)sd->n_tris, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    ShapesF64 ex;
    std::memset(&ex, 0, sizeof(ex));
    for (uint32_t i = 0; i < sd->n_rects; ++i) {
      double n[3];
      cross3(sd->rects[i].side_a, sd->rects[i].side_b, n);
      const double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      for (int k = 0; k < 3; ++k) { ex.rect_pos[i][k] = sd->rects[i].pos[k]; ex.rect_normal[i][k] = n[k] / l; }
    }
    for (uint32_t i = 0; i < sd->n_planes; ++i)
      for (int k = 0; k < 3; ++k) { ex.plane_pos[i][k] = sd->planes[i].pos[k]; ex.plane_normal[i][k] = sd->planes[i].normal[k]; }
    for (uint32_t i = 0; i < sd->n_spheres; ++i) {
      for (int k = 0; k < 3; ++k) ex.sphere_center[i][k] = sd->spheres[i].center[k];
      ex.sphere_radius[i] = sd->spheres[i].radius;
    }
    e = hipMemcpy(ctx->d_exact, &ex, sizeof(ex), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && st.has_env) {
    // one 4-byte texel per lookup instead of three byte loads
    std::vector<uint8_t> rgba((size_t)sd->env_w * sd->env_h * 4);
    for (size_t i = 0; i < (size_t)sd->env_w * sd->env_h; ++i) {
      rgba[4 * i + 0] = sd->env_rgb8[3 * i + 0];
      rgba[4 * i + 1] = sd->env_rgb8[3 * i + 1];
      rgba[4 * i + 2] = sd->env_rgb8[3 * i + 2];
      rgba[4 * i + 3] = 0;
    }
    e = hipMemcpy(ctx->d_env, rgba.data(), rgba.size(), hipMemcpyHostToDevice);
  }
  free_bvh(&bvh);
  if (e != hipSuccess) { free_scene(ctx); return fail(ctx, RRT_E_DEVICE, std::string("scene upload: ") + hipGetErrorString(e)); }
  ctx->bvh_bytes = sizeof(BvhNode) * (uint64_t)ctx->n_nodes + (sizeof(TriIsect) + sizeof(TriShade)) * (uint64_t)sd->n_tris;
  ctx->n_tris = sd->n_tris;
  ctx->have_scene = true;
  return RRT_OK;
}
}  // namespace

int rrt_upload_scene(rrt_ctx* ctx, const rrt_scene_desc* sd) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  try {
    return upload_scene_impl(ctx, sd);
  } catch (...) {  // std::bad_alloc from a host container: nothing may unwind across the C boundary
    free_scene(ctx);
    return fail(ctx, RRT_E_NOMEM, "rrt_upload_scene: out of host memory");
  }
}

int rrt_set_camera(rrt_ctx* ctx, const double raster_to_camera[16], const double camera_to_world[16], double aperture,
                   double focal_distance) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  if (!raster_to_camera || !camera_to_world) return fail(ctx, RRT_E_INVALID, "null matrix");
  std::memcpy(ctx->camera.raster_to_camera, raster_to_camera, sizeof(double) * 16);
  std::memcpy(ctx->camera.camera_to_world, camera_to_world, sizeof(double) * 16);
  ctx->camera.aperture = aperture;
  ctx->camera.focal_distance = focal_distance;
  for (int i = 0; i < 12; ++i) ctx->camera.camera_to_world_f[i] = (float)camera_to_world[i];
  ctx->camera.aperture_f = (float)aperture;
  ctx->camera.focal_distance_f = (float)focal_distance;
  {
    const double* m = ctx->camera.raster_to_camera;
    ctx->camera.const_inv_w = (m[12] == 0.0 && m[13] == 0.0 && m[15] != 0.0) ? 1.0 / m[15] : 0.0;
  }
  ctx->have_camera = true;
  return RRT_OK;
}

namespace {
size_t aov_bytes(const rrt_ctx* ctx) { return sizeof(float) * 8 * (size_t)ctx->aov_w * ctx->aov_h; }

// The first-hit buffers follow the film's size; allocated (and zeroed) on first need.
int ensure_aov(rrt_ctx* ctx) {
  if (ctx->d_aov && ctx->aov_w == ctx->film_w && ctx->aov_h == ctx->film_h) return RRT_OK;
  if (ctx->d_aov) (void)hipFree(ctx->d_aov);
  ctx->d_aov = nullptr;
  ctx->aov_w = ctx->film_w;
  ctx->aov_h = ctx->film_h;
  int rc = dev_alloc(ctx, &ctx->d_aov, (size_t)ctx->aov_w * ctx->aov_h * 8);
  if (rc != RRT_OK) return rc;
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_aov, 0, aov_bytes(ctx), ctx->stream));
  return RRT_OK;
}
}  // namespace

int rrt_film_alloc(rrt_ctx* ctx, uint32_t width, uint32_t height) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  if (!width || !height) return fail(ctx, RRT_E_INVALID, "empty film");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;
  if (ctx->film_owned) (void)hipFree(ctx->d_film);
  ctx->d_film = nullptr;
  ctx->film_owned = false;
  rc = dev_alloc(ctx, &ctx->d_film, (size_t)width * height * 4);
  if (rc != RRT_OK) return rc;
  ctx->film_owned = true;
  ctx->film_w = width;
  ctx->film_h = height;
  return rrt_film_clear(ctx);
}

int rrt_film_bind(rrt_ctx* ctx, float* device_ptr, uint32_t width, uint32_t height) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  if (!device_ptr || !width || !height) return fail(ctx, RRT_E_INVALID, "bad film binding");
  if (((uintptr_t)device_ptr & 15u) != 0) return fail(ctx, RRT_E_INVALID, "film pointer must be 16-byte aligned");
  if (ctx->film_owned) {
    (void)hipSetDevice(ctx->device);
    (void)hipFree(ctx->d_film);
  }
  ctx->d_film = device_ptr;
  ctx->film_owned = false;
  ctx->film_w = width;
  ctx->film_h = height;
  return RRT_OK;
}

int rrt_film_clear(rrt_ctx* ctx) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  if (!ctx->d_film) return fail(ctx, RRT_E_STATE, "no film");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_film, 0, sizeof(float) * 4 * (size_t)ctx->film_w * ctx->film_h, ctx->stream));
  if (ctx->d_aov && ctx->aov_w == ctx->film_w && ctx->aov_h == ctx->film_h)
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_aov, 0, aov_bytes(ctx), ctx->stream));
  return RRT_OK;
}

int rrt_aov_download(rrt_ctx* ctx, float* host_normal_sum, float* host_albedo_sum) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  if (!host_normal_sum || !host_albedo_sum) return fail(ctx, RRT_E_INVALID, "null destination");
  if (!ctx->d_film) return fail(ctx, RRT_E_STATE, "no film");
  if (!ctx->aovs) return fail(ctx, RRT_E_STATE, "RRT_OPT_AOVS is off");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;
  if ((rc = ensure_aov(ctx)) != RRT_OK) return rc;
  const size_t half = aov_bytes(ctx) / 2;
  HIP_TRY(ctx, hipMemcpyAsync(host_normal_sum, ctx->d_aov, half, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(host_albedo_sum, (const char*)ctx->d_aov + half, half, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return RRT_OK;
}

void* rrt_aov_device_ptr(rrt_ctx* ctx) {
  if (!ctx || !ctx->aovs || !ctx->d_film || bind_device(ctx) != RRT_OK || ensure_aov(ctx) != RRT_OK) return nullptr;
  return ctx->d_aov;
}

int rrt_film_download(rrt_ctx* ctx, float* host) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  if (!host) return fail(ctx, RRT_E_INVALID, "null destination");
  if (!ctx->d_film) return fail(ctx, RRT_E_STATE, "no film");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(host, ctx->d_film, sizeof(float) * 4 * (size_t)ctx->film_w * ctx->film_h,
                              hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return RRT_OK;
}

void* rrt_film_device_ptr(rrt_ctx* ctx) { return ctx ? ctx->d_film : nullptr; }

namespace {
// src/film.rs:326-358; GAUSSIAN_ALPHA = 1.5 (src/film.rs:17)
double evaluate_gaussian(double px, double py, double radius, double alpha) {
  const double expv = std::exp(-alpha * radius * radius);
  return std::max(std::exp(-alpha * px * px) - expv, 0.0) * std::max(std::exp(-alpha * py * py) - expv, 0.0);
}
double evaluate_mitchell_1d(double input) {
  const double b = 0.33, c = 0.33;
  const double x = std::fabs(2.0 * input);
  if (x > 1.0)
    return ((-b - 6.0 * c) * x * x * x + (6.0 * b + 30.0 * c) * x * x + (-12.0 * b - 48.0 * c) * x + (8.0 * b + 24.0 * c)) *
           (1.0 / 6.0);
  return ((12.0 - 9.0 * b - 6.0 * c) * x * x * x + (-18.0 + 12.0 * b + 6.0 * c) * x * x + (6.0 - 2.0 * b)) * (1.0 / 6.0);
}
}  // namespace

int rrt_set_filter(rrt_ctx* ctx, uint32_t method, double radius) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  if (method > RRT_FILTER_MITCHELL) return fail(ctx, RRT_E_INVALID, "unknown filter method");
  if (method == RRT_FILTER_NONE) {
    ctx->filter_method = method;
    ctx->filter_radius = 0.0;  // src/film.rs:110-112
    return RRT_OK;
  }
  if (!(radius > 0.0) || radius > 64.0) return fail(ctx, RRT_E_INVALID, "filter radius out of range");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;
  float table[256];
  const int n = 16;  // filter_table_size, src/film.rs:88
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) {
      const double xp = (x + 0.5) * radius / n, yp = (y + 0.5) * radius / n;
      table[y * n + x] = (float)(method == RRT_FILTER_GAUSSIAN
                                     ? evaluate_gaussian(xp, yp, radius, 1.5)
                                     : evaluate_mitchell_1d(xp / radius) * evaluate_mitchell_1d(yp / radius));
    }
  if (!ctx->d_filter_table && (rc = dev_alloc(ctx, &ctx->d_filter_table, 256)) != RRT_OK) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // a render still using the old table
  HIP_TRY(ctx, hipMemcpy(ctx->d_filter_table, table, sizeof(table), hipMemcpyHostToDevice));
  ctx->filter_method = method;
  ctx->filter_radius = radius;
  return RRT_OK;
}

int rrt_film_tonemap(rrt_ctx* ctx, uint8_t* host_rgb8) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  if (!host_rgb8) return fail(ctx, RRT_E_INVALID, "null destination");
  if (!ctx->d_film) return fail(ctx, RRT_E_STATE, "no film");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;
  const size_t n = (size_t)ctx->film_w * ctx->film_h;
  uint8_t* d_rgb = nullptr;
  if ((rc = dev_alloc(ctx, &d_rgb, n * 3)) != RRT_OK) return rc;
  hipLaunchKernelGGL(k_tonemap, dim3(grid_for((uint32_t)n)), dim3(kBlock), 0, ctx->stream,
                     reinterpret_cast<const float4*>(ctx->d_film), (uint32_t)n, d_rgb);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(host_rgb8, d_rgb, n * 3, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_rgb);
  if (e != hipSuccess) return fail(ctx, RRT_E_DEVICE, std::string("tone map: ") + hipGetErrorString(e));
  return RRT_OK;
}

namespace {
int render_impl(rrt_ctx* ctx, const rrt_settings* s, uint32_t sample_begin, uint32_t sample_end) {
  if (!s) return fail(ctx, RRT_E_INVALID, "null settings");
  if (!ctx->have_scene) return fail(ctx, RRT_E_STATE, "no scene uploaded");
  if (!ctx->have_camera) return fail(ctx, RRT_E_STATE, "no camera set");
  if (!ctx->d_film) return fail(ctx, RRT_E_STATE, "no film");
  if (s->width != ctx->film_w || s->height != ctx->film_h) return fail(ctx, RRT_E_INVALID, "settings do not match the film size");
  if (s->crop_x0 > s->crop_x1 || s->crop_y0 > s->crop_y1 || s->crop_x1 > s->width || s->crop_y1 > s->height)
    return fail(ctx, RRT_E_INVALID, "crop window outside the film");
  if (sample_end < sample_begin) return fail(ctx, RRT_E_INVALID, "empty or reversed sample range");
  if ((uint64_t)s->width * s->height > 0xffffffffull) return fail(ctx, RRT_E_UNSUPPORTED, "film too large");
  const uint32_t win_w = s->crop_x1 - s->crop_x0, win_h = s->crop_y1 - s->crop_y0;
  const uint32_t n_samples = sample_end - sample_begin;
  if (win_w == 0 || win_h == 0 || n_samples == 0 || s->depth_limit == 0) return RRT_OK;
  if (s->depth_limit > 4096) return fail(ctx, RRT_E_UNSUPPORTED, "depth_limit above 4096");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;

  const uint32_t tiles_x = (win_w + 7) / 8, tiles_y = (win_h + 7) / 8;
  const uint64_t total_tiles = (uint64_t)tiles_x * tiles_y;
  // Pool shape: as many slots per pixel as fit, up to one per sample (all of a pixel's slots are
  // consecutive), and if one slot per pixel of the whole window does not fit, tile ranges.
  // The pool is as large as memory allows: a pass has a fixed cost -- the dozen last iterations
  // with few paths left, each waiting for its longest rays -- and wide launches keep the rays of
  // a wave together (one pixel, one bounce).  Measured at 1080p, 256 spp: 1.6 G samples/s with 16
  // slots per pixel, 2.4 G with 64, 3.2 G with 256.
  const bool need_mis = ctx->h_scene.has_env != 0;
  // One sample per slot (nothing regenerates) needs less state per slot: sized for that first, and
  // for regenerating slots if the samples then do not all fit.
  const bool filtered = ctx->filter_method != RRT_FILTER_NONE;
  uint32_t k = 0, tiles_per_pass = 0;
  bool regenerating = false;
  for (int attempt = filtered ? 1 : 0; attempt < 2; ++attempt) {
    const bool regenerates = attempt == 1;
    regenerating = regenerates;
    uint64_t max_slots = ctx->max_slots;
    if (max_slots == 0) {
      size_t free_bytes = 0, total_bytes = 0;
      HIP_TRY(ctx, hipMemGetInfo(&free_bytes, &total_bytes));
      const double have = (double)free_bytes + (double)state_bytes(ctx->state_slots, ctx->state_has_mis, ctx->state_regenerates);
      max_slots = (uint64_t)(have * kAutoMemoryShare / (double)state_bytes(1u << 20, need_mis, regenerates) * (double)(1u << 20));
      max_slots = std::min<uint64_t>(std::max<uint64_t>(max_slots, 1u << 16), kSlotLimit);
    }
    bool fits = true;
    for (;;) {
      k = ctx->batch_samples ? ctx->batch_samples
                             : (uint32_t)std::min<uint64_t>(kMaxSlotsPerPixel, std::max<uint64_t>(1, max_slots / (total_tiles * 64)));
      k = std::min(k, n_samples);
      // Not all samples at once: with at least kMinRoundSamples slots per pixel the samples are rendered
      // in rounds of one sample per slot (measured at 1080p, 256 spp: four rounds of 64 take 129 ms, 64
      // slots per pixel working through four samples each 159 ms; at 16 the two are level); below that,
      // or with an explicit RRT_OPT_BATCH_SAMPLES, the slots regenerate.
      if (!regenerates && k < n_samples) {
        if (ctx->batch_samples || k < kMinRoundSamples) { fits = false; break; }
        const uint32_t rounds = (n_samples + k - 1) / k;
        k = (n_samples + rounds - 1) / rounds;
      }
      tiles_per_pass = (uint32_t)std::min<uint64_t>(total_tiles, std::max<uint64_t>(1, max_slots / (64ull * k)));
      rc = ensure_state(ctx, tiles_per_pass * 64 * k, s->depth_limit, need_mis, regenerates);
      // an automatic size that the allocator cannot deliver after all (fragmentation, another
      // process on the device): try again with half
      if (rc == RRT_E_NOMEM && ctx->max_slots == 0 && max_slots > (1u << 16)) {
        free_state(ctx);
        max_slots /= 2;
        continue;
      }
      if (rc != RRT_OK) return rc;
      break;
    }
    if (fits) break;
  }

  RenderParams p;
  std::memset(&p, 0, sizeof(p));
  p.a = ctx->arrays;
  p.nodes = ctx->d_nodes;
  p.grid = ctx->grid;
  p.tris = ctx->d_tris;
  p.tri_shade = ctx->d_tri_shade;
  p.materials = ctx->d_materials;
  p.scene = ctx->d_scene;
  p.env = ctx->d_env;
  p.stats = ctx->d_stats;
  p.film = ctx->d_film;
  p.aov = nullptr;
  if (ctx->aovs) {
    if ((rc = ensure_aov(ctx)) != RRT_OK) return rc;
    p.aov = ctx->d_aov;
  }
  p.seed = s->seed;
  p.width = s->width;
  p.height = s->height;
  p.win_x0 = s->crop_x0;
  p.win_y0 = s->crop_y0;
  p.win_w = win_w;
  p.win_h = win_h;
  p.tiles_x = tiles_x;
  p.div_tiles_x = make_fast_div(tiles_x);
  p.depth_limit = s->depth_limit;
  p.clamp = (float)s->clamp;
  p.filter_table = ctx->filter_method != RRT_FILTER_NONE ? ctx->d_filter_table : nullptr;
  p.filter_radius = ctx->filter_radius;
  p.escape_lit = (ctx->h_scene.has_env || ctx->h_scene.distant_env[0] != 0.0f || ctx->h_scene.distant_env[1] != 0.0f ||
                  ctx->h_scene.distant_env[2] != 0.0f) ? 1u : 0u;
  // rounds of one sample per slot (see above), or one round in which the slots take every k-th sample
  const bool rounds = !regenerating;
  for (uint64_t s0 = sample_begin; s0 < sample_end; s0 += rounds ? k : n_samples) {
    const uint32_t s1 = rounds ? (uint32_t)std::min<uint64_t>(sample_end, s0 + k) : sample_end;
    const uint32_t per_pixel = rounds ? s1 - (uint32_t)s0 : k;
    for (uint64_t t0 = 0; t0 < total_tiles; t0 += tiles_per_pass) {
      p.tile_begin = (uint32_t)t0;
      p.n_tiles = (uint32_t)std::min<uint64_t>(tiles_per_pass, total_tiles - t0);
      p.sample_begin = (uint32_t)s0;
      p.sample_end = s1;
      p.n_samples = per_pixel;
      p.div_samples = make_fast_div(per_pixel);
      p.n_slots = p.n_tiles * 64 * per_pixel;
      if ((rc = launch_pass(ctx, p)) != RRT_OK) return rc;
      // the primary kernel traced one ray for every slot of a pixel inside the window
      for (uint64_t t = t0; t < t0 + p.n_tiles; ++t) {
        const uint32_t tx = (uint32_t)(t % tiles_x), ty = (uint32_t)(t / tiles_x);
        ctx->rays_primary += (uint64_t)std::min(8u, win_w - tx * 8) * std::min(8u, win_h - ty * 8) * per_pixel;
      }
    }
  }
  return RRT_OK;
}
}  // namespace

int rrt_render(rrt_ctx* ctx, const rrt_settings* s, uint32_t sample_begin, uint32_t sample_end) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  try {
    return render_impl(ctx, s, sample_begin, sample_end);
  } catch (...) {  // std::bad_alloc (the list of state allocations, the timing events, an error string)
    return fail_static(ctx, RRT_E_NOMEM, "rrt_render: out of host memory");
  }
}

int rrt_synchronize(rrt_ctx* ctx) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return RRT_OK;
}

int rrt_intersect(rrt_ctx* ctx, const float* org_dir, uint32_t n, int32_t* prim_id, float* t) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  if (!ctx->have_scene) return fail(ctx, RRT_E_STATE, "no scene uploaded");
  if (n == 0) return RRT_OK;
  if (!org_dir || !prim_id || !t) return fail(ctx, RRT_E_INVALID, "null array");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;
  float* d_rays = nullptr;
  int32_t* d_prim = nullptr;
  float* d_t = nullptr;
  rc = dev_alloc(ctx, &d_rays, (size_t)n * 6);
  if (rc == RRT_OK) rc = dev_alloc(ctx, &d_prim, n);
  if (rc == RRT_OK) rc = dev_alloc(ctx, &d_t, n);
  hipError_t e = hipSuccess;
  if (rc == RRT_OK) {
    e = hipMemcpyAsync(d_rays, org_dir, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      e = hipMemsetAsync(ctx->d_work, 0, sizeof(uint32_t), ctx->stream);
      hipLaunchKernelGGL(k_intersect, dim3(std::min(grid_for(n), kTraceGridBlocks)), dim3(kBlock), 0, ctx->stream, ctx->d_nodes,
                         ctx->grid, ctx->d_tris, ctx->d_tri_pos, ctx->d_exact, ctx->d_scene, d_rays, n, d_prim, d_t, ctx->d_work);
      if (e == hipSuccess) e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(prim_id, d_prim, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(t, d_t, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  (void)hipFree(d_rays);
  (void)hipFree(d_prim);
  (void)hipFree(d_t);
  if (rc != RRT_OK) return rc;
  if (e != hipSuccess) return fail(ctx, RRT_E_DEVICE, std::string("rrt_intersect: ") + hipGetErrorString(e));
  return RRT_OK;
}

int rrt_intersect_noalloc(rrt_ctx* ctx, const float* org_dir, uint32_t n, int32_t* prim_id, float* t) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  if (!ctx->have_scene) return fail(ctx, RRT_E_STATE, "no scene uploaded");
  if (n == 0) return RRT_OK;
  if (!org_dir || !prim_id || !t) return fail(ctx, RRT_E_INVALID, "null array");
  if (n > RRT_NOALLOC_MAX_RAYS)
    return fail(ctx, RRT_E_UNSUPPORTED, "rrt_intersect_noalloc: " + std::to_string(n) + " rays exceed RRT_NOALLOC_MAX_RAYS = " +
                                            std::to_string(RRT_NOALLOC_MAX_RAYS) + "; split the batch or call rrt_intersect");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;
  // everything below addresses the device's copy of the array through the pointer looked up in rrt_create
  unsigned char* const base = ctx->scratch;
  uint32_t* const d_cursor = reinterpret_cast<uint32_t*>(base);
  float* const d_rays = reinterpret_cast<float*>(base + kScratchRaysAt);
  int32_t* const d_prim = reinterpret_cast<int32_t*>(base + kScratchPrimAt);
  float* const d_t = reinterpret_cast<float*>(base + kScratchTAt);
  // the array keeps the previous call's contents: the cursor must start at zero; the staging areas
  // are overwritten (rays) or fully written by the kernel (ids, distances) for the n entries read back
  HIP_TRY(ctx, hipMemsetAsync(d_cursor, 0, kScratchCursorBytes, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_rays, org_dir, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_intersect, dim3(std::min(grid_for(n), kTraceGridBlocks)), dim3(kBlock), 0, ctx->stream, ctx->d_nodes,
                     ctx->grid, ctx->d_tris, ctx->d_tri_pos, ctx->d_exact, ctx->d_scene, d_rays, n, d_prim, d_t, d_cursor);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(prim_id, d_prim, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(t, d_t, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return RRT_OK;
}

int rrt_occluded(rrt_ctx* ctx, const float* org_dir_tmax, uint32_t n, uint8_t* hit) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  if (!ctx->have_scene) return fail(ctx, RRT_E_STATE, "no scene uploaded");
  if (n == 0) return RRT_OK;
  if (!org_dir_tmax || !hit) return fail(ctx, RRT_E_INVALID, "null array");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;
  float* d_rays = nullptr;
  uint8_t* d_hit = nullptr;
  rc = dev_alloc(ctx, &d_rays, (size_t)n * 7);
  if (rc == RRT_OK) rc = dev_alloc(ctx, &d_hit, n);
  hipError_t e = hipSuccess;
  if (rc == RRT_OK) {
    e = hipMemcpyAsync(d_rays, org_dir_tmax, sizeof(float) * 7 * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      e = hipMemsetAsync(ctx->d_work, 0, sizeof(uint32_t), ctx->stream);
      hipLaunchKernelGGL(k_occluded, dim3(std::min(grid_for(n), kTraceGridBlocks)), dim3(kBlock), 0, ctx->stream, ctx->d_nodes,
                         ctx->grid, ctx->d_tris, ctx->d_scene, d_rays, n, d_hit, ctx->d_work);
      if (e == hipSuccess) e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hit, d_hit, (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  }
  (void)hipFree(d_rays);
  (void)hipFree(d_hit);
  if (rc != RRT_OK) return rc;
  if (e != hipSuccess) return fail(ctx, RRT_E_DEVICE, std::string("rrt_occluded: ") + hipGetErrorString(e));
  return RRT_OK;
}

int rrt_set_option(rrt_ctx* ctx, int option, int64_t value) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  switch (option) {
    case RRT_OPT_TIMING: ctx->timing = value != 0; return RRT_OK;
    case RRT_OPT_BATCH_SAMPLES:
      if (value < 0 || value > 4096) return fail(ctx, RRT_E_INVALID, "batch samples out of range");
      ctx->batch_samples = (uint32_t)value;
      return RRT_OK;
    case RRT_OPT_MAX_SLOTS:
      if (value != 0 && (value < 64 || value > (int64_t)kSlotLimit)) return fail(ctx, RRT_E_INVALID, "max slots out of range");
      ctx->max_slots = (uint32_t)value;
      return RRT_OK;
    case RRT_OPT_COUNT_TRAVERSAL: ctx->count_traversal = value != 0; return RRT_OK;
    case RRT_OPT_SIDE_STREAM: ctx->side_stream_allowed = value != 0; return RRT_OK;
    case RRT_OPT_AOVS: ctx->aovs = value != 0; return RRT_OK;
    case RRT_OPT_DEVICE_BVH:
      if (value < 0 || value > 2) return fail(ctx, RRT_E_INVALID, "device BVH option out of range");
      ctx->device_bvh = (int)value;
      return RRT_OK;
    case RRT_OPT_TAIL_PATHS:
      if (value < 0 || value > (int64_t)kSlotLimit) return fail(ctx, RRT_E_INVALID, "tail paths out of range");
      ctx->tail_paths = (uint32_t)value;
      return RRT_OK;
    default: return fail(ctx, RRT_E_INVALID, "unknown option");
  }
}

int rrt_get_stats(rrt_ctx* ctx, rrt_stats* out) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  if (!out) return fail(ctx, RRT_E_INVALID, "null destination");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  DeviceStats ds;
  HIP_TRY(ctx, hipMemcpy(&ds, ctx->d_stats, sizeof(ds), hipMemcpyDeviceToHost));
  std::memset(out, 0, sizeof(*out));
  out->paths = ds.paths;
  out->rays_extend = ds.rays_extend;
  out->rays_shadow = ds.rays_shadow;
  out->path_bounces = ds.path_bounces;
  for (int k = 0; k < 3; ++k) {
    out->nodes_by_kernel[k] = ds.nodes[k];
    out->tris_by_kernel[k] = ds.tris[k];
    out->rays_entered_by_kernel[k] = ds.entered[k];
    out->nodes_visited += ds.nodes[k];
    out->tris_tested += ds.tris[k];
  }
  out->ms_primary = ctx->ms[5];
  out->launches_primary = ctx->launches_primary;
  out->rays_primary = ctx->rays_primary;
  out->ms_raygen = ctx->ms[0];
  out->ms_extend = ctx->ms[1];
  out->ms_shade = ctx->ms[2];
  out->ms_connect = ctx->ms[3];
  out->ms_film = ctx->ms[4];
  out->launches_extend = ctx->launches_extend;
  out->ms_tail = ctx->ms[6];
  out->launches_tail = ctx->launches_tail;
  out->bvh_nodes = ctx->n_nodes;
  out->bvh_bytes = ctx->bvh_bytes;
#ifdef RRT_PHASE_CLOCKS
  {
    unsigned long long dbg[96];
    if (hipMemcpyFromSymbol(dbg, HIP_SYMBOL(rrt_phase_dbg), sizeof(dbg)) == hipSuccess) {
      for (int k = 0; k < 96; k += 32) {
        std::fprintf(stderr, "phase clocks (%s): refill %llu (advance %llu begin %llu flush %llu) node %llu leaf %llu | node iters %llu leaf iters %llu | lanes stepping %llu at leaf %llu | refills %llu total %llu | longest ray %llu node steps, %llu rays above 256, %llu above 1024\n",
                     k == 0 ? "continuation rays" : k == 32 ? "visibility rays" : "camera rays", dbg[k + 0], dbg[k + 10], dbg[k + 11], dbg[k + 12], dbg[k + 1], dbg[k + 2], dbg[k + 3], dbg[k + 4], dbg[k + 5], dbg[k + 6], dbg[k + 7], dbg[k + 8], dbg[k + 13], dbg[k + 14], dbg[k + 15]);
        std::fprintf(stderr, "  node iterations by lanes stepping (1-8 ... 57-64):");
        for (int i = 0; i < 8; ++i) std::fprintf(stderr, " %llu", dbg[k + 16 + i]);
        std::fprintf(stderr, "\n  leaf iterations by lanes at a leaf:");
        for (int i = 0; i < 8; ++i) std::fprintf(stderr, " %llu", dbg[k + 24 + i]);
        std::fprintf(stderr, "\n");
      }
      std::memset(dbg, 0, sizeof(dbg));
      (void)hipMemcpyToSymbol(HIP_SYMBOL(rrt_phase_dbg), dbg, sizeof(dbg));
    }
  }
#endif
  return RRT_OK;
}

int64_t rrt_bvh_selfcheck(const float* tri_pos, uint32_t n_tris, uint32_t* n_nodes, uint32_t* max_depth) {
  if (!tri_pos && n_tris) return RRT_E_INVALID;
  thread_switch_from_environment();
  BvhBuildResult bvh;
  if (!build_bvh(tri_pos, n_tris, 0, &bvh)) return RRT_E_NOMEM;
  if (n_nodes) *n_nodes = bvh.n_nodes;
  if (max_depth) *max_depth = bvh.max_depth;
  uint64_t bad = 0;
  bool ok = true;
  try {
    bad = validate_bvh(tri_pos, n_tris, bvh);
  } catch (...) {  // its table of visited triangles
    ok = false;
  }
  free_bvh(&bvh);
  if (!ok) return RRT_E_NOMEM;
  return (int64_t)std::min<uint64_t>(bad, (uint64_t)INT64_MAX);
}

int64_t rrt_bvh_check_uploaded(rrt_ctx* ctx, const float* tri_pos, uint32_t n_tris, uint32_t* n_nodes, uint32_t* max_depth,
                               int* built_on_device, double* build_seconds) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  if (!ctx->have_scene) return fail(ctx, RRT_E_STATE, "no scene uploaded");
  if ((!tri_pos && n_tris) || n_tris != ctx->n_tris) return fail(ctx, RRT_E_INVALID, "triangle array does not match the uploaded scene");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;
  if (n_nodes) *n_nodes = ctx->n_nodes;
  if (max_depth) *max_depth = ctx->bvh_max_depth;
  if (built_on_device) *built_on_device = ctx->bvh_built_on_device ? 1 : 0;
  if (build_seconds) *build_seconds = ctx->bvh_build_seconds;
  BvhBuildResult b;
  b.n_nodes = ctx->n_nodes;
  b.max_depth = ctx->bvh_max_depth;
  b.grid = ctx->grid;
  b.nodes = (BvhNode*)std::malloc(sizeof(BvhNode) * std::max<size_t>(ctx->n_nodes, 1));
  b.order = (uint32_t*)std::malloc(sizeof(uint32_t) * std::max<size_t>(n_tris, 1));
  TriIsect* recs = (TriIsect*)std::malloc(sizeof(TriIsect) * std::max<size_t>(n_tris, 1));
  int64_t result = RRT_E_NOMEM;
  if (b.nodes && b.order && recs) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(b.nodes, ctx->d_nodes, sizeof(BvhNode) * (size_t)ctx->n_nodes, hipMemcpyDeviceToHost);
    if (e == hipSuccess && n_tris) e = hipMemcpy(recs, ctx->d_tris, sizeof(TriIsect) * (size_t)n_tris, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
      result = fail(ctx, RRT_E_DEVICE, std::string("rrt_bvh_check_uploaded: ") + hipGetErrorString(e));
    } else {
      uint64_t bad = 0;
      for (uint32_t i = 0; i < n_tris; ++i) {
        uint32_t id;
        std::memcpy(&id, &recs[i].v0.w, 4);
        if (id >= n_tris) { ++bad; id = 0; }
        b.order[i] = id;
        // the record must hold the triangle it names
        const float* p = tri_pos + (size_t)id * 9;
        if (recs[i].v0.x != p[0] || recs[i].v0.y != p[1] || recs[i].v0.z != p[2] || recs[i].e1.x != p[3] - p[0] ||
            recs[i].e2.z != p[8] - p[2])
          ++bad;
      }
      try {
        bad += validate_bvh(tri_pos, n_tris, b);
        result = (int64_t)std::min<uint64_t>(bad, (uint64_t)INT64_MAX);
      } catch (...) {
        result = RRT_E_NOMEM;
      }
    }
  }
  std::free(recs);
  free_bvh(&b);
  return result;
}

int rrt_reset_stats(rrt_ctx* ctx) {
  if (!ctx) return fail(nullptr, RRT_E_INVALID, "null context");
  int rc = bind_device(ctx);
  if (rc != RRT_OK) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemset(ctx->d_stats, 0, sizeof(DeviceStats)));
  for (double& m : ctx->ms) m = 0.0;
  ctx->launches_extend = ctx->launches_primary = ctx->rays_primary = ctx->launches_tail = 0;
  return RRT_OK;
}

// RCCL is loaded on first use so that single-GPU users never pay for it.
extern "C++" {
namespace {
typedef void* comm_t;
struct RcclApi {
  int (*comm_init_all)(comm_t*, int, const int*) = nullptr;
  int (*reduce)(const void*, void*, size_t, int, int, int, comm_t, hipStream_t) = nullptr;
  int (*group_start)(void) = nullptr;
  int (*group_end)(void) = nullptr;
  int (*comm_destroy)(comm_t) = nullptr;
  bool ok = false;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;
const RcclApi& rccl_api() {
  std::call_once(g_rccl_once, []() {
    void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return;
    g_rccl.comm_init_all = (int (*)(comm_t*, int, const int*))dlsym(lib, "ncclCommInitAll");
    g_rccl.reduce = (int (*)(const void*, void*, size_t, int, int, int, comm_t, hipStream_t))dlsym(lib, "ncclReduce");
    g_rccl.group_start = (int (*)(void))dlsym(lib, "ncclGroupStart");
    g_rccl.group_end = (int (*)(void))dlsym(lib, "ncclGroupEnd");
    g_rccl.comm_destroy = (int (*)(comm_t))dlsym(lib, "ncclCommDestroy");
    g_rccl.ok = g_rccl.comm_init_all && g_rccl.reduce && g_rccl.group_start && g_rccl.group_end && g_rccl.comm_destroy;
  });
  return g_rccl;
}

// Communicators are kept per list of devices: creating them takes hundreds of milliseconds, a
// reduce of a 1080p film about one.  They go when the last context of the process is destroyed.
struct CommSet {
  std::vector<int> devices;
  std::vector<comm_t> comms;
};
std::mutex g_comm_mutex;           // also serialises reduces: a communicator runs one collective at a time
std::vector<CommSet> g_comm_sets;
void destroy_comm_sets() {
  std::lock_guard<std::mutex> lock(g_comm_mutex);
  if (g_comm_sets.empty()) return;
  int before = -1;
  if (hipGetDevice(&before) != hipSuccess) before = -1;
  for (CommSet& set : g_comm_sets)
    for (size_t i = 0; i < set.comms.size(); ++i) {
      (void)hipSetDevice(set.devices[i]);
      g_rccl.comm_destroy(set.comms[i]);
    }
  g_comm_sets.clear();
  if (before >= 0) (void)hipSetDevice(before);
}

int film_reduce_impl(rrt_ctx** ctxs, int n, int root) {
  const RcclApi& api = rccl_api();
  if (!api.ok) return fail(ctxs[root], RRT_E_UNSUPPORTED, "librccl.so not found or incomplete");
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j)
      if (devs[i] == devs[j]) return fail(ctxs[i], RRT_E_INVALID, "two contexts on one device: one rank per GPU");
  std::lock_guard<std::mutex> lock(g_comm_mutex);
  // the calling thread gets its current device back, whichever way this returns
  struct DeviceGuard {
    int device = -1;
    DeviceGuard() { if (hipGetDevice(&device) != hipSuccess) device = -1; }
    ~DeviceGuard() { if (device >= 0) (void)hipSetDevice(device); }
  } device_guard;
  CommSet* set = nullptr;
  for (CommSet& s : g_comm_sets)
    if (s.devices == devs) set = &s;
  if (!set) {
    CommSet fresh;
    fresh.devices = devs;
    fresh.comms.assign(n, nullptr);
    if (api.comm_init_all(fresh.comms.data(), n, devs.data()) != 0) return fail(ctxs[root], RRT_E_DEVICE, "ncclCommInitAll failed");
    g_comm_sets.push_back(std::move(fresh));
    set = &g_comm_sets.back();
  }
  const size_t count = (size_t)ctxs[0]->film_w * ctxs[0]->film_h * 4;
  const int kFloat32 = 7, kSum = 0;  // ncclFloat32, ncclSum (nccl.h: ncclDataType_t, ncclRedOp_t)
  int rc = RRT_OK;
  api.group_start();
  for (int i = 0; i < n; ++i) {
    (void)hipSetDevice(ctxs[i]->device);
    if (api.reduce(ctxs[i]->d_film, ctxs[i]->d_film, count, kFloat32, kSum, root, set->comms[i], ctxs[i]->stream) != 0) rc = RRT_E_DEVICE;
  }
  if (api.group_end() != 0) rc = RRT_E_DEVICE;
  // the first-hit buffers as well, when every context keeps them (not after a failure: the communicators are
  // in an unknown state then)
  bool all_aovs = true;
  for (int i = 0; i < n; ++i)
    all_aovs = all_aovs && ctxs[i]->aovs && ctxs[i]->d_aov && ctxs[i]->aov_w == ctxs[0]->film_w && ctxs[i]->aov_h == ctxs[0]->film_h;
  if (all_aovs && rc == RRT_OK) {
    api.group_start();
    for (int i = 0; i < n; ++i) {
      (void)hipSetDevice(ctxs[i]->device);
      if (api.reduce(ctxs[i]->d_aov, ctxs[i]->d_aov, 2 * count, kFloat32, kSum, root, set->comms[i], ctxs[i]->stream) != 0) rc = RRT_E_DEVICE;
    }
    if (api.group_end() != 0) rc = RRT_E_DEVICE;
  }
  if (rc == RRT_OK)
    for (int i = 0; i < n; ++i) {
      (void)hipSetDevice(ctxs[i]->device);
      if (hipStreamSynchronize(ctxs[i]->stream) != hipSuccess) rc = RRT_E_DEVICE;
    }
  if (rc != RRT_OK) {
    // A communicator that has seen a failed or half-enqueued collective is not used again: the set goes, and
    // the next reduce over these devices makes a fresh one.  (No wait on the streams: a rank that was only
    // partly enqueued would never finish.)
    for (size_t i = 0; i < set->comms.size(); ++i) {
      (void)hipSetDevice(set->devices[i]);
      if (set->comms[i]) api.comm_destroy(set->comms[i]);
    }
    g_comm_sets.erase(g_comm_sets.begin() + (set - g_comm_sets.data()));
    return fail(ctxs[root], rc, "ncclReduce failed");
  }
  return RRT_OK;
}
}  // namespace
}  // extern "C++"

int rrt_film_reduce(rrt_ctx** ctxs, int n, int root) {
  if (!ctxs || n <= 0 || root < 0 || root >= n) return fail(nullptr, RRT_E_INVALID, "bad reduce arguments");
  for (int i = 0; i < n; ++i)
    if (!ctxs[i] || !ctxs[i]->d_film) return fail(ctxs[i], RRT_E_STATE, "context without a film");
  for (int i = 1; i < n; ++i)
    if (ctxs[i]->film_w != ctxs[0]->film_w || ctxs[i]->film_h != ctxs[0]->film_h)
      return fail(ctxs[i], RRT_E_INVALID, "films differ in size");
  // One context: nothing to add.  (A context created with RRT_DEBUG_REDUCE_SINGLE_THROUGH_RCCL in the environment
  // sends even that through a one-rank communicator -- the film comes back unchanged -- so that a machine with one
  // GPU can run the library loading, the communicator cache and the collective.)
  if (n == 1 && !ctxs[0]->debug_reduce_through_rccl) return RRT_OK;
  try {
    return film_reduce_impl(ctxs, n, root);
  } catch (...) {
    return fail_static(ctxs[root], RRT_E_NOMEM, "rrt_film_reduce: out of host memory");
  }
}

const char* rrt_last_error(rrt_ctx* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

}  // extern "C"
