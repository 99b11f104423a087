/*
 * rrt.h -- C ABI of librrt_hip.so, the MI355X (gfx950) path-tracing core.
 *
 * The reference program has no FFI; its integrator is entered through one Rust
 * call, renderer::render(scene, settings, sampler, camera) (src/renderer.rs:92-97,
 * called from src/main.rs:315).  The entry points below are what a Rust `ffi.rs`
 * for that seam binds: the scene the loader built (src/scene.rs:55-207) goes
 * down once as flat arrays, the camera as the two matrices Camera::new derives
 * (src/camera.rs:39-101), and each call of rrt_render does the work of
 * render_work + trace (src/renderer.rs:184-211, src/tracer.rs:24-139) for a
 * range of sample indices, accumulating the per-pixel sums that
 * Film::write_bucket_pixels / merge_bucket_pixels_to_image_buffer keep
 * (src/film.rs:150-251).
 *
 * Conventions: every function returns 0 on success or a negative RRT_E_* code
 * and never throws or aborts across the boundary; all pointers are caller
 * owned and only read during the call unless stated; a context is driven by
 * one host thread at a time, different contexts may be driven concurrently.
 */
#ifndef RRT_H
#define RRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRT_OK 0
#define RRT_E_INVALID (-1)  /* bad argument (null pointer, zero size, bad enum) */
#define RRT_E_DEVICE (-2)   /* a HIP call failed; see rrt_last_error */
#define RRT_E_STATE (-3)    /* call order: no scene / no camera / no film */
#define RRT_E_NOMEM (-4)    /* host or device allocation failed */
#define RRT_E_UNSUPPORTED (-5)

typedef struct rrt_ctx rrt_ctx;

/* Material kinds: src/materials.rs:14-20. */
enum { RRT_MAT_MATTE = 0, RRT_MAT_PLASTIC = 1, RRT_MAT_MIRROR = 2, RRT_MAT_GLASS = 3 };

/* One material.  Field use per kind (constructor argument order of the reference):
 *   MATTE   (materials/matte.rs:17-23):   kd = reflectance, roughness = sigma (raw value)
 *   PLASTIC (materials/plastic.rs:24-33): kd = diffuse, ks = specular, roughness, ior
 *   MIRROR  (materials/mirror.rs:14-17):  kd = reflectance
 *   GLASS   (materials/glass.rs:17-29):   ior, kd = reflection colour, ks = refraction colour */
typedef struct rrt_material {
  uint32_t kind;
  uint32_t _pad;
  double kd[3];
  double ks[3];
  double roughness;
  double ior;
} rrt_material;

/* Light kinds: src/lights.rs:15-21. */
enum { RRT_LIGHT_POINT = 0, RRT_LIGHT_AREA = 1, RRT_LIGHT_DISTANT = 2, RRT_LIGHT_INFINITE = 3 };

/* One light, in scene.lights order (src/scene.rs:72-132: yaml order, the
 * environment light last).
 *   POINT    (lights/point.rs:66-72):   v0 = position
 *   AREA     (lights/area.rs:66-69 over a Rectangle): v0 = position, v1 = side_a, v2 = side_b
 *   DISTANT  (lights/distant.rs:71-84): v0 = direction towards the light (normalised by the library)
 *   INFINITE (lights/infinite_area.rs:113-139): texels come from env_rgb8; intensity unused */
typedef struct rrt_light {
  uint32_t kind;
  uint32_t _pad;
  double intensity[3];
  double v0[3];
  double v1[3];
  double v2[3];
} rrt_light;

/* Rectangle (src/objects/rectangle.rs:17-24). light = index into lights[] of
 * the area light this quad belongs to, or -1. */
typedef struct rrt_rect {
  double pos[3];
  double side_a[3];
  double side_b[3];
  uint32_t material;
  int32_t light;
} rrt_rect;

/* Plane (src/objects/plane.rs:18-23). */
typedef struct rrt_plane {
  double pos[3];
  double normal[3];
  uint32_t material;
  uint32_t _pad;
} rrt_plane;

/* Sphere (src/objects/sphere.rs:15-20). */
typedef struct rrt_sphere {
  double center[3];
  double radius;
  uint32_t material;
  uint32_t _pad;
} rrt_sphere;

/* How the reference's f32 bounding-box pre-filter (bvh crate, leaf AABB of
 * each shape, src/objects/rectangle.rs:136-146) treats quads whose box is flat:
 * STRICT = slab test tmax > max(tmin, 0): a zero-thickness box is never hit;
 * INCLUSIVE = tmax >= max(tmin, 0); OFF = no pre-filter. */
enum { RRT_RECT_AABB_STRICT = 0, RRT_RECT_AABB_INCLUSIVE = 1, RRT_RECT_AABB_OFF = 2 };

/* The flattened scene.  Object ids (as reported by rrt_intersect) follow the
 * order of Scene.objects (src/scene.rs:64-106, 315-325): triangles in OBJ
 * model / face / fan order, then rectangles, then planes, then spheres. */
typedef struct rrt_scene_desc {
  uint32_t n_tris;
  uint32_t n_rects;
  uint32_t n_planes;
  uint32_t n_spheres;
  uint32_t n_materials;
  uint32_t n_lights;
  const float* tri_pos;     /* [n_tris][3 vertices][xyz], f32 as in the OBJ (objects/triangle.rs:59-82) */
  const float* tri_nrm;     /* [n_tris][3 vertices][xyz] vertex normals (objects/triangle.rs:84-107) */
  const uint32_t* tri_mat;  /* [n_tris] material index */
  const rrt_rect* rects;
  const rrt_plane* planes;
  const rrt_sphere* spheres;
  const rrt_material* materials;
  const rrt_light* lights;
  const uint8_t* env_rgb8;  /* [env_h][env_w][3] texels as InfiniteAreaLight::new stores them; may be null */
  uint32_t env_w;
  uint32_t env_h;
  uint32_t rect_aabb_mode;  /* RRT_RECT_AABB_* */
  uint32_t _pad;
} rrt_scene_desc;

/* Settings (src/renderer.rs:24-30) plus the film geometry the integrator needs
 * (src/film.rs:67-74, crop: src/main.rs:263-275).  Pixels outside
 * [crop_x0,crop_x1) x [crop_y0,crop_y1) are not rendered. */
typedef struct rrt_settings {
  uint32_t width;
  uint32_t height;
  uint32_t crop_x0, crop_y0, crop_x1, crop_y1;
  uint32_t depth_limit;
  uint32_t _pad;
  double clamp;   /* <= 0: no clamp (src/tracer.rs:116) */
  uint64_t seed;  /* counter-based RNG seed; replaces the Sobol / thread RNG of src/sampler.rs */
} rrt_settings;

typedef struct rrt_stats {
  uint64_t paths;            /* camera paths started */
  uint64_t rays_extend;      /* closest-hit queries (src/renderer.rs:213) */
  uint64_t rays_shadow;      /* any-hit queries (src/renderer.rs:251) incl. MIS escape rays */
  uint64_t path_bounces;     /* surface interactions shaded */
  double ms_raygen, ms_extend, ms_shade, ms_connect, ms_film; /* kernel time, valid when timing is on; raygen = start kernels,
                                                                 extend = continuation-ray kernels only (see ms_primary) */
  uint64_t launches_extend;  /* extend launches ms_extend covers */
  uint64_t nodes_visited;    /* BVH nodes fetched, counted only while RRT_OPT_COUNT_TRAVERSAL is on */
  uint64_t tris_tested;      /* triangle records fetched, counted likewise */
  uint64_t bvh_nodes;
  uint64_t bvh_bytes;        /* node + triangle bytes resident on the device */
  /* the same counts per traversal kernel: [0] primary (first camera ray of every slot), [1] extend
   * (continuation rays and later camera rays), [2] connect (visibility rays) */
  uint64_t nodes_by_kernel[3], tris_by_kernel[3];
  double ms_primary;         /* kernel time of the primary kernels */
  uint64_t launches_primary;
  uint64_t rays_primary;     /* closest-hit queries made by the primary kernels (part of rays_extend) */
  double ms_tail;            /* kernel time of the tail kernels (RRT_OPT_TAIL_PATHS) */
  uint64_t launches_tail;
  /* rays per traversal kernel (as nodes_by_kernel) that passed the test against the box around all triangles
   * and went into the hierarchy; counted only while RRT_OPT_COUNT_TRAVERSAL is on.  Camera rays that look
   * past the model never get this far. */
  uint64_t rays_entered_by_kernel[3];
} rrt_stats;

/* Number of visible HIP devices, or a negative code. */
int rrt_device_count(void);

/* Creates a context on `device`; null on failure (rrt_last_error(NULL) explains). */
rrt_ctx* rrt_create(int device);
void rrt_destroy(rrt_ctx* ctx);

/* Launch all later work on this hipStream_t (null = the default stream). */
int rrt_set_stream(rrt_ctx* ctx, void* hip_stream);

/* Copies the scene, builds the BVH and uploads both.  Replaces Arc<Scene> +
 * Bvh::build (src/scene.rs:196, src/renderer.rs:98). */
int rrt_upload_scene(rrt_ctx* ctx, const rrt_scene_desc* scene);

/* Camera::new's raster_to_camera and camera_to_world (row-major 4x4,
 * src/camera.rs:60-88), aperture and focal_distance (src/camera.rs:103-125). */
int rrt_set_camera(rrt_ctx* ctx, const double raster_to_camera[16], const double camera_to_world[16],
                   double aperture, double focal_distance);

/* Film accumulator: float [height][width][4] = (sum_r, sum_g, sum_b, sum_weight)
 * per pixel (Pixel.sum_radiance / sum_weight, src/film.rs:46-52), resident on
 * the device.  rrt_film_alloc makes the library own one (zeroed);
 * rrt_film_bind makes it use caller-owned device memory instead (not zeroed). */
int rrt_film_alloc(rrt_ctx* ctx, uint32_t width, uint32_t height);
int rrt_film_bind(rrt_ctx* ctx, float* device_ptr, uint32_t width, uint32_t height);
int rrt_film_clear(rrt_ctx* ctx);
int rrt_film_download(rrt_ctx* ctx, float* host_sum_rgbw);
void* rrt_film_device_ptr(rrt_ctx* ctx);

/* The reconstruction filter of Film::new (src/film.rs:59-132: the `film.filter_method` and
 * `film.filter_radius` keys of render_settings.yaml, src/main.rs:286-287).  With a method other
 * than NONE every finished sample is splatted over the pixels within `radius` of its film position
 * with the weights of the 16 x 16 table (Film::write_bucket_pixels, src/film.rs:169-215); with
 * NONE (the default) it lands in its own pixel with weight 1 and `radius` is ignored.  Applies to
 * the rrt_render calls that follow. */
enum { RRT_FILTER_NONE = 0, RRT_FILTER_GAUSSIAN = 1, RRT_FILTER_MITCHELL = 2 };
int rrt_set_filter(rrt_ctx* ctx, uint32_t method, double radius);

/* The tone map of Film::merge_bucket_pixels_to_image_buffer (src/film.rs:233-249) on the device:
 * sum / weight through the XYZ -> sRGB matrix and the sRGB transfer curve, truncated to 8 bits;
 * a pixel without weight is black.  host_rgb8 receives [height][width][3]. */
int rrt_film_tonemap(rrt_ctx* ctx, uint8_t* host_rgb8);

/* Renders sample indices [sample_begin, sample_end) of every pixel in the crop
 * window and adds them to the film: the pixel x max_samples loop of render_work
 * (src/renderer.rs:191-207) with trace (src/tracer.rs:24) inside.  The work
 * runs on the context's stream; how many rounds of kernels it takes depends on
 * the paths, so the call reads a counter back every few rounds and returns once
 * the last round has been enqueued and the queues are known to be empty.  The
 * film kernel may still be running then: rrt_synchronize waits for it, and work
 * enqueued on the same stream afterwards is ordered behind it. */
int rrt_render(rrt_ctx* ctx, const rrt_settings* settings, uint32_t sample_begin, uint32_t sample_end);
int rrt_synchronize(rrt_ctx* ctx);

/* Sums the film accumulators of n contexts (one per GPU, one process) into
 * ctxs[root] with RCCL.  Multi-process callers reduce the bound film buffer
 * with their own communicator instead. */
int rrt_film_reduce(rrt_ctx** ctxs, int n, int root);  /* (and the first-hit buffers, if every context keeps them) */

/* check_intersect_scene (src/renderer.rs:213-249) for n rays given as host
 * [n][6] = origin xyz, direction xyz: object id of the closest hit (-1 = miss)
 * and its distance.  Which object: the f32 traversal the renderer uses.  The
 * distance: that one object evaluated once more in f64 by the reference's own
 * formulas (objects/triangle.rs:119-181 on the f32 vertices as uploaded,
 * rectangle.rs:58-71, plane.rs:45-57, sphere.rs:116-151), rounded to f32. */
int rrt_intersect(rrt_ctx* ctx, const float* org_dir, uint32_t n, int32_t* prim_id, float* t);

/* rrt_intersect without any allocation.  Has no counterpart in the reference: it is for callers that
 * cannot allocate device memory where they call -- tight serving loops, harnesses that forbid hipMalloc,
 * code that must not disturb a memory pool.  (Not for use inside a HIP graph capture: like rrt_intersect
 * it copies to and from host arrays and waits for the stream.)  Same arguments and results as rrt_intersect, but the staging of
 * the rays, of the results and the kernel's job cursor live in a static device array of the library
 * (2 MiB + 256 B per GPU, held until the process ends) instead of buffers allocated per call, so n is
 * limited: more than RRT_NOALLOC_MAX_RAYS rays are rejected with RRT_E_UNSUPPORTED.  The call makes
 * no hipMalloc / hipFree; with host arrays in pinned memory the copies allocate nothing in the
 * runtime either.  All contexts of one GPU share that GPU's array: calls that overlap in time on one
 * GPU (two contexts, two host threads) race -- serialise them. */
#define RRT_NOALLOC_MAX_RAYS 65536u
int rrt_intersect_noalloc(rrt_ctx* ctx, const float* org_dir, uint32_t n, int32_t* prim_id, float* t);

/* check_intersect_scene_simple (src/renderer.rs:251-275) for n rays given as
 * host [n][7] = origin, direction, max_dist: 1 if anything is hit closer than
 * max_dist. */
int rrt_occluded(rrt_ctx* ctx, const float* org_dir_tmax, uint32_t n, uint8_t* hit);

/* Builds the hierarchy for n_tris triangles (host [n][3][3] positions) exactly as
 * rrt_upload_scene would, on the host alone, and checks it: every triangle inside the
 * (8-bit quantised) box of every node on the way to it, every triangle in exactly one
 * leaf, depth within bounds.  Returns the number of violations (0 = sound) or a negative
 * error code; optionally reports the node count and the depth.  No device is touched.
 * Has no counterpart in the reference (its hierarchy is the bvh crate's): a self-check. */
int64_t rrt_bvh_selfcheck(const float* tri_pos, uint32_t n_tris, uint32_t* n_nodes, uint32_t* max_depth);

/* The same check on the hierarchy the context holds on the device (downloaded for the purpose), whichever
 * way it was built: tri_pos / n_tris must be the triangles of the uploaded scene.  Also checks that every
 * leaf-ordered triangle record holds the triangle it names.  Optionally reports the node count, the depth,
 * whether kernels built it (RRT_OPT_DEVICE_BVH) and how long they took.  Returns the number of violations. */
int64_t rrt_bvh_check_uploaded(rrt_ctx* ctx, const float* tri_pos, uint32_t n_tris, uint32_t* n_nodes, uint32_t* max_depth,
                               int* built_on_device, double* build_seconds);

/* First-hit buffers for a denoiser (the reference's SampleResult carries normal and albedo
 * but leaves them zero, src/tracer.rs:36-37,65-68; its denoiser reads them from the film,
 * src/denoise.rs:9-23).  With RRT_OPT_AOVS on, rrt_render also adds, per pixel, the shading
 * normal and the albedo (diffuse colour, else specular, else transmission colour) of the surface
 * each camera ray hits; rays that escape add nothing.  host arrays are [height][width][4]:
 * (sum nx, sum ny, sum nz, number of hits) and (sum r, sum g, sum b, 0); divide by the film's
 * sum of weights for per-pixel means.  rrt_film_clear clears them too.  rrt_aov_device_ptr: the
 * two arrays back to back on the device, for hosts that reduce them across GPUs themselves. */
int rrt_aov_download(rrt_ctx* ctx, float* host_normal_sum, float* host_albedo_sum);
void* rrt_aov_device_ptr(rrt_ctx* ctx);

/* Options. */
enum {
  RRT_OPT_TIMING = 1,          /* 1: time each kernel with HIP events (the host waits at the end of every pass) */
  RRT_OPT_BATCH_SAMPLES = 2,   /* path slots per pixel, each working through its share of the samples in place (0 = automatic:
                                  one slot per sample if memory allows, else rounds of one sample per slot, and slots that
                                  work through several samples only below 16 slots per pixel) */
  RRT_OPT_MAX_SLOTS = 3,       /* upper bound on paths in flight (sizes the path-state arrays; at most 2^30; 0 = 60 % of the
                                  free device memory) */
  RRT_OPT_COUNT_TRAVERSAL = 4, /* 1: count nodes_visited / tris_tested (slower kernels; for calibration runs) */
  RRT_OPT_AOVS = 6,            /* 1: accumulate first-hit normal and albedo (rrt_aov_download); default 0 */
  RRT_OPT_DEVICE_BVH = 8,      /* how rrt_upload_scene builds the hierarchy: 0 on the host (binned SAH, threads), 1 by kernels on
                                  the device (Morton order + locally-ordered clustering; the triangles go up unsorted and the
                                  leaf-ordered records are written there too), 2 (default) by kernels from 4 Mi triangles on,
                                  where the host build would take seconds.  Replaces Bvh::build (src/scene.rs:196) either way. */
  RRT_OPT_TAIL_PATHS = 7,      /* once at most this many paths are still alive, one kernel runs each of them to its end
                                  (continuation ray, shading, visibility rays, next bounce ...) instead of further rounds of
                                  the separate kernels, each of which would wait for its longest ray; same film.  0: never.
                                  Default 2^21.  Only with one path slot per sample. */
  RRT_OPT_SIDE_STREAM = 5      /* 1 (default): rrt_render runs the visibility rays of an iteration on a stream of its own
                                  beside the next iteration's closest-hit rays and joins it again before it returns
                                  work to the caller's stream; 0: everything on the one stream (per-kernel times in
                                  rrt_stats then do not overlap) */
};
int rrt_set_option(rrt_ctx* ctx, int option, int64_t value);

int rrt_get_stats(rrt_ctx* ctx, rrt_stats* out);
int rrt_reset_stats(rrt_ctx* ctx);

/* Last error text of this context (or of rrt_create when ctx is null). */
const char* rrt_last_error(rrt_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RRT_H */
