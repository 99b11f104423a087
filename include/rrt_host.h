/*
 * rrt_host.h -- C facade over librrt_host.so, the C++ host mirror of the
 * reference's loader / Camera / Film (rust-raytracer_amd/host/rrt_host.hpp).
 * It exists so that tools without a C++ compiler at hand (the Python tests and
 * bench.py via ctypes) can load a scene folder the way main() does
 * (src/main.rs:236-315) and hand the result to the device library in rrt.h.
 * Nothing here computes radiance.
 */
#ifndef RRT_HOST_H
#define RRT_HOST_H

#include <stdint.h>

#include "rrt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rrth_scene rrth_scene;

/* Error text of the last failed rrth_* call on this thread. */
const char* rrth_last_error(void);

/* Scene::load_from_folder (src/scene.rs:55).  Null on failure. */
rrth_scene* rrth_scene_load(const char* folder);
/* Same lights / environment, mesh given in memory as [n_tris][3][3] floats. */
rrth_scene* rrth_scene_load_with_mesh(const char* folder, const float* positions, const float* normals,
                                      uint64_t n_tris, const rrt_material* material);
void rrth_scene_free(rrth_scene* scene);
/* Fills *out with pointers into the scene (valid until rrth_scene_free). */
int rrth_scene_desc(const rrth_scene* scene, rrt_scene_desc* out);
int rrth_scene_set_rect_aabb_mode(rrth_scene* scene, uint32_t mode);
/* Replaces material `index` (e.g. the microfacet variant of a shipped MTL). */
int rrth_scene_set_material(rrth_scene* scene, uint32_t index, const rrt_material* material);

/* render_settings.yaml as main() reads it (src/main.rs:252-296). */
typedef struct rrth_render_settings {
  uint32_t threads, depth_limit, max_samples;
  uint32_t image_width, image_height, bucket_width, bucket_height;
  uint32_t crop_x0, crop_y0, crop_x1, crop_y1;
  uint32_t filter_method; /* 0 none, 1 gaussian, 2 mitchell */
  uint32_t denoise;
  uint32_t has_focal_distance;
  double clamp;
  double filter_radius;
  double camera_position[3], camera_target[3];
  double fov, aperture, focal_distance;
  double window_scale;
} rrth_render_settings;
int rrth_load_render_settings(const char* folder, rrth_render_settings* out);

/* Camera::new (src/camera.rs:39-101): the two matrices rrt_set_camera takes
 * (row-major) and the effective focal distance. */
int rrth_camera_matrices(const double position[3], const double target[3], double fov, double aperture,
                         int has_focal_distance, double focal_distance, uint32_t image_width, uint32_t image_height,
                         double raster_to_camera[16], double camera_to_world[16], double* focal_distance_out);
/* Camera::generate_ray (src/camera.rs:103-125) with an explicit unit-disk lens point. */
int rrth_camera_generate_ray(const double position[3], const double target[3], double fov, double aperture,
                             int has_focal_distance, double focal_distance, uint32_t image_width,
                             uint32_t image_height, double film_x, double film_y, double lens_x, double lens_y,
                             double origin[3], double direction[3]);

/* Film tone map (src/film.rs:233-249): accumulators float[h][w][4] -> sRGB u8 [h][w][3]. */
int rrth_tonemap(const float* sum_rgbw, uint32_t width, uint32_t height, uint8_t* rgb8);

/* Image output (the reference has none: it only shows a window): 8-bit RGB rows, top to bottom,
 * as a PNG whose image data is stored, not compressed. */
int rrth_write_png(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height);

/* main() without the window: load `folder`, render it on the given devices
 * with render() and write <out_prefix>.ppm / .png / .pfm.  samples <= 0 keeps the
 * file's max_samples.  Returns camera paths per second through *samples_per_sec. */
int rrth_render_folder(const char* folder, const int* devices, int n_devices, uint64_t seed, int samples,
                       const char* out_prefix, double* samples_per_sec);
/* The same, and the two maps the reference's denoise() hands to OIDN (src/denoise.rs:9-23) filled
 * from the devices' first-hit buffers: float [height][width][3] each, per-pixel means of the shading
 * normal and the albedo of what the camera rays hit (either may be null). */
int rrth_render_folder_first_hits(const char* folder, const int* devices, int n_devices, uint64_t seed, int samples,
                                  const char* out_prefix, double* samples_per_sec, float* normal_map,
                                  float* albedo_map);

#ifdef __cplusplus
}
#endif
#endif /* RRT_HOST_H */
