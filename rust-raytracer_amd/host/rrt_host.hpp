// Host side of the MI355X path-tracing core: the C++ mirror of the pieces of
// the reference host that sit either side of renderer::render -- scene and
// settings loading (src/scene.rs, src/main.rs:244-302), Camera
// (src/camera.rs), Film (src/film.rs) and the render() entry itself
// (src/renderer.rs:92-182) -- written against the C ABI in include/rrt.h.
// Names and argument meaning follow the reference so that host code and tests
// read like the reference's.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rrt.h"

namespace rrt_host {

struct Vec3 {
  double x = 0, y = 0, z = 0;
};

// 4x4 row-major matrix with the nalgebra operations Camera::new uses.
struct Mat4 {
  double m[16];
  static Mat4 identity();
  static Mat4 scaling(double x, double y, double z);       // Matrix4::new_nonuniform_scaling
  static Mat4 translation(double x, double y, double z);   // Matrix4::new_translation
  Mat4 operator*(const Mat4& o) const;
  bool try_inverse(Mat4* out) const;                        // Matrix4::try_inverse
  Vec3 transform_point(const Vec3& p) const;                // homogeneous divide when w != 0
  Vec3 transform_vector(const Vec3& v) const;               // upper-left 3x3 only
};

// src/renderer.rs:24-30
struct Settings {
  uint32_t thread_count = 1;
  uint32_t depth_limit = 8;
  uint32_t max_samples = 1;
  double clamp = 0.0;
};

// src/film.rs:10-26
enum class FilterMethod { None, Gaussian, Mitchell };
FilterMethod filter_method_from_str(const std::string& s);

// src/film.rs:46-52.  normal / albedo: the reference has the two fields and hands them to its
// denoiser (src/denoise.rs:9-23) but never fills them (src/tracer.rs:36-37); here they hold the sums
// of the device's first-hit buffers when the film keeps them.
struct Pixel {
  double sum_weight = 0;
  double sum_radiance[3] = {0, 0, 0};
  double sum_normal[3] = {0, 0, 0};
  double sum_albedo[3] = {0, 0, 0};
};

// src/film.rs:54-251.  Buckets are the CPU scheduler's unit of work and are
// not needed when a GPU renders whole sample ranges; what remains is the
// accumulator, the tone map and the 8-bit image.
// 8-bit RGB rows, top to bottom, as a PNG with stored (uncompressed) image data.
bool write_png_rgb8(const std::string& path, const uint8_t* rgb, uint32_t width, uint32_t height);

class Film {
 public:
  Film(uint32_t width, uint32_t height, uint32_t bucket_w, uint32_t bucket_h, bool has_crop, uint32_t crop_x0,
       uint32_t crop_y0, uint32_t crop_x1, uint32_t crop_y1, FilterMethod filter_method, double filter_radius);

  uint32_t width() const { return width_; }
  uint32_t height() const { return height_; }
  // Render window [x0,x1) x [y0,y1): the crop if given, else the image (src/film.rs:263-268).
  void render_window(uint32_t* x0, uint32_t* y0, uint32_t* x1, uint32_t* y1) const;
  FilterMethod filter_method() const { return filter_method_; }
  double filter_radius() const { return filter_radius_; }
  const std::vector<double>& filter_table() const { return filter_table_; }

  // Adds device accumulators float[h][w][4] = (sum_rgb, sum_weight) into
  // pixels and refreshes the 8-bit image for every touched pixel: the two
  // halves of merge_bucket_pixels_to_image_buffer (src/film.rs:221-251).
  void merge_accumulators(const float* sum_rgbw);
  void clear();

  // First-hit normal and albedo for a denoiser: when on, render() has the devices collect them and
  // merges them here; first_hit_maps() gives the two float[h][w][3] arrays denoise() builds for OIDN
  // (src/denoise.rs:9-23), as per-pixel means over the samples.
  void keep_first_hits(bool on) { keep_first_hits_ = on; }
  bool keeps_first_hits() const { return keep_first_hits_; }
  void merge_first_hits(const float* normal_sum, const float* albedo_sum);  // device arrays [h][w][4]
  void first_hit_maps(std::vector<float>* normal_map, std::vector<float>* albedo_map) const;

  std::vector<Pixel> pixels;
  std::vector<uint8_t> image_buffer;  // [h][w][3] sRGB

  bool write_ppm(const std::string& path) const;   // 8-bit image
  bool write_png(const std::string& path) const;   // the same image as PNG (stored, not compressed)
  bool write_pfm(const std::string& path) const;   // linear sum_radiance / sum_weight

 private:
  uint32_t width_, height_, bucket_w_, bucket_h_;
  bool has_crop_;
  uint32_t crop_[4];
  FilterMethod filter_method_;
  double filter_radius_;
  std::vector<double> filter_table_;  // 16 x 16 (src/film.rs:87-112)
  bool keep_first_hits_ = false;
};

// Tone map pieces (src/film.rs:360-406).
void xyz_to_srgb(const double xyz[3], double rgb[3]);
void srgb_to_xyz(const double srgb[3], double xyz[3]);
double gamma_correct_srgb(double v);
double inverse_gamma_correct_srgb(double v);
double evaluate_gaussian(double px, double py, double radius, double alpha);
double evaluate_mitchell(double px, double py, double radius);

struct Ray {
  Vec3 point, direction;
};

// src/camera.rs:14-125
class Camera {
 public:
  // aspect_ratio is accepted and ignored, as in the reference (src/camera.rs:42).
  Camera(Vec3 position, Vec3 target, double aspect_ratio, double fov, double aperture, bool has_focal_distance,
         double focal_distance, double screen_min_x, double screen_min_y, double screen_max_x, double screen_max_y,
         uint32_t image_width, uint32_t image_height);

  // Pinhole part of generate_ray (src/camera.rs:103-125); with aperture > 0
  // the lens point is (lens_x, lens_y) * aperture on the unit disk sample.
  Ray generate_ray(double film_x, double film_y, double lens_x = 0.0, double lens_y = 0.0) const;

  Vec3 position, target;
  double fov, aperture, focal_distance;
  Mat4 camera_to_world, camera_to_screen, screen_to_raster, raster_to_screen, raster_to_camera;
};

// The scene as the device wants it: flat arrays in Scene.objects order
// (src/scene.rs:32-37, 64-106, 315-325).
class Scene {
 public:
  // Scene::load_from_folder (src/scene.rs:55).  Throws std::runtime_error
  // where the reference panics (missing files, missing keys, mesh without
  // normals).
  static std::unique_ptr<Scene> load_from_folder(const std::string& folder);
  // As above but reads the mesh from memory instead of world.file (for
  // generated stand-in meshes): positions / normals are [n_tris][3][3].
  static std::unique_ptr<Scene> load_from_folder_with_mesh(const std::string& folder, const float* positions,
                                                           const float* normals, size_t n_tris,
                                                           const rrt_material& material);

  std::vector<float> tri_pos, tri_nrm;
  std::vector<uint32_t> tri_mat;
  std::vector<rrt_rect> rects;
  std::vector<rrt_plane> planes;
  std::vector<rrt_sphere> spheres;
  std::vector<rrt_material> materials;
  std::vector<rrt_light> lights;
  std::vector<uint8_t> env_rgb8;
  uint32_t env_w = 0, env_h = 0;
  uint32_t rect_aabb_mode = RRT_RECT_AABB_STRICT;
  std::vector<std::string> model_names;
  double bg_color[3] = {0.5, 0.5, 0.5};  // src/scene.rs:202, never read by the integrator

  size_t n_tris() const { return tri_mat.size(); }
  rrt_scene_desc desc() const;

  // InfiniteAreaLight::new's texel conversion (src/lights/infinite_area.rs:113-128)
  // of a decoded RGB8 image, and registration of the light (src/scene.rs:120-132).
  void set_environment_rgb8(const uint8_t* rgb, uint32_t w, uint32_t h);
};

// render_settings.yaml as main() reads it (src/main.rs:244-302).
struct RenderSettingsFile {
  Settings settings;
  uint32_t image_width = 0, image_height = 0, bucket_width = 128, bucket_height = 128;
  bool has_crop_start = false, has_crop_end = false;
  uint32_t crop_start[2] = {0, 0}, crop_end[2] = {0, 0};
  FilterMethod filter_method = FilterMethod::None;
  double filter_radius = 0;
  bool denoise = false;
  Vec3 camera_position, camera_target;
  double fov = 0, aperture = 0;
  bool has_focal_distance = false;
  double focal_distance = 0;
  double window_scale = 1.5;

  static RenderSettingsFile load(const std::string& folder);  // throws std::runtime_error
  std::unique_ptr<Film> make_film() const;
  std::unique_ptr<Camera> make_camera() const;
};

// src/renderer.rs:54-57
struct ThreadMessage {
  bool exit = false;
  bool finished = false;
  std::string error;  // non-empty when the worker failed (the reference would have panicked)
};

// What render() hands back: the worker threads and the receiving end of the
// completion channel (src/renderer.rs:97).
class RenderJob {
 public:
  ~RenderJob();
  bool try_recv(ThreadMessage* out);   // Receiver::try_recv
  ThreadMessage recv();                // blocking
  void join();
  size_t thread_count() const { return threads_.size(); }
  double seconds() const { return seconds_; }
  rrt_stats stats() const { return stats_; }

 private:
  friend std::unique_ptr<RenderJob> render(const Scene&, const Settings&, uint64_t, const Camera&, Film*,
                                           const std::vector<int>&, uint32_t);
  void send(const ThreadMessage& m);
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<ThreadMessage> queue_;
  std::mutex film_mu_;
  double seconds_ = 0;
  rrt_stats stats_{};
};

// renderer::render (src/renderer.rs:92-182) for GPUs: one worker thread per
// entry of `devices`; worker g renders sample indices
// [g*max_samples/G, (g+1)*max_samples/G) of every pixel in chunks of
// `samples_per_merge` and merges each chunk into `film` under a lock, then
// sends ThreadMessage{finished}.  settings.thread_count is ignored (it sized
// the CPU pool).  `seed` replaces the SobolSampler argument.
std::unique_ptr<RenderJob> render(const Scene& scene, const Settings& settings, uint64_t seed, const Camera& camera,
                                  Film* film, const std::vector<int>& devices, uint32_t samples_per_merge = 0);

}  // namespace rrt_host
