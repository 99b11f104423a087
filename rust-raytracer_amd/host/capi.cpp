// C facade declared in include/rrt_host.h.
#include <chrono>
#include <cstring>
#include <exception>
#include <string>

#include "../../include/rrt_host.h"
#include "rrt_host.hpp"

using namespace rrt_host;

struct rrth_scene {
  std::unique_ptr<Scene> scene;
};

namespace {
thread_local std::string g_error;

template <typename F>
int guarded(F&& f) {
  try {
    f();
    return RRT_OK;
  } catch (const std::exception& e) {
    g_error = e.what();
  } catch (...) {
    g_error = "unknown error";
  }
  return RRT_E_INVALID;
}

Camera make_camera(const double position[3], const double target[3], double fov, double aperture, int has_fd,
                   double fd, uint32_t w, uint32_t h) {
  return Camera(Vec3{position[0], position[1], position[2]}, Vec3{target[0], target[1], target[2]},
                (double)w / (double)h, fov, aperture, has_fd != 0, fd, -1.0, -1.0, 1.0, 1.0, w, h);
}
}  // namespace

extern "C" {

const char* rrth_last_error(void) { return g_error.c_str(); }

rrth_scene* rrth_scene_load(const char* folder) {
  rrth_scene* s = nullptr;
  guarded([&] {
    if (!folder) throw std::runtime_error("null folder");
    auto scene = Scene::load_from_folder(folder);
    s = new rrth_scene{std::move(scene)};
  });
  return s;
}

rrth_scene* rrth_scene_load_with_mesh(const char* folder, const float* positions, const float* normals,
                                      uint64_t n_tris, const rrt_material* material) {
  rrth_scene* s = nullptr;
  guarded([&] {
    if (!folder || !positions || !normals || !material) throw std::runtime_error("null argument");
    auto scene = Scene::load_from_folder_with_mesh(folder, positions, normals, (size_t)n_tris, *material);
    s = new rrth_scene{std::move(scene)};
  });
  return s;
}

void rrth_scene_free(rrth_scene* scene) { delete scene; }

int rrth_scene_desc(const rrth_scene* scene, rrt_scene_desc* out) {
  if (!scene || !out) { g_error = "null argument"; return RRT_E_INVALID; }
  *out = scene->scene->desc();
  return RRT_OK;
}

int rrth_scene_set_rect_aabb_mode(rrth_scene* scene, uint32_t mode) {
  if (!scene || mode > RRT_RECT_AABB_OFF) { g_error = "bad argument"; return RRT_E_INVALID; }
  scene->scene->rect_aabb_mode = mode;
  return RRT_OK;
}

int rrth_scene_set_material(rrth_scene* scene, uint32_t index, const rrt_material* material) {
  if (!scene || !material || index >= scene->scene->materials.size()) { g_error = "bad argument"; return RRT_E_INVALID; }
  scene->scene->materials[index] = *material;
  return RRT_OK;
}

int rrth_load_render_settings(const char* folder, rrth_render_settings* out) {
  return guarded([&] {
    if (!folder || !out) throw std::runtime_error("null argument");
    const RenderSettingsFile r = RenderSettingsFile::load(folder);
    std::memset(out, 0, sizeof(*out));
    out->threads = r.settings.thread_count;
    out->depth_limit = r.settings.depth_limit;
    out->max_samples = r.settings.max_samples;
    out->clamp = r.settings.clamp;
    out->image_width = r.image_width;
    out->image_height = r.image_height;
    out->bucket_width = r.bucket_width;
    out->bucket_height = r.bucket_height;
    auto film = r.make_film();
    film->render_window(&out->crop_x0, &out->crop_y0, &out->crop_x1, &out->crop_y1);
    out->filter_method = (uint32_t)r.filter_method;
    out->filter_radius = r.filter_radius;
    out->denoise = r.denoise;
    out->has_focal_distance = r.has_focal_distance;
    out->focal_distance = r.focal_distance;
    out->camera_position[0] = r.camera_position.x;
    out->camera_position[1] = r.camera_position.y;
    out->camera_position[2] = r.camera_position.z;
    out->camera_target[0] = r.camera_target.x;
    out->camera_target[1] = r.camera_target.y;
    out->camera_target[2] = r.camera_target.z;
    out->fov = r.fov;
    out->aperture = r.aperture;
    out->window_scale = r.window_scale;
  });
}

int rrth_camera_matrices(const double position[3], const double target[3], double fov, double aperture,
                         int has_focal_distance, double focal_distance, uint32_t image_width, uint32_t image_height,
                         double raster_to_camera[16], double camera_to_world[16], double* focal_distance_out) {
  return guarded([&] {
    if (!position || !target || !raster_to_camera || !camera_to_world || !image_width || !image_height)
      throw std::runtime_error("bad argument");
    const Camera c = make_camera(position, target, fov, aperture, has_focal_distance, focal_distance, image_width,
                                 image_height);
    std::memcpy(raster_to_camera, c.raster_to_camera.m, sizeof(double) * 16);
    std::memcpy(camera_to_world, c.camera_to_world.m, sizeof(double) * 16);
    if (focal_distance_out) *focal_distance_out = c.focal_distance;
  });
}

int rrth_camera_generate_ray(const double position[3], const double target[3], double fov, double aperture,
                             int has_focal_distance, double focal_distance, uint32_t image_width,
                             uint32_t image_height, double film_x, double film_y, double lens_x, double lens_y,
                             double origin[3], double direction[3]) {
  return guarded([&] {
    if (!position || !target || !origin || !direction || !image_width || !image_height)
      throw std::runtime_error("bad argument");
    const Camera c = make_camera(position, target, fov, aperture, has_focal_distance, focal_distance, image_width,
                                 image_height);
    const Ray r = c.generate_ray(film_x, film_y, lens_x, lens_y);
    origin[0] = r.point.x; origin[1] = r.point.y; origin[2] = r.point.z;
    direction[0] = r.direction.x; direction[1] = r.direction.y; direction[2] = r.direction.z;
  });
}

int rrth_tonemap(const float* sum_rgbw, uint32_t width, uint32_t height, uint8_t* rgb8) {
  return guarded([&] {
    if (!sum_rgbw || !rgb8 || !width || !height) throw std::runtime_error("bad argument");
    Film film(width, height, 128, 128, false, 0, 0, 0, 0, FilterMethod::None, 0.0);
    film.merge_accumulators(sum_rgbw);
    std::memcpy(rgb8, film.image_buffer.data(), film.image_buffer.size());
  });
}

int rrth_render_folder(const char* folder, const int* devices, int n_devices, uint64_t seed, int samples,
                       const char* out_prefix, double* samples_per_sec) {
  return rrth_render_folder_first_hits(folder, devices, n_devices, seed, samples, out_prefix, samples_per_sec, nullptr,
                                       nullptr);
}

int rrth_render_folder_first_hits(const char* folder, const int* devices, int n_devices, uint64_t seed, int samples,
                                  const char* out_prefix, double* samples_per_sec, float* normal_map,
                                  float* albedo_map) {
  return guarded([&] {
    if (!folder || !devices || n_devices <= 0) throw std::runtime_error("bad argument");
    auto scene = Scene::load_from_folder(folder);
    RenderSettingsFile rs = RenderSettingsFile::load(folder);
    if (samples > 0) rs.settings.max_samples = (uint32_t)samples;
    auto film = rs.make_film();
    film->keep_first_hits(normal_map || albedo_map);
    auto camera = rs.make_camera();
    std::vector<int> devs(devices, devices + n_devices);
    const auto t0 = std::chrono::steady_clock::now();
    auto job = render(*scene, rs.settings, seed, *camera, film.get(), devs);
    std::string error;
    for (size_t running = job->thread_count(); running > 0; --running) {
      ThreadMessage m = job->recv();
      if (!m.error.empty()) error = m.error;
    }
    job->join();
    if (!error.empty()) throw std::runtime_error(error);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint32_t x0, y0, x1, y1;
    film->render_window(&x0, &y0, &x1, &y1);
    if (samples_per_sec) *samples_per_sec = (double)(x1 - x0) * (y1 - y0) * rs.settings.max_samples / secs;
    if (out_prefix) {
      film->write_ppm(std::string(out_prefix) + ".ppm");
      film->write_png(std::string(out_prefix) + ".png");
      film->write_pfm(std::string(out_prefix) + ".pfm");
    }
    if (film->keeps_first_hits()) {
      std::vector<float> n, a;
      film->first_hit_maps(&n, &a);
      if (normal_map) std::memcpy(normal_map, n.data(), n.size() * sizeof(float));
      if (albedo_map) std::memcpy(albedo_map, a.data(), a.size() * sizeof(float));
    }
  });
}

int rrth_write_png(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height) {
  return guarded([&] {
    if (!path || !rgb8 || !width || !height) throw std::runtime_error("bad argument");
    if (!write_png_rgb8(path, rgb8, width, height)) throw std::runtime_error(std::string("cannot write ") + path);
  });
}

}  // extern "C"
