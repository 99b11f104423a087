// Scene and render-settings loading: src/scene.rs:55-338 and src/main.rs:244-302.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "obj_loader.h"
#include "rrt_host.hpp"
#include "yaml_lite.h"

namespace rrt_host {

namespace {

[[noreturn]] void panic(const std::string& msg) { throw std::runtime_error(msg); }

// helpers::yaml_array_into_point3 / _vector3 (src/helpers.rs:162-176): each
// element must be a real; `.unwrap()` on anything else.
void yaml_vec3(const Yaml& y, const char* what, double out[3]) {
  for (size_t i = 0; i < 3; ++i)
    if (!y[i].as_f64(&out[i])) panic(std::string("scene yaml: '") + what + "' must be three reals");
}

// helpers::yaml_array_into_point2 / yaml_into_u32 (src/helpers.rs:155-160,178-180)
uint32_t yaml_u32(const Yaml& y, const char* what) {
  long long v;
  if (!y.as_i64(&v)) panic(std::string("yaml: '") + what + "' must be an integer");
  return (uint32_t)v;
}

rrt_material make_material(uint32_t kind, const double kd[3], const double ks[3], double roughness, double ior) {
  rrt_material m{};
  m.kind = kind;
  for (int k = 0; k < 3; ++k) { m.kd[k] = kd[k]; m.ks[k] = ks[k]; }
  m.roughness = roughness;
  m.ior = ior;
  return m;
}

// The MTL -> material mapping of load_model (src/scene.rs:249-310).
rrt_material material_from_mtl(const ObjMaterial* mtl) {
  double color[3] = {0.8, 0.8, 0.8}, specular[3] = {0, 0, 0}, translucence[3] = {0, 0, 0};
  double ior = 1.5, roughness = 0.5;
  bool is_translucent = false;
  if (mtl) {
    for (int k = 0; k < 3; ++k) { color[k] = (double)mtl->diffuse[k]; specular[k] = (double)mtl->specular[k]; }
    ior = (double)mtl->optical_density;
    roughness = 32.0;
    auto pr = mtl->unknown_param.find("Pr");
    if (pr != mtl->unknown_param.end()) {
      char* end = nullptr;
      double v = std::strtod(pr->second.c_str(), &end);
      if (end != pr->second.c_str() && *end == 0) roughness = v;
    }
    auto tf = mtl->unknown_param.find("Tf");
    if (tf != mtl->unknown_param.end()) {
      std::istringstream ss(tf->second);
      double v[3];
      if (!(ss >> v[0] >> v[1] >> v[2])) panic("mtl: Tf needs three numbers");
      is_translucent = true;
      for (int k = 0; k < 3; ++k) translucence[k] = v[k];
    }
  }
  if (is_translucent) return make_material(RRT_MAT_GLASS, color, translucence, 0.0, ior);
  return make_material(RRT_MAT_PLASTIC, color, specular, roughness, ior);
}

bool ends_with(const std::string& s, const std::string& suffix) {
  return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

// Binary PPM (P6, maxval 255).
bool read_ppm(const std::string& path, std::vector<uint8_t>* rgb, uint32_t* w, uint32_t* h) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::string magic;
  f >> magic;
  if (magic != "P6") return false;
  auto next_int = [&](long* out) {
    for (;;) {
      int c = f.peek();
      if (c == '#') { std::string skip; std::getline(f, skip); }
      else if (c == ' ' || c == '\n' || c == '\r' || c == '\t') f.get();
      else break;
    }
    return (bool)(f >> *out);
  };
  long W, H, maxv;
  if (!next_int(&W) || !next_int(&H) || !next_int(&maxv) || maxv != 255 || W <= 0 || H <= 0) return false;
  f.get();
  rgb->resize((size_t)W * H * 3);
  f.read((char*)rgb->data(), (std::streamsize)rgb->size());
  if ((size_t)f.gcount() != rgb->size()) return false;
  *w = (uint32_t)W;
  *h = (uint32_t)H;
  return true;
}

// Everything of load_from_folder after the mesh (src/scene.rs:72-132).
void load_lights_and_environment(Scene* scene, const Yaml& scene_yaml, const std::string& folder) {
  const Yaml& lights = scene_yaml["lights"];
  for (size_t i = 0; i < lights.size(); ++i) {
    const Yaml& cfg = lights[i];
    std::string type;
    if (!cfg["type"].as_str(&type)) panic("scene.yaml: light without a type");
    if (type == "area") {
      rrt_light l{};
      l.kind = RRT_LIGHT_AREA;
      yaml_vec3(cfg["position"], "position", l.v0);
      yaml_vec3(cfg["side_a"], "side_a", l.v1);
      yaml_vec3(cfg["side_b"], "side_b", l.v2);
      yaml_vec3(cfg["intensity"], "intensity", l.intensity);
      // the quad that goes into Scene.objects carries Matte(0.9, 20.0) and the light (src/scene.rs:91-102)
      const double kd[3] = {0.9, 0.9, 0.9}, zero[3] = {0, 0, 0};
      scene->materials.push_back(make_material(RRT_MAT_MATTE, kd, zero, 20.0, 1.0));
      rrt_rect r{};
      for (int k = 0; k < 3; ++k) { r.pos[k] = l.v0[k]; r.side_a[k] = l.v1[k]; r.side_b[k] = l.v2[k]; }
      r.material = (uint32_t)scene->materials.size() - 1;
      r.light = (int32_t)scene->lights.size();
      scene->lights.push_back(l);
      scene->rects.push_back(r);
    }
    if (type == "distant") {
      rrt_light l{};
      l.kind = RRT_LIGHT_DISTANT;
      yaml_vec3(cfg["direction"], "direction", l.v0);
      yaml_vec3(cfg["intensity"], "intensity", l.intensity);
      scene->lights.push_back(l);
    }
  }
  std::string env;
  if (scene_yaml["environment_map"].as_str(&env)) {
    // The reference decodes any format the `image` crate knows.  Here a
    // pre-decoded binary PPM of the same name stands in for the JPEG.
    std::string path = folder + "/" + env;
    std::string ppm = path;
    if (!ends_with(path, ".ppm")) {
      size_t dot = path.find_last_of('.');
      ppm = (dot == std::string::npos ? path : path.substr(0, dot)) + ".ppm";
    }
    std::vector<uint8_t> rgb;
    uint32_t w = 0, h = 0;
    if (!read_ppm(ppm, &rgb, &w, &h))
      panic("Environment map not found: " + ppm + " (decode " + env + " to a binary PPM next to it)");
    scene->set_environment_rgb8(rgb.data(), w, h);
  }
}

}  // namespace

void Scene::set_environment_rgb8(const uint8_t* rgb, uint32_t w, uint32_t h) {
  env_w = w;
  env_h = h;
  env_rgb8.resize((size_t)w * h * 3);
  const double intensity[3] = {1.0, 1.0, 1.0};  // Vector3::repeat(1.0), src/scene.rs:126
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    double srgb[3] = {rgb[3 * i] / 255.0, rgb[3 * i + 1] / 255.0, rgb[3 * i + 2] / 255.0};
    double xyz[3];
    srgb_to_xyz(srgb, xyz);
    for (int k = 0; k < 3; ++k) {
      double v = std::fmin(xyz[k] * intensity[k] * 255.0, 255.0);
      env_rgb8[3 * i + k] = (uint8_t)v;  // `as u8`: truncation, saturating at 0
    }
  }
  // one infinite light, pushed after the yaml lights; re-registering replaces it
  for (size_t i = 0; i < lights.size(); ++i)
    if (lights[i].kind == RRT_LIGHT_INFINITE) return;
  rrt_light l{};
  l.kind = RRT_LIGHT_INFINITE;
  l.intensity[0] = l.intensity[1] = l.intensity[2] = 1.0;
  lights.push_back(l);
}

rrt_scene_desc Scene::desc() const {
  rrt_scene_desc d{};
  d.n_tris = (uint32_t)n_tris();
  d.n_rects = (uint32_t)rects.size();
  d.n_planes = (uint32_t)planes.size();
  d.n_spheres = (uint32_t)spheres.size();
  d.n_materials = (uint32_t)materials.size();
  d.n_lights = (uint32_t)lights.size();
  d.tri_pos = tri_pos.data();
  d.tri_nrm = tri_nrm.data();
  d.tri_mat = tri_mat.data();
  d.rects = rects.data();
  d.planes = planes.data();
  d.spheres = spheres.data();
  d.materials = materials.data();
  d.lights = lights.data();
  d.env_rgb8 = env_rgb8.empty() ? nullptr : env_rgb8.data();
  d.env_w = env_w;
  d.env_h = env_h;
  d.rect_aabb_mode = rect_aabb_mode;
  return d;
}

std::unique_ptr<Scene> Scene::load_from_folder(const std::string& folder) {
  Yaml scene_yaml;
  std::string err;
  if (!Yaml::parse_file(folder + "/scene.yaml", &scene_yaml, &err)) panic("Unable to open scene.yaml file: " + err);

  auto scene = std::make_unique<Scene>();
  std::string filename;
  if (scene_yaml["world"]["file"].as_str(&filename)) {
    std::string up_axis;
    if (!scene_yaml["world"]["up_axis"].as_str(&up_axis)) panic("scene.yaml: world.up_axis missing");  // src/scene.rs:66
    ObjFile obj;
    if (!load_obj(folder + "/" + filename, &obj, &err)) panic(err);
    for (const ObjModel& m : obj.models) {
      if (m.normals.empty())
        panic("model '" + m.name + "' has no vertex normals (the reference indexes mesh.normals unconditionally)");
      const ObjMaterial* mtl = m.material_id >= 0 ? &obj.materials[m.material_id] : nullptr;
      scene->materials.push_back(material_from_mtl(mtl));
      const uint32_t mat = (uint32_t)scene->materials.size() - 1;
      scene->model_names.push_back(m.name);
      scene->tri_pos.insert(scene->tri_pos.end(), m.positions.begin(), m.positions.end());
      scene->tri_nrm.insert(scene->tri_nrm.end(), m.normals.begin(), m.normals.end());
      scene->tri_mat.insert(scene->tri_mat.end(), m.n_tris(), mat);
    }
  }
  load_lights_and_environment(scene.get(), scene_yaml, folder);
  return scene;
}

std::unique_ptr<Scene> Scene::load_from_folder_with_mesh(const std::string& folder, const float* positions,
                                                         const float* normals, size_t n_tris,
                                                         const rrt_material& material) {
  Yaml scene_yaml;
  std::string err;
  if (!Yaml::parse_file(folder + "/scene.yaml", &scene_yaml, &err)) panic("Unable to open scene.yaml file: " + err);
  auto scene = std::make_unique<Scene>();
  scene->materials.push_back(material);
  scene->model_names.push_back("generated");
  scene->tri_pos.assign(positions, positions + n_tris * 9);
  scene->tri_nrm.assign(normals, normals + n_tris * 9);
  scene->tri_mat.assign(n_tris, 0u);
  load_lights_and_environment(scene.get(), scene_yaml, folder);
  return scene;
}

RenderSettingsFile RenderSettingsFile::load(const std::string& folder) {
  Yaml y;
  std::string err;
  if (!Yaml::parse_file(folder + "/render_settings.yaml", &y, &err))
    panic("Unable to open render_settings.yaml file: " + err);
  RenderSettingsFile r;
  r.settings.thread_count = yaml_u32(y["renderer"]["threads"], "renderer.threads");
  r.settings.depth_limit = yaml_u32(y["renderer"]["depth_limit"], "renderer.depth_limit");
  r.settings.max_samples = yaml_u32(y["sampler"]["max_samples"], "sampler.max_samples");
  if (!y["renderer"]["clamp"].as_f64(&r.settings.clamp)) r.settings.clamp = 0.0;  // unwrap_or(0.0)
  r.image_width = yaml_u32(y["film"]["image_width"], "film.image_width");
  r.image_height = yaml_u32(y["film"]["image_height"], "film.image_height");
  if (!y["window"]["scale"].as_f64(&r.window_scale)) r.window_scale = 1.5;
  const Yaml& crop = y["film"]["crop"];
  if (!crop["start"].is_badvalue()) {
    r.has_crop_start = true;
    r.crop_start[0] = yaml_u32(crop["start"][0], "film.crop.start");
    r.crop_start[1] = yaml_u32(crop["start"][1], "film.crop.start");
  }
  if (!crop["end"].is_badvalue()) {
    r.has_crop_end = true;
    r.crop_end[0] = yaml_u32(crop["end"][0], "film.crop.end");
    r.crop_end[1] = yaml_u32(crop["end"][1], "film.crop.end");
  } else {
    r.crop_end[0] = r.image_width;
    r.crop_end[1] = r.image_height;
  }
  bool dn;
  r.denoise = y["film"]["denoise"].as_bool(&dn) ? dn : false;
  r.bucket_width = yaml_u32(y["film"]["bucket_width"], "film.bucket_width");
  r.bucket_height = yaml_u32(y["film"]["bucket_height"], "film.bucket_height");
  std::string fm;
  if (!y["film"]["filter_method"].as_str(&fm)) panic("render_settings.yaml: film.filter_method missing");
  r.filter_method = filter_method_from_str(fm);
  if (!y["film"]["filter_radius"].as_f64(&r.filter_radius)) panic("render_settings.yaml: film.filter_radius missing");
  double v[3];
  yaml_vec3(y["camera"]["position"], "camera.position", v);
  r.camera_position = {v[0], v[1], v[2]};
  yaml_vec3(y["camera"]["target"], "camera.target", v);
  r.camera_target = {v[0], v[1], v[2]};
  if (!y["camera"]["fov"].as_f64(&r.fov)) panic("render_settings.yaml: camera.fov missing");
  if (!y["camera"]["aperture"].as_f64(&r.aperture)) panic("render_settings.yaml: camera.aperture missing");
  r.has_focal_distance = y["camera"]["focal_distance"].as_f64(&r.focal_distance);
  return r;
}

std::unique_ptr<Film> RenderSettingsFile::make_film() const {
  // main() always passes Some(crop_start), Some(crop_end) (src/main.rs:279-290)
  return std::make_unique<Film>(image_width, image_height, bucket_width, bucket_height, true, crop_start[0],
                                crop_start[1], crop_end[0], crop_end[1], filter_method, filter_radius);
}

std::unique_ptr<Camera> RenderSettingsFile::make_camera() const {
  const double aspect = (double)image_width / (double)image_height;
  return std::make_unique<Camera>(camera_position, camera_target, aspect, fov, aperture, has_focal_distance,
                                  focal_distance, -1.0, -1.0, 1.0, 1.0, image_width, image_height);
}

}  // namespace rrt_host
