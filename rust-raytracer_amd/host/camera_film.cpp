// Camera (src/camera.rs) and Film (src/film.rs) on the host.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "rrt_host.hpp"

namespace rrt_host {

// ---------------------------------------------------------------- Mat4
Mat4 Mat4::identity() {
  Mat4 r{};
  for (int i = 0; i < 4; ++i) r.m[5 * i] = 1.0;
  return r;
}

Mat4 Mat4::scaling(double x, double y, double z) {
  Mat4 r = identity();
  r.m[0] = x;
  r.m[5] = y;
  r.m[10] = z;
  return r;
}

Mat4 Mat4::translation(double x, double y, double z) {
  Mat4 r = identity();
  r.m[3] = x;
  r.m[7] = y;
  r.m[11] = z;
  return r;
}

Mat4 Mat4::operator*(const Mat4& o) const {
  Mat4 r{};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += m[4 * i + k] * o.m[4 * k + j];
      r.m[4 * i + j] = s;
    }
  return r;
}

bool Mat4::try_inverse(Mat4* out) const {
  // Gauss-Jordan with partial pivoting
  double a[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      a[i][j] = m[4 * i + j];
      a[i][4 + j] = i == j ? 1.0 : 0.0;
    }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    for (int r = c + 1; r < 4; ++r)
      if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
    if (a[piv][c] == 0.0) return false;
    if (piv != c)
      for (int j = 0; j < 8; ++j) std::swap(a[piv][j], a[c][j]);
    const double inv = 1.0 / a[c][c];
    for (int j = 0; j < 8; ++j) a[c][j] *= inv;
    for (int r = 0; r < 4; ++r) {
      if (r == c) continue;
      const double f = a[r][c];
      if (f != 0.0)
        for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
    }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) out->m[4 * i + j] = a[i][4 + j];
  return true;
}

Vec3 Mat4::transform_point(const Vec3& p) const {
  const double x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
  const double y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
  const double z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
  const double w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
  if (w != 0.0) return {x / w, y / w, z / w};
  return {x, y, z};
}

Vec3 Mat4::transform_vector(const Vec3& v) const {
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
          m[8] * v.x + m[9] * v.y + m[10] * v.z};
}

// ---------------------------------------------------------------- Camera
namespace {
Vec3 sub(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
Vec3 cross(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
double length(Vec3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
Vec3 normalize(Vec3 a) {
  const double l = length(a);
  return {a.x / l, a.y / l, a.z / l};
}

// perspective() of src/camera.rs:128-158
Mat4 perspective(double fov_deg, double n, double f) {
  Mat4 p{};
  p.m[0] = 1.0;
  p.m[5] = 1.0;
  p.m[10] = f / (f - n);
  p.m[11] = -f * n / (f - n);
  p.m[14] = 1.0;
  const double fov_rad = fov_deg * (M_PI / 180.0);
  const double inv_tan_ang = 1.0 / std::tan(fov_rad / 2.0);
  return p * Mat4::scaling(inv_tan_ang, inv_tan_ang, 1.0);
}
}  // namespace

Camera::Camera(Vec3 position_, Vec3 target_, double /*aspect_ratio*/, double fov_, double aperture_,
               bool has_focal_distance, double focal_distance_, double screen_min_x, double screen_min_y,
               double screen_max_x, double screen_max_y, uint32_t image_width, uint32_t image_height)
    : position(position_), target(target_), fov(fov_), aperture(aperture_) {
  const Vec3 direction = sub(target, position);
  focal_distance = has_focal_distance ? focal_distance_ : length(direction);

  // Rotation3::face_towards(&direction, &Vector3::y()): columns are the camera
  // x, y, z axes in world space, z along the view direction; then
  // append_translation(position)  (src/camera.rs:57-62)
  const Vec3 zaxis = normalize(direction);
  const Vec3 xaxis = normalize(cross(Vec3{0, 1, 0}, zaxis));
  const Vec3 yaxis = normalize(cross(zaxis, xaxis));
  camera_to_world = Mat4::identity();
  camera_to_world.m[0] = xaxis.x; camera_to_world.m[1] = yaxis.x; camera_to_world.m[2] = zaxis.x;
  camera_to_world.m[4] = xaxis.y; camera_to_world.m[5] = yaxis.y; camera_to_world.m[6] = zaxis.y;
  camera_to_world.m[8] = xaxis.z; camera_to_world.m[9] = yaxis.z; camera_to_world.m[10] = zaxis.z;
  camera_to_world.m[3] = position.x;
  camera_to_world.m[7] = position.y;
  camera_to_world.m[11] = position.z;

  camera_to_screen = perspective(fov, 0.01, 1000.0);
  screen_to_raster = Mat4::scaling((double)image_width, (double)image_height, 1.0) *
                     Mat4::scaling(1.0 / (screen_max_x - screen_min_x), 1.0 / (screen_min_y - screen_max_y), 1.0) *
                     Mat4::translation(-screen_min_x, -screen_max_y, 0.0);
  screen_to_raster.try_inverse(&raster_to_screen);
  Mat4 screen_to_camera;
  camera_to_screen.try_inverse(&screen_to_camera);
  raster_to_camera = screen_to_camera * raster_to_screen;
}

Ray Camera::generate_ray(double film_x, double film_y, double lens_x, double lens_y) const {
  Vec3 origin{0, 0, 0};
  Vec3 direction = raster_to_camera.transform_point(Vec3{film_x, film_y, 0.0});
  if (aperture > 0.0) {
    const double px = aperture * lens_x, py = aperture * lens_y;
    const double ft = focal_distance / direction.z;
    const Vec3 p_focus{ft * direction.x, ft * direction.y, ft * direction.z};
    origin = {px, py, 0.0};
    direction = normalize(sub(p_focus, origin));
  }
  Ray r;
  r.point = camera_to_world.transform_point(origin);
  r.direction = normalize(camera_to_world.transform_vector(direction));
  return r;
}

// ---------------------------------------------------------------- Film
FilterMethod filter_method_from_str(const std::string& s) {
  if (s == "gaussian") return FilterMethod::Gaussian;
  if (s == "mitchell") return FilterMethod::Mitchell;
  return FilterMethod::None;
}

double evaluate_gaussian(double px, double py, double radius, double alpha) {
  const double expv = std::exp(-alpha * radius * radius);
  const double x = std::fmax(std::exp(-alpha * px * px) - expv, 0.0);
  const double y = std::fmax(std::exp(-alpha * py * py) - expv, 0.0);
  return x * y;
}

static double evaluate_mitchell_1d(double input) {
  const double b = 0.33, c = 0.33;
  const double x = std::fabs(2.0 * input);
  if (x > 1.0)
    return ((-b - 6.0 * c) * x * x * x + (6.0 * b + 30.0 * c) * x * x + (-12.0 * b - 48.0 * c) * x +
            (8.0 * b + 24.0 * c)) * (1.0 / 6.0);
  return ((12.0 - 9.0 * b - 6.0 * c) * x * x * x + (-18.0 + 12.0 * b + 6.0 * c) * x * x + (6.0 - 2.0 * b)) *
         (1.0 / 6.0);
}

double evaluate_mitchell(double px, double py, double radius) {
  const double inv_radius = 1.0 / radius;
  return evaluate_mitchell_1d(px * inv_radius) * evaluate_mitchell_1d(py * inv_radius);
}

void xyz_to_srgb(const double xyz[3], double rgb[3]) {
  const double x = xyz[0], y = xyz[1], z = xyz[2];
  rgb[0] = 3.240479 * x - 1.537150 * y - 0.498535 * z;
  rgb[1] = -0.969256 * x + 1.875991 * y + 0.041556 * z;
  rgb[2] = 0.055648 * x - 0.204043 * y + 1.057311 * z;
}

void srgb_to_xyz(const double srgb[3], double xyz[3]) {
  const double r = inverse_gamma_correct_srgb(srgb[0]);
  const double g = inverse_gamma_correct_srgb(srgb[1]);
  const double b = inverse_gamma_correct_srgb(srgb[2]);
  xyz[0] = 0.412453 * r + 0.357580 * g + 0.180423 * b;
  xyz[1] = 0.212671 * r + 0.715160 * g + 0.072169 * b;
  xyz[2] = 0.019334 * r + 0.119193 * g + 0.950227 * b;
}

double gamma_correct_srgb(double v) {
  if (v <= 0.0) return 0.0;
  if (v < 0.0031308) return v * 12.92;
  if (v < 1.0) return std::pow(v, 1.0 / 2.4) * 1.055 - 0.055;
  return 1.0;
}

double inverse_gamma_correct_srgb(double v) {
  if (v <= 0.0) return 0.0;
  if (v < 0.04045) return v / 12.92;
  if (v < 1.0) return std::pow((v + 0.055) / 1.055, 2.4);
  return 1.0;
}

Film::Film(uint32_t width, uint32_t height, uint32_t bucket_w, uint32_t bucket_h, bool has_crop, uint32_t crop_x0,
           uint32_t crop_y0, uint32_t crop_x1, uint32_t crop_y1, FilterMethod filter_method, double filter_radius)
    : width_(width), height_(height), bucket_w_(bucket_w), bucket_h_(bucket_h), has_crop_(has_crop),
      crop_{crop_x0, crop_y0, crop_x1, crop_y1}, filter_method_(filter_method), filter_radius_(filter_radius) {
  pixels.assign((size_t)width * height, Pixel{});
  image_buffer.assign((size_t)width * height * 3, 0);
  const int n = 16;  // filter_table_size
  if (filter_method_ != FilterMethod::None) {
    for (int y = 0; y < n; ++y)
      for (int x = 0; x < n; ++x) {
        const double xp = (x + 0.5) * filter_radius_ / n, yp = (y + 0.5) * filter_radius_ / n;
        filter_table_.push_back(filter_method_ == FilterMethod::Gaussian
                                    ? evaluate_gaussian(xp, yp, filter_radius_, 1.5)
                                    : evaluate_mitchell(xp, yp, filter_radius_));
      }
  } else {
    filter_radius_ = 0.0;  // src/film.rs:110-112
  }
}

void Film::render_window(uint32_t* x0, uint32_t* y0, uint32_t* x1, uint32_t* y1) const {
  if (has_crop_) {
    // Film::init_buckets (src/film.rs:263-283): ceil((crop_end - crop_start) / bucket) WHOLE buckets
    // are laid from crop_start and capped only at the image size, so with a crop that is not a
    // multiple of the bucket size the reference also renders the pixels between crop_end and the
    // next bucket edge.  The window handed to the device is that union of buckets.
    const uint32_t bw = bucket_w_ ? bucket_w_ : 1, bh = bucket_h_ ? bucket_h_ : 1;
    const uint32_t cw = crop_[2] > crop_[0] ? crop_[2] - crop_[0] : 0, ch = crop_[3] > crop_[1] ? crop_[3] - crop_[1] : 0;
    const uint64_t ex = (uint64_t)crop_[0] + (uint64_t)((cw + bw - 1) / bw) * bw;
    const uint64_t ey = (uint64_t)crop_[1] + (uint64_t)((ch + bh - 1) / bh) * bh;
    *x0 = crop_[0] < width_ ? crop_[0] : width_;
    *y0 = crop_[1] < height_ ? crop_[1] : height_;
    *x1 = (uint32_t)(ex < width_ ? ex : width_);
    *y1 = (uint32_t)(ey < height_ ? ey : height_);
    if (*x1 < *x0) *x1 = *x0;
    if (*y1 < *y0) *y1 = *y0;
  } else {
    *x0 = 0;
    *y0 = 0;
    *x1 = width_;
    *y1 = height_;
  }
}

void Film::clear() {
  pixels.assign(pixels.size(), Pixel{});
  image_buffer.assign(image_buffer.size(), 0);
}

void Film::merge_accumulators(const float* acc) {
  for (size_t i = 0; i < pixels.size(); ++i) {
    const float* a = acc + 4 * i;
    if (a[3] == 0.0f) continue;
    Pixel& p = pixels[i];
    p.sum_weight += (double)a[3];
    for (int k = 0; k < 3; ++k) p.sum_radiance[k] += (double)a[k];
    uint8_t* out = &image_buffer[3 * i];
    if (p.sum_weight < 2.220446049250313e-16) {
      out[0] = out[1] = out[2] = 0;
      continue;
    }
    const double radiance[3] = {p.sum_radiance[0] / p.sum_weight, p.sum_radiance[1] / p.sum_weight,
                                p.sum_radiance[2] / p.sum_weight};
    double rgb[3];
    xyz_to_srgb(radiance, rgb);  // radiance is handed to the XYZ matrix as is (src/film.rs:238-241)
    for (int k = 0; k < 3; ++k) out[k] = (uint8_t)(gamma_correct_srgb(rgb[k]) * 255.0);
  }
}

void Film::merge_first_hits(const float* normal_sum, const float* albedo_sum) {
  for (size_t i = 0; i < pixels.size(); ++i)
    for (int k = 0; k < 3; ++k) {
      pixels[i].sum_normal[k] += (double)normal_sum[4 * i + k];
      pixels[i].sum_albedo[k] += (double)albedo_sum[4 * i + k];
    }
}

void Film::first_hit_maps(std::vector<float>* normal_map, std::vector<float>* albedo_map) const {
  normal_map->assign(pixels.size() * 3, 0.0f);
  albedo_map->assign(pixels.size() * 3, 0.0f);
  for (size_t i = 0; i < pixels.size(); ++i) {
    const Pixel& p = pixels[i];
    if (!(p.sum_weight > 0.0)) continue;
    for (int k = 0; k < 3; ++k) {
      (*normal_map)[3 * i + k] = (float)(p.sum_normal[k] / p.sum_weight);
      (*albedo_map)[3 * i + k] = (float)(p.sum_albedo[k] / p.sum_weight);
    }
  }
}

bool Film::write_ppm(const std::string& path) const {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  std::fprintf(f, "P6\n%u %u\n255\n", width_, height_);
  const bool ok = std::fwrite(image_buffer.data(), 1, image_buffer.size(), f) == image_buffer.size();
  std::fclose(f);
  return ok;
}

namespace {

uint32_t crc32_update(uint32_t crc, const uint8_t* data, size_t n) {
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready = true;
  }
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ data[i]) & 0xffu] ^ (crc >> 8);
  return crc;
}

void put_be32(std::vector<uint8_t>* out, uint32_t v) {
  for (int s = 24; s >= 0; s -= 8) out->push_back((uint8_t)(v >> s));
}

bool write_chunk(FILE* f, const char type[4], const std::vector<uint8_t>& body) {
  std::vector<uint8_t> head;
  put_be32(&head, (uint32_t)body.size());
  head.insert(head.end(), type, type + 4);
  uint32_t crc = crc32_update(0xffffffffu, head.data() + 4, 4);
  crc = crc32_update(crc, body.data(), body.size()) ^ 0xffffffffu;
  std::vector<uint8_t> tail;
  put_be32(&tail, crc);
  return std::fwrite(head.data(), 1, head.size(), f) == head.size() &&
         (body.empty() || std::fwrite(body.data(), 1, body.size(), f) == body.size()) &&
         std::fwrite(tail.data(), 1, tail.size(), f) == tail.size();
}

}  // namespace

// 8-bit RGB PNG.  The image data goes into stored (uncompressed) deflate blocks: a valid PNG that
// every reader opens, without a compressor in the host library.
bool Film::write_png(const std::string& path) const { return write_png_rgb8(path, image_buffer.data(), width_, height_); }

bool write_png_rgb8(const std::string& path, const uint8_t* rgb, uint32_t width_, uint32_t height_) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  static const uint8_t signature[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  bool ok = std::fwrite(signature, 1, 8, f) == 8;
  std::vector<uint8_t> ihdr;
  put_be32(&ihdr, width_);
  put_be32(&ihdr, height_);
  const uint8_t fmt[5] = {8, 2, 0, 0, 0};  // 8 bits, RGB, deflate, no filter method, no interlace
  ihdr.insert(ihdr.end(), fmt, fmt + 5);
  ok = ok && write_chunk(f, "IHDR", ihdr);
  // scanlines, each led by filter type 0
  std::vector<uint8_t> raw;
  raw.reserve(((size_t)width_ * 3 + 1) * height_);
  for (uint32_t y = 0; y < height_; ++y) {
    raw.push_back(0);
    const uint8_t* row = rgb + (size_t)y * width_ * 3;
    raw.insert(raw.end(), row, row + (size_t)width_ * 3);
  }
  uint32_t a = 1, b = 0;  // Adler-32 of the raw data
  for (uint8_t v : raw) {
    a = (a + v) % 65521u;
    b = (b + a) % 65521u;
  }
  std::vector<uint8_t> z;
  z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
  z.push_back(0x78);
  z.push_back(0x01);
  size_t at = 0;
  do {
    const size_t n = std::min<size_t>(65535, raw.size() - at);
    z.push_back(at + n == raw.size() ? 1 : 0);  // last block?
    z.push_back((uint8_t)(n & 0xff));
    z.push_back((uint8_t)(n >> 8));
    z.push_back((uint8_t)(~n & 0xff));
    z.push_back((uint8_t)((~n >> 8) & 0xff));
    z.insert(z.end(), raw.begin() + (std::ptrdiff_t)at, raw.begin() + (std::ptrdiff_t)(at + n));
    at += n;
  } while (at < raw.size());
  put_be32(&z, (b << 16) | a);
  ok = ok && write_chunk(f, "IDAT", z);
  ok = ok && write_chunk(f, "IEND", std::vector<uint8_t>());
  std::fclose(f);
  return ok;
}

bool Film::write_pfm(const std::string& path) const {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  std::fprintf(f, "PF\n%u %u\n-1.0\n", width_, height_);
  std::vector<float> row((size_t)width_ * 3);
  bool ok = true;
  for (uint32_t y = height_; y-- > 0;) {  // PFM rows go bottom to top
    for (uint32_t x = 0; x < width_; ++x) {
      const Pixel& p = pixels[(size_t)y * width_ + x];
      for (int k = 0; k < 3; ++k) row[3 * x + k] = p.sum_weight > 0 ? (float)(p.sum_radiance[k] / p.sum_weight) : 0.0f;
    }
    ok = ok && std::fwrite(row.data(), sizeof(float), row.size(), f) == row.size();
  }
  std::fclose(f);
  return ok;
}

}  // namespace rrt_host
