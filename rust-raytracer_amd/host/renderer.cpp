// render(): the GPU counterpart of renderer::render (src/renderer.rs:92-182).
#include <chrono>
#include <stdexcept>

#include "rrt_host.hpp"

namespace rrt_host {

RenderJob::~RenderJob() { join(); }

void RenderJob::send(const ThreadMessage& m) {
  {
    std::lock_guard<std::mutex> lock(mu_);
    queue_.push_back(m);
  }
  cv_.notify_all();
}

bool RenderJob::try_recv(ThreadMessage* out) {
  std::lock_guard<std::mutex> lock(mu_);
  if (queue_.empty()) return false;
  *out = queue_.front();
  queue_.pop_front();
  return true;
}

ThreadMessage RenderJob::recv() {
  std::unique_lock<std::mutex> lock(mu_);
  cv_.wait(lock, [&] { return !queue_.empty(); });
  ThreadMessage m = queue_.front();
  queue_.pop_front();
  return m;
}

void RenderJob::join() {
  for (auto& t : threads_)
    if (t.joinable()) t.join();
}

namespace {

struct CtxGuard {
  rrt_ctx* ctx = nullptr;
  ~CtxGuard() {
    if (ctx) rrt_destroy(ctx);
  }
};

void check(rrt_ctx* ctx, int rc, const char* what) {
  if (rc != RRT_OK) throw std::runtime_error(std::string(what) + ": " + rrt_last_error(ctx));
}

}  // namespace

std::unique_ptr<RenderJob> render(const Scene& scene, const Settings& settings, uint64_t seed, const Camera& camera,
                                  Film* film, const std::vector<int>& devices, uint32_t samples_per_merge) {
  if (devices.empty()) throw std::runtime_error("render: no devices given");
  auto job = std::make_unique<RenderJob>();
  RenderJob* j = job.get();
  const size_t n_dev = devices.size();
  const auto t0 = std::chrono::steady_clock::now();

  for (size_t g = 0; g < n_dev; ++g) {
    const int device = devices[g];
    j->threads_.emplace_back([=, &scene, &camera]() {
      ThreadMessage done;
      done.finished = true;
      try {
        CtxGuard guard;
        guard.ctx = rrt_create(device);
        if (!guard.ctx) throw std::runtime_error(std::string("rrt_create: ") + rrt_last_error(nullptr));
        rrt_ctx* ctx = guard.ctx;
        const rrt_scene_desc desc = scene.desc();
        check(ctx, rrt_upload_scene(ctx, &desc), "rrt_upload_scene");
        check(ctx, rrt_set_camera(ctx, camera.raster_to_camera.m, camera.camera_to_world.m, camera.aperture,
                                  camera.focal_distance),
              "rrt_set_camera");
        check(ctx, rrt_film_alloc(ctx, film->width(), film->height()), "rrt_film_alloc");
        const uint32_t method = film->filter_method() == FilterMethod::Gaussian   ? RRT_FILTER_GAUSSIAN
                                : film->filter_method() == FilterMethod::Mitchell ? RRT_FILTER_MITCHELL
                                                                                  : RRT_FILTER_NONE;
        check(ctx, rrt_set_filter(ctx, method, film->filter_radius()), "rrt_set_filter");
        const bool first_hits = film->keeps_first_hits();
        check(ctx, rrt_set_option(ctx, RRT_OPT_AOVS, first_hits ? 1 : 0), "rrt_set_option");

        rrt_settings s{};
        s.width = film->width();
        s.height = film->height();
        film->render_window(&s.crop_x0, &s.crop_y0, &s.crop_x1, &s.crop_y1);
        s.depth_limit = settings.depth_limit;
        s.clamp = settings.clamp;
        s.seed = seed;

        // this worker's share of the sample index range
        const uint32_t begin = (uint32_t)((uint64_t)settings.max_samples * g / n_dev);
        const uint32_t end = (uint32_t)((uint64_t)settings.max_samples * (g + 1) / n_dev);
        const uint32_t step = samples_per_merge ? samples_per_merge : (end - begin ? end - begin : 1);
        std::vector<float> acc((size_t)film->width() * film->height() * 4);
        std::vector<float> hit_normal(first_hits ? acc.size() : 0), hit_albedo(first_hits ? acc.size() : 0);
        for (uint32_t b = begin; b < end; b += step) {
          const uint32_t e = b + step < end ? b + step : end;
          check(ctx, rrt_film_clear(ctx), "rrt_film_clear");
          check(ctx, rrt_render(ctx, &s, b, e), "rrt_render");
          check(ctx, rrt_film_download(ctx, acc.data()), "rrt_film_download");
          if (first_hits) check(ctx, rrt_aov_download(ctx, hit_normal.data(), hit_albedo.data()), "rrt_aov_download");
          std::lock_guard<std::mutex> lock(j->film_mu_);  // film.write() in the reference
          film->merge_accumulators(acc.data());
          if (first_hits) film->merge_first_hits(hit_normal.data(), hit_albedo.data());
        }
        rrt_stats st{};
        rrt_get_stats(ctx, &st);
        std::lock_guard<std::mutex> lock(j->film_mu_);
        j->stats_.paths += st.paths;
        j->stats_.rays_extend += st.rays_extend;
        j->stats_.rays_shadow += st.rays_shadow;
        j->stats_.path_bounces += st.path_bounces;
        j->stats_.bvh_nodes = st.bvh_nodes;
        j->stats_.bvh_bytes = st.bvh_bytes;
        j->seconds_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      } catch (const std::exception& e) {
        done.error = e.what();
      }
      j->send(done);
    });
  }
  return job;
}

}  // namespace rrt_host
