// rrt_render <scene_folder> [out_prefix] [samples] [devices]
//
// main() of the reference without the window (src/main.rs:236-347): loads
// scene.yaml + render_settings.yaml from the folder, renders on the GPUs with
// render() and writes <out_prefix>.ppm (tone-mapped) and <out_prefix>.pfm
// (linear).  The reference only ever shows the image in its ggez window.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/rrt_host.h"

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s <scene_folder> [out_prefix] [samples] [n_devices]\n", argv[0]);
    return 2;
  }
  const std::string folder = argv[1];
  const std::string out = argc > 2 ? argv[2] : "render";
  const int samples = argc > 3 ? std::atoi(argv[3]) : 0;
  int n_dev = argc > 4 ? std::atoi(argv[4]) : 1;
  if (n_dev <= 0) n_dev = 1;
  std::vector<int> devices;
  for (int i = 0; i < n_dev; ++i) devices.push_back(i);
  std::printf("Load scene from \"%s\"\nStart rendering...\n", folder.c_str());
  double rate = 0.0;
  const int rc = rrth_render_folder(folder.c_str(), devices.data(), n_dev, 0, samples, out.c_str(), &rate);
  if (rc != 0) {
    std::fprintf(stderr, "error: %s\n", rrth_last_error());
    return 1;
  }
  std::printf("done: %.3f M samples/s on %d device(s); wrote %s.ppm / %s.pfm\n", rate * 1e-6, n_dev, out.c_str(),
              out.c_str());
  return 0;
}
