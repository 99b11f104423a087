// Device-resident data layout of the path-tracing core (shared by the host
// BVH builder, the upload code and the kernels).  Everything the kernels read
// is structure-of-arrays or 16-byte records so that one lane fetches a whole
// record with dwordx4 loads.
#pragma once
#include <math.h>
#include <stdint.h>

#include <string>

#include "../../include/rrt.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RRT_HD __host__ __device__
#else
#define RRT_HD
struct float4 { float x, y, z, w; };
struct uint4 { uint32_t x, y, z, w; };
#endif

namespace rrt {

// The grid of one hierarchy, on which the corners of its nodes lie: 65535 steps of a power of two per
// axis from the minimum corner of all triangles span the scene.  Wave-uniform in every kernel.
struct BvhGrid {
  float base[3];      // minimum corner of all triangles
  float step[3];      // 2^e, the smallest with base + 65535 * step >= the maximum corner
  float inv_step[3];  // 2^-e
};

// One 4-wide BVH node, 48 B = three 16-byte loads.  The node's own corner is a point of the scene's
// 16-bit grid, corner[k] steps from its base (rounded down, so at most one step below the node's true
// minimum).  The boxes of up to four children are 8-bit offsets from that corner on a grid of the
// node's own: plane = base[k] + (corner[k] + q * 2^shift[k]) * step[k], with shift[k] -- signed, 5 bits
// -- the smallest for which 255 steps reach the node's maximum, and q rounded outwards by the builder,
// so a decoded box always contains the exact one (at most 1/255 of the node's extent looser per side).
// Every term is a dyadic rational of few bits: the decoded planes are exact real numbers in double
// arithmetic, and the kernels never form them -- they go from q straight to the distance along the ray.
// child >= 0: index of an inner node.  child < 0: a leaf of 1..4 triangles of the leaf-ordered triangle
// arrays, ~child = first << 2 | (count - 1).  A slot without a child holds an inverted box (qlo = 255,
// qhi = 0 on every axis), which no ray enters, and repeats the link of child 0: should rounding ever
// let a ray into it -- only a node without any extent can -- the ray visits that child twice, which is
// harmless.
// Four wide rather than two: half the steps per ray.  Compressed rather than six floats per child, and
// the corner on a scene-wide grid rather than three floats: what bounds traversal is the number of
// 16-byte loads the lanes of a wave issue to scattered addresses -- the memory pipeline of a CU
// retires about one such load per lane and clock, whatever the cache level it is served from -- and a
// node costs three of them instead of seven.
struct alignas(16) BvhNode {
  int32_t child[4];
  uint8_t qlo[3][4];    // [axis][child]: node-grid steps from the corner to the child's minimum (rounded down)
  uint8_t qhi[3][4];    // ... to its maximum (rounded up)
  uint16_t corner[3];   // scene-grid steps from BvhGrid::base to the node's corner
  uint16_t shift;       // three signed 5-bit fields, x lowest: log2 of the node-grid step in scene-grid steps
};
static_assert(sizeof(BvhNode) == 48, "BvhNode must be 48 bytes");
constexpr int kNodeShiftMin = -16, kNodeShiftMax = 15;
RRT_HD inline bool child_present(const BvhNode& n, int i) { return n.qlo[0][i] <= n.qhi[0][i]; }
RRT_HD inline int node_shift(const BvhNode& n, int k) {
  const int f = (n.shift >> (5 * k)) & 31;
  return f >= 16 ? f - 32 : f;
}
// Plane q (0..255) of the node's grid on axis k.  corner + q * 2^shift has at most 40 significant bits and the
// step is a power of two, so the only rounding is that of the final sum, and the result is monotonic in q.
RRT_HD inline double node_plane(const BvhGrid& g, uint32_t corner, int shift, int k, uint32_t q) {
  return (double)g.base[k] + ((double)corner + ldexp((double)q, shift)) * (double)g.step[k];
}
RRT_HD inline double node_plane(const BvhGrid& g, const BvhNode& n, int k, uint32_t q) {
  return node_plane(g, n.corner[k], node_shift(n, k), k, q);
}

// Fills the boxes of a node from those of its n (0..4) children: corner, shifts, qlo, qhi; slots from n on
// get the inverted box.  Every decoded plane is checked with node_plane itself, the expression the
// validator uses, so the boxes contain their children whatever the rounding on the way.
RRT_HD inline void encode_node_boxes(const BvhGrid& g, const float (*mn)[3], const float (*mx)[3], int n, BvhNode* node) {
  node->shift = 0;
  for (int k = 0; k < 3; ++k) {
    node->corner[k] = 0;
    for (int i = 0; i < 4; ++i) { node->qlo[k][i] = 255; node->qhi[k][i] = 0; }
    if (n <= 0) continue;
    float all_mn = mn[0][k], all_mx = mx[0][k];
    for (int i = 1; i < n; ++i) {
      all_mn = mn[i][k] < all_mn ? mn[i][k] : all_mn;
      all_mx = mx[i][k] > all_mx ? mx[i][k] : all_mx;
    }
    double c = floor(((double)all_mn - (double)g.base[k]) * (double)g.inv_step[k]);
    c = c >= 0.0 ? (c > 65535.0 ? 65535.0 : c) : 0.0;  // (a NaN from a vertex that is not a number goes to 0)
    uint32_t corner = (uint32_t)c;
    while (corner > 0 && node_plane(g, corner, 0, k, 0) > (double)all_mn) --corner;
    // the smallest shift whose 255 steps reach the far side
    const double reach = ((double)all_mx - node_plane(g, corner, 0, k, 0)) * (double)g.inv_step[k] / 255.0;
    int shift = kNodeShiftMin;
    if (reach > 0.0) {
      int e = 0;
      (void)frexp(reach, &e);  // reach = m * 2^e with m in [0.5, 1): 2^e >= reach
      shift = e < kNodeShiftMin ? kNodeShiftMin : (e > kNodeShiftMax ? kNodeShiftMax : e);
    }
    while (shift < kNodeShiftMax && node_plane(g, corner, shift, k, 255) < (double)all_mx) ++shift;
    node->corner[k] = (uint16_t)corner;
    node->shift = (uint16_t)(node->shift | ((uint32_t)(shift & 31) << (5 * k)));
    const double origin = node_plane(g, corner, shift, k, 0);
    const double inv_node_step = ldexp((double)g.inv_step[k], -shift);
    for (int i = 0; i < n; ++i) {
      double lo = floor(((double)mn[i][k] - origin) * inv_node_step);
      double hi = ceil(((double)mx[i][k] - origin) * inv_node_step);
      lo = lo >= 0.0 ? (lo > 255.0 ? 255.0 : lo) : 0.0;
      hi = hi >= 0.0 ? (hi > 255.0 ? 255.0 : hi) : 0.0;
      uint32_t qlo = (uint32_t)lo, qhi = (uint32_t)hi;
      while (qlo > 0 && node_plane(g, corner, shift, k, qlo) > (double)mn[i][k]) --qlo;
      while (qhi < 255 && node_plane(g, corner, shift, k, qhi) < (double)mx[i][k]) ++qhi;
      node->qlo[k][i] = (uint8_t)qlo;
      node->qhi[k][i] = (uint8_t)qhi;
    }
  }
}

// Intersection record of a triangle, 48 B, leaf order: vertex 0 and the two
// edges Moeller-Trumbore needs.  v0.w carries the scene-order object id.
struct TriIsect {
  float4 v0;  // xyz = p0, w = bits of scene-order id
  float4 e1;  // xyz = p1 - p0
  float4 e2;  // xyz = p2 - p0
};

// Shading record of a triangle, one 128-byte line, leaf order: all that the surface interaction of a
// hit needs -- vertex 0 and the edges once more, the three vertex normals (objects/triangle.rs:84-107),
// the material -- so that a shaded bounce fetches one line for its triangle instead of parts of three
// or four (the shading kernel runs at the rate at which the chip fetches scattered lines).
// v0.w carries the material index bits.
struct alignas(128) TriShade {
  float4 v0, e1, e2;  // as in TriIsect
  float4 n0, n1, n2;
  float4 unused[2];
};
static_assert(sizeof(TriShade) == 128, "TriShade must be one cache line");

enum DiffuseKind : int { DIFF_NONE = 0, DIFF_LAMBERT = 1, DIFF_OREN_NAYAR = 2 };
enum SpecKind : int { SPEC_NONE = 0, SPEC_MIRROR_DIELECTRIC = 1, SPEC_MICROFACET = 2, SPEC_MIRROR_NOOP = 3 };

// A material pre-digested into the lobes compute_scattering_functions would
// add (materials/*.rs), in the order they are added: transmission, diffuse,
// specular.
struct DeviceMaterial {
  float diffuse[3];
  int diffuse_kind;
  float spec[3];
  int spec_kind;
  float trans[3];
  int has_trans;
  float oren_a, oren_b;
  float alpha;  // Trowbridge-Reitz alpha_x = alpha_y
  float ior;
  int n_lobes;
  int _pad[3];
};

struct DeviceRect {
  float pos[3], a[3], b[3];
  float normal[3];     // normalize(a x b)
  float ss[3], ts[3];  // coordinate_system(normal)
  float inv_aa, inv_bb;
  float bmin[3], bmax[3];  // the f32 box the reference's pre-filter tests
  int material;
  int light;
};

struct DevicePlane {
  float pos[3], normal[3];
  float ng[3];  // face_forward(normal, (0,1,0))
  int material;
};

struct DeviceSphere {
  float center[3];
  float radius;
  int material;
};

struct DeviceLight {
  int kind;
  float intensity[3];
  float v0[3], v1[3], v2[3];  // see rrt_light
  float normal[3];            // area: normalize(a x b)
  float area;                 // area: |side_b|^2 (objects/rectangle.rs:131-133)
  float skew_a, skew_b;       // area: (a.b)/|a|^2 and (a.b)/|b|^2, zero for a rectangle proper
  int rect;                   // area: index of its quad in rects[], -1 if none
};

#define RRT_MAX_RECTS 16
#define RRT_MAX_PLANES 4
#define RRT_MAX_SPHERES 16
#define RRT_MAX_LIGHTS 16

// Small tables, passed to kernels by pointer to one device copy.
struct SceneTables {
  int n_tris, n_rects, n_planes, n_spheres, n_lights, n_materials;
  int rect_aabb_mode;
  int has_env;
  int env_w, env_h;
  float distant_env[3];  // sum of DistantLight intensities: what every escaping ray picks up (lights/distant.rs:62-64)
  int infinite_light;    // index of the InfiniteArea light or -1
  // Box around every quad and sphere (their true extent, slightly padded; infinite when the scene has a plane): a ray
  // whose segment of interest misses it needs none of the per-shape tests.
  float others_bmin[3], others_bmax[3];
  DeviceRect rects[RRT_MAX_RECTS];
  DevicePlane planes[RRT_MAX_PLANES];
  DeviceSphere spheres[RRT_MAX_SPHERES];
  DeviceLight lights[RRT_MAX_LIGHTS];
};

struct CameraParams {
  double raster_to_camera[16];
  double camera_to_world[16];
  double aperture, focal_distance;
  double const_inv_w;  // 1 / w of raster_to_camera * (x, y, 0, 1) when that w does not depend on x and y (any perspective or orthographic projection), else 0
  // the f32 values the ray set-up works with, rounded once on the host instead of once per ray
  float camera_to_world_f[12];
  float aperture_f, focal_distance_f;
};

// Host BVH builder (bvh_build.cpp).  Produces nodes and the leaf order.
struct BvhBuildResult {
  BvhNode* nodes = nullptr;      // malloc'ed, n_nodes entries, node 0 is the root
  uint32_t n_nodes = 0;
  uint32_t* order = nullptr;     // malloc'ed, n_tris entries: leaf slot -> scene-order triangle
  uint32_t max_depth = 0;        // inner nodes on the longest root-to-leaf chain
  BvhGrid grid = {};
};
// The grid for a scene whose triangles span [mn, mx] (any finite box; an empty one gives a unit grid).
BvhGrid make_bvh_grid(const float mn[3], const float mx[3]);
// tri_pos: [n][3][3] floats.  threads = 0: hardware concurrency.  Returns false on allocation failure.
bool build_bvh(const float* tri_pos, uint32_t n_tris, uint32_t threads, BvhBuildResult* out);
void free_bvh(BvhBuildResult* r);
// Number of violations of the hierarchy's invariants (0 = sound): see bvh_build.cpp.
uint64_t validate_bvh(const float* tri_pos, uint32_t n_tris, const BvhBuildResult& b);

#if defined(__HIPCC__)
// Device BVH builder (bvh_build_device.hip): the hierarchy and the leaf-ordered triangle records,
// built by kernels and left in device memory (hipMalloc'ed; the caller owns them, also after an error).
constexpr uint32_t kDeviceBuildMinTris = 64;
struct DeviceBvhResult {
  BvhNode* d_nodes = nullptr;
  TriIsect* d_tris = nullptr;
  TriShade* d_tri_shade = nullptr;
  uint32_t n_nodes = 0;
  uint32_t max_depth = 0;   // inner nodes on the longest root-to-leaf chain
  BvhGrid grid = {};
  uint32_t rounds = 0;      // clustering rounds it took
  double seconds = 0.0;     // wall clock, upload of the triangles included
};
// tri_pos / tri_nrm: host [n][3][3] floats, tri_mat: host [n].  Returns an RRT_* code; *error says what failed.
int build_bvh_device(const float* tri_pos, const float* tri_nrm, const uint32_t* tri_mat, uint32_t n, hipStream_t stream,
                     DeviceBvhResult* out, std::string* error);
#endif

}  // namespace rrt
