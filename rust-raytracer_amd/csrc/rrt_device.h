// Device-resident data layout of the path-tracing core (shared by the host
// BVH builder, the upload code and the kernels).  Everything the kernels read
// is structure-of-arrays or 16-byte records so that one lane fetches a whole
// record with dwordx4 loads.
#pragma once
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

// One 4-wide BVH node, 64 B = four 16-byte loads.  The boxes of up to four children are stored as
// 8-bit offsets from the node's own corner: plane = origin[k] + q * 2^(exp[k] - 127), rounded outwards
// by the builder, so a decoded box always contains the exact one (at most 1/255 of the node's extent
// looser per side).  child >= 0: index of an inner node.  child < 0: a leaf of 1..4 triangles of the
// leaf-ordered triangle arrays, ~child = first << 2 | (count - 1).  A slot without a child holds an
// inverted box (qlo = 255, qhi = 0 on every axis), which no ray enters, and repeats the link of
// child 0: should rounding ever let a ray into it -- only a node without any extent can -- the ray
// visits that child twice, which is harmless.
// Four wide rather than two: half the steps per ray.  Compressed rather than six floats per child:
// what bounds traversal is the number of 16-byte loads the lanes of a wave issue to scattered
// addresses -- the memory pipeline of a CU retires about one such load per lane and clock, whatever
// the cache level it is served from -- and a node costs four of them instead of seven.
struct alignas(64) BvhNode {
  float origin[3];      // minimum corner of the union of the children
  uint8_t exp[3];       // per axis: biased exponent of the grid step (the exponent field of a float)
  uint8_t pad0;
  int32_t child[4];
  uint8_t qlo[3][4];    // [axis][child]: grid steps from origin to the child's minimum (rounded down)
  uint8_t qhi[3][4];    // ... to its maximum (rounded up)
  uint32_t pad1[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");
RRT_HD inline bool child_present(const BvhNode& n, int i) { return n.qlo[0][i] <= n.qhi[0][i]; }

// Intersection record of a triangle, 48 B, leaf order: vertex 0 and the two
// edges Moeller-Trumbore needs.  v0.w carries the scene-order object id.
struct TriIsect {
  float4 v0;  // xyz = p0, w = bits of scene-order id
  float4 e1;  // xyz = p1 - p0
  float4 e2;  // xyz = p2 - p0
};

// Shading record of a triangle, 48 B, leaf order: the three vertex normals
// (objects/triangle.rs:84-107).  n0.w carries the material index bits.
struct TriShade {
  float4 n0, n1, n2;
};

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
};
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
  uint32_t rounds = 0;      // clustering rounds it took
  double seconds = 0.0;     // wall clock, upload of the triangles included
};
// tri_pos / tri_nrm: host [n][3][3] floats, tri_mat: host [n].  Returns an RRT_* code; *error says what failed.
int build_bvh_device(const float* tri_pos, const float* tri_nrm, const uint32_t* tri_mat, uint32_t n, hipStream_t stream,
                     DeviceBvhResult* out, std::string* error);
#endif

}  // namespace rrt
