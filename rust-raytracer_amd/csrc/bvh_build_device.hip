// BVH build on the device, for scenes whose host build (bvh_build.cpp) would dominate the upload.
//
// The reference builds its hierarchy on one host thread (`Bvh::build(&mut objects)`, src/scene.rs:196)
// and uses it only as a conservative candidate filter (src/renderer.rs:226-246), so any sound
// hierarchy over the same triangles gives the same hits.  This builder produces the same kind of
// result as the host one -- 48-byte 4-wide nodes with 8-bit child boxes in depth-first order, leaves
// of up to four triangles stored contiguously, leaf-ordered triangle records -- from kernels:
//
//   1. centroid bounds, 63-bit Morton codes, radix sort (hipCUB)
//   2. a binary tree by parallel locally-ordered clustering (Meister & Bittner 2018): every cluster
//      looks at its 2 x kRadius neighbours in Morton order for the one whose union with it has the
//      smallest surface, pairs that choose each other merge; repeated until one cluster is left.
//      While merging, every node gets its triangle count and its SAH cost (a subtree of at most four
//      triangles becomes a leaf when that is cheaper than splitting it).
//   3. top-down over the rounds of step 2 in reverse (nodes made in one round are never each
//      other's ancestors, and parents are made in later rounds than their children): which binary
//      nodes become 4-wide nodes -- each adopts the children of its largest inner children until it
//      has four, as the host collapse does -- and where every leaf's triangles go
//   4. bottom-up: 4-wide nodes below each 4-wide node; top-down again: depth-first positions, and
//      each node is written with its quantised child boxes
//   5. triangle records in leaf order
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rrt_device.h"

namespace rrt {

namespace {

constexpr int kBuildBlock = 256;
#ifndef RRT_PLOC_RADIUS
#define RRT_PLOC_RADIUS 4
#endif
// Neighbours examined on either side in Morton order.  Measured on a 1.09 M-triangle statue and on 50 M triangles of
// scattered small meshes: nodes fetched per ray fall slightly from 64 down to 4 (15.9 -> 14.9 and 30.9 -> 26.0).
constexpr int kRadius = RRT_PLOC_RADIUS;
constexpr uint32_t kMaxLeafTris = 4;  // a leaf reference keeps count - 1 in two bits
constexpr float kNodeCost = 1.0f;   // cost of a traversal step in units of one triangle test (as the host builder)

// Binary-tree node: two 16-byte words.  Ids below the triangle count are single triangles
// (left = scene-order triangle id, right = -1); the others are merges (left, right = node ids).
struct PNode {
  float mn[3];
  int32_t left;
  float mx[3];
  int32_t right;
};
static_assert(sizeof(PNode) == 32, "PNode must be 32 bytes");

enum : uint8_t { kRoleNone = 0, kRoleWide = 1, kRoleAbsorbed = 2, kRoleLeaf = 3 };

__device__ __forceinline__ uint32_t ordered_bits(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
inline float from_ordered_bits(uint32_t u) {
  const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
  std::memcpy(&f, &b, 4);
  return f;
}

__device__ __forceinline__ void tri_bounds(const float* __restrict__ pos, uint32_t tri, float* lo, float* hi) {
  const float* p = pos + (size_t)tri * 9;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    lo[k] = fminf(p[k], fminf(p[3 + k], p[6 + k]));
    hi[k] = fmaxf(p[k], fmaxf(p[3 + k], p[6 + k]));
  }
}

// bounds[0..2] = min, bounds[3..5] = max of the triangle centroids, as ordered integers
__global__ void __launch_bounds__(kBuildBlock) k_centroid_bounds(const float* __restrict__ pos, uint32_t n, uint32_t* bounds) {
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float lo[3], hi[3];
    tri_bounds(pos, i, lo, hi);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float c = 0.5f * lo[k] + 0.5f * hi[k];
      mn[k] = fminf(mn[k], c);
      mx[k] = fmaxf(mx[k], c);
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    for (int d = 32; d > 0; d >>= 1) {
      mn[k] = fminf(mn[k], __shfl_xor(mn[k], d));
      mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], d));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      atomicMin(&bounds[k], ordered_bits(mn[k]));
      atomicMax(&bounds[3 + k], ordered_bits(mx[k]));
    }
  }
}

__device__ __forceinline__ uint64_t spread21(uint64_t v) {  // bit i -> bit 3 i
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

__global__ void __launch_bounds__(kBuildBlock) k_morton(const float* __restrict__ pos, uint32_t n, float3 cmin, float3 scale,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float lo[3], hi[3];
  tri_bounds(pos, i, lo, hi);
  const float c[3] = {0.5f * lo[0] + 0.5f * hi[0], 0.5f * lo[1] + 0.5f * hi[1], 0.5f * lo[2] + 0.5f * hi[2]};
  const float s[3] = {scale.x, scale.y, scale.z}, m[3] = {cmin.x, cmin.y, cmin.z};
  uint64_t q[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float f = fminf(fmaxf((c[k] - m[k]) * s[k], 0.0f), 2097151.0f);
    q[k] = (uint64_t)f;
  }
  keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
  vals[i] = i;
}

__global__ void __launch_bounds__(kBuildBlock) k_init_leaves(const float* __restrict__ pos, const uint32_t* __restrict__ sorted,
                                                             uint32_t n, PNode* __restrict__ nodes, uint32_t* __restrict__ cnt,
                                                             float* __restrict__ cost, uint8_t* __restrict__ leaf_flag,
                                                             uint32_t* __restrict__ cid) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t tri = sorted[i];
  PNode nd;
  tri_bounds(pos, tri, nd.mn, nd.mx);
  nd.left = (int32_t)tri;
  nd.right = -1;
  nodes[i] = nd;
  cnt[i] = 1;
  cost[i] = 1.0f;
  leaf_flag[i] = 1;
  cid[i] = i;
}

__device__ __forceinline__ float half_area(const float* mn, const float* mx) {
  const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
  return dx * dy + dy * dz + dz * dx;
}

// Nearest neighbour of every cluster among the kRadius clusters on either side: smallest surface
// of the union.  Equal surfaces (runs of coincident or zero-area triangles, strips of equal boxes)
// are ordered by a key of the unordered pair {i, j}: the distance |i - j|, then whether the lower
// index is odd, then the lower index.  Any strict order of the pairs makes the smallest pair choose
// each other, so every round merges at least one; this one makes a run of equal surfaces pair up as
// (0,1), (2,3), ... and halve per round, where "the lower index wins" would merge one pair per round
// and build a tree as deep as the run is long.
__global__ void __launch_bounds__(kBuildBlock) k_nearest(const uint32_t* __restrict__ cid, uint32_t n_c,
                                                         const PNode* __restrict__ nodes, uint32_t* __restrict__ nn) {
  __shared__ float4 smn[kBuildBlock + 2 * kRadius], smx[kBuildBlock + 2 * kRadius];
  const int base = (int)(blockIdx.x * kBuildBlock) - kRadius;
  for (int t = threadIdx.x; t < kBuildBlock + 2 * kRadius; t += kBuildBlock) {
    const int j = base + t;
    float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a;
    if (j >= 0 && j < (int)n_c) {
      const float4* p = reinterpret_cast<const float4*>(nodes + cid[j]);
      a = p[0];
      b = p[1];
    }
    smn[t] = a;
    smx[t] = b;
  }
  __syncthreads();
  const int i = (int)(blockIdx.x * kBuildBlock + threadIdx.x);
  if (i >= (int)n_c) return;
  const float4 mn = smn[threadIdx.x + kRadius], mx = smx[threadIdx.x + kRadius];
  float best = INFINITY;
  int best_j = -1;
  uint32_t best_tie = 0;
#pragma unroll
  for (int d = -kRadius; d <= kRadius; ++d) {
    if (d == 0) continue;
    const int j = i + d;
    if (j < 0 || j >= (int)n_c) continue;
    const float4 a = smn[threadIdx.x + kRadius + d], b = smx[threadIdx.x + kRadius + d];
    const float lo[3] = {fminf(mn.x, a.x), fminf(mn.y, a.y), fminf(mn.z, a.z)};
    const float hi[3] = {fmaxf(mx.x, b.x), fmaxf(mx.y, b.y), fmaxf(mx.z, b.z)};
    const float area = half_area(lo, hi);
    const uint32_t tie = ((uint32_t)(d < 0 ? -d : d) << 1) | ((uint32_t)(d < 0 ? j : i) & 1u);
    if (best_j < 0 || area < best || (area == best && tie < best_tie)) {  // (j ascends: of equal keys the lower j stays)
      best = area;
      best_j = j;
      best_tie = tie;
    }
  }
  nn[i] = (uint32_t)best_j;
}

// low word: the cluster stays in the list (alone, or as the merged pair's new node);
// high word: it is the lower half of a pair and becomes a new node
__global__ void __launch_bounds__(kBuildBlock) k_pair_flags(const uint32_t* __restrict__ nn, uint32_t n_c,
                                                            uint64_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_c) return;
  const uint32_t j = nn[i];
  const bool mutual = nn[j] == i;
  const uint64_t stays = (!mutual || i < j) ? 1ull : 0ull;
  const uint64_t merges = (mutual && i < j) ? 1ull : 0ull;
  flags[i] = stays | (merges << 32);
}

__global__ void __launch_bounds__(kBuildBlock) k_merge(const uint32_t* __restrict__ cid, uint32_t* __restrict__ cid_out,
                                                       const uint32_t* __restrict__ nn, const uint64_t* __restrict__ flags,
                                                       const uint64_t* __restrict__ scanned, uint32_t n_c, uint32_t next_node,
                                                       PNode* __restrict__ nodes, uint32_t* __restrict__ cnt,
                                                       float* __restrict__ cost, uint8_t* __restrict__ leaf_flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_c) return;
  const uint64_t f = flags[i];
  if ((f & 1ull) == 0) return;
  const uint64_t s = scanned[i];
  const uint32_t at = (uint32_t)s;
  if ((f >> 32) == 0) {
    cid_out[at] = cid[i];
    return;
  }
  const uint32_t id = next_node + (uint32_t)(s >> 32);
  const uint32_t l = cid[i], r = cid[nn[i]];
  const PNode a = nodes[l], b = nodes[r];
  PNode nd;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    nd.mn[k] = fminf(a.mn[k], b.mn[k]);
    nd.mx[k] = fmaxf(a.mx[k], b.mx[k]);
  }
  nd.left = (int32_t)l;
  nd.right = (int32_t)r;
  nodes[id] = nd;
  const uint32_t c = cnt[l] + cnt[r];
  cnt[id] = c;
  // SAH: a traversal step plus the children weighted by the chance of reaching them, against testing
  // all triangles here (only up to kMaxLeafTris can share a leaf)
  const float area = half_area(nd.mn, nd.mx);
  const float below = cost[l] * half_area(a.mn, a.mx) + cost[r] * half_area(b.mn, b.mx);
  const float split = kNodeCost + (area > 0.0f ? below / area : 0.0f);
  const bool leaf = c <= kMaxLeafTris && (float)c <= split;
  cost[id] = leaf ? (float)c : split;
  leaf_flag[id] = leaf ? 1 : 0;
  cid_out[at] = id;
}

__global__ void k_set_root(uint32_t root, uint8_t* role, uint32_t* first, uint32_t* offset, uint32_t* depth) {
  role[root] = kRoleWide;
  first[root] = 0;
  offset[root] = 0;
  depth[root] = 1;
}

// Writes the scene-order ids of the triangles below `id` (a subtree of at most kMaxLeafTris
// triangles) to order[at...], left subtrees first.
__device__ void write_leaf_order(const PNode* __restrict__ nodes, uint32_t n_tris, uint32_t id, uint32_t at,
                                 uint32_t* __restrict__ order) {
  uint32_t stack[2 * kMaxLeafTris];
  int sp = 0;
  stack[sp++] = id;
  while (sp > 0) {
    const uint32_t x = stack[--sp];
    const PNode nd = nodes[x];
    if (x < n_tris) {
      order[at++] = (uint32_t)nd.left;
    } else if (sp + 2 <= (int)(2 * kMaxLeafTris)) {
      stack[sp++] = (uint32_t)nd.right;  // popped after the left one
      stack[sp++] = (uint32_t)nd.left;
    }
  }
}

// Top-down, one launch per clustering round (latest first): every node [begin, end) that has turned
// out to be a 4-wide node picks its up to four children -- it adopts the two children of its largest
// inner child until it has four or only leaves are left -- marks what it absorbed, and tells each
// child where its triangles start in leaf order.
__global__ void __launch_bounds__(kBuildBlock) k_roles(uint32_t begin, uint32_t end, uint32_t n_tris, const PNode* __restrict__ nodes,
                                                       const uint32_t* __restrict__ cnt, const uint8_t* __restrict__ leaf_flag,
                                                       uint8_t* __restrict__ role, uint32_t* __restrict__ first,
                                                       int4* __restrict__ wide_children, uint32_t* __restrict__ order) {
  const uint32_t id = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= end || role[id] != kRoleWide) return;
  int32_t c[4] = {-1, -1, -1, -1};
  uint32_t f[4] = {0, 0, 0, 0};
  const PNode me = nodes[id];
  const uint32_t f0 = first[id];
  c[0] = me.left;
  f[0] = f0;
  c[1] = me.right;
  f[1] = f0 + cnt[me.left];
  int k = 2;
  while (k < 4) {
    int best = -1;
    float best_area = -1.0f;
    for (int i = 0; i < k; ++i) {
      if (leaf_flag[c[i]]) continue;
      const PNode x = nodes[c[i]];
      const float a = half_area(x.mn, x.mx);
      if (a > best_area) { best = i; best_area = a; }
    }
    if (best < 0) break;
    const uint32_t x = (uint32_t)c[best];
    role[x] = kRoleAbsorbed;
    const PNode nd = nodes[x];
    const uint32_t fx = f[best];
    c[best] = nd.left;
    f[best] = fx;
    c[k] = nd.right;
    f[k] = fx + cnt[nd.left];
    ++k;
  }
  for (int i = 0; i < k; ++i) {
    const uint32_t x = (uint32_t)c[i];
    first[x] = f[i];
    if (leaf_flag[x]) {
      role[x] = kRoleLeaf;
      write_leaf_order(nodes, n_tris, x, f[i], order);
    } else {
      role[x] = kRoleWide;
    }
  }
  wide_children[id] = make_int4(c[0], c[1], c[2], c[3]);
}

// Bottom-up, one launch per round (earliest first): 4-wide nodes in the subtree of every 4-wide node.
__global__ void __launch_bounds__(kBuildBlock) k_below(uint32_t begin, uint32_t end, const uint8_t* __restrict__ role,
                                                       const int4* __restrict__ wide_children, uint32_t* __restrict__ below) {
  const uint32_t id = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= end || role[id] != kRoleWide) return;
  const int4 ch = wide_children[id];
  const int32_t c[4] = {ch.x, ch.y, ch.z, ch.w};
  uint32_t total = 1;
  for (int i = 0; i < 4; ++i)
    if (c[i] >= 0 && role[c[i]] == kRoleWide) total += below[c[i]];
  below[id] = total;
}

// Top-down again: the children's positions in depth-first order, and the node itself
// (corner on the scene's grid, per-axis shift, child boxes rounded outwards: encode_node_boxes).
__global__ void __launch_bounds__(kBuildBlock) k_emit(uint32_t begin, uint32_t end, const PNode* __restrict__ nodes,
                                                      const uint32_t* __restrict__ cnt, const uint8_t* __restrict__ role,
                                                      const uint32_t* __restrict__ first, const int4* __restrict__ wide_children,
                                                      const uint32_t* __restrict__ below, uint32_t* __restrict__ offset,
                                                      uint32_t* __restrict__ depth, uint32_t* __restrict__ max_depth,
                                                      BvhGrid grid, BvhNode* __restrict__ out) {
  const uint32_t id = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= end || role[id] != kRoleWide) return;
  const int4 ch = wide_children[id];
  const int32_t c[4] = {ch.x, ch.y, ch.z, ch.w};
  const uint32_t me = offset[id], d = depth[id];
  atomicMax(max_depth, d);
  float mn[4][3], mx[4][3];
  int n = 0;
  for (int i = 0; i < 4; ++i) {
    if (c[i] < 0) break;
    const PNode box = nodes[c[i]];
    for (int k = 0; k < 3; ++k) {
      mn[i][k] = box.mn[k];
      mx[i][k] = box.mx[k];
    }
    ++n;
  }
  BvhNode node;
  encode_node_boxes(grid, mn, mx, n, &node);
  uint32_t next = me + 1;
  for (int i = 0; i < 4; ++i) {
    node.child[i] = 0;
    if (i >= n) continue;
    const uint32_t x = (uint32_t)c[i];
    if (role[x] == kRoleWide) {
      node.child[i] = (int32_t)next;
      offset[x] = next;
      depth[x] = d + 1;
      next += below[x];
    } else {
      node.child[i] = ~(int32_t)((first[x] << 2) | (cnt[x] - 1));
    }
  }
  for (int i = n; i < 4; ++i) node.child[i] = node.child[0];  // a slot without a child repeats the first link
  out[me] = node;
}

// Leaf-ordered records from the scene-order arrays.
__global__ void __launch_bounds__(kBuildBlock) k_records(const float* __restrict__ pos, const float* __restrict__ nrm,
                                                         const uint32_t* __restrict__ mat, const uint32_t* __restrict__ order,
                                                         uint32_t n, TriIsect* __restrict__ tris, TriShade* __restrict__ shade) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n) return;
  const uint32_t id = order[slot];
  const float* p = pos + (size_t)id * 9;
  const float* q = nrm + (size_t)id * 9;
  TriIsect t;
  t.v0 = make_float4(p[0], p[1], p[2], __uint_as_float(id));
  t.e1 = make_float4(p[3] - p[0], p[4] - p[1], p[5] - p[2], 0.0f);
  t.e2 = make_float4(p[6] - p[0], p[7] - p[1], p[8] - p[2], 0.0f);
  tris[slot] = t;
  TriShade s;
  s.v0 = make_float4(p[0], p[1], p[2], __uint_as_float(mat[id]));
  s.e1 = t.e1;
  s.e2 = t.e2;
  s.n0 = make_float4(q[0], q[1], q[2], 0.0f);
  s.n1 = make_float4(q[3], q[4], q[5], 0.0f);
  s.n2 = make_float4(q[6], q[7], q[8], 0.0f);
  s.unused[0] = s.unused[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  shade[slot] = s;
}

inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + kBuildBlock - 1) / kBuildBlock); }

struct Temp {
  std::vector<void*> ptrs;
  ~Temp() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  template <typename T>
  bool alloc(T** out, size_t count) {
    void* p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return false;
    ptrs.push_back(p);
    *out = (T*)p;
    return true;
  }
};

}  // namespace

#define BUILD_TRY(expr)                                                                \
  do {                                                                                 \
    const hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess) {                                                            \
      *error = std::string("device BVH build: ") + #expr + ": " + hipGetErrorString(e_); \
      return RRT_E_DEVICE;                                                             \
    }                                                                                  \
  } while (0)

int build_bvh_device(const float* tri_pos, const float* tri_nrm, const uint32_t* tri_mat, uint32_t n, hipStream_t stream,
                     DeviceBvhResult* out, std::string* error) {
  *out = DeviceBvhResult();
  if (n < kDeviceBuildMinTris) {
    *error = "device BVH build: too few triangles";
    return RRT_E_INVALID;
  }
  const auto t_begin = std::chrono::steady_clock::now();
  Temp tmp;
  const size_t n_all = 2 * (size_t)n;  // n triangles + n - 1 merges
  float *d_pos = nullptr, *d_nrm = nullptr;
  uint32_t* d_mat = nullptr;
  uint32_t *d_bounds = nullptr, *d_vals = nullptr, *d_vals_sorted = nullptr, *d_cid[2] = {nullptr, nullptr}, *d_nn = nullptr;
  uint64_t *d_keys = nullptr, *d_keys_sorted = nullptr, *d_flags = nullptr, *d_scanned = nullptr;
  PNode* d_pnodes = nullptr;
  uint32_t *d_cnt = nullptr, *d_first = nullptr, *d_below = nullptr, *d_offset = nullptr, *d_depth = nullptr, *d_order = nullptr,
           *d_max_depth = nullptr;
  float* d_cost = nullptr;
  uint8_t *d_leaf_flag = nullptr, *d_role = nullptr;
  int4* d_wide = nullptr;
  bool ok = tmp.alloc(&d_pos, (size_t)n * 9) && tmp.alloc(&d_nrm, (size_t)n * 9) && tmp.alloc(&d_mat, n) &&
            tmp.alloc(&d_bounds, 8) && tmp.alloc(&d_keys, n) && tmp.alloc(&d_keys_sorted, n) && tmp.alloc(&d_vals, n) &&
            tmp.alloc(&d_vals_sorted, n) && tmp.alloc(&d_cid[0], n) && tmp.alloc(&d_cid[1], n) && tmp.alloc(&d_nn, n) &&
            tmp.alloc(&d_flags, n) && tmp.alloc(&d_scanned, n) && tmp.alloc(&d_pnodes, n_all) && tmp.alloc(&d_cnt, n_all) &&
            tmp.alloc(&d_cost, n_all) && tmp.alloc(&d_leaf_flag, n_all) && tmp.alloc(&d_role, n_all) &&
            tmp.alloc(&d_first, n_all) && tmp.alloc(&d_below, n_all) && tmp.alloc(&d_offset, n_all) &&
            tmp.alloc(&d_depth, n_all) && tmp.alloc(&d_order, n) && tmp.alloc(&d_wide, n_all) && tmp.alloc(&d_max_depth, 1);
  if (!ok) {
    *error = "device BVH build: out of device memory";
    return RRT_E_NOMEM;
  }
  BUILD_TRY(hipMemcpyAsync(d_pos, tri_pos, sizeof(float) * 9 * (size_t)n, hipMemcpyHostToDevice, stream));
  BUILD_TRY(hipMemcpyAsync(d_nrm, tri_nrm, sizeof(float) * 9 * (size_t)n, hipMemcpyHostToDevice, stream));
  BUILD_TRY(hipMemcpyAsync(d_mat, tri_mat, sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, stream));

  // ---- 1. Morton order
  {
    const uint32_t init[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u, 0u, 0u};
    BUILD_TRY(hipMemcpyAsync(d_bounds, init, sizeof(init), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_centroid_bounds, dim3(std::min(blocks_for(n), 4096u)), dim3(kBuildBlock), 0, stream, d_pos, n, d_bounds);
    uint32_t h[8];
    BUILD_TRY(hipMemcpyAsync(h, d_bounds, sizeof(h), hipMemcpyDeviceToHost, stream));
    BUILD_TRY(hipStreamSynchronize(stream));
    float3 cmin, scale;
    float* cm = &cmin.x;
    float* sc = &scale.x;
    float widest = 0.0f;
    for (int k = 0; k < 3; ++k) widest = std::max(widest, from_ordered_bits(h[3 + k]) - from_ordered_bits(h[k]));
    for (int k = 0; k < 3; ++k) {
      const float lo = from_ordered_bits(h[k]), hi = from_ordered_bits(h[3 + k]);
      cm[k] = lo;
      // one scale for the three axes: the cells of the Morton grid are cubes, whatever the shape of the scene
      sc[k] = widest > 0.0f ? 2097152.0f / widest : 0.0f;
    }
    hipLaunchKernelGGL(k_morton, dim3(blocks_for(n)), dim3(kBuildBlock), 0, stream, d_pos, n, cmin, scale, d_keys, d_vals);
    size_t sort_bytes = 0;
    BUILD_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, d_keys, d_keys_sorted, d_vals, d_vals_sorted, (int)n, 0, 63, stream));
    char* d_sort = nullptr;
    if (!tmp.alloc(&d_sort, sort_bytes)) {
      *error = "device BVH build: out of device memory";
      return RRT_E_NOMEM;
    }
    BUILD_TRY(hipcub::DeviceRadixSort::SortPairs(d_sort, sort_bytes, d_keys, d_keys_sorted, d_vals, d_vals_sorted, (int)n, 0, 63, stream));
  }
  hipLaunchKernelGGL(k_init_leaves, dim3(blocks_for(n)), dim3(kBuildBlock), 0, stream, d_pos, d_vals_sorted, n, d_pnodes, d_cnt,
                     d_cost, d_leaf_flag, d_cid[0]);

  // ---- 2. clustering rounds
  size_t scan_bytes = 0;
  BUILD_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_flags, d_scanned, (int)n, stream));
  char* d_scan = nullptr;
  if (!tmp.alloc(&d_scan, scan_bytes)) {
    *error = "device BVH build: out of device memory";
    return RRT_E_NOMEM;
  }
  std::vector<uint32_t> round_begin;  // first node id made in each round; round_begin.back() = one past the last node
  uint32_t n_c = n, next_node = n;
  int cur = 0;
  // Meshes take 70-120 rounds (a round merges a third to a half of the clusters at first, fewer near the top).
  // Input on which the clustering crawls -- a round costs three kernels, a scan and a synchronisation, and
  // the tree comes out as deep as the rounds are many -- is left to the host builder.
  uint32_t log2n = 0;
  while ((1ull << log2n) < n) ++log2n;
  const size_t max_rounds = 16 * (size_t)log2n + 64;
  while (n_c > 1) {
    if (round_begin.size() >= max_rounds) {
      *error = "device BVH build: clustering takes too many rounds (degenerate input)";
      return RRT_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_nearest, dim3(blocks_for(n_c)), dim3(kBuildBlock), 0, stream, d_cid[cur], n_c, d_pnodes, d_nn);
    hipLaunchKernelGGL(k_pair_flags, dim3(blocks_for(n_c)), dim3(kBuildBlock), 0, stream, d_nn, n_c, d_flags);
    BUILD_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan, scan_bytes, d_flags, d_scanned, (int)n_c, stream));
    uint64_t last_scan = 0, last_flag = 0;
    BUILD_TRY(hipMemcpyAsync(&last_scan, d_scanned + (n_c - 1), 8, hipMemcpyDeviceToHost, stream));
    BUILD_TRY(hipMemcpyAsync(&last_flag, d_flags + (n_c - 1), 8, hipMemcpyDeviceToHost, stream));
    hipLaunchKernelGGL(k_merge, dim3(blocks_for(n_c)), dim3(kBuildBlock), 0, stream, d_cid[cur], d_cid[cur ^ 1], d_nn, d_flags,
                       d_scanned, n_c, next_node, d_pnodes, d_cnt, d_cost, d_leaf_flag);
    BUILD_TRY(hipStreamSynchronize(stream));
    const uint64_t total = last_scan + last_flag;
    const uint32_t stays = (uint32_t)total, merges = (uint32_t)(total >> 32);
    if (merges == 0 || stays >= n_c || (uint64_t)next_node + merges > n_all - 1) {
      *error = "device BVH build: clustering made no progress";
      return RRT_E_DEVICE;
    }
    round_begin.push_back(next_node);
    next_node += merges;
    n_c = stays;
    cur ^= 1;
  }
  round_begin.push_back(next_node);
  if (next_node != 2 * n - 1) {
    *error = "device BVH build: wrong number of merges";
    return RRT_E_DEVICE;
  }
  const uint32_t root = next_node - 1;
  const size_t rounds = round_begin.size() - 1;

  // ---- 3. roles and leaf order, top-down
  BUILD_TRY(hipMemsetAsync(d_role, 0, n_all, stream));
  BUILD_TRY(hipMemsetAsync(d_max_depth, 0, sizeof(uint32_t), stream));
  hipLaunchKernelGGL(k_set_root, dim3(1), dim3(1), 0, stream, root, d_role, d_first, d_offset, d_depth);
  for (size_t r = rounds; r-- > 0;) {
    const uint32_t b = round_begin[r], e = round_begin[r + 1];
    hipLaunchKernelGGL(k_roles, dim3(blocks_for(e - b)), dim3(kBuildBlock), 0, stream, b, e, n, d_pnodes, d_cnt, d_leaf_flag, d_role,
                       d_first, d_wide, d_order);
  }
  // ---- 4. subtree sizes bottom-up, then positions and nodes top-down
  for (size_t r = 0; r < rounds; ++r) {
    const uint32_t b = round_begin[r], e = round_begin[r + 1];
    hipLaunchKernelGGL(k_below, dim3(blocks_for(e - b)), dim3(kBuildBlock), 0, stream, b, e, d_role, d_wide, d_below);
  }
  uint32_t n_nodes = 0;
  PNode root_box;  // the last merge: its box spans every triangle
  BUILD_TRY(hipMemcpyAsync(&n_nodes, d_below + root, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  BUILD_TRY(hipMemcpyAsync(&root_box, d_pnodes + root, sizeof(PNode), hipMemcpyDeviceToHost, stream));
  BUILD_TRY(hipStreamSynchronize(stream));
  out->grid = make_bvh_grid(root_box.mn, root_box.mx);
  if (n_nodes == 0 || n_nodes > n) {
    *error = "device BVH build: bad node count";
    return RRT_E_DEVICE;
  }
  BvhNode* d_nodes = nullptr;
  TriIsect* d_tris = nullptr;
  TriShade* d_shade = nullptr;
  if (hipMalloc((void**)&d_nodes, sizeof(BvhNode) * (size_t)n_nodes) != hipSuccess ||
      hipMalloc((void**)&d_tris, sizeof(TriIsect) * (size_t)n) != hipSuccess ||
      hipMalloc((void**)&d_shade, sizeof(TriShade) * (size_t)n) != hipSuccess) {
    (void)hipFree(d_nodes);
    (void)hipFree(d_tris);
    (void)hipFree(d_shade);
    *error = "device BVH build: out of device memory";
    return RRT_E_NOMEM;
  }
  out->d_nodes = d_nodes;  // (the caller frees these when it gets an error back)
  out->d_tris = d_tris;
  out->d_tri_shade = d_shade;
  for (size_t r = rounds; r-- > 0;) {
    const uint32_t b = round_begin[r], e = round_begin[r + 1];
    hipLaunchKernelGGL(k_emit, dim3(blocks_for(e - b)), dim3(kBuildBlock), 0, stream, b, e, d_pnodes, d_cnt, d_role, d_first, d_wide,
                       d_below, d_offset, d_depth, d_max_depth, out->grid, d_nodes);
  }
  // ---- 5. triangle records
  hipLaunchKernelGGL(k_records, dim3(blocks_for(n)), dim3(kBuildBlock), 0, stream, d_pos, d_nrm, d_mat, d_order, n, d_tris, d_shade);
  uint32_t max_depth = 0;
  BUILD_TRY(hipMemcpyAsync(&max_depth, d_max_depth, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  BUILD_TRY(hipStreamSynchronize(stream));
  BUILD_TRY(hipGetLastError());
  out->n_nodes = n_nodes;
  out->max_depth = max_depth;
  out->rounds = (uint32_t)rounds;
  out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  return RRT_OK;
}

}  // namespace rrt
