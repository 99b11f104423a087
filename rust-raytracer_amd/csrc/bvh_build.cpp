// Host-side BVH builder for the device traversal kernels.
//
// The reference builds its hierarchy with bvh 0.8.0 (`Bvh::build(&mut objects)`,
// src/scene.rs:196) and only ever uses it as a conservative candidate filter
// (src/renderer.rs:226-246): the closest hit is the minimum over exact tests,
// so any correct hierarchy over the same triangles gives the same answer.
// This one is built for the GPU: a binary tree by binned SAH (16 bins, all three
// axes; subtrees built on separate threads) with leaves of up to 4 triangles
// stored contiguously, then collapsed into 4-wide nodes -- each node adopts the
// grandchildren of its largest children until it has four -- laid out in
// depth-first order.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "rrt_device.h"
#include "rrt_threads.h"

namespace rrt {

std::atomic<int> g_thread_creations_allowed{-1};

namespace {

struct Box {
  float mn[3], mx[3];
  void reset() {
    for (int k = 0; k < 3; ++k) { mn[k] = INFINITY; mx[k] = -INFINITY; }
  }
  void grow(const float* lo, const float* hi) {
    for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], lo[k]); mx[k] = std::max(mx[k], hi[k]); }
  }
  void grow(const Box& b) { grow(b.mn, b.mx); }
  void grow_point(const float* p) { grow(p, p); }
  float half_area() const {
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (!(dx >= 0.0f)) return 0.0f;
    return dx * dy + dy * dz + dz * dx;
  }
};

// Node of the binary tree the SAH pass builds.  child >= 0: inner node; child < 0: leaf
// reference (see BvhNode); lcount / rcount: triangles in a leaf child, 0 for an inner child.
struct Node2 {
  float lmin[3], lmax[3];
  float rmin[3], rmax[3];
  int32_t left, right;
  int32_t lcount, rcount;
};

#ifndef RRT_SAH_BINS
#define RRT_SAH_BINS 16
#endif
constexpr int kBins = RRT_SAH_BINS;
#ifndef RRT_SAH_NODE_COST
#define RRT_SAH_NODE_COST 1.0  // cost of a traversal step in units of one triangle test
#endif
#ifndef RRT_MAX_LEAF
#define RRT_MAX_LEAF 4
#endif
constexpr uint32_t kMaxLeaf = RRT_MAX_LEAF;  // at most 4: a leaf reference keeps count - 1 in two bits
constexpr uint32_t kSahDepthLimit = 40;     // below this depth splits fall back to the object median
constexpr uint32_t kSpawnThreshold = 1u << 16;
constexpr uint32_t kParallelPassThreshold = 1u << 21;  // triangles in a node from which its passes are split over threads

struct Builder {
  const float* pos;
  uint32_t n;
  std::vector<float> tmin, tmax, cent;  // [n][3]
  uint32_t* idx;
  Node2* nodes;
  std::atomic<uint32_t> node_count{0};
  std::atomic<int> free_threads{0};
  std::atomic<uint32_t> max_depth{0};
  std::atomic<bool> failed{false};  // a worker thread ran out of memory: the tree is incomplete

  void prepare() {
    tmin.resize((size_t)n * 3);
    tmax.resize((size_t)n * 3);
    cent.resize((size_t)n * 3);
    for (uint32_t i = 0; i < n; ++i) {
      const float* p = pos + (size_t)i * 9;
      for (int k = 0; k < 3; ++k) {
        const float lo = std::min(p[k], std::min(p[3 + k], p[6 + k]));
        const float hi = std::max(p[k], std::max(p[3 + k], p[6 + k]));
        tmin[(size_t)i * 3 + k] = lo;
        tmax[(size_t)i * 3 + k] = hi;
        cent[(size_t)i * 3 + k] = 0.5f * lo + 0.5f * hi;
      }
    }
  }

  // Up to `want` of the free threads (the caller gives them back).
  int take_threads(int want) {
    int got = 0;
    while (got < want) {
      if (free_threads.fetch_sub(1, std::memory_order_relaxed) > 0) ++got;
      else { free_threads.fetch_add(1, std::memory_order_relaxed); break; }
    }
    return got;
  }
  // fn(worker, begin, end) over [lo, hi) cut into helpers + 1 ranges, the last one on this thread (and
  // any range whose thread could not be started).  fn must not throw: it only reads and writes
  // memory its caller has set up.
  template <class Fn>
  static void for_ranges(uint32_t lo, uint32_t hi, int helpers, Fn fn) {
    const uint64_t per = ((uint64_t)(hi - lo) + (uint64_t)helpers) / ((uint64_t)helpers + 1);
    std::thread pool[8];
    if (helpers > 8) helpers = 8;
    JoinGuard join{pool, helpers};
    for (int w = 0; w < helpers; ++w) {
      const uint32_t a = (uint32_t)std::min<uint64_t>(hi, lo + per * (uint64_t)w), b = (uint32_t)std::min<uint64_t>(hi, (uint64_t)a + per);
      auto run = [=]() { fn(w, a, b); };
      if (!try_spawn(&pool[w], run)) run();
    }
    const uint32_t a = (uint32_t)std::min<uint64_t>(hi, lo + per * (uint64_t)helpers);
    fn(helpers, a, hi);
  }

  // Builds the subtree over idx[lo, hi).  Reports how the parent must refer
  // to it: an inner node index (count 0) or a leaf (~lo, count).
  void build(uint32_t lo, uint32_t hi, uint32_t depth, int32_t* child, int32_t* count, Box* box) {
    const uint32_t cnt = hi - lo;
    Box bounds, cbounds;
    bounds.reset();
    cbounds.reset();
    // The passes over a large node's triangles are cut into ranges for whatever threads are free: near
    // the root there are few subtrees to hand out yet, and one thread walking 50 M triangles per level
    // was a third of the build.
    const int helpers = cnt >= kParallelPassThreshold ? take_threads(7) : 0;
    {
      struct Partial { Box bounds, cbounds; };
      std::vector<Partial> part((size_t)helpers + 1);
      for_ranges(lo, hi, helpers, [&](int w, uint32_t a, uint32_t b) {
        Partial& pt = part[(size_t)w];
        pt.bounds.reset();
        pt.cbounds.reset();
        for (uint32_t i = a; i < b; ++i) {
          const size_t t = (size_t)idx[i] * 3;
          pt.bounds.grow(&tmin[t], &tmax[t]);
          pt.cbounds.grow_point(&cent[t]);
        }
      });
      for (const Partial& pt : part) {
        bounds.grow(pt.bounds);
        cbounds.grow(pt.cbounds);
      }
    }
    struct GiveBack {  // the helpers go back before the subtrees are handed out (or at any earlier return)
      Builder* b; int n;
      void now() { if (n) b->free_threads.fetch_add(n, std::memory_order_relaxed); n = 0; }
      ~GiveBack() { now(); }
    } give_back{this, helpers};
    *box = bounds;
    uint32_t d = max_depth.load(std::memory_order_relaxed);
    while (depth > d && !max_depth.compare_exchange_weak(d, depth, std::memory_order_relaxed)) {}

    auto make_leaf = [&]() {
      *child = ~(int32_t)((lo << 2) | (cnt - 1));  // leaf reference: first triangle and count - 1 in one word
      *count = (int32_t)cnt;
    };
    if (cnt == 1) { make_leaf(); return; }

    uint32_t mid = lo;
    bool have_split = false;
    if (depth < kSahDepthLimit) {
      float best_cost = INFINITY;
      int best_axis = -1, best_bin = -1;
      // one pass over the triangles fills the bins of all three axes
      Box bin_box[3][kBins];
      uint32_t bin_cnt[3][kBins];
      float c0[3], scale[3];
      for (int axis = 0; axis < 3; ++axis) {
        c0[axis] = cbounds.mn[axis];
        const float extent = cbounds.mx[axis] - c0[axis];
        scale[axis] = extent > 0.0f ? (float)kBins / extent : 0.0f;  // 0: everything in bin 0, no split on this axis
        for (int b = 0; b < kBins; ++b) { bin_box[axis][b].reset(); bin_cnt[axis][b] = 0; }
      }
      {
        struct Bins { Box box[3][kBins]; uint32_t cnt[3][kBins]; };
        std::vector<Bins> part((size_t)helpers + 1);
        for_ranges(lo, hi, helpers, [&](int w, uint32_t a, uint32_t e) {
          Bins& bn = part[(size_t)w];
          for (int axis = 0; axis < 3; ++axis)
            for (int b = 0; b < kBins; ++b) { bn.box[axis][b].reset(); bn.cnt[axis][b] = 0; }
          for (uint32_t i = a; i < e; ++i) {
            const size_t t = (size_t)idx[i] * 3;
            for (int axis = 0; axis < 3; ++axis) {
              int b = (int)((cent[t + axis] - c0[axis]) * scale[axis]);
              b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
              bn.cnt[axis][b]++;
              bn.box[axis][b].grow(&tmin[t], &tmax[t]);
            }
          }
        });
        for (const Bins& bn : part)
          for (int axis = 0; axis < 3; ++axis)
            for (int b = 0; b < kBins; ++b) {
              bin_cnt[axis][b] += bn.cnt[axis][b];
              if (bn.cnt[axis][b]) bin_box[axis][b].grow(bn.box[axis][b]);
            }
        give_back.now();
      }
      for (int axis = 0; axis < 3; ++axis) {
        if (!(scale[axis] > 0.0f)) continue;
        float right_area[kBins];
        uint32_t right_cnt[kBins];
        Box acc;
        acc.reset();
        uint32_t c = 0;
        for (int b = kBins - 1; b > 0; --b) {
          acc.grow(bin_box[axis][b]);
          c += bin_cnt[axis][b];
          right_area[b] = acc.half_area();
          right_cnt[b] = c;
        }
        acc.reset();
        c = 0;
        for (int b = 0; b < kBins - 1; ++b) {
          acc.grow(bin_box[axis][b]);
          c += bin_cnt[axis][b];
          if (c == 0 || right_cnt[b + 1] == 0) continue;
          const float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
        }
      }
      if (best_axis >= 0) {
        // leaf cost n * A versus 1 traversal step + children
        const float parent_area = bounds.half_area();
        const float split_cost = (float)RRT_SAH_NODE_COST + (parent_area > 0.0f ? best_cost / parent_area : 0.0f);
        if (cnt <= kMaxLeaf && (float)cnt <= split_cost) { make_leaf(); return; }
        const float c0 = cbounds.mn[best_axis];
        const float scale = (float)kBins / (cbounds.mx[best_axis] - c0);
        uint32_t* m = std::partition(idx + lo, idx + hi, [&](uint32_t t) {
          int b = (int)((cent[(size_t)t * 3 + best_axis] - c0) * scale);
          b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
          return b <= best_bin;
        });
        mid = (uint32_t)(m - idx);
        have_split = mid > lo && mid < hi;
      }
    }
    if (!have_split) {
      if (cnt <= kMaxLeaf) { make_leaf(); return; }
      // object median along the widest centroid axis (all centroids equal, or too deep for SAH)
      int axis = 0;
      for (int k = 1; k < 3; ++k)
        if (cbounds.mx[k] - cbounds.mn[k] > cbounds.mx[axis] - cbounds.mn[axis]) axis = k;
      mid = lo + cnt / 2;
      std::nth_element(idx + lo, idx + mid, idx + hi, [&](uint32_t a, uint32_t b) {
        return cent[(size_t)a * 3 + axis] < cent[(size_t)b * 3 + axis];
      });
    }

    const uint32_t me = node_count.fetch_add(1, std::memory_order_relaxed);
    Node2 node;
    std::memset(&node, 0, sizeof(node));
    Box lb, rb;
    lb.reset();
    rb.reset();
    int32_t lc = ~0, lcount = 1, rc = ~0, rcount = 1;  // (what a subtree abandoned on an allocation failure leaves behind)
    bool spawned = false;
    std::thread worker;
    if (cnt >= kSpawnThreshold) {
      if (free_threads.fetch_sub(1, std::memory_order_relaxed) > 0) {
        JoinGuard join{&worker, 1};  // joined also when the right subtree below throws
        auto left = [&]() { guarded(&failed, [&]() { build(lo, mid, depth + 1, &lc, &lcount, &lb); }); };
        spawned = try_spawn(&worker, left);
        if (!spawned) build(lo, mid, depth + 1, &lc, &lcount, &lb);
        struct Return {  // the thread goes back to the pool once the left subtree is done or abandoned
          Builder* b;
          ~Return() { b->free_threads.fetch_add(1, std::memory_order_relaxed); }
        } give{this};
        build(mid, hi, depth + 1, &rc, &rcount, &rb);
        if (worker.joinable()) worker.join();
      } else {
        free_threads.fetch_add(1, std::memory_order_relaxed);
        build(lo, mid, depth + 1, &lc, &lcount, &lb);
        build(mid, hi, depth + 1, &rc, &rcount, &rb);
      }
    } else {
      build(lo, mid, depth + 1, &lc, &lcount, &lb);
      build(mid, hi, depth + 1, &rc, &rcount, &rb);
    }
    (void)spawned;
    for (int k = 0; k < 3; ++k) {
      node.lmin[k] = lb.mn[k]; node.lmax[k] = lb.mx[k];
      node.rmin[k] = rb.mn[k]; node.rmax[k] = rb.mx[k];
    }
    node.left = lc; node.lcount = lcount;
    node.right = rc; node.rcount = rcount;
    nodes[me] = node;
    *child = (int32_t)me;
    *count = 0;
  }
};

// Collapses the binary tree into 4-wide nodes in depth-first pre-order.  Two passes, both spread
// over threads near the top of the tree: the first counts the 4-wide nodes below every binary node
// that becomes one (so that the second knows where each subtree goes), the second writes them.
struct Collapser {
  const Node2* in;
  uint32_t* below = nullptr;   // per binary node: 4-wide nodes in the subtree it roots (0 if it is adopted away)
  BvhNode* out = nullptr;
  BvhGrid grid = {};
  std::atomic<uint32_t> max_depth{0};
  std::atomic<int> free_threads{0};
  std::atomic<bool> failed{false};

  struct Child {
    Box box;
    int32_t ref;    // binary-tree reference: inner index or leaf code
    bool inner;
  };

  static void children_of(const Node2& n, Child* l, Child* r) {
    for (int k = 0; k < 3; ++k) {
      l->box.mn[k] = n.lmin[k]; l->box.mx[k] = n.lmax[k];
      r->box.mn[k] = n.rmin[k]; r->box.mx[k] = n.rmax[k];
    }
    l->ref = n.left; l->inner = n.lcount == 0 && n.left >= 0;
    r->ref = n.right; r->inner = n.rcount == 0 && n.right >= 0;
  }

  // The up to four children of the 4-wide node that binary node `old_i` becomes.
  int gather(uint32_t old_i, Child* c) const {
    int n = 2;
    children_of(in[old_i], &c[0], &c[1]);
    // an absent right child (single-leaf scene) has an inverted box: drop it
    if (!(c[1].box.mx[0] >= c[1].box.mn[0])) n = 1;
    while (n < 4) {
      int best = -1;
      float best_area = -1.0f;
      for (int i = 0; i < n; ++i)
        if (c[i].inner && c[i].box.half_area() > best_area) { best = i; best_area = c[i].box.half_area(); }
      if (best < 0) break;
      Child l, r;
      children_of(in[c[best].ref], &l, &r);
      c[best] = l;
      c[n++] = r;
    }
    return n;
  }

  // Runs fn(i) for the inner children, the first ones on threads of their own while any are free
  // and the tree is still shallow enough for the subtrees to be worth it.
  template <class Fn>
  void for_inner_children(const Child* c, int n, uint32_t depth, Fn fn) {
    std::thread workers[4];
    int n_workers = 0;
    struct Finish {  // joins the workers and gives their threads back, on every way out
      Collapser* self; std::thread* workers; int* n;
      ~Finish() {
        for (int w = 0; w < *n; ++w) {
          if (workers[w].joinable()) workers[w].join();
          self->free_threads.fetch_add(1, std::memory_order_relaxed);
        }
      }
    } finish{this, workers, &n_workers};
    int last_inner = -1;
    for (int i = 0; i < n; ++i)
      if (c[i].inner) last_inner = i;
    for (int i = 0; i < n; ++i) {
      if (!c[i].inner) continue;
      bool started = false;
      if (i != last_inner && depth <= 6) {
        if (free_threads.fetch_sub(1, std::memory_order_relaxed) > 0) {
          auto run = [=]() { guarded(&failed, [&]() { fn(i); }); };
          started = try_spawn(&workers[n_workers], run);
          if (started) ++n_workers;
          else free_threads.fetch_add(1, std::memory_order_relaxed);
        } else {
          free_threads.fetch_add(1, std::memory_order_relaxed);
        }
      }
      if (!started) fn(i);
    }
  }

  uint32_t count(uint32_t old_i, uint32_t depth) {
    Child c[4];
    const int n = gather(old_i, c);
    uint32_t sizes[4] = {0, 0, 0, 0};
    for_inner_children(c, n, depth, [&](int i) { sizes[i] = count((uint32_t)c[i].ref, depth + 1); });
    const uint32_t total = 1 + sizes[0] + sizes[1] + sizes[2] + sizes[3];
    below[old_i] = total;
    return total;
  }

  // Writes the 4-wide node for binary node `old_i` at out[me], its subtrees behind it.
  void emit(uint32_t old_i, uint32_t me, uint32_t depth) {
    uint32_t d = max_depth.load(std::memory_order_relaxed);
    while (depth > d && !max_depth.compare_exchange_weak(d, depth, std::memory_order_relaxed)) {}
    Child c[4];
    const int n = gather(old_i, c);
    const bool empty = !(c[0].box.mx[0] >= c[0].box.mn[0]);  // a scene without triangles
    BvhNode node;
    std::memset(&node, 0, sizeof(node));
    float mn[4][3], mx[4][3];
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < 3; ++k) { mn[i][k] = c[i].box.mn[k]; mx[i][k] = c[i].box.mx[k]; }
    encode_node_boxes(grid, mn, mx, empty ? 0 : n, &node);
    uint32_t at[4] = {0, 0, 0, 0};
    uint32_t next = me + 1;
    for (int i = 0; i < 4; ++i) {
      // a slot without a child: replaced by child 0's link below (stays as it is in a scene without
      // triangles, which is never traversed)
      node.child[i] = ~0;
      if (i >= n || empty) continue;
      node.child[i] = c[i].ref;
      if (c[i].inner) {
        at[i] = next;
        node.child[i] = (int32_t)next;
        next += below[c[i].ref];
      }
    }
    if (!empty)
      for (int i = n; i < 4; ++i) node.child[i] = node.child[0];
    out[me] = node;
    for_inner_children(c, empty ? 0 : n, depth, [&](int i) { emit((uint32_t)c[i].ref, at[i], depth + 1); });
  }
};

}  // namespace

namespace {
bool build_bvh_impl(const float* tri_pos, uint32_t n_tris, uint32_t threads, BvhBuildResult* out, Node2* scratch, uint32_t* order, uint32_t** below_out) {
  const uint32_t cap = n_tris > 1 ? n_tris : 1;

  auto empty_child = [](float* mn, float* mx) {
    for (int k = 0; k < 3; ++k) { mn[k] = INFINITY; mx[k] = -INFINITY; }
  };
  if (n_tris == 0) {
    Node2 root;
    std::memset(&root, 0, sizeof(root));
    empty_child(root.lmin, root.lmax);
    empty_child(root.rmin, root.rmax);
    root.left = root.right = ~0;
    root.lcount = root.rcount = 1;  // not inner
    scratch[0] = root;
  } else {
    Builder b;
    b.pos = tri_pos;
    b.n = n_tris;
    b.idx = order;
    b.nodes = scratch;
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
    if (threads > 64) threads = 64;
    b.free_threads = (int)threads - 1;
    for (uint32_t i = 0; i < n_tris; ++i) order[i] = i;
    const auto t0 = std::chrono::steady_clock::now();
    b.prepare();
    const auto t1 = std::chrono::steady_clock::now();
    int32_t child, count;
    Box box;
    b.build(0, n_tris, 0, &child, &count, &box);
    if (b.failed) return false;
    if (std::getenv("RRT_BVH_TIMING"))
      std::fprintf(stderr, "bvh build: %u triangles, %u threads: bounds %.2f s, binary tree %.2f s\n", n_tris, threads,
                   std::chrono::duration<double>(t1 - t0).count(),
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    if (count > 0) {  // the whole scene is one leaf: wrap it in a root node
      Node2 root;
      std::memset(&root, 0, sizeof(root));
      for (int k = 0; k < 3; ++k) { root.lmin[k] = box.mn[k]; root.lmax[k] = box.mx[k]; }
      empty_child(root.rmin, root.rmax);
      root.left = child;
      root.lcount = count;
      root.right = ~0;
      root.rcount = 1;
      scratch[0] = root;
    }
  }
  // the root of a multi-node binary tree is whichever node the top-level call allocated: index 0
  const auto t_collapse = std::chrono::steady_clock::now();
  Collapser col;
  col.in = scratch;
  col.below = (uint32_t*)std::calloc(cap, sizeof(uint32_t));
  *below_out = col.below;  // (freed by the caller, whichever way this function is left)
  if (!col.below) return false;
  if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
  if (threads > 64) threads = 64;
  col.free_threads = (int)threads - 1;
  BvhNode* nodes = nullptr;
  uint32_t n_nodes = 0;
  n_nodes = col.count(0, 1);
  if (col.failed) return false;
  if (posix_memalign((void**)&nodes, alignof(BvhNode), sizeof(BvhNode) * (size_t)n_nodes) != 0) return false;
  out->nodes = nodes;  // (the caller frees it if anything below fails)
  col.out = nodes;
  {
    // the root's two children span every triangle (an absent one has an inverted box and adds nothing)
    const Node2& root = scratch[0];
    float mn[3], mx[3];
    for (int k = 0; k < 3; ++k) {
      mn[k] = std::min(root.lmin[k], root.rmin[k]);
      mx[k] = std::max(root.lmax[k], root.rmax[k]);
    }
    col.grid = out->grid = make_bvh_grid(mn, mx);
  }
  col.emit(0, 0, 1);
  if (col.failed) return false;
  if (std::getenv("RRT_BVH_TIMING"))
    std::fprintf(stderr, "bvh build: collapse to %u 4-wide nodes %.2f s\n", n_nodes,
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t_collapse).count());
  out->n_nodes = n_nodes;
  out->max_depth = col.max_depth;
  return true;
}
}  // namespace

BvhGrid make_bvh_grid(const float mn[3], const float mx[3]) {
  BvhGrid g;
  for (int k = 0; k < 3; ++k) {
    const bool finite = std::isfinite(mn[k]) && std::isfinite(mx[k]) && mx[k] >= mn[k];
    g.base[k] = finite ? mn[k] : 0.0f;
    int e = 0;
    if (finite && mx[k] > mn[k]) {
      std::frexp(((double)mx[k] - (double)mn[k]) / 65535.0, &e);  // = m * 2^e with m in [0.5, 1): 2^e is at least as large
      // far inside the range of a float, so that neither the step times a ray's reciprocal direction
      // nor the smallest node step underflows
      e = std::min(std::max(e, -80), 100);
      // (rounding of base + 65535 * step is seen to by the node encoder, which checks every plane it writes)
    }
    g.step[k] = (float)std::ldexp(1.0, e);
    g.inv_step[k] = (float)std::ldexp(1.0, -e);
  }
  return g;
}

// Never throws: running out of memory or of threads anywhere below gives false (a thread that cannot
// be started is not an error: its share of the work runs on the thread that wanted it).
bool build_bvh(const float* tri_pos, uint32_t n_tris, uint32_t threads, BvhBuildResult* out) {
  *out = BvhBuildResult();
  const uint32_t cap = n_tris > 1 ? n_tris : 1;
  Node2* scratch = (Node2*)std::malloc(sizeof(Node2) * (size_t)cap);
  uint32_t* order = (uint32_t*)std::malloc(sizeof(uint32_t) * (size_t)cap);
  uint32_t* below = nullptr;
  bool ok = scratch && order;
  if (ok) {
    try {
      ok = build_bvh_impl(tri_pos, n_tris, threads, out, scratch, order, &below);
    } catch (...) {  // std::bad_alloc from the builder's vectors
      ok = false;
    }
  }
  std::free(below);
  std::free(scratch);
  if (!ok) {
    std::free(out->nodes);
    std::free(order);
    *out = BvhBuildResult();
    return false;
  }
  out->order = order;
  return true;
}


namespace {

// The box a traversing lane decodes for child i of a node, in double.
void decoded_box(const BvhGrid& g, const BvhNode& n, int i, double* mn, double* mx) {
  for (int k = 0; k < 3; ++k) {
    mn[k] = node_plane(g, n, k, n.qlo[k][i]);
    mx[k] = node_plane(g, n, k, n.qhi[k][i]);
  }
}

// Walks the hierarchy and checks what traversal relies on: every triangle lies inside the decoded
// box of every ancestor's child slot that leads to it, every leaf slot is reached exactly once, and
// no chain of inner nodes is longer than max_depth.
struct Validator {
  const float* pos;
  const BvhBuildResult& b;
  uint32_t n_tris;
  std::vector<uint8_t> seen;
  uint64_t bad = 0;

  // Exact bounds of the triangles behind `ref` (depth = inner nodes on the way here, this one included).
  void bounds(int32_t ref, uint32_t depth, double* mn, double* mx) {
    for (int k = 0; k < 3; ++k) { mn[k] = INFINITY; mx[k] = -INFINITY; }
    if (ref < 0) {
      const uint32_t code = (uint32_t)~ref, first = code >> 2, cnt = (code & 3u) + 1u;
      if ((uint64_t)first + cnt > n_tris) { ++bad; return; }
      for (uint32_t t = first; t < first + cnt; ++t) {
        if (seen[t]++) ++bad;
        const float* p = pos + 9 * (size_t)b.order[t];
        for (int v = 0; v < 3; ++v)
          for (int k = 0; k < 3; ++k) {
            mn[k] = std::min(mn[k], (double)p[3 * v + k]);
            mx[k] = std::max(mx[k], (double)p[3 * v + k]);
          }
      }
      return;
    }
    if ((uint32_t)ref >= b.n_nodes || depth > b.max_depth) { ++bad; return; }
    const BvhNode& n = b.nodes[ref];
    for (int i = 0; i < 4; ++i) {
      if (!child_present(n, i)) continue;
      double cmn[3], cmx[3], dmn[3], dmx[3];
      bounds(n.child[i], depth + (n.child[i] >= 0 ? 1 : 0), cmn, cmx);
      decoded_box(b.grid, n, i, dmn, dmx);
      for (int k = 0; k < 3; ++k) {
        if (!(cmn[k] >= dmn[k] && cmx[k] <= dmx[k])) ++bad;
        mn[k] = std::min(mn[k], cmn[k]);
        mx[k] = std::max(mx[k], cmx[k]);
      }
    }
  }
};

}  // namespace

uint64_t validate_bvh(const float* tri_pos, uint32_t n_tris, const BvhBuildResult& b) {
  if (n_tris == 0 || b.n_nodes == 0) return 0;
  Validator v{tri_pos, b, n_tris, std::vector<uint8_t>(n_tris, 0), 0};  // (may throw std::bad_alloc: callers catch)
  double mn[3], mx[3];
  v.bounds(0, 1, mn, mx);
  for (uint32_t t = 0; t < n_tris; ++t)
    if (v.seen[t] != 1) ++v.bad;
  return v.bad;
}

void free_bvh(BvhBuildResult* r) {
  std::free(r->nodes);
  std::free(r->order);
  *r = BvhBuildResult();
}

}  // namespace rrt
