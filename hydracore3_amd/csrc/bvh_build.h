// Host-side BVH2 construction for the gfx950 traversal kernel (hpt_device.h: traceRay).
//
// Replaces what the reference delegates to Embree in EmbreeRT::AddGeom_Triangles3f / AddInstance / CommitScene
// (external/CrossRT/EmbreeRT.cpp:138-190, 242-298): one BLAS per mesh over object-space triangles, one TLAS over the
// instances' world boxes. Binned SAH (16 bins, 3 axes); leaves hold <= 4 triangles (BLAS) or exactly one instance (TLAS).
// The tree depth is capped so that the traversal stack (LDS, one 32-bit slot per level and lane) can never overflow:
// when the SAH split would leave a side too large for the remaining levels, the builder falls back to a median split.
//
// Node layout: 64 bytes holding BOTH child boxes as (lo, hi) pairs per axis (see hpt_types.h); boxes are padded by a relative 1e-5 so that the
// slab test can only err on the conservative side with respect to the exact triangle test.
#pragma once
#include "hpt_types.h"
#include <vector>
#include <algorithm>
#include <cmath>
#include <cfloat>
#include <cstring>
#include <thread>
#include <atomic>

namespace hpt {

struct Aabb
{
  float lo[3], hi[3];
  void reset() { lo[0] = lo[1] = lo[2] = FLT_MAX; hi[0] = hi[1] = hi[2] = -FLT_MAX; }
  void grow(const float* p) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
  void merge(const Aabb& b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
  float halfArea() const
  {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return (dx < 0.0f) ? 0.0f : dx * dy + dy * dz + dz * dx;
  }
  void pad()
  {
    const float ex = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    float mag = 0.0f;
    for (int a = 0; a < 3; a++) mag = std::max(mag, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
    const float p = 1e-5f * std::max(ex, mag) + 1e-30f;
    for (int a = 0; a < 3; a++) { lo[a] -= p; hi[a] += p; }
  }
};

struct Bvh2
{
  std::vector<BvhNode> nodes;     // inner nodes, child references local to this tree (see patch())
  std::vector<uint>    order;     // primitive ids in leaf order
  uint                 rootRef = REF_NONE;
  uint                 depth = 0; // number of inner levels above the deepest leaf
  Aabb                 bounds;
};

class Bvh2Builder
{
public:
  // leafMax: max primitives per leaf (<= 4 for triangles, 1 for instances); maxDepth: cap on inner levels.
  // instanceLeaves: encode leaves as instance references (count field 0).
  // threads > 1: the top of the tree is built by the caller's thread down to subtrees of at most n / (8 * threads) primitives; those subtrees
  // - disjoint ranges of the primitive order - are built by worker threads into node arrays of their own and appended afterwards. The
  // splits, and so the tree, are the ones the sequential build makes; only the numbering of the nodes differs.
  static Bvh2 build(const std::vector<Aabb>& boxes, int leafMax, int maxDepth, bool instanceLeaves, int threads = 1)
  {
    Bvh2 out;
    out.bounds.reset();
    const size_t n = boxes.size();
    if (n == 0) return out;
    std::vector<float> cent(3 * n);
    out.order.resize(n);
    for (size_t i = 0; i < n; i++) { out.order[i] = (uint)i; out.bounds.merge(boxes[i]); }
    for (size_t i = 0; i < n; i++) for (int a = 0; a < 3; a++) cent[3 * i + a] = 0.5f * (boxes[i].lo[a] + boxes[i].hi[a]);
    out.nodes.reserve(n);
    Bvh2Builder b(boxes, cent, leafMax, maxDepth, instanceLeaves, out.order, out.nodes, out.depth);
    if (threads > 1 && n >= 65536) b.jobLimit = std::max<size_t>(n / (8 * (size_t)threads), 4096);
    out.rootRef = b.buildRange(0, (uint)n, 0);
    if (!b.jobs.empty()) {
      struct Sub { std::vector<BvhNode> nodes; uint depth = 0, ref = REF_NONE; };
      std::vector<Sub> subs(b.jobs.size());
      std::atomic<size_t> next(0);
      auto work = [&]() {
        for (size_t j = next.fetch_add(1); j < b.jobs.size(); j = next.fetch_add(1)) {
          Bvh2Builder sb(boxes, cent, leafMax, maxDepth, instanceLeaves, out.order, subs[j].nodes, subs[j].depth);
          subs[j].nodes.reserve(b.jobs[j].count);
          subs[j].ref = sb.buildRange(b.jobs[j].first, b.jobs[j].count, b.jobs[j].depth);
        }
      };
      std::vector<std::thread> pool;
      try { for (int t = 1; t < threads; t++) pool.emplace_back(work); } catch (...) {}   // jobs are drawn dynamically: the caller's thread does what no worker takes
      work();
      for (std::thread& t : pool) t.join();
      for (size_t j = 0; j < subs.size(); j++) {                             // append, shifting the subtree's inner references
        const uint base = (uint)out.nodes.size();
        for (BvhNode nd : subs[j].nodes) {
          if (nd.ref0 != REF_NONE && !(nd.ref0 & REF_LEAF)) nd.ref0 += base;
          if (nd.ref1 != REF_NONE && !(nd.ref1 & REF_LEAF)) nd.ref1 += base;
          out.nodes.push_back(nd);
        }
        const uint ref = (subs[j].ref != REF_NONE && !(subs[j].ref & REF_LEAF)) ? subs[j].ref + base : subs[j].ref;
        if (b.jobs[j].parent == 0xFFFFFFFFu) out.rootRef = ref;
        else if (b.jobs[j].side == 0) out.nodes[b.jobs[j].parent].ref0 = ref; else out.nodes[b.jobs[j].parent].ref1 = ref;
        out.depth = std::max(out.depth, subs[j].depth);
      }
    }
    return out;
  }

private:
  Bvh2Builder(const std::vector<Aabb>& bx, const std::vector<float>& ce, int lm, int md, bool il, std::vector<uint>& ord, std::vector<BvhNode>& nd, uint& dp)
    : boxes(bx), cent(ce), leafMax(lm), maxDepth(md), instLeaves(il), order(ord), nodes(nd), depthOut(dp) {}

  const std::vector<Aabb>& boxes;
  const std::vector<float>& cent;
  int leafMax, maxDepth; bool instLeaves;
  std::vector<uint>& order;          // shared: every builder works on its own range of it
  std::vector<BvhNode>& nodes;       // this builder's node array
  uint& depthOut;
  // deferred subtrees (parallel build): ranges of at most jobLimit primitives are not built by the top-level pass
  struct Job { uint first, count; int depth; uint parent; int side; };
  std::vector<Job> jobs;
  size_t jobLimit = 0;
  static const uint REF_JOB = 0x7FF00000u;                                   // placeholder reference: REF_JOB + job index (node ids stay far below)

  uint leafRef(uint first, uint count) const
  {
    if (instLeaves) return REF_LEAF | (order[first] & 0x0FFFFFFFu);
    return REF_LEAF | (count << 28) | (first & 0x0FFFFFFFu);
  }

  // largest primitive count a subtree rooted `levels` inner levels above the depth cap may hold
  size_t capacity(int levelsLeft) const { return levelsLeft >= 40 ? ~size_t(0) : (size_t)leafMax << levelsLeft; }

  uint buildRange(uint first, uint count, int depth)
  {
    if ((int)count <= leafMax) { depthOut = std::max(depthOut, (uint)depth); return leafRef(first, count); }
    if (jobLimit != 0 && count <= jobLimit && jobs.size() < 0xFFFFFu) {       // a subtree for the workers
      jobs.push_back({ first, count, depth, 0xFFFFFFFFu, 0 });
      return REF_JOB + (uint)(jobs.size() - 1);
    }

    // centroid bounds
    float clo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, chi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (uint i = first; i < first + count; i++) {
      const float* c = &cent[3 * order[i]];
      for (int a = 0; a < 3; a++) { clo[a] = std::min(clo[a], c[a]); chi[a] = std::max(chi[a], c[a]); }
    }
    // binned SAH over the three axes
#ifndef HPT_SAH_BINS
#define HPT_SAH_BINS 16
#endif
    const int NB = HPT_SAH_BINS;
    float bestCost = FLT_MAX; int bestAxis = -1, bestBin = -1;
    for (int a = 0; a < 3; a++) {
      const float ext = chi[a] - clo[a];
      if (!(ext > 0.0f)) continue;
      Aabb bb[NB]; uint bc[NB];
      for (int k = 0; k < NB; k++) { bb[k].reset(); bc[k] = 0; }
      const float scale = float(NB) / ext;
      for (uint i = first; i < first + count; i++) {
        const uint p = order[i];
        int k = (int)((cent[3 * p + a] - clo[a]) * scale);
        k = std::min(std::max(k, 0), NB - 1);
        bb[k].merge(boxes[p]); bc[k]++;
      }
      float rightArea[NB]; uint rightCnt[NB];
      Aabb acc; acc.reset(); uint cnt = 0;
      for (int k = NB - 1; k > 0; k--) { acc.merge(bb[k]); cnt += bc[k]; rightArea[k] = acc.halfArea(); rightCnt[k] = cnt; }
      acc.reset(); cnt = 0;
      for (int k = 0; k < NB - 1; k++) {
        acc.merge(bb[k]); cnt += bc[k];
        if (cnt == 0 || rightCnt[k + 1] == 0) continue;
        const float cost = acc.halfArea() * float(cnt) + rightArea[k + 1] * float(rightCnt[k + 1]);
        if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = k; }
      }
    }

    uint mid = first;
    bool useMedian = (bestAxis < 0);
    if (!useMedian) {
      const float ext = chi[bestAxis] - clo[bestAxis];
      const float scale = float(NB) / ext;
      const float lo = clo[bestAxis];
      uint* b = order.data() + first;
      uint* e = std::partition(b, b + count, [&](uint p) {
        int k = (int)((cent[3 * p + bestAxis] - lo) * scale);
        k = std::min(std::max(k, 0), NB - 1);
        return k <= bestBin;
      });
      mid = first + (uint)(e - b);
      const size_t cap = capacity(maxDepth - depth - 1);
      if (mid == first || mid == first + count || (size_t)(mid - first) > cap || (size_t)(first + count - mid) > cap) useMedian = true;
    }
    if (useMedian) {
      int axis = 0;
      float ex[3] = { chi[0] - clo[0], chi[1] - clo[1], chi[2] - clo[2] };
      if (ex[1] > ex[axis]) axis = 1;
      if (ex[2] > ex[axis]) axis = 2;
      mid = first + count / 2;
      std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                       [&](uint p, uint q) { return cent[3 * p + axis] < cent[3 * q + axis]; });
    }

    const uint id = (uint)nodes.size();
    nodes.push_back(BvhNode());
    Aabb b0, b1; b0.reset(); b1.reset();
    for (uint i = first; i < mid; i++) b0.merge(boxes[order[i]]);
    for (uint i = mid; i < first + count; i++) b1.merge(boxes[order[i]]);
    b0.pad(); b1.pad();
    const uint r0 = buildRange(first, mid - first, depth + 1);
    const uint r1 = buildRange(mid, first + count - mid, depth + 1);
    if (r0 >= REF_JOB && r0 < REF_LEAF) { jobs[r0 - REF_JOB].parent = id; jobs[r0 - REF_JOB].side = 0; }
    if (r1 >= REF_JOB && r1 < REF_LEAF) { jobs[r1 - REF_JOB].parent = id; jobs[r1 - REF_JOB].side = 1; }
    BvhNode& nd = nodes[id];
    // (lo, hi) pairs per axis: child 0 = q[0..5], child 1 = q[6..11] - the layout the packed slab test consumes (hpt_device.h: nodeSlabs)
    for (int a = 0; a < 3; a++) { nd.q[2 * a] = b0.lo[a]; nd.q[2 * a + 1] = b0.hi[a]; nd.q[6 + 2 * a] = b1.lo[a]; nd.q[6 + 2 * a + 1] = b1.hi[a]; }
    nd.ref0 = r0; nd.ref1 = r1; nd.pad0 = nd.pad1 = 0;
    return id;
  }
};

// The same tree as 4-wide compressed nodes (BvhNode4, hpt_types.h) for the heavy-scene kernels. A node adopts its grandchildren, largest surface
// first, until it has four children or only leaves are left. Child boxes are the (padded) BVH2 child boxes, quantised outwards in the node's frame
// (quantizeNode4); src[4 * node + child] = (BVH2 node << 1 | side) remembers where each box lives so that a refit can requantise. `tree` must have
// an inner root. depth = inner levels of the wide tree (a traversal leaves at most three children waiting per level).
inline void collapseToWide(const Bvh2& tree, std::vector<BvhNode4>& n4, std::vector<uint>& src4, uint& depth4)
{
  n4.clear(); src4.clear(); depth4 = 1;
  n4.reserve(tree.nodes.size() / 2 + 1); src4.reserve(2 * tree.nodes.size() + 4);
  struct Item { uint node2, idx4, depth; };
  std::vector<Item> todo; todo.push_back({ tree.rootRef, 0u, 1u });
  n4.push_back(BvhNode4()); src4.resize(4, 0xFFFFFFFFu);
  auto area = [&](uint node2, uint side) { const float* q = tree.nodes[node2].q + 6 * side; const float dx = q[1] - q[0], dy = q[3] - q[2], dz = q[5] - q[4]; return dx * dy + dy * dz + dz * dx; };
  auto refOf = [&](uint k) { const BvhNode& n = tree.nodes[k >> 1]; return (k & 1u) ? n.ref1 : n.ref0; };
  while (!todo.empty()) {
    const Item it = todo.back(); todo.pop_back();
    depth4 = std::max(depth4, it.depth);
    uint kids[4]; int nk = 2;                                      // each kid = (BVH2 node << 1 | side)
    kids[0] = it.node2 << 1; kids[1] = (it.node2 << 1) | 1u;
    while (nk < 4) {
      int best = -1; float bestA = -1.0f;
      for (int k = 0; k < nk; k++) { const uint r = refOf(kids[k]); if (r != REF_NONE && !(r & REF_LEAF)) { const float a = area(kids[k] >> 1, kids[k] & 1u); if (a > bestA) { bestA = a; best = k; } } }
      if (best < 0) break;
      const uint inner = refOf(kids[best]);
      kids[best] = inner << 1; kids[nk++] = (inner << 1) | 1u;
    }
    float lo[4][3], hi[4][3]; uint valid = 0;
    BvhNode4 nd; std::memset(&nd, 0, sizeof(nd));
    for (int k = 0; k < 4; k++) nd.ref[k] = REF_NONE;
    for (int k = 0; k < nk; k++) {
      const uint r = refOf(kids[k]);
      if (r == REF_NONE) continue;
      const float* q = tree.nodes[kids[k] >> 1].q + 6 * (kids[k] & 1u);
      for (int a = 0; a < 3; a++) { lo[k][a] = q[2 * a]; hi[k][a] = q[2 * a + 1]; }
      valid |= 1u << k;
      src4[4 * (size_t)it.idx4 + k] = kids[k];
      if (r & REF_LEAF) nd.ref[k] = r;
      else { nd.ref[k] = (uint)n4.size(); todo.push_back({ r, (uint)n4.size(), it.depth + 1u }); n4.push_back(BvhNode4()); src4.resize(src4.size() + 4, 0xFFFFFFFFu); }
    }
    uint keep[4]; for (int k = 0; k < 4; k++) keep[k] = nd.ref[k];
    quantizeNode4(lo, hi, valid, nd);
    for (int k = 0; k < 4; k++) nd.ref[k] = keep[k];
    n4[it.idx4] = nd;
  }
}

// shift a tree's local references so that it can live at nodeBase / triBase of the shared arrays
inline uint patchRef(uint ref, uint nodeBase, uint triBase)
{
  if (ref == REF_NONE) return ref;
  if ((ref & REF_LEAF) == 0u) return ref + nodeBase;
  const uint cnt = (ref >> 28) & 7u;
  if (cnt == 0u) return ref;                                   // instance leaf: ids are global already
  return REF_LEAF | (cnt << 28) | (((ref & 0x0FFFFFFFu) + triBase) & 0x0FFFFFFFu);
}

} // namespace hpt
