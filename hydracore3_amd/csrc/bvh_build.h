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
  static Bvh2 build(const std::vector<Aabb>& boxes, int leafMax, int maxDepth, bool instanceLeaves)
  {
    Bvh2 out;
    out.bounds.reset();
    const size_t n = boxes.size();
    if (n == 0) return out;
    Bvh2Builder b(boxes, leafMax, maxDepth, instanceLeaves, out);
    out.order.resize(n);
    for (size_t i = 0; i < n; i++) { out.order[i] = (uint)i; out.bounds.merge(boxes[i]); }
    b.cent.resize(3 * n);
    for (size_t i = 0; i < n; i++) for (int a = 0; a < 3; a++) b.cent[3 * i + a] = 0.5f * (boxes[i].lo[a] + boxes[i].hi[a]);
    out.nodes.reserve(n);
    out.rootRef = b.buildRange(0, (uint)n, 0);
    return out;
  }

private:
  Bvh2Builder(const std::vector<Aabb>& bx, int lm, int md, bool il, Bvh2& o) : boxes(bx), leafMax(lm), maxDepth(md), instLeaves(il), out(o) {}

  const std::vector<Aabb>& boxes;
  std::vector<float> cent;
  int leafMax, maxDepth; bool instLeaves;
  Bvh2& out;

  uint leafRef(uint first, uint count) const
  {
    if (instLeaves) return REF_LEAF | (out.order[first] & 0x0FFFFFFFu);
    return REF_LEAF | (count << 28) | (first & 0x0FFFFFFFu);
  }

  // largest primitive count a subtree rooted `levels` inner levels above the depth cap may hold
  size_t capacity(int levelsLeft) const { return levelsLeft >= 40 ? ~size_t(0) : (size_t)leafMax << levelsLeft; }

  uint buildRange(uint first, uint count, int depth)
  {
    if ((int)count <= leafMax) { out.depth = std::max(out.depth, (uint)depth); return leafRef(first, count); }

    // centroid bounds
    float clo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, chi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (uint i = first; i < first + count; i++) {
      const float* c = &cent[3 * out.order[i]];
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
        const uint p = out.order[i];
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
      uint* b = out.order.data() + first;
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
      std::nth_element(out.order.begin() + first, out.order.begin() + mid, out.order.begin() + first + count,
                       [&](uint p, uint q) { return cent[3 * p + axis] < cent[3 * q + axis]; });
    }

    const uint id = (uint)out.nodes.size();
    out.nodes.push_back(BvhNode());
    Aabb b0, b1; b0.reset(); b1.reset();
    for (uint i = first; i < mid; i++) b0.merge(boxes[out.order[i]]);
    for (uint i = mid; i < first + count; i++) b1.merge(boxes[out.order[i]]);
    b0.pad(); b1.pad();
    const uint r0 = buildRange(first, mid - first, depth + 1);
    const uint r1 = buildRange(mid, first + count - mid, depth + 1);
    BvhNode& nd = out.nodes[id];
    // (lo, hi) pairs per axis: child 0 = q[0..5], child 1 = q[6..11] - the layout the packed slab test consumes (hpt_device.h: nodeSlabs)
    for (int a = 0; a < 3; a++) { nd.q[2 * a] = b0.lo[a]; nd.q[2 * a + 1] = b0.hi[a]; nd.q[6 + 2 * a] = b1.lo[a]; nd.q[6 + 2 * a + 1] = b1.hi[a]; }
    nd.ref0 = r0; nd.ref1 = r1; nd.pad0 = nd.pad1 = 0;
    return id;
  }
};

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
