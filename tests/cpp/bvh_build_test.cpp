// Host check of the BVH2 builder's thread pool (hydracore3_amd/csrc/bvh_build.h): the tree built with worker threads - the top of the tree by the
// caller, subtrees of disjoint primitive ranges by the workers, appended afterwards - is the tree the sequential build makes: same primitive
// order, same depth, and node for node the same child boxes and leaf references (only the numbering of the nodes differs).
// Plain g++, no GPU:  g++ -std=c++17 -O2 -pthread bvh_build_test.cpp -o bvh_build_test
#include <cstdio>
#include <cstring>
#include <random>
struct float4 { float x, y, z, w; };
#include "../../hydracore3_amd/csrc/bvh_build.h"
using namespace hpt;

static bool same(const Bvh2& a, uint ra, const Bvh2& b, uint rb, size_t& nodes)
{
  if ((ra & REF_LEAF) || ra == REF_NONE) return ra == rb;
  if ((rb & REF_LEAF) || rb == REF_NONE) return false;
  const BvhNode& x = a.nodes[ra]; const BvhNode& y = b.nodes[rb];
  nodes++;
  if (std::memcmp(x.q, y.q, sizeof(x.q)) != 0) return false;
  return same(a, x.ref0, b, y.ref0, nodes) && same(a, x.ref1, b, y.ref1, nodes);
}

// the 4-wide collapse: every BVH2 leaf is reachable exactly once, every node has two to four children, every decoded child box contains the
// BVH2 box it was made from, the recorded sources are right, the depth bound holds
static bool checkWide(const Bvh2& t)
{
  std::vector<BvhNode4> n4; std::vector<uint> src; uint depth4 = 0;
  collapseToWide(t, n4, src, depth4);
  size_t leaves2 = 0;
  for (const BvhNode& n : t.nodes) { if (n.ref0 & REF_LEAF) leaves2++; if (n.ref1 & REF_LEAF) leaves2++; }
  size_t leaves4 = 0, visited = 0; uint maxDepth = 0;
  struct It { uint node, depth; };
  std::vector<It> st; st.push_back({0u, 1u});
  std::vector<char> seen(n4.size(), 0);
  while (!st.empty()) {
    const It it = st.back(); st.pop_back();
    if (it.node >= n4.size() || seen[it.node]) { std::printf("wide tree: bad or repeated node reference\n"); return false; }
    seen[it.node] = 1; visited++; maxDepth = std::max(maxDepth, it.depth);
    const BvhNode4& nd = n4[it.node];
    const uint valid = nd.exps >> 24; int nk = 0;
    for (int c = 0; c < 4; c++) {
      if (!(valid & (1u << c))) { if (nd.ref[c] != REF_NONE) { std::printf("wide tree: reference in an empty slot\n"); return false; } continue; }
      nk++;
      const uint sidx = src[4 * (size_t)it.node + c];
      const BvhNode& b2 = t.nodes[sidx >> 1];
      const float* q = b2.q + 6 * (sidx & 1u);
      const uint r2 = (sidx & 1u) ? b2.ref1 : b2.ref0;
      for (int a = 0; a < 3; a++) {
        const float s = hptBitsToFloat(((nd.exps >> (8 * a)) & 0xFFu) << 23);
        const float dl = std::fmaf(float((nd.q[a] >> (8 * c)) & 0xFFu), s, nd.org[a]), dh = std::fmaf(float((nd.q[3 + a] >> (8 * c)) & 0xFFu), s, nd.org[a]);
        if (!(dl <= q[2 * a]) || !(dh >= q[2 * a + 1])) { std::printf("wide tree: decoded box does not contain its BVH2 box\n"); return false; }
      }
      if (nd.ref[c] & REF_LEAF) { if (nd.ref[c] != r2) { std::printf("wide tree: leaf reference differs from its source\n"); return false; } leaves4++; }
      else { if ((r2 & REF_LEAF) || r2 == REF_NONE) { std::printf("wide tree: inner child made from a leaf\n"); return false; } st.push_back({nd.ref[c], it.depth + 1u}); }
    }
    if (nk < 2) { std::printf("wide tree: a node with %d children\n", nk); return false; }
  }
  if (visited != n4.size() || leaves4 != leaves2 || maxDepth != depth4 || depth4 > t.depth) { std::printf("wide tree: %zu of %zu nodes, %zu of %zu leaves, depth %u / %u (BVH2 %u)\n", visited, n4.size(), leaves4, leaves2, maxDepth, depth4, t.depth); return false; }
  std::printf("   4-wide tree: %zu nodes for %zu, depth %u for %u, every leaf once, every box contains its source\n", n4.size(), t.nodes.size(), depth4, t.depth);
  return true;
}

int main()
{
  std::mt19937 rng(99);
  std::uniform_real_distribution<float> U(0.0f, 1.0f);
  for (int round = 0; round < 3; round++) {
    const size_t n = round == 0 ? 70000 : (round == 1 ? 200000 : 65536);
    std::vector<Aabb> boxes(n);
    for (size_t i = 0; i < n; i++) {
      const float cx = round == 2 ? float(i % 256) : 100.0f * U(rng), cy = round == 2 ? float(i / 256) : 40.0f * U(rng), cz = round == 2 ? 0.0f : 100.0f * U(rng) * U(rng);
      const float e = 0.01f + 0.5f * U(rng) * U(rng);
      Aabb b; b.lo[0] = cx; b.lo[1] = cy; b.lo[2] = cz; b.hi[0] = cx + e; b.hi[1] = cy + e; b.hi[2] = cz + (round == 2 ? 0.0f : e);
      boxes[i] = b;
    }
    const Bvh2 seq = Bvh2Builder::build(boxes, 2, 40, false, 1);
    for (int threads : { 2, 5, 8 }) {
      const Bvh2 par = Bvh2Builder::build(boxes, 2, 40, false, threads);
      size_t visited = 0;
      if (par.order != seq.order) { std::printf("round %d, %d threads: primitive order differs\n", round, threads); return 1; }
      if (par.depth != seq.depth || par.nodes.size() != seq.nodes.size()) { std::printf("round %d, %d threads: depth %u vs %u, nodes %zu vs %zu\n", round, threads, par.depth, seq.depth, par.nodes.size(), seq.nodes.size()); return 1; }
      if (!same(seq, seq.rootRef, par, par.rootRef, visited) || visited != seq.nodes.size()) { std::printf("round %d, %d threads: trees differ (%zu of %zu nodes compared)\n", round, threads, visited, seq.nodes.size()); return 1; }
    }
    if (!checkWide(seq)) return 1;
    std::printf("round %d: %zu primitives, %zu nodes, depth %u: the builds with 2, 5 and 8 threads equal the sequential one\n", round, n, seq.nodes.size(), seq.depth);
  }
  return 0;
}
