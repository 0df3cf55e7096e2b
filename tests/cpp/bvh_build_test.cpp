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
    std::printf("round %d: %zu primitives, %zu nodes, depth %u: the builds with 2, 5 and 8 threads equal the sequential one\n", round, n, seq.nodes.size(), seq.depth);
  }
  return 0;
}
