// CommitScene on the device (CrossRT.h:109 CommitScene, :85-86 UpdateGeom_Triangles3f, :134 UpdateInstance; Embree backend: EmbreeRT.cpp:294-307):
// the single-level tree of hpt_host.hip's flat layout - ONE BVH2 over all instanced triangles, world-space boxes, object-space triangle
// records - and its 4-wide compressed form, built by kernels instead of the host's binned-SAH builder. For scenes whose topology changes
// from frame to frame a quarter of a second of host build against a 0.44 s frame (1 M triangles) is the bottleneck; this build takes a few
// milliseconds. It is a linear BVH: triangles sorted along a 63-bit Morton curve of their world-space centroids (hipCUB radix sort), the
// hierarchy of Karras (HPG 2012: every inner node finds its key range and split from the sorted codes, no dependencies between nodes), boxes
// fitted bottom-up by the second thread to arrive at each node, subtrees of <= BVH_LEAF_MAX triangles folded into leaves, then the same
// collapse to 4-wide nodes the host does (a node adopts its grandchildren, largest surface first) level by level, quantised by the shared
// quantizeNode4. Tree quality is below the SAH build's (hpt_set_option("device_build", ..) / CommitScene's BUILD_LOW | BUILD_MEDIUM choose it,
// BUILD_HIGH keeps the host's SAH tree). Hits do not depend on the tree: the closest hit is min t with ties broken by (instId, primId) and
// triangles are tested in object space by the shared triangleTest, so frames, generators and ray queries are bit-identical to the host-built
// tree's (tests/test_gpu_parity.py::test_device_built_tree_*).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstring>
#include <vector>
#include "hpt_lbvh.h"

namespace hpt {

namespace {

struct Scratch
{
  // per instanced triangle (unsorted): world box, Morton key, index; sorted copies
  float* triBox = nullptr;                // [6][n]: lo.xyz, hi.xyz (unpadded), SoA
  unsigned long long *keysIn = nullptr, *keysOut = nullptr;
  uint *valsIn = nullptr, *valsOut = nullptr;
  void* sortTemp = nullptr; size_t sortTempBytes = 0;
  // hierarchy: inner node i in [0, n - 1)
  uint *childL = nullptr, *childR = nullptr, *parent = nullptr, *leafParent = nullptr, *first = nullptr, *last = nullptr, *flags = nullptr, *depth = nullptr;
  float* nodeBox = nullptr;               // [6][n - 1]
  uint2 *frontA = nullptr, *frontB = nullptr;
  uint* counters = nullptr;               // see LbvhCounters
  // inputs mirrored on the device
  float* geomPos = nullptr; uint* geomIdx = nullptr; size_t geomPosCap = 0, geomIdxCap = 0;
  float* instMat = nullptr; uint4* instInfo = nullptr; size_t instCap = 0;
  size_t cap = 0;
  void freeAll()
  {
    void* ps[] = { triBox, keysIn, keysOut, valsIn, valsOut, sortTemp, childL, childR, parent, leafParent, first, last, flags, depth, nodeBox, frontA, frontB, counters, geomPos, geomIdx, instMat, instInfo };
    for (void* p : ps) if (p) (void)hipFree(p);
    *this = Scratch();
  }
};

// device-side result words
struct LbvhCounters { uint sceneLo[3], sceneHi[3]; uint count4; uint frontCount[2]; uint depth4; float sahSum; uint pad; };

HPT_HD uint floatKey(float f) { const uint b = hptFloatToBits(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }       // monotonic float -> uint
HPT_HD float keyFloat(uint k) { return hptBitsToFloat((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

__device__ inline void padBox(float lo[3], float hi[3])      // Aabb::pad (bvh_build.h): the same formula as the host build's
{
  const float ex = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
  float mag = 0.0f;
  for (int a = 0; a < 3; a++) mag = fmaxf(mag, fmaxf(fabsf(lo[a]), fabsf(hi[a])));
  const float p = 1e-5f * fmaxf(ex, mag) + 1e-30f;
  for (int a = 0; a < 3; a++) { lo[a] -= p; hi[a] += p; }
}

// ---- 1. instanced triangles: world box + scene bounds ---------------------------------------------------------------------------------------
// instInfo[i] = {first instanced triangle of instance i, its triangle count, offset of its mesh's positions (floats), offset of its mesh's indices}
__device__ inline uint findInstance(const uint4* info, uint ni, uint k)
{
  uint lo = 0, hi = ni;                       // last i with info[i].x <= k
  while (hi - lo > 1u) { const uint mid = (lo + hi) >> 1; if (info[mid].x <= k) lo = mid; else hi = mid; }
  return lo;
}
__global__ void __launch_bounds__(256) lbvhTriBoxKernel(const float* __restrict__ geomPos, const uint* __restrict__ geomIdx, const float* __restrict__ instMat, const uint4* __restrict__ instInfo,
                                                        uint ni, uint n, float* __restrict__ triBox, LbvhCounters* C)
{
  __shared__ uint sLo[3], sHi[3];
  if (threadIdx.x < 3) { sLo[threadIdx.x] = 0xFFFFFFFFu; sHi[threadIdx.x] = 0u; }
  __syncthreads();
  const uint k = blockIdx.x * 256u + threadIdx.x;
  if (k < n) {
    const uint i = findInstance(instInfo, ni, k);
    const uint4 inf = instInfo[i];
    const uint t = k - inf.x;
    const float* m = instMat + 12 * (size_t)i;                           // object -> world rows (3 x 4)
    float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    for (int v = 0; v < 3; v++) {
      const float* p = geomPos + inf.z + 3 * (size_t)geomIdx[inf.w + 3 * (size_t)t + v];
      // (the host build's expression order: m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12])
      const float q[3] = { m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3], m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7], m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11] };
      for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], q[a]); hi[a] = fmaxf(hi[a], q[a]); }
    }
    for (int a = 0; a < 3; a++) { triBox[(size_t)a * n + k] = lo[a]; triBox[(size_t)(3 + a) * n + k] = hi[a]; atomicMin(&sLo[a], floatKey(lo[a])); atomicMax(&sHi[a], floatKey(hi[a])); }
  }
  __syncthreads();
  if (threadIdx.x < 3) { atomicMin(&C->sceneLo[threadIdx.x], sLo[threadIdx.x]); atomicMax(&C->sceneHi[threadIdx.x], sHi[threadIdx.x]); }
}

// ---- 2. Morton keys of the centroids -----------------------------------------------------------------------------------------------------------
__device__ inline unsigned long long expand21(unsigned long long v)   // bit k of v -> bit 3k
{
  v &= 0x1FFFFFull;
  v = (v | (v << 32)) & 0x1F00000000FFFFull;
  v = (v | (v << 16)) & 0x1F0000FF0000FFull;
  v = (v | (v << 8))  & 0x100F00F00F00F00Full;
  v = (v | (v << 4))  & 0x10C30C30C30C30C3ull;
  v = (v | (v << 2))  & 0x1249249249249249ull;
  return v;
}
__global__ void __launch_bounds__(256) lbvhMortonKernel(const float* __restrict__ triBox, uint n, const LbvhCounters* C, unsigned long long* keys, uint* vals)
{
  const uint k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n) return;
  unsigned long long key = 0ull;
  // one scale for the three axes (the scene's longest extent): the Morton cells are cubes whatever the scene's proportions, so a split never
  // halves an axis that is already the shortest of its cell (a 10 x 4 x 10 room normalised per axis splits y as often as x and z)
  float ext = 0.0f;
  for (int a = 0; a < 3; a++) ext = fmaxf(ext, keyFloat(C->sceneHi[a]) - keyFloat(C->sceneLo[a]));
  for (int a = 0; a < 3; a++) {
    const float lo = keyFloat(C->sceneLo[a]);
    const float c = 0.5f * (triBox[(size_t)a * n + k] + triBox[(size_t)(3 + a) * n + k]);
    float u = ext > 0.0f ? (c - lo) / ext : 0.0f;
    u = fminf(fmaxf(u, 0.0f), 1.0f);
    const unsigned long long q = (unsigned long long)fminf(u * 2097152.0f, 2097151.0f);
    key |= expand21(q) << (2 - a);
  }
  keys[k] = key; vals[k] = k;
}

// ---- 3. hierarchy (Karras 2012) ---------------------------------------------------------------------------------------------------------------------
__device__ inline int delta(const unsigned long long* keys, int n, int i, int j)
{
  if (j < 0 || j >= n) return -1;
  const unsigned long long a = keys[i], b = keys[j];
  if (a != b) return __clzll((long long)(a ^ b));
  return 64 + __clz(i ^ j);                                             // equal keys: the index decides, so every pair differs
}
__global__ void __launch_bounds__(256) lbvhHierarchyKernel(const unsigned long long* __restrict__ keys, int n, uint* childL, uint* childR, uint* parent, uint* leafParent, uint* first, uint* last)
{
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  if (i >= n - 1) return;
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1) if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) >> 1; ; t = (t + 1) >> 1) { if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t; if (t == 1) break; }
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const bool leafL = lo == gamma, leafR = hi == gamma + 1;
  childL[i] = leafL ? (0x80000000u | (uint)gamma) : (uint)gamma;
  childR[i] = leafR ? (0x80000000u | (uint)(gamma + 1)) : (uint)(gamma + 1);
  if (leafL) leafParent[gamma] = (uint)i; else parent[gamma] = (uint)i;
  if (leafR) leafParent[gamma + 1] = (uint)i; else parent[gamma + 1] = (uint)i;
  first[i] = (uint)lo; last[i] = (uint)hi;
  if (i == 0) parent[0] = 0xFFFFFFFFu;
}

// ---- 4. boxes bottom-up: the second thread to arrive at a node fits it ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lbvhFitKernel(const float* __restrict__ triBox, const uint* __restrict__ vals, uint n, const uint* __restrict__ childL, const uint* __restrict__ childR,
                                                     const uint* __restrict__ parent, const uint* __restrict__ leafParent, uint* flags, float* nodeBox, uint* depth)
{
  const uint k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n || n < 2u) return;
  const size_t m = n - 1;
  uint node = leafParent[k];
  while (node != 0xFFFFFFFFu) {
    __threadfence();
    if (atomicAdd(&flags[node], 1u) == 0u) return;                       // the first arrival leaves; its sibling's subtree is not ready yet
    __threadfence();
    float lo[3], hi[3]; uint dep = 0;
    const uint cs[2] = { childL[node], childR[node] };
    for (int a = 0; a < 3; a++) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; }
    for (int c = 0; c < 2; c++) {
      const uint ch = cs[c];
      if (ch & 0x80000000u) {
        const uint t = vals[ch & 0x7FFFFFFFu];
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], triBox[(size_t)a * n + t]); hi[a] = fmaxf(hi[a], triBox[(size_t)(3 + a) * n + t]); }
      } else {
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], __hip_atomic_load(&nodeBox[(size_t)a * m + ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); hi[a] = fmaxf(hi[a], __hip_atomic_load(&nodeBox[(size_t)(3 + a) * m + ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
        dep = max(dep, __hip_atomic_load(&depth[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
    }
    for (int a = 0; a < 3; a++) { __hip_atomic_store(&nodeBox[(size_t)a * m + node], lo[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&nodeBox[(size_t)(3 + a) * m + node], hi[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __hip_atomic_store(&depth[node], dep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    node = parent[node];
  }
}

// ---- 5. BvhNode records (both child boxes in the parent, hpt_types.h), leaves of <= BVH_LEAF_MAX triangles; triangle records in leaf order ----
__device__ inline uint childRef(uint ch, const uint* first, const uint* last, uint& leafFirst, uint& leafCount)
{
  if (ch & 0x80000000u) { leafFirst = ch & 0x7FFFFFFFu; leafCount = 1u; return REF_LEAF | (1u << 28) | leafFirst; }
  const uint cnt = last[ch] - first[ch] + 1u;
  if (cnt <= (uint)BVH_LEAF_MAX) { leafFirst = first[ch]; leafCount = cnt; return REF_LEAF | (cnt << 28) | leafFirst; }
  leafCount = 0u; return ch;
}
__global__ void __launch_bounds__(256) lbvhEmitKernel(uint n, const uint* __restrict__ childL, const uint* __restrict__ childR, const uint* __restrict__ first, const uint* __restrict__ last,
                                                      const float* __restrict__ triBox, const uint* __restrict__ vals, const float* __restrict__ nodeBox, BvhNode* nodes, LbvhCounters* C)
{
  const uint i = blockIdx.x * 256u + threadIdx.x;
  float sah = 0.0f;
  if (i + 1u < n && last[i] - first[i] + 1u > (uint)BVH_LEAF_MAX) {
    const size_t m = n - 1;
    BvhNode nd;
    const uint cs[2] = { childL[i], childR[i] };
    for (int c = 0; c < 2; c++) {
      uint lf, lc; const uint ref = childRef(cs[c], first, last, lf, lc);
      float lo[3], hi[3];
      if (cs[c] & 0x80000000u) { const uint t = vals[cs[c] & 0x7FFFFFFFu]; for (int a = 0; a < 3; a++) { lo[a] = triBox[(size_t)a * n + t]; hi[a] = triBox[(size_t)(3 + a) * n + t]; } }
      else for (int a = 0; a < 3; a++) { lo[a] = nodeBox[(size_t)a * m + cs[c]]; hi[a] = nodeBox[(size_t)(3 + a) * m + cs[c]]; }
      padBox(lo, hi);
      for (int a = 0; a < 3; a++) { nd.q[6 * c + 2 * a] = lo[a]; nd.q[6 * c + 2 * a + 1] = hi[a]; }
      if (c == 0) nd.ref0 = ref; else nd.ref1 = ref;
      if ((ref & REF_LEAF) == 0u) { const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; sah += dx * dy + dy * dz + dz * dx; }
    }
    nd.pad0 = nd.pad1 = 0u;
    nodes[i] = nd;
  }
  // sum of the inner children's half-areas (sah_node_visits' numerator): wave-reduced, one atomic per wave
  for (int o = 32; o > 0; o >>= 1) sah += __shfl_down(sah, o);
  if ((threadIdx.x & 63u) == 0u && sah != 0.0f) atomicAdd(&C->sahSum, sah);
}
__global__ void __launch_bounds__(256) lbvhGatherTrisKernel(const float* __restrict__ geomPos, const uint* __restrict__ geomIdx, const uint4* __restrict__ instInfo, uint ni, uint n,
                                                            const uint* __restrict__ vals, BvhTri* tris)
{
  const uint k = blockIdx.x * 256u + threadIdx.x;
  if (k >= n) return;
  const uint src = vals[k];
  const uint i = findInstance(instInfo, ni, src);
  const uint4 inf = instInfo[i];
  const uint p = src - inf.x;
  const uint* ix = geomIdx + inf.w + 3 * (size_t)p;
  const float* A = geomPos + inf.z + 3 * (size_t)ix[0]; const float* B = geomPos + inf.z + 3 * (size_t)ix[1]; const float* Cc = geomPos + inf.z + 3 * (size_t)ix[2];
  BvhTri t;
  for (int a = 0; a < 3; a++) { t.v0[a] = A[a]; t.e1[a] = B[a] - A[a]; t.e2[a] = Cc[a] - A[a]; }   // the same object-space record as the host build's
  t.primId = p; t.instId = i; t.pad1 = 0u;
  tris[k] = t;
}

// ---- 6. collapse to 4-wide compressed nodes, one level per launch (collapseToWide of bvh_build.h as a frontier kernel) -------------------------------
__global__ void __launch_bounds__(256) lbvhCollapseKernel(const BvhNode* __restrict__ nodes, const uint2* __restrict__ frontIn, uint2* frontOut, uint level, BvhNode4* nodes4, uint cap4, LbvhCounters* C)
{
  const uint e = blockIdx.x * 256u + threadIdx.x;
  const uint cnt = C->frontCount[level & 1u];
  if (e == 0u && cnt != 0u) C->depth4 = level + 1u;
  if (e >= cnt) return;
  const uint2 it = frontIn[e];                                           // {BVH2 node, index of the wide node to fill}
  uint kids[4]; int nk = 2;                                              // each kid = (BVH2 node << 1 | side)
  kids[0] = it.x << 1; kids[1] = (it.x << 1) | 1u;
  auto refOf = [&](uint k) { const BvhNode& nd = nodes[k >> 1]; return (k & 1u) ? nd.ref1 : nd.ref0; };
  auto area = [&](uint k) { const float* q = nodes[k >> 1].q + 6 * (k & 1u); const float dx = q[1] - q[0], dy = q[3] - q[2], dz = q[5] - q[4]; return dx * dy + dy * dz + dz * dx; };
  while (nk < 4) {
    int best = -1; float bestA = -1.0f;
    for (int k = 0; k < nk; k++) { const uint r = refOf(kids[k]); if (r != REF_NONE && !(r & REF_LEAF)) { const float a = area(kids[k]); if (a > bestA) { bestA = a; best = k; } } }
    if (best < 0) break;
    const uint inner = refOf(kids[best]);
    kids[best] = inner << 1; kids[nk++] = (inner << 1) | 1u;
  }
  float lo[4][3], hi[4][3]; uint valid = 0u, refs[4] = { REF_NONE, REF_NONE, REF_NONE, REF_NONE };
  uint nInner = 0u;
  for (int k = 0; k < nk; k++) {
    const uint r = refOf(kids[k]);
    if (r == REF_NONE) continue;
    const float* q = nodes[kids[k] >> 1].q + 6 * (kids[k] & 1u);
    for (int a = 0; a < 3; a++) { lo[k][a] = q[2 * a]; hi[k][a] = q[2 * a + 1]; }
    valid |= 1u << k;
    refs[k] = r;
    if (!(r & REF_LEAF)) nInner++;
  }
  if (nInner) {
    const uint base4 = atomicAdd(&C->count4, nInner);
    const uint baseF = atomicAdd(&C->frontCount[(level + 1u) & 1u], nInner);
    uint j = 0;
    for (int k = 0; k < nk; k++) if (refs[k] != REF_NONE && !(refs[k] & REF_LEAF)) {
      if (base4 + j < cap4) frontOut[baseF + j] = make_uint2(refs[k], base4 + j);
      refs[k] = base4 + j; j++;
    }
  }
  BvhNode4 nd;
  nd.pad[0] = nd.pad[1] = 0u;
  quantizeNode4(lo, hi, valid, nd);
  for (int k = 0; k < 4; k++) nd.ref[k] = refs[k];
  if (it.y < cap4) nodes4[it.y] = nd;
}
__global__ void lbvhResetFrontKernel(LbvhCounters* C, uint level) { C->frontCount[level & 1u] = 0u; }

template <class T> bool ensure(T*& p, size_t count) { if (p) (void)hipFree(p); p = nullptr; return hipMalloc((void**)&p, count * sizeof(T)) == hipSuccess; }

} // namespace

void* lbvhCreate() { return new Scratch(); }
void lbvhDestroy(void* s) { if (s) { ((Scratch*)s)->freeAll(); delete (Scratch*)s; } }

#define LCHK(call) do { hipError_t _e = (call); if (_e != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(_e); return false; } } while (0)

bool lbvhUploadGeometry(void* scratch, const std::vector<const float*>& pos, const std::vector<size_t>& posFloats, const std::vector<const uint*>& idx, const std::vector<size_t>& idxCount,
                        std::vector<size_t>& posOffset, std::vector<size_t>& idxOffset, std::string& err)
{
  Scratch& S = *(Scratch*)scratch;
  size_t np = 0, nx = 0;
  posOffset.resize(pos.size()); idxOffset.resize(idx.size());
  for (size_t g = 0; g < pos.size(); g++) { posOffset[g] = np; idxOffset[g] = nx; np += posFloats[g]; nx += idxCount[g]; }
  if (np > S.geomPosCap) { if (!ensure(S.geomPos, np + 1)) { err = "hipMalloc (mesh positions)"; return false; } S.geomPosCap = np; }
  if (nx > S.geomIdxCap) { if (!ensure(S.geomIdx, nx + 1)) { err = "hipMalloc (mesh indices)"; return false; } S.geomIdxCap = nx; }
  for (size_t g = 0; g < pos.size(); g++) {
    if (posFloats[g]) LCHK(hipMemcpyAsync(S.geomPos + posOffset[g], pos[g], posFloats[g] * sizeof(float), hipMemcpyHostToDevice, nullptr));
    if (idxCount[g]) LCHK(hipMemcpyAsync(S.geomIdx + idxOffset[g], idx[g], idxCount[g] * sizeof(uint), hipMemcpyHostToDevice, nullptr));
  }
  return true;
}

bool lbvhBuild(void* scratch, const LbvhInstance* insts, uint ni, uint n, BvhNode* nodes, BvhTri* tris, BvhNode4* nodes4, uint cap4, bool wantWide, LbvhResult& out, std::string& err)
{
  Scratch& S = *(Scratch*)scratch;
  out = LbvhResult();
  if (n == 0u) { out.rootRef = REF_NONE; return true; }
  if (n > S.cap) {
    const size_t c = n;
    bool ok = ensure(S.triBox, 6 * c) && ensure(S.keysIn, c) && ensure(S.keysOut, c) && ensure(S.valsIn, c) && ensure(S.valsOut, c) && ensure(S.childL, c) && ensure(S.childR, c) &&
              ensure(S.parent, c) && ensure(S.leafParent, c) && ensure(S.first, c) && ensure(S.last, c) && ensure(S.flags, c) && ensure(S.depth, c) && ensure(S.nodeBox, 6 * c) &&
              ensure(S.frontA, c) && ensure(S.frontB, c);
    if (!ok) { err = "hipMalloc (build scratch)"; S.cap = 0; return false; }
    S.cap = c;
    size_t bytes = 0;
    LCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, S.keysIn, S.keysOut, S.valsIn, S.valsOut, (int)n, 0, 63, nullptr));
    if (S.sortTemp) (void)hipFree(S.sortTemp);
    S.sortTemp = nullptr;
    LCHK(hipMalloc(&S.sortTemp, bytes + 16)); S.sortTempBytes = bytes;
  }
  if (!S.counters) LCHK(hipMalloc((void**)&S.counters, sizeof(LbvhCounters)));
  if (ni > S.instCap) { if (!ensure(S.instMat, 12 * (size_t)ni) || !ensure(S.instInfo, (size_t)ni)) { err = "hipMalloc (instances)"; return false; } S.instCap = ni; }
  {
    std::vector<float> mats(12 * (size_t)ni); std::vector<uint4> info(ni);
    uint acc = 0;
    for (uint i = 0; i < ni; i++) {
      for (int k = 0; k < 12; k++) mats[12 * (size_t)i + k] = insts[i].objectToWorld[k];
      info[i] = make_uint4(acc, insts[i].triCount, (uint)insts[i].posOffset, (uint)insts[i].idxOffset);
      acc += insts[i].triCount;
    }
    LCHK(hipMemcpyAsync(S.instMat, mats.data(), mats.size() * sizeof(float), hipMemcpyHostToDevice, nullptr));
    LCHK(hipMemcpyAsync(S.instInfo, info.data(), info.size() * sizeof(uint4), hipMemcpyHostToDevice, nullptr));
    LCHK(hipStreamSynchronize(nullptr));                               // (the staging vectors leave scope)
  }
  LbvhCounters init; std::memset(&init, 0, sizeof(init));
  for (int a = 0; a < 3; a++) { init.sceneLo[a] = 0xFFFFFFFFu; init.sceneHi[a] = 0u; }
  init.count4 = 1u; init.frontCount[0] = 1u;
  LCHK(hipMemcpyAsync(S.counters, &init, sizeof(init), hipMemcpyHostToDevice, nullptr));
  LbvhCounters* C = (LbvhCounters*)S.counters;
  const dim3 blk(256), grdN((n + 255u) / 256u);
  lbvhTriBoxKernel<<<grdN, blk>>>(S.geomPos, S.geomIdx, S.instMat, S.instInfo, ni, n, S.triBox, C);
  lbvhMortonKernel<<<grdN, blk>>>(S.triBox, n, C, S.keysIn, S.valsIn);
  size_t bytes = S.sortTempBytes;
  LCHK(hipcub::DeviceRadixSort::SortPairs(S.sortTemp, bytes, S.keysIn, S.keysOut, S.valsIn, S.valsOut, (int)n, 0, 63, nullptr));
  lbvhGatherTrisKernel<<<grdN, blk>>>(S.geomPos, S.geomIdx, S.instInfo, ni, n, S.valsOut, tris);
  if (n <= (uint)BVH_LEAF_MAX) {                                      // the whole scene is one leaf
    LCHK(hipStreamSynchronize(nullptr));
    out.rootRef = REF_LEAF | (n << 28); out.numNodes = 0; out.depth = 0; out.sahVisits = 1.0f;
    return true;
  }
  LCHK(hipMemsetAsync(S.flags, 0, (size_t)n * sizeof(uint), nullptr));
  LCHK(hipMemsetAsync(S.depth, 0, (size_t)n * sizeof(uint), nullptr));
  lbvhHierarchyKernel<<<grdN, blk>>>(S.keysOut, (int)n, S.childL, S.childR, S.parent, S.leafParent, S.first, S.last);
  lbvhFitKernel<<<grdN, blk>>>(S.triBox, S.valsOut, n, S.childL, S.childR, S.parent, S.leafParent, S.flags, S.nodeBox, S.depth);
  lbvhEmitKernel<<<grdN, blk>>>(n, S.childL, S.childR, S.first, S.last, S.triBox, S.valsOut, S.nodeBox, nodes, C);
  const uint MAX_LEVELS = 64u;
  if (wantWide && nodes4 && cap4) {
    const uint2 root = make_uint2(0u, 0u);
    LCHK(hipMemcpyAsync(S.frontA, &root, sizeof(root), hipMemcpyHostToDevice, nullptr));
    for (uint level = 0; level < MAX_LEVELS; level++) {
      lbvhResetFrontKernel<<<1, 1>>>(C, level + 1u);
      lbvhCollapseKernel<<<grdN, blk>>>(nodes, (level & 1u) ? S.frontB : S.frontA, (level & 1u) ? S.frontA : S.frontB, level, nodes4, cap4, C);
    }
  }
  LCHK(hipGetLastError());
  LbvhCounters res; uint rootDepth = 0; float rootBox[6];
  LCHK(hipMemcpy(&res, S.counters, sizeof(res), hipMemcpyDeviceToHost));
  LCHK(hipMemcpy(&rootDepth, S.depth, sizeof(uint), hipMemcpyDeviceToHost));
  for (int a = 0; a < 6; a++) LCHK(hipMemcpy(&rootBox[a], S.nodeBox + (size_t)a * (n - 1), sizeof(float), hipMemcpyDeviceToHost));
  out.rootRef = 0u; out.numNodes = n - 1u; out.depth = rootDepth;
  float lo[3] = { rootBox[0], rootBox[1], rootBox[2] }, hi[3] = { rootBox[3], rootBox[4], rootBox[5] };
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  const float a0 = dx * dy + dy * dz + dz * dx;
  out.sahVisits = a0 > 0.0f ? 1.0f + res.sahSum / a0 : 1.0f;
  if (wantWide && nodes4 && cap4) {
    if (res.frontCount[MAX_LEVELS & 1u] != 0u || res.count4 > cap4) { out.nodes4Count = 0; out.depth4 = 0; }     // (deeper than the level loop / more nodes than room: no wide tree)
    else { out.nodes4Count = res.count4; out.depth4 = res.depth4; }
  }
  return true;
}

} // namespace hpt
