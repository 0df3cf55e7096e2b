// Small helper kernels of the C ABI (compiled into the host translation unit): PackXY, InitRandomGens, batched ray queries,
// the texture regulariser's gradient and AdamOptimizer::step. The path-tracing kernels themselves live in hpt_kernels.hip /
// hpt_wavefront.hip and are compiled as separate translation units (hpt_decl.h declares them).
#include <hip/hip_runtime.h>
#include "hpt_decl.h"

namespace hpt {

// ---- helper kernels --------------------------------------------------------------------------------------------------------------
// kernel_PackXY over the window (integrator_rt.cpp:13-31)
__global__ void packXYKernel(uint* out, int W, int H, uint ts)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) return;
  uint offset = (uint)y * (uint)W + (uint)x;
  if (ts != 1u) {
    const uint inX = (uint)x % ts, inY = (uint)y % ts;
    const uint wBlocks = (uint)W / ts;
    offset = (((uint)x / ts) + ((uint)y / ts) * wBlocks) * ts * ts + inY * ts + inX;
  }
  if (offset < (uint)W * (uint)H) out[offset] = (((uint)y << 16) & 0xFFFF0000u) | ((uint)x & 0x0000FFFFu);
}

// InitRandomGens (integrator_pt.cpp:13-21)
__global__ void initRandomGensKernel(Rng* gens, uint n, uint firstSeed)
{
  const uint i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) gens[i] = rng_init(firstSeed + i);
}

// batched RayQuery_NearestHit / RayQuery_AnyHit for the ISceneObject entry points
template <bool FLAT, bool MOTION = false, bool SWEEP = false>
__global__ void __launch_bounds__(256) rayQueryKernel(const DevScene S, const float4* posNear, const float4* dirFar, uint n, void* out, int anyHit, uint* stackOverflow, float time = 0.0f)
{
  __shared__ uint stackMem[LDS_STACK * 256];
  const uint i = blockIdx.x * 256u + threadIdx.x;
  TravStack stk; stk.lds = &stackMem[threadIdx.x]; stk.ovf = stackOverflow + i; stk.ovfStride = gridDim.x * 256u;
  if (i >= n) return;
  const float4 p = posNear[i], d = dirFar[i];
  HitRec h; TravStats st; st.nodes = st.tris = st.insts = st.waveNodeIters = st.waveTriIters = 0;
  if (anyHit) {
    const bool occ = traceAny<true, false, true, FLAT, MOTION, SWEEP>(S, v3(p.x, p.y, p.z), v3(d.x, d.y, d.z), p.w, d.w, h, stk, st, time);
    ((uint*)out)[i] = occ ? 1u : 0u;
  } else {
    const bool found = traceAny<false, false, true, FLAT, MOTION, SWEEP>(S, v3(p.x, p.y, p.z), v3(d.x, d.y, d.z), p.w, d.w, h, stk, st, time);
    // CRT_Hit (CrossRT.h:23-30) as the Embree backend fills it (EmbreeRT.cpp:343-360)
    float4* o = (float4*)out + 2 * (size_t)i;
    if (found) {
      o[0] = make_float4(h.t, __uint_as_float(h.prim), __uint_as_float(h.inst), __uint_as_float(S.insts[h.inst].geomId));
      o[1] = make_float4(h.v, h.u, 1.0f - h.v - h.u, 0.0f);
    } else {
      o[0] = make_float4(d.w, __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu));
      o[1] = make_float4(0, 0, 0, 0);
    }
  }
}

// Image2D4fRegularizer (diff_render/integrator_dr.cpp:317-367): grad += d/d data of  sum_{interior pixels} sqrt(sum_{4 neighbours} |p0 - p_k|^2_rgb).
// Hand-derived instead of Enzyme, gather form (no atomics): texel q receives its own term (4 q - sum nb)/sqrt(S_q) when it is interior,
// and -(n - q)/sqrt(S_n) from each interior neighbour n; terms with S == 0 contribute nothing.
HPT_DEV float regS(const float4* d, int w, int x, int y)
{
  const float4 p0 = d[y * w + x], a = d[(y + 1) * w + x], b = d[(y - 1) * w + x], c = d[y * w + x - 1], e = d[y * w + x + 1];
  float S = 0.0f;
  const float4 nb[4] = { a, b, c, e };
  for (int k = 0; k < 4; k++) { const float dx = p0.x - nb[k].x, dy = p0.y - nb[k].y, dz = p0.z - nb[k].z; S += dx * dx + dy * dy + dz * dz; }
  return S;
}
__global__ void image2D4fRegularizerKernel(int w, int h, const float4* data, float4* grad)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const float4 q = data[y * w + x];
  float gx = 0.0f, gy = 0.0f, gz = 0.0f;
  const bool interior = x >= 1 && x < w - 1 && y >= 1 && y < h - 1;
  if (interior) {
    const float S = regS(data, w, x, y);
    if (S > 0.0f) {
      const float inv = 1.0f / __builtin_sqrtf(S);
      const float4 a = data[(y + 1) * w + x], b = data[(y - 1) * w + x], c = data[y * w + x - 1], e = data[y * w + x + 1];
      gx += (4.0f * q.x - (a.x + b.x + c.x + e.x)) * inv; gy += (4.0f * q.y - (a.y + b.y + c.y + e.y)) * inv; gz += (4.0f * q.z - (a.z + b.z + c.z + e.z)) * inv;
    }
  }
  const int nx[4] = { x, x, x - 1, x + 1 }, ny[4] = { y + 1, y - 1, y, y };
  for (int k = 0; k < 4; k++) {
    const int X = nx[k], Y = ny[k];
    if (X >= 1 && X < w - 1 && Y >= 1 && Y < h - 1) {              // neighbour n is an interior pixel: q is one of ITS four neighbours
      const float S = regS(data, w, X, Y);
      if (S > 0.0f) {
        const float inv = 1.0f / __builtin_sqrtf(S);
        const float4 n = data[Y * w + X];
        gx -= (n.x - q.x) * inv; gy -= (n.y - q.y) * inv; gz -= (n.z - q.z) * inv;
      }
    }
  }
  float4 g = grad[y * w + x];
  g.x += gx; g.y += gy; g.z += gz;
  grad[y * w + x] = g;
}

// AdamOptimizer<float>::step (diff_render/adam.h:43-62): HBM-bound, 16 bytes per lane per array
__global__ void adamStepKernel(float* state, const float* grad, float* momentum, float* gsq, size_t n, float gamma)
{
  const float alpha = 0.5f, beta = 0.25f, epsilon = 1e-8f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float g = grad[i];
    const float mo = momentum[i] * beta + g * (1.0f - beta);
    const float gs = 2.0f * (gsq[i] * alpha + (g * g) * (1.0f - alpha));
    momentum[i] = mo; gsq[i] = gs;
    state[i] -= (gamma * mo / (__builtin_sqrtf(gs + epsilon)));
  }
}

// ---- device refit of the single-level BVH (UpdateInstance / UpdateGeom_Triangles3f + CommitScene on a committed scene) --------------------
// The topology the host built stays; only the boxes are recomputed, bottom-up, on the GPU: (1) one lane per triangle record takes the
// record's object-space vertices to world space with the instance's CURRENT matrix and writes the padded box the builder would have
// given it; (2) level by level from the deepest, one lane per node: a child's box is the padded union of the primitive boxes below it
// (kept unpadded per node in `bounds`, so the padding does not compound with the depth). Hits do not depend on the boxes as long as
// they contain their triangles, so the refitted tree returns bit for bit what a fresh build returns.
HPT_DEV void padBox(float* lo, float* hi)              // Aabb::pad (bvh_build.h)
{
  const float ex = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
  float mag = 0.0f;
  for (int a = 0; a < 3; a++) mag = fmaxf(mag, fmaxf(absf(lo[a]), absf(hi[a])));
  const float p = 1e-5f * fmaxf(ex, mag) + 1e-30f;
  for (int a = 0; a < 3; a++) { lo[a] -= p; hi[a] += p; }
}
__global__ void refitTriBoxesKernel(const BvhTri* tris, const float* instMat, uint n, float* triBox)
{
  const uint k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const BvhTri t = tris[k];
  const float* m = instMat + 12u * t.instId;            // object -> world rows (3 x 4)
  float lo[3] = { HPT_FLT_MAX, HPT_FLT_MAX, HPT_FLT_MAX }, hi[3] = { -HPT_FLT_MAX, -HPT_FLT_MAX, -HPT_FLT_MAX };
  for (int v = 0; v < 3; v++) {
    const float p[3] = { t.v0[0] + (v == 1 ? t.e1[0] : v == 2 ? t.e2[0] : 0.0f), t.v0[1] + (v == 1 ? t.e1[1] : v == 2 ? t.e2[1] : 0.0f), t.v0[2] + (v == 1 ? t.e1[2] : v == 2 ? t.e2[2] : 0.0f) };
    for (int a = 0; a < 3; a++) {
      const float q = m[4 * a + 0] * p[0] + m[4 * a + 1] * p[1] + m[4 * a + 2] * p[2] + m[4 * a + 3];
      lo[a] = fminf(lo[a], q); hi[a] = fmaxf(hi[a], q);
    }
  }
  padBox(lo, hi);                                        // covers the rounding of v0 + e and of the transform, as the builder's per-triangle pad does
  float* o = triBox + 6u * (size_t)k;
  o[0] = lo[0]; o[1] = hi[0]; o[2] = lo[1]; o[3] = hi[1]; o[4] = lo[2]; o[5] = hi[2];
}
// DevScene::shadeTris: one 64-byte record per triangle record of the single-level layout (see hpt_types.h)
__global__ void buildShadeTrisKernel(const DevScene S, uint n, float4* out)
{
  const uint k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const BvhTri t = S.tris[k];
  const uint instId = t.instId, prim = t.primId;
  const uint geomId = S.insts[instId].geomId;
  const uint triOffset = S.matVertOffset[2 * geomId + 0], vertOffset = S.matVertOffset[2 * geomId + 1];
  const uint A = S.triIndices[(triOffset + prim) * 3 + 0], B = S.triIndices[(triOffset + prim) * 3 + 1], C = S.triIndices[(triOffset + prim) * 3 + 2];
  const float4 nA = ((const float4*)S.vData8f)[2 * (A + vertOffset)], nB = ((const float4*)S.vData8f)[2 * (B + vertOffset)], nC = ((const float4*)S.vData8f)[2 * (C + vertOffset)];
  const float tyA = S.vData8f[8 * (A + vertOffset) + 7], tyB = S.vData8f[8 * (B + vertOffset) + 7], tyC = S.vData8f[8 * (C + vertOffset) + 7];
  const uint matId = remapMaterialId(S, S.matIdByPrimId[triOffset + prim], instId) & 0x00FFFFFFu;
  float4* o = out + 4u * (size_t)k;
  o[0] = nA; o[1] = nB; o[2] = nC; o[3] = make_float4(tyA, tyB, tyC, __uint_as_float(matId));
}

// after the level passes: every 4-wide node requantises its children's boxes from the BVH2 nodes they come from (src = node << 1 | side)
__global__ void refitNodes4Kernel(BvhNode4* nodes4, const uint* src, const BvhNode* nodes2, uint count)
{
  const uint i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  BvhNode4 nd = nodes4[i];
  float lo[4][3], hi[4][3];
  const uint valid = nd.exps >> 24;
  for (int c = 0; c < 4; c++) {
    if (!(valid & (1u << c))) continue;
    const uint sidx = src[4u * (size_t)i + c];
    const float* q = nodes2[sidx >> 1].q + 6u * (sidx & 1u);
    for (int a = 0; a < 3; a++) { lo[c][a] = q[2 * a]; hi[c][a] = q[2 * a + 1]; }
  }
  uint keep[4]; for (int c = 0; c < 4; c++) keep[c] = nd.ref[c];
  quantizeNode4(lo, hi, valid, nd);
  for (int c = 0; c < 4; c++) nd.ref[c] = keep[c];
  nodes4[i] = nd;
}
__global__ void refitLevelKernel(BvhNode* nodes, const uint* ids, uint count, const float* triBox, float* bounds)
{
  const uint i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint id = ids[i];
  BvhNode nd = nodes[id];
  float nlo[3] = { HPT_FLT_MAX, HPT_FLT_MAX, HPT_FLT_MAX }, nhi[3] = { -HPT_FLT_MAX, -HPT_FLT_MAX, -HPT_FLT_MAX };
  for (int c = 0; c < 2; c++) {
    const uint ref = c ? nd.ref1 : nd.ref0;
    if (ref == REF_NONE) continue;
    float lo[3] = { HPT_FLT_MAX, HPT_FLT_MAX, HPT_FLT_MAX }, hi[3] = { -HPT_FLT_MAX, -HPT_FLT_MAX, -HPT_FLT_MAX };
    if (ref & REF_LEAF) {
      const uint cnt = (ref >> 28) & 7u, first = ref & 0x0FFFFFFFu;
      for (uint k = 0; k < cnt; k++) { const float* b = triBox + 6u * (size_t)(first + k); for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], b[2 * a]); hi[a] = fmaxf(hi[a], b[2 * a + 1]); } }
    } else {
      const float* b = bounds + 6u * (size_t)ref;       // the child's level is deeper: already refitted
      for (int a = 0; a < 3; a++) { lo[a] = b[2 * a]; hi[a] = b[2 * a + 1]; }
    }
    for (int a = 0; a < 3; a++) { nlo[a] = fminf(nlo[a], lo[a]); nhi[a] = fmaxf(nhi[a], hi[a]); }
    padBox(lo, hi);
    for (int a = 0; a < 3; a++) { nd.q[6 * c + 2 * a] = lo[a]; nd.q[6 * c + 2 * a + 1] = hi[a]; }
  }
  float* o = bounds + 6u * (size_t)id;
  for (int a = 0; a < 3; a++) { o[2 * a] = nlo[a]; o[2 * a + 1] = nhi[a]; }
  nodes[id] = nd;
}

} // namespace hpt
