// Surface shading shared by the persistent megakernel (hpt_kernels.hip) and the wavefront pipeline (hpt_wavefront.hip):
// one path vertex = kernel_GetRayColor's surface fetch + kernel_SampleLightSource + kernel_NextBounce of the reference
// (integrator_pt.cpp:238-548), with the light sample's BSDF evaluated BEFORE the shadow ray is traced so that only the
// candidate contribution has to survive the any-hit traversal.
#pragma once
#include "hpt_device.h"
#include "hpt_film.h"

namespace hpt {


HPT_DEV uint lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
HPT_DEV uint mbcnt64(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint)m, 0u)); }

// differentiable texture fetch: Tex2DFetchAD (diff_render/integrator_dr.cpp:95-161); returns texColor and tap data
HPT_DEV V4 texFetchAD(const DevScene& S, const float* data, uint texId, V2 uv, Taps& taps, bool& isParam)
{
  const TexRec t = S.textures[texId];
  isParam = false;
  if (t.diffOffset != ~0ull && data != nullptr) {
    taps = bilinearTaps(t.diffW, t.diffH, t.addrU, t.addrV, uv);
    taps.base = (uint)t.diffOffset; taps.ch = t.diffChannels;
    isParam = true;
    const float* d = data + t.diffOffset;
    if (t.diffChannels == 4) {
      const float4 a = ((const float4*)d)[taps.off[0]], b = ((const float4*)d)[taps.off[1]], c = ((const float4*)d)[taps.off[2]], e = ((const float4*)d)[taps.off[3]];
      return v4(a.x * taps.w[0] + b.x * taps.w[1] + c.x * taps.w[2] + e.x * taps.w[3],
                a.y * taps.w[0] + b.y * taps.w[1] + c.y * taps.w[2] + e.y * taps.w[3],
                a.z * taps.w[0] + b.z * taps.w[1] + c.z * taps.w[2] + e.z * taps.w[3],
                a.w * taps.w[0] + b.w * taps.w[1] + c.w * taps.w[2] + e.w * taps.w[3]);
    }
    const float o = d[taps.off[0]] * taps.w[0] + d[taps.off[1]] * taps.w[1] + d[taps.off[2]] * taps.w[2] + d[taps.off[3]] * taps.w[3];
    return v4(o, o, o, o);
  }
  return texSample(S.textures, texId, uv);
}

// ---- lens simulation: TraceLensesFromFilm and its helpers (integrator_pt.cpp:806-938, after pbrt's RealisticCamera) --------------------------
HPT_DEV bool lensQuadratic(float A, float B, float C, float& t0, float& t1)
{
  const float discrim = B * B - 4.0f * A * C;
  if (discrim < 0.f) return false;
  const float rootDiscrim = sqrtf_(discrim);
  const float q = (B < 0.0f) ? -.5f * (B - rootDiscrim) : -.5f * (B + rootDiscrim);
  t0 = q / A; t1 = C / q;
  if (t0 > t1) { const float t = t0; t0 = t1; t1 = t; }
  return true;
}
HPT_DEV bool lensRefract(V3 wi, V3 n, float eta, V3& wt)
{
  const float cosThetaI = dot(n, wi);
  const float sin2ThetaI = smax(0.0f, 1.0f - cosThetaI * cosThetaI);
  const float sin2ThetaT = eta * eta * sin2ThetaI;
  if (sin2ThetaT >= 1) return false;
  const float cosThetaT = sqrtf_(1 - sin2ThetaT);
  wt = eta * (-1.0f) * wi + (eta * cosThetaI - cosThetaT) * n;
  return true;
}
HPT_DEV bool intersectSphericalElement(float radius, float zCenter, V3 rayPos, V3 rayDir, float& t, V3& n)
{
  const V3 o = rayPos - v3(0, 0, zCenter);
  const float A = rayDir.x * rayDir.x + rayDir.y * rayDir.y + rayDir.z * rayDir.z;
  const float B = 2 * (rayDir.x * o.x + rayDir.y * o.y + rayDir.z * o.z);
  const float C = o.x * o.x + o.y * o.y + o.z * o.z - radius * radius;
  float t0, t1;
  if (!lensQuadratic(A, B, C, t0, t1)) return false;
  const bool useCloserT = (rayDir.z > 0.0f) != (radius < 0.0f);
  t = useCloserT ? smin(t0, t1) : smax(t0, t1);
  if (t < 0.0f) return false;
  n = normalize(o + t * rayDir);
  n = (dot(n, -1.0f * rayDir) < 0.f) ? (-1.0f) * n : n;               // faceforward
  return true;
}
HPT_DEV bool traceLensesFromFilm(const DevScene& S, V3& rayPos, V3& rayDir)
{
  float elementZ = 0;
  V3 p = v3(rayPos.x, rayPos.y, -rayPos.z), d = v3(rayDir.x, rayDir.y, -rayDir.z);     // camera -> lens-system space
  for (uint i = 0; i < S.lensCount; i++) {
    const float4 e = S.lensLines[i];                                   // {curvatureRadius, thickness, eta, apertureRadius}
    elementZ -= e.y;
    float t; V3 n = v3(0, 0, 0);
    const bool isStop = (e.x == 0.0f);
    if (isStop) {
      if (d.z >= 0.0f) return false;
      t = (elementZ - p.z) / d.z;
    } else {
      if (!intersectSphericalElement(e.x, elementZ + e.x, p, d, t, n)) return false;
    }
    const V3 pHit = p + t * d;
    if (pHit.x * pHit.x + pHit.y * pHit.y > e.w * e.w) return false;
    p = pHit;
    if (!isStop) {
      const float etaI = e.z;
      float etaT = (i == S.lensCount - 1u) ? 1.0f : S.lensLines[i + 1u].z;
      if (etaT == 0.0f) etaT = 1.0f;
      V3 wt;
      if (!lensRefract(normalize((-1.0f) * d), n, etaI / etaT, wt)) return false;
      d = wt;
    }
  }
  rayPos = v3(p.x, p.y, -p.z); rayDir = v3(d.x, d.y, -d.z);
  return true;
}

// SampleCameraRay + kernel_InitEyeRay2 (integrator_pt.cpp:44-157), RGB subset
// LENS: the lens-simulation branch exists only in the kernels with every BSDF branch (the host routes scenes with m_enableOpticSim to them):
// compiled into the lean kernels it cost the Cornell benchmark 0.6 % without ever running (profiles/ab.sh, base vs -DHPT_NO_LENS)
template <bool LENS>
HPT_DEV void cameraRay(const DevScene& S, uint x, uint y, V4 pixelOffsets, V3& rayPos, V3& rayDir)
{
  const float fx = float(x) + pixelOffsets.x, fy = float(y) + pixelOffsets.y;
  const float xn = (fx + float(S.winStartX)) / float(S.fbWidth);
  const float yn = (fy + float(S.winStartY)) / float(S.fbHeight);
  V4 pos = v4(2.0f * xn - 1.0f, 2.0f * yn - 1.0f, 0.0f, 1.0f);          // EyeRayDirNormalized (cglobals.h:49-55)
  pos = mul4x4(S.projInv, pos);
  V3 dir = normalize(v3(pos.x / pos.w, pos.y / pos.w, pos.z / pos.w));
  V3 org = v3(0, 0, 0);
  if (S.camLensRadius > 0.0f) {
    const float tFocus = S.camTargetDist / (-dir.z);
    const V3 focusPosition = org + dir * tFocus;
    const V2 d2 = mapSamplesToDisc(v2(pixelOffsets.z - 0.5f, pixelOffsets.w - 0.5f));
    const float k = S.camLensRadius * 2.0f;
    org.x += k * d2.x; org.y += k * d2.y;
    dir = normalize(focusPosition - org);
  }
  else if (LENS && S.lensCount != 0u) {                                  // m_enableOpticSim (:79-103): a film point, a point on the rear element, the lens stack
    org = v3(0.25f * S.physSize[0] * (2.0f * xn - 1.0f), 0.25f * S.physSize[1] * (2.0f * yn - 1.0f), 0.0f);
    const float4 rear = S.lensLines[0];                                  // LensRearZ() = thickness, LensRearRadius() = apertureRadius of the first line
    const V2 rs = mapSamplesToDisc(v2(pixelOffsets.z - 0.5f, pixelOffsets.w - 0.5f));
    const float k = rear.w * 2.0f;
    dir = normalize(v3(k * rs.x, k * rs.y, rear.y) - org);
    if (!traceLensesFromFilm(S, org, dir)) { org = v3(0, -10000000.0f, 0.0f); dir = v3(0, -1, 0); }   // "shoot ray under the floor"
    else { dir = (-1.0f) * normalize(dir); org = (-1.0f) * org; }
  }
  const V3 p1 = mul4x3(S.worldViewInv, org);                             // transform_ray3f (cglobals.h:254-263)
  const V3 p2 = mul4x3(S.worldViewInv, org + 100.0f * dir);
  rayPos = p1;
  rayDir = normalize(p2 - p1);
}

// ---- normal-map bump (integrator_pt_mat.cpp:94-107, 131-139, 298-303, 336-355; NormalMapTransform in include/cmaterial.h) -------------------
// world-space tangent of the hit (integrator_pt.cpp:270-302: interpolated, through the normal matrix, normalised, flipped with the normal)
HPT_DEV V3 hitTangent(const DevScene& S, uint A, uint B, uint C, uint vertOffset, float wA, float uvx, float uvy, const float* nm, float flipNorm,
                      const float* nm2 = nullptr, float time = 0.0f)
{
  const float4 tA = ((const float4*)S.vData8f)[2 * (A + vertOffset) + 1], tB = ((const float4*)S.vData8f)[2 * (B + vertOffset) + 1], tC = ((const float4*)S.vData8f)[2 * (C + vertOffset) + 1];
  const V3 tO = v3(wA * tA.x + uvy * tB.x + uvx * tC.x, wA * tA.y + uvy * tB.y + uvx * tC.y, wA * tA.z + uvy * tB.z + uvx * tC.z);
  V3 t = v3(nm[0] * tO.x + nm[1] * tO.y + nm[2] * tO.z, nm[4] * tO.x + nm[5] * tO.y + nm[6] * tO.z, nm[8] * tO.x + nm[9] * tO.y + nm[10] * tO.z);
  if (nm2) {                                                             // motion blur: lerp towards the end-of-motion matrix applied to the result (integrator_pt.cpp:285-292)
    const V3 t2 = v3(nm2[0] * t.x + nm2[1] * t.y + nm2[2] * t.z, nm2[4] * t.x + nm2[5] * t.y + nm2[6] * t.z, nm2[8] * t.x + nm2[9] * t.y + nm2[10] * t.z);
    t = t + time * (t2 - t);
  }
  return flipNorm * normalize(t);
}
// BumpMapping: the tangent-space normal of the map through the inverse of the matrix with rows (tan, bitan, n) - by cofactors,
// inv(M) = [r1 x r2 | r2 x r0 | r0 x r1] / det (LiteMath's make_float3x3 / inverse3x3 are absent from the reference tree)
HPT_DEV V3 bumpNormal(const DevScene& S, const MaterialRec& m, V3 n, V3 tan, V2 uv)
{
  const V4 nt = texSample(S.textures, m.texid[1], mulRows2x4(m.row0[1], m.row1[1], uv));
  V3 ts = v3(2.0f * nt.x - 1.0f, 2.0f * nt.y - 1.0f, nt.z);
  if ((m.cflags & FLAG_NMAP_INVERT_X) != 0) ts.x *= -1.0f;
  if ((m.cflags & FLAG_NMAP_INVERT_Y) != 0) ts.y *= -1.0f;
  if ((m.cflags & FLAG_NMAP_SWAP_XY) != 0) { const float t = ts.x; ts.x = ts.y; ts.y = t; }
  const V3 bitan = cross(n, tan);
  const V3 c0 = cross(bitan, n), c1 = cross(n, tan), c2 = cross(tan, bitan);
  const float det = dot(tan, c0);
  const V3 w = (ts.x * c0 + ts.y * c1 + ts.z * c2) / det;
  return normalize(w);
}
// MaterialEval's cosine correction for a bent shading normal (:341-355)
HPT_DEV float bumpCosMult(V3 l, V3 geomNormal, V3 shadeNormal)
{
  const float c1 = smax(dot(l, geomNormal), 0.0f), c2 = smax(dot(l, shadeNormal), 0.0f);
  return (c1 <= 0.0f) ? 0.0f : c2 / smax(c1, 1e-6f);
}
// ---- blend materials (integrator_pt_mat.cpp:23-77, 123-130, 316-333, 511-527); only in the non-LEAN kernels ---------------------------------
// texture colour and the "four scalar parameters" of a leaf material (:139-167)
HPT_DEV void leafTextures(const DevScene& S, const MaterialRec& m, V2 uv, V3& tex3, V3& four)
{
  const V4 tc = texSample(S.textures, m.texid[0], mulRows2x4(m.row0[0], m.row1[0], uv));
  tex3 = v3(tc.x, tc.y, tc.z); four = v3(1, 1, 1);
  if ((m.cflags & FLAG_FOUR_TEXTURES) != 0) {
    const V4 c2 = texSample(S.textures, m.texid[2], mulRows2x4(m.row0[2], m.row1[2], uv));
    const V4 c3 = texSample(S.textures, m.texid[3], mulRows2x4(m.row0[3], m.row1[3], uv));
    four = ((m.cflags & FLAG_PACK_FOUR_PARAMS_IN_TEXTURE) != 0) ? v3(c2.x, c2.y, c2.z) : v3(c2.x, c3.x, 1.0f);
  }
}
// MaterialEval of a blend tree: a stack of (material id, weight) pairs, BLEND_STACK_SIZE deep, in the reference's visiting order
template <bool FILM>
HPT_DEV void blendTreeEval(const DevScene& S, uint rootId, V2 uv, V3 l, V3 v, V3 gn, V3 tan, BsdfE& res)
{
  uint stackId[BLEND_STACK_SIZE]; float stackW[BLEND_STACK_SIZE];
  uint curId = rootId; float curW = 1.0f;
  stackId[0] = curId; stackW[0] = curW;
  int top = 0; bool needPop = false;
  do {
    if (needPop) { top--; const int t = top > 0 ? top : 0; curId = stackId[t]; curW = stackW[t]; } else needPop = true;
    const MaterialRec& m = S.materials[curId];
    V3 tex3, four; leafTextures(S, m, uv, tex3, four);
    BsdfE cv; cv.val = v3(0, 0, 0); cv.pdf = 0.0f; cv.dval = v3(0, 0, 0);
    const uint t = m.mtype;
    V3 n = gn; float bm = 1.0f;
    if (m.texid[1] != 0xFFFFFFFFu && t != MAT_TYPE_BLEND) { n = bumpNormal(S, m, gn, tan, uv); bm = bumpCosMult(l, gn, n); }
    if (t == MAT_TYPE_GLTF) { gltfEval(m, l, v, n, ld3(m.colors[GLTF_COLOR_BASE]) * tex3, four, cv); res.val = res.val + cv.val * curW * bm; res.pdf += cv.pdf * curW; }
    else if (t == MAT_TYPE_CONDUCTOR) {
      if (!(smax(m.data[1], m.data[0]) < 1e-3f)) conductorRoughEval(m, m.data[2], m.data[3], l, v, n, tex3, cv);
      res.val = res.val + cv.val * curW * bm; res.pdf += cv.pdf * curW;
    }
    else if (t == MAT_TYPE_DIFFUSE) { diffuseEval(m, ld3(m.colors[0]) * tex3, l, v, n, cv); res.val = res.val + cv.val * curW * bm; res.pdf += cv.pdf * curW; }
    else if (t == MAT_TYPE_PLASTIC) { plasticEval(m, ld3(m.colors[0]) * tex3, l, v, n, cv, S.arrays1f, m.datai[0]); res.val = res.val + cv.val * curW * bm; res.pdf += cv.pdf * curW; }
    else if (FILM && t == MAT_TYPE_THIN_FILM) {              // the geometric normal (integrator_pt_mat.cpp:464)
      filmEvalBranch(S, m, uv, 0.0f, l, v, gn, tex3, cv);
      res.val = res.val + cv.val * curW * bm; res.pdf += cv.pdf * curW;
    }
    else if (t == MAT_TYPE_BLEND) {                          // BlendEval: first child next (no pop), second child waits on the stack
      const float w = m.data[0] * tex3.x;
      const uint id1 = m.datai[0], id2 = m.datai[1];
      const float w1 = curW * (1.0f - w), w2 = curW * w;
      curId = id1; curW = w1; needPop = false;
      if (top + 1 <= (int)BLEND_STACK_SIZE) { stackId[top] = id2; stackW[top] = w2; top++; }
    }                                                        // glass / dielectric leaves add zero
  } while (top > 0);
}

// Shades the vertex a closest-hit query returned for one path.  All path registers are passed by reference and the
// function is always inlined, so both callers keep them in VGPRs.  Returns true when the path continues (didBounce).
// A miss only sets the OUT_OF_SCENE flags.  The caller traces the shadow ray (if wantShadow) and adds `contrib`.
// MOTION: moving instances - the normal (and tangent) are interpolated at the path's `time` (see hitTangent).
// LEAN: the scene holds gltf and emissive materials only (the host checked): the conductor / diffuse / glass / dielectric branches are
// compiled out - fewer live registers and spills in the kernels every benchmark scene runs (the DR variant is lean by definition).
// FILM: the scene holds thin films (MAT_TYPE_THIN_FILM, hpt_film.h): their branches exist in the FILM variants only.
template <bool DR, bool NAIVE, bool LEAN = false, bool MOTION = false, bool FILM = false>
HPT_DEV bool shadeVertex(const DevScene& S, const float* diffData, const HitRec& hit,
                         V3& rpos, V3& rdir, V3& accum, V3& thr, float& misPdf, float& misIor, uint& flags, const uint bounce, Rng& gen,
                         bool& wantShadow, V3& shPos, V3& shDir, float& shFar, V3& contrib,
                         V3& recA, V3& recS, V3& recdA, V3& recdS, Taps& recTaps, uint& recTex, V3& tailR, const float time = 0.0f)
{
  bool didBounce = false;
  if (hit.inst == 0xFFFFFFFFu) {
    flags |= (bounce == 0) ? (RAY_FLAG_PRIME_RAY_MISS | RAY_FLAG_IS_DEAD | RAY_FLAG_OUT_OF_SCENE) : (RAY_FLAG_IS_DEAD | RAY_FLAG_OUT_OF_SCENE);
    return false;
  }
  {
    // -- surface attributes (integrator_pt.cpp:238-311) --
    const uint instId = hit.inst;
    const V3 hitPos = rpos + hit.t * (1.f - 1e-6f) * rdir;
    const float uvx = hit.v, uvy = hit.u;                              // coords[0] = v, coords[1] = u (EmbreeRT.cpp:350-352)
    // gltf / emissive kernels on the single-level layout: the three vertices' shading data and the material id come from ONE 64-byte record
    // per triangle (DevScene::shadeTris) instead of the chain instance -> mesh offsets -> indices -> vertices / primitive -> id -> remap
    const bool packed = (DR || LEAN) && !MOTION && S.shadeTris != nullptr && hit.slot != 0xFFFFFFFFu;
    uint geomId = 0, triOffset = 0, vertOffset = 0, A = 0, B = 0, C = 0, packedMat = 0;
    float4 nA, nB, nC; float tyA, tyB, tyC;
    if (packed) {
      const float4* sp = S.shadeTris + 4u * (size_t)hit.slot;
      const float4 q3 = sp[3];
      nA = sp[0]; nB = sp[1]; nC = sp[2]; tyA = q3.x; tyB = q3.y; tyC = q3.z; packedMat = __float_as_uint(q3.w);
    } else {
      geomId = S.insts[instId].geomId;
      triOffset = S.matVertOffset[2 * geomId + 0]; vertOffset = S.matVertOffset[2 * geomId + 1];
      A = S.triIndices[(triOffset + hit.prim) * 3 + 0];
      B = S.triIndices[(triOffset + hit.prim) * 3 + 1];
      C = S.triIndices[(triOffset + hit.prim) * 3 + 2];
      nA = ((const float4*)S.vData8f)[2 * (A + vertOffset)]; nB = ((const float4*)S.vData8f)[2 * (B + vertOffset)]; nC = ((const float4*)S.vData8f)[2 * (C + vertOffset)];
      tyA = S.vData8f[8 * (A + vertOffset) + 7]; tyB = S.vData8f[8 * (B + vertOffset) + 7]; tyC = S.vData8f[8 * (C + vertOffset) + 7];
    }
    const float wA = 1.0f - uvx - uvy;
    const V3 nrmO = v3(wA * nA.x + uvy * nB.x + uvx * nC.x, wA * nA.y + uvy * nB.y + uvx * nC.y, wA * nA.z + uvy * nB.z + uvx * nC.z);
    const V2 uv = v2(wA * nA.w + uvy * nB.w + uvx * nC.w, wA * tyA + uvy * tyB + uvx * tyC);
    const float* nm = S.normMat + 12 * instId;
    V3 hitNorm = v3(nm[0] * nrmO.x + nm[1] * nrmO.y + nm[2] * nrmO.z,
                    nm[4] * nrmO.x + nm[5] * nrmO.y + nm[6] * nrmO.z,
                    nm[8] * nrmO.x + nm[9] * nrmO.y + nm[10] * nrmO.z);
    if (MOTION && (S.motion & 2u) == 0u) {                               // integrator_pt.cpp:285-292: m_normMatrices[m_normMatrices2Offs + inst] applied to the
      const float* nm2 = S.normMat2 + 12 * instId;                       // already transformed normal, then lerp(hitNorm, hitNorm2, time)
      const V3 n2 = v3(nm2[0] * hitNorm.x + nm2[1] * hitNorm.y + nm2[2] * hitNorm.z,
                       nm2[4] * hitNorm.x + nm2[5] * hitNorm.y + nm2[6] * hitNorm.z,
                       nm2[8] * hitNorm.x + nm2[9] * hitNorm.y + nm2[10] * hitNorm.z);
      hitNorm = hitNorm + time * (n2 - hitNorm);
    }
    hitNorm = normalize(hitNorm);
    const float flipNorm = dot(rdir, hitNorm) > 0.001f ? -1.0f : 1.0f;
    hitNorm = flipNorm * hitNorm;
    if (flipNorm < 0.0f) flags |= RAY_FLAG_HAS_INV_NORMAL; else flags &= ~RAY_FLAG_HAS_INV_NORMAL;
    const uint matId = packed ? packedMat : (remapMaterialId(S, S.matIdByPrimId[triOffset + hit.prim], instId) & 0x00FFFFFFu);
    const MaterialRec& m = S.materials[matId];
    const uint mtype = m.mtype;
    const V3 vdir = (-1.0f) * rdir;
    V3 hitTang = v3(0, 0, 0);                                          // only materials with a normal map (or a blend that may hold one) read it
    if (!(DR || LEAN) && (mtype == MAT_TYPE_BLEND || (mtype != MAT_TYPE_LIGHT_SOURCE && m.texid[1] != 0xFFFFFFFFu)))
      hitTang = hitTangent(S, A, B, C, vertOffset, wA, uvx, uvy, nm, flipNorm, (MOTION && (S.motion & 2u) == 0u) ? S.normMat2 + 12 * instId : nullptr, time);

    // -- kernel_SampleLightSource (integrator_pt.cpp:350-424): the randoms are drawn for every surface hit --
    V3 shade = v3(0, 0, 0), dshade = v3(0, 0, 0);
    V4 texColor = v4(1, 1, 1, 1); V3 four = v3(1, 1, 1);
    bool isParam = false;
    if (mtype != MAT_TYPE_LIGHT_SOURCE) {
      const V2 tcT = mulRows2x4(m.row0[0], m.row1[0], uv);
      if (DR) { texColor = texFetchAD(S, diffData, m.texid[0], tcT, recTaps, isParam); if (isParam) recTex = m.texid[0]; }
      else    texColor = texSample(S.textures, m.texid[0], tcT);
      if ((m.cflags & FLAG_FOUR_TEXTURES) != 0) {                      // integrator_pt_mat.cpp:151-167
        const V4 c2 = texSample(S.textures, m.texid[2], mulRows2x4(m.row0[2], m.row1[2], uv));
        const V4 c3 = texSample(S.textures, m.texid[3], mulRows2x4(m.row0[3], m.row1[3], uv));
        four = ((m.cflags & FLAG_PACK_FOUR_PARAMS_IN_TEXTURE) != 0) ? v3(c2.x, c2.y, c2.z) : v3(c2.x, c3.x, 1.0f);
      }
    }
    const V3 tex3 = v3(texColor.x, texColor.y, texColor.z);
    const V3 baseCol = ld3(m.colors[GLTF_COLOR_BASE]);

    if (!NAIVE) {
      const float rndId = rng_float1(gen);                             // GetRandomNumbersLgts: two generator steps, in this order
      const V4 r4 = rng_float4(gen);
      const int nLights = (int)S.numLights;
      const int lightId = min((int)floorf(rndId * float(nLights)), nLights - 1);
      if (lightId >= 0 && mtype != MAT_TYPE_LIGHT_SOURCE) {
        const LightRec& L = S.lights[lightId];
        const LightSam ls = (L.geomType == LIGHT_GEOM_ENV) ? envLightSampleRev(S, L, v3(r4.x, r4.y, r4.z), hitPos) : lightSampleRev(L, v3(r4.x, r4.y, r4.z), hitPos);
        const V3 dlt = hitPos - ls.pos;
        const float hitDist = sqrtf_(dot(dlt, dlt));
        const V3 shadowRayDir = normalize(ls.pos - hitPos);
        const V3 shadowRayPos = hitPos + hitNorm * smax(maxcomp(hitPos), 1.0f) * 5e-6f;
        const bool inIllumArea = (dot(shadowRayDir, ls.norm) < 0.0f) || ls.isOmni || ls.hasIES;
        if (inIllumArea) {
          // MaterialEval (integrator_pt_mat.cpp:308-528), evaluated before the shadow ray so that nothing but the
          // candidate contribution has to stay live across the any-hit traversal
          BsdfE bv; bv.val = v3(0, 0, 0); bv.pdf = 0.0f; bv.dval = v3(0, 0, 0);
          V3 evalNorm = hitNorm; float bumpMult = 1.0f;
          if (!(DR || LEAN) && mtype != MAT_TYPE_BLEND && m.texid[1] != 0xFFFFFFFFu) {
            evalNorm = bumpNormal(S, m, hitNorm, hitTang, uv); bumpMult = bumpCosMult(shadowRayDir, hitNorm, evalNorm);
          }
          if (mtype == MAT_TYPE_GLTF) gltfEval(m, shadowRayDir, vdir, evalNorm, baseCol * tex3, four, bv);
          else if (!(DR || LEAN) && mtype == MAT_TYPE_CONDUCTOR) {
            if (!(smax(m.data[1], m.data[0]) < 1e-3f)) conductorRoughEval(m, m.data[2], m.data[3], shadowRayDir, vdir, evalNorm, tex3, bv);
          }
          else if (!(DR || LEAN) && mtype == MAT_TYPE_DIFFUSE) diffuseEval(m, ld3(m.colors[0]) * tex3, shadowRayDir, vdir, evalNorm, bv);
          else if (!(DR || LEAN) && mtype == MAT_TYPE_PLASTIC) plasticEval(m, ld3(m.colors[0]) * tex3, shadowRayDir, vdir, evalNorm, bv, S.arrays1f, m.datai[0]);
          else if (FILM && mtype == MAT_TYPE_THIN_FILM) {                 // rough films only; the geometric normal (integrator_pt_mat.cpp:422-470)
            filmEvalBranch(S, m, uv, 0.0f, shadowRayDir, vdir, hitNorm, tex3, bv);
          }
          else if (!(DR || LEAN) && mtype == MAT_TYPE_BLEND) blendTreeEval<FILM>(S, matId, uv, shadowRayDir, vdir, hitNorm, hitTang, bv);
          if (!(DR || LEAN) && mtype != MAT_TYPE_BLEND) bv.val = bv.val * bumpMult;     // res.val += currVal.val * weight * bumpCosMult, weight 1
          const float cosThetaOut = smax(dot(shadowRayDir, hitNorm), 0.0f);
          float lgtPdfW = (1.0f / float(nLights)) * lightEvalPDF(L, shadowRayPos, shadowRayDir, ls.pos, ls.norm, ls.pdf);
          float misWeight = (S.integratorType == INTEGRATOR_MIS_PT) ? misWeightHeuristic(lgtPdfW, bv.pdf) : 1.0f;
          if (L.geomType == LIGHT_GEOM_DIRECT) { misWeight = 1.0f; lgtPdfW = 1.0f; }
          else if (L.geomType == LIGHT_GEOM_POINT) misWeight = 1.0f;
          const bool isDirectLight = (flags & RAY_FLAG_HAS_NON_SPEC) == 0;
          if ((S.renderLayer == FB_DIRECT && !isDirectLight) || (S.renderLayer == FB_INDIRECT && isDirectLight)) misWeight = 0.0f;
          const V3 lightColor = lightIntensity(S, L, shadowRayPos, shadowRayDir);
          shade = ((lightColor * bv.val) / lgtPdfW) * cosThetaOut * misWeight;
          if (DR) dshade = ((lightColor * bv.dval) / lgtPdfW) * cosThetaOut * misWeight;
          wantShadow = true;
          shPos = shadowRayPos; shDir = shadowRayDir; shFar = hitDist * 0.9995f;
        }
      }
    }

    // -- kernel_NextBounce (integrator_pt.cpp:426-548) --
    if (mtype == MAT_TYPE_LIGHT_SOURCE) {
      const V4 tc = texSample(S.textures, m.texid[0], mulRows2x4(m.row0[0], m.row1[0], uv));
      const uint lightId = (uint)S.remapInst[2 * instId + 1];
      V3 lightInt = ld3(m.colors[0]) * v3(tc.x, tc.y, tc.z);
      float misWeight = 1.0f;
      if (lightId != 0xFFFFFFFFu) {
        const LightRec& L = S.lights[lightId];
        const float lightCos = dot(rdir, ld3(L.norm));
        const float atten = (lightCos < 0.0f || L.geomType == LIGHT_GEOM_SPHERE) ? 1.0f : 0.0f;
        lightInt = lightIntensity(S, L, rpos, rdir) * atten;
      }
      if (S.integratorType == INTEGRATOR_MIS_PT) {
        if (bounce > 0 && lightId != 0xFFFFFFFFu) {
          const float lgtPdf = (1.0f / float(S.numLights)) * lightEvalPDF(S.lights[lightId], rpos, rdir, hitPos, hitNorm, 1.0f);
          misWeight = misWeightHeuristic(misPdf, lgtPdf);
          if (misPdf <= 0.0f) misWeight = 1.0f;
        }
      } else if (S.integratorType == INTEGRATOR_SHADOW_PT && (flags & RAY_FLAG_HAS_NON_SPEC) != 0) misWeight = 0.0f;
      const bool isDirectLight = (flags & RAY_FLAG_HAS_NON_SPEC) == 0;
      const bool isFirstNonSpec = (flags & RAY_FLAG_FIRST_NON_SPEC) != 0;
      if (S.renderLayer == FB_INDIRECT && (isDirectLight || isFirstNonSpec)) misWeight = 0.0f;
      accum = accum + thr * lightInt * misWeight;
      if (DR) tailR = lightInt * misWeight;
      flags |= (RAY_FLAG_IS_DEAD | RAY_FLAG_HIT_LIGHT);
    } else {
      // The reference keeps the hit's material id in the low 24 bits of the ray flags (packMatId, integrator_pt.h:340-341, set at
      // integrator_pt.cpp:307) and hands that word to MaterialSampleAndEval as the initial sample flags (integrator_pt_mat.cpp:118).
      // The glass and dielectric samplers OR their events into it, so RAY_EVENT_S (1) / RAY_EVENT_T (8) tested at integrator_pt.cpp:514,534
      // also read bits 0 and 3 of the material id.  Restated here bit for bit: the event bits of the previous bounce do not survive.
      BsdfS ms; ms.val = v3(0, 0, 0); ms.pdf = 1.0f; ms.dir = v3(0, 1, 0); ms.ior = 1.0f; ms.flags = (flags & 0xFF000000u) | matId; ms.dval = v3(0, 0, 0);
      // blend descent (BlendSampleAndEval): one generator step per layer BEFORE the float4; the leaf then overwrites val / pdf as in the
      // reference, whose leaf samplers assign them
      const MaterialRec* lm = &m; uint lt = mtype; V3 ltex3 = tex3, lfour = four;
      if (!(DR || LEAN) && mtype == MAT_TYPE_BLEND) {
        while (lt == MAT_TYPE_BLEND) {
          const V4 wd = texSample(S.textures, lm->texid[0], mulRows2x4(lm->row0[0], lm->row1[0], uv));
          const float weight = lm->data[0] * wd.x;
          const float select = rng_float1(gen);                        // GetRandomNumbersMatB (integrator_pt.cpp:37)
          if (select < weight) { ms.pdf *= weight; ms.val = ms.val * weight; lm = &S.materials[lm->datai[1]]; }
          else                 { ms.pdf *= 1.0f - weight; ms.val = ms.val * (1.0f - weight); lm = &S.materials[lm->datai[0]]; }
          lt = lm->mtype;
        }
        leafTextures(S, *lm, uv, ltex3, lfour);
      }
      const MaterialRec& ml = *lm;
      const bool leafBump = !(DR || LEAN) && ml.texid[1] != 0xFFFFFFFFu;        // the leaf's normal map bends the shading normal (:131-139)
      V3 sNorm = hitNorm;
      if (leafBump) sNorm = bumpNormal(S, ml, hitNorm, hitTang, uv);
      const V4 rands = rng_float4(gen);                                // GetRandomNumbersMats: drawn for every material type (integrator_pt_mat.cpp:147)
      if (lt == MAT_TYPE_GLTF) gltfSampleAndEval(ml, rands, vdir, sNorm, ld3(ml.colors[GLTF_COLOR_BASE]) * ltex3, lfour, ms);
      else if (!(DR || LEAN) && lt == MAT_TYPE_CONDUCTOR) {
        if (smax(ml.data[1], ml.data[0]) < 1e-3f) conductorSmoothSampleAndEval(ml, ml.data[2], ml.data[3], vdir, sNorm, ms);
        else                                      conductorRoughSampleAndEval(ml, ml.data[2], ml.data[3], rands, vdir, sNorm, ltex3, ms);
      }
      else if (!(DR || LEAN) && lt == MAT_TYPE_DIFFUSE) diffuseSampleAndEval(ml, ld3(ml.colors[0]) * ltex3, rands, vdir, sNorm, ms);
      else if (!(DR || LEAN) && lt == MAT_TYPE_PLASTIC) plasticSampleAndEval(ml, ld3(ml.colors[0]) * ltex3, rands, vdir, sNorm, ms, S.arrays1f, ml.datai[0]);
      else if (!(DR || LEAN) && lt == MAT_TYPE_GLASS) glassSampleAndEval(ml, rands, vdir, hitNorm, ms, misIor);      // the geometric normal (:182)
      else if (!(DR || LEAN) && lt == MAT_TYPE_DIELECTRIC) {
        dielectricSmoothSampleAndEval(ml, ml.data[1], misIor, rands, vdir, sNorm, ms);
        ms.flags |= (ml.spdid[0] < 0xFFFFFFFFu) ? RAY_FLAG_WAVES_DIVERGED : 0u;
        misIor = ms.ior;
      }
      else if (FILM && lt == MAT_TYPE_THIN_FILM) {                       // integrator_pt_mat.cpp:197-249: the geometric normal, always "diverged"
        const FilmArgs fa = filmArgs(S, ml, uv, 0.0f);
        if (smax(ml.data[1], ml.data[0]) < 1e-3f) filmSmoothSampleAndEval(ml, fa, misIor, rands, vdir, hitNorm, ms);
        else                                      filmRoughSampleAndEval(ml, fa, misIor, rands, vdir, hitNorm, ltex3, ms);
        ms.flags |= RAY_FLAG_WAVES_DIVERGED;
        misIor = ms.ior;
      }
      if (leafBump) {                                                  // the caller multiplies by the cosine to the geometric normal (:298-303)
        const float c1 = absf(dot(ms.dir, hitNorm)), c2 = absf(dot(ms.dir, sNorm));
        ms.val = ms.val * (c2 / smax(c1, 1e-10f));
      }
      const float invPdf = 1.0f / smax(ms.pdf, 1e-20f);
      const V3 bxdfVal = ms.val * invPdf;
      const float cosTheta = absf(dot(ms.dir, hitNorm));
      misPdf = (ms.flags & RAY_EVENT_S) != 0 ? -1.0f : ms.pdf;
      if (S.integratorType == INTEGRATOR_STUPID_PT) thr = thr * (cosTheta * bxdfVal);
      else {
        contrib = thr * shade;
        thr = thr * cosTheta * bxdfVal;
      }
      if (DR) {
        recS = shade; recdS = dshade * baseCol;                          // d shade / d texColor
        recA = cosTheta * bxdfVal; recdA = (cosTheta * invPdf) * (ms.dval * baseCol);
      }
      V3 hp = hitPos;
      if ((ms.flags & RAY_EVENT_T) != 0) hp = hp + hit.t * rdir * 2.0f * 1e-6f;
      rpos = offsRayPos(hp, hitNorm, ms.dir);
      rdir = ms.dir;
      uint nextFlags = ((flags & ~RAY_FLAG_FIRST_NON_SPEC) | ms.flags);
      if (S.renderLayer == FB_DIRECT && (flags & RAY_FLAG_HAS_NON_SPEC) != 0) nextFlags |= RAY_FLAG_IS_DEAD;
      else if ((flags & RAY_FLAG_HAS_NON_SPEC) == 0 && (nextFlags & RAY_FLAG_HAS_NON_SPEC) != 0) nextFlags |= RAY_FLAG_FIRST_NON_SPEC;
      flags = nextFlags;
      didBounce = true;
    }
  }
  return didBounce;
}

// ---- differentiable rendering: per-bounce adjoint record and the reverse sweep (shared by the megakernel and the wavefront shade pass) ----
// Record of bounce b for path slot `idx`: 17 dwords in five planes laid out [bounce][plane][slot] - four float4 planes and one dword plane,
// so a wave stores a record with four 1-KB global_store_dwordx4 and one 256-B store (it was 21 dword stores: every store stays counted in
// vmcnt until the memory side acknowledges it, and the next load the wave waits for waits for all of them):
//   q0 = (A.xyz, S.x)  q1 = (S.yz, T*dA.xy)  q2 = (T*dA.z, T*dS.xyz)  q3 = (first gradient element of tap 0, element step to the tap in +x,
//   element step to the tap in +y, fx)  e = fy
// The four bilinear taps are kept as tap 0's element index in a_dataGrad plus the two steps (negative where the footprint wraps) and the
// two fractions the weights are products of (bilinearTaps: w = {fx1 fy1, fx fy1, fx1 fy, fx fy}, recomputed with the same two products), so
// the sweep needs neither the texture descriptor nor eight more dwords. q3.x = 0xFFFFFFFF: no parameter texture at this bounce; bit 31 set
// otherwise: a one-channel texture (its rgb sum goes to one float).
static const int REC_PLANES = 5;
static const int REC_FIELDS = 4 * REC_PLANES;   // floats of buffer per slot and bounce (the last plane holds one dword per slot)
struct DrRec { V3 A, S, TdA, TdS; uint e0; int dx, dy; float fx, fy; };
HPT_DEV DrRec drEmptyRecord() { DrRec r; r.A = r.S = r.TdA = r.TdS = v3(0, 0, 0); r.e0 = 0xFFFFFFFFu; r.dx = r.dy = 0; r.fx = r.fy = 0.0f; return r; }
HPT_DEV DrRec drMakeRecord(V3 recA, V3 recS, V3 recdA, V3 recdS, V3 thrBefore, uint recTex, const Taps& t)
{
  DrRec r;
  r.A = recA; r.S = recS; r.TdA = recdA * thrBefore; r.TdS = recdS * thrBefore;       // T_b * dA_b/dtex, T_b * dS_b/dtex
  r.e0 = 0xFFFFFFFFu; r.dx = 0; r.dy = 0; r.fx = 0.0f; r.fy = 0.0f;
  if (recTex != 0xFFFFFFFFu) {
    const int ch = (int)t.ch;
    r.e0 = (t.base + (uint)t.off[0] * t.ch) | (t.ch == 4u ? 0u : 0x80000000u);
    r.dx = (t.off[1] - t.off[0]) * ch; r.dy = (t.off[2] - t.off[0]) * ch;
    r.fx = t.fx; r.fy = t.fy;
  }
  return r;
}
HPT_DEV void drStoreRecord(float* record, size_t s, size_t idx, uint bounce, const DrRec& r)
{
  float4* q = (float4*)record + ((size_t)bounce * REC_PLANES) * s + idx;
  q[0 * s] = make_float4(r.A.x, r.A.y, r.A.z, r.S.x);
  q[1 * s] = make_float4(r.S.y, r.S.z, r.TdA.x, r.TdA.y);
  if (r.e0 != 0xFFFFFFFFu) q[2 * s] = make_float4(r.TdA.z, r.TdS.x, r.TdS.y, r.TdS.z);   // (without a parameter texture the sweep reads neither derivative)
  q[3 * s] = make_float4(__uint_as_float(r.e0), __int_as_float(r.dx), __int_as_float(r.dy), r.fx);
  if (r.e0 != 0xFFFFFFFFu) *(float*)(q + 4 * s) = r.fy;
}
// the light sample of bounce b turned out occluded: its S term and derivative vanish
HPT_DEV void drClearShadowTerm(float* record, size_t s, size_t idx, uint bounce)
{
  float* q = (float*)((float4*)record + ((size_t)bounce * REC_PLANES) * s + idx);
  q[3] = 0.0f;                                                              // S.x
  q[4 * s + 0] = 0.0f; q[4 * s + 1] = 0.0f;                                 // S.yz
  q[8 * s + 1] = 0.0f; q[8 * s + 2] = 0.0f; q[8 * s + 3] = 0.0f;            // T*dS
}
// Hand-derived reverse sweep replacing __enzyme_autodiff (integrator_dr.cpp:1172-1183). With T_0 = 1, T_{b+1} = T_b A_b and
// C = sum_b T_b S_b + T_n tail:  dC/dtex_b = T_b dS_b + T_b dA_b R_{b+1},  R_b = S_b + A_b R_{b+1},  R_n = tail;  the loss gradient
// 2 (C - ref) dC/dtex_b is scattered to the four bilinear taps with float atomics.
//
// Called by ALL lanes of a wave, in wave-uniform control flow; `closing`: this lane has a path to close. Float atomics execute at the
// memory side, one request per 64-byte line an instruction touches, every instruction stays counted in vmcnt for ~3000 cycles with the chip
// busy and a wave stalls once 16..32 are outstanding (MI355X_MICROARCH.md, "Global float atomics"), so what counts is the NUMBER of atomic
// instructions and the lines each one touches. The closing lanes that have a gradient at a bounce stage their 12 values (4 taps x rgb) and
// tap 0's element index and steps in LDS, and ALL 64 lanes of the wave - also those whose paths go on - then walk the staged values in order:
// consecutive lanes take consecutive (tap, channel) elements of one source lane, so an instruction covers 64 values in a few 16-byte runs.
// (Round 2 ran this under the divergent "my path ended" branch: 13 of 64 lanes per atomic instruction on the test_228 class, 50
// instructions per wave-trip; 360 -> see profiles/r3_measurements.md.) `last`: the record of the closing lane's last bounce when it is
// still in registers (lastInRegs; it was never stored). `stage`: the wave's 1024-dword LDS area. Same sums, another order.
static const uint DR_STAGE_DWORDS = 16u * 64u;
HPT_DEV void drReverseSweep(const DevScene& S, const float* record, size_t s, size_t idx, const bool closing, const uint bounceIn, V3 Rn, const V3 diff,
                            float* grad, const bool skipNonFinite, uint* stage, const DrRec& last, const bool lastInRegs,
                            unsigned long long* statAtomics = nullptr, const uint ss = 64u, const bool defer = true)      // ss: dwords between the staging area's rows (64: an area of its own; 256: the wave's columns of a [16][256] array)
{
  const uint bounce = closing ? bounceIn : 0u;
  const uint lane = lane_id();
  // The scatter can be DEFERRED (the megakernels): every level appends its (taps, values) columns to the staging area and the atomics of all levels go out together at the
  // end (or when the 64 columns are full). Atomics, stores and loads share one in-order counter (vmcnt), so a record load issued behind a level's
  // atomics would wait for them to complete at the memory side (~1 ... 3 k cycles each time); this way the levels' loads only wait for each other:
  // PathTraceDR on the test_228 class 467 -> 477 (block-local), 415 -> 431 (megakernel). The wavefront shade kernel scatters level by level (defer = false):
  // with its many short sweeps in flight the bunched atomics stall the issue instead (256 -> 249, 1 M triangles 229 -> 224 when deferred).
  uint cnt = 0;                                                             // columns staged so far (wave-uniform)
  auto flush = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (statAtomics && lane == 0u) *statAtomics += (12u * cnt + 63u) / 64u;       // (instrumented build: atomic wave-instructions of this sweep)
    for (uint p = lane; p < 12u * cnt; p += 64u) {
      const uint src = (p * 0xAAABu) >> 19;                               // p / 12 (p < 768)
      const uint e = p - 12u * src, tap = (e * 11u) >> 5, ch = e - 3u * tap;   // e / 3, e % 3 (e < 12)
      const uint ix0 = stage[12 * ss + src];
      const uint ix = (ix0 & 0x7FFFFFFFu) + ((tap & 1u) ? stage[13 * ss + src] : 0u) + ((tap & 2u) ? stage[14 * ss + src] : 0u);   // (two's complement: negative steps wrap back)
      const float val = ((const float*)stage)[e * ss + src];
#ifndef HPT_DR_NO_ATOMICS    // diagnostic build only: how much of PathTraceDR is the gradient scatter?
      if ((ix0 & 0x80000000u) == 0u) atomicAdd(grad + (size_t)ix + ch, val);
      else if (ch == 0u) atomicAdd(grad + (size_t)ix, val);
#else
      if (ix == 0x7FFFFFFFu) grad[0] = val;
#endif
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    cnt = 0;
  };
  for (int b = (int)S.traceDepth - 1; b >= 0; b--) {                      // wave-uniform trip count; a lane joins at its own last bounce
    const bool mine = (uint)b < bounce;
    if (__ballot(mine) == 0ull) continue;
    const bool inRegs = mine && lastInRegs && (uint)b + 1u == bounce;
    const bool load = mine && !inRegs;
    V3 A = v3(0, 0, 0), Sb = v3(0, 0, 0), TdA = v3(0, 0, 0), TdS = v3(0, 0, 0);
    uint e0 = 0xFFFFFFFFu; int dx = 0, dy = 0; float fx = 0.0f, fy = 0.0f;
    const float4* q = (const float4*)record + ((size_t)b * REC_PLANES) * s + idx;
    if (load) {
      const float4 q0 = q[0 * s], q1 = q[1 * s], q3 = q[3 * s];
      A = v3(q0.x, q0.y, q0.z); Sb = v3(q0.w, q1.x, q1.y); TdA.x = q1.z; TdA.y = q1.w;
      e0 = __float_as_uint(q3.x); dx = __float_as_int(q3.y); dy = __float_as_int(q3.z); fx = q3.w;
      if (e0 != 0xFFFFFFFFu) { const float4 q2 = q[2 * s]; TdA.z = q2.x; TdS = v3(q2.y, q2.z, q2.w); fy = *(const float*)(q + 4 * s); }
    } else if (inRegs) { A = last.A; Sb = last.S; TdA = last.TdA; TdS = last.TdS; e0 = last.e0; dx = last.dx; dy = last.dy; fx = last.fx; fy = last.fy; }
    const bool has = mine && e0 != 0xFFFFFFFFu;
    const unsigned long long hm = __ballot(has);
    if (hm != 0ull) {
      const uint n = (uint)__popcll(hm);
      if (cnt + n > 64u) flush();
      const uint col = cnt + mbcnt64(hm);
      if (has) {
        const V3 dC = TdS + TdA * Rn;
        V3 g = v3(2.0f * diff.x * dC.x, 2.0f * diff.y * dC.y, 2.0f * diff.z * dC.z);
        if (skipNonFinite && !__builtin_isfinite(g.x + g.y + g.z)) g = v3(0, 0, 0);   // (only with dr_skip_nonfinite: the reference scatters whatever comes out)
        const bool four = (e0 & 0x80000000u) == 0u;
        const float fx1 = 1.0f - fx, fy1 = 1.0f - fy;
        const float w[4] = { fx1 * fy1, fx * fy1, fx1 * fy, fx * fy };      // bilinearTaps' weights, the same products
        float* sv = (float*)stage;
        for (int k = 0; k < 4; k++) {
          sv[(3 * k + 0) * ss + col] = four ? g.x * w[k] : (g.x + g.y + g.z) * w[k];
          sv[(3 * k + 1) * ss + col] = g.y * w[k];
          sv[(3 * k + 2) * ss + col] = g.z * w[k];
        }
        stage[12 * ss + col] = e0; stage[13 * ss + col] = (uint)dx; stage[14 * ss + col] = (uint)dy;
      }
      cnt += n;
      if (!defer) flush();
    }
    if (mine) Rn = Sb + A * Rn;
  }
  if (cnt != 0u) flush();
}

} // namespace hpt
