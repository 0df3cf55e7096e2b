// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).  PARITY UNPINNED.
//
// RNG, sampling helpers, BSDFs and light samplers of HydraCore3, restated for the CPU checker.
// Every function names the reference lines it follows; arithmetic is kept in the reference's order so that
// the result is what the reference's CPU build (USE_VULKAN=OFF) computes, up to libm.
#pragma once
#include "orc_scene.h"

namespace orc {

// ---- include/cglobals.h:9-16 ----------------------------------------------------------------------------------
static const uint RAY_FLAG_IS_DEAD        = 0x80000000u;
static const uint RAY_FLAG_OUT_OF_SCENE   = 0x40000000u;
static const uint RAY_FLAG_HIT_LIGHT      = 0x20000000u;
static const uint RAY_FLAG_HAS_NON_SPEC   = 0x10000000u;
static const uint RAY_FLAG_HAS_INV_NORMAL = 0x08000000u;
static const uint RAY_FLAG_WAVES_DIVERGED = 0x04000000u;
static const uint RAY_FLAG_PRIME_RAY_MISS = 0x02000000u;
static const uint RAY_FLAG_FIRST_NON_SPEC = 0x01000000u;

// ---- include/cmaterial.h:26-56 --------------------------------------------------------------------------------
static const uint GLTF_COMPONENT_METAL = 4, GLTF_COMPONENT_ORENNAYAR = 16, FLAG_FOUR_TEXTURES = 256, FLAG_PACK_FOUR_PARAMS_IN_TEXTURE = 512;
static const uint FLAG_NMAP_INVERT_X = 32, FLAG_NMAP_INVERT_Y = 64, FLAG_NMAP_SWAP_XY = 128;                 // include/cmaterial.h:31-33
static const uint MAT_TYPE_BLEND = 6, BLEND_WEIGHT = 0, BLEND_STACK_SIZE = 4;   // include/cmaterial.h:43,155; integrator_pt.h:599
static const uint MAT_TYPE_GLASS = 2;   // include/cmaterial.h:39; slots :85-92
static const uint GLASS_COLOR_REFLECT = 0, GLASS_COLOR_TRANSP = 1, GLASS_FLOAT_IOR = 2;
static const uint MAT_TYPE_GLTF = 1, MAT_TYPE_CONDUCTOR = 3, MAT_TYPE_DIFFUSE = 4, MAT_TYPE_DIELECTRIC = 7, MAT_TYPE_LIGHT_SOURCE = 0xEFFFFFFFu;
static const uint RAY_EVENT_S = 1, RAY_EVENT_T = 8;
// include/cmaterial.h:67-147
static const int GLTF_COLOR_BASE = 0, GLTF_COLOR_COAT = 1, GLTF_COLOR_METAL = 2;
static const int GLTF_FLOAT_MI_FDR_INT = 0, GLTF_FLOAT_ALPHA = 3, GLTF_FLOAT_GLOSINESS = 4, GLTF_FLOAT_IOR = 5, GLTF_FLOAT_REFL_COAT = 7;
static const int DIELECTRIC_ETA_EXT = 0, DIELECTRIC_ETA_INT = 1;
static const int EMISSION_COLOR = 0;
static const int CONDUCTOR_COLOR = 0, CONDUCTOR_ROUGH_U = 0, CONDUCTOR_ROUGH_V = 1, CONDUCTOR_ETA = 2, CONDUCTOR_K = 3;
static const int DIFFUSE_COLOR = 0, DIFFUSE_ROUGHNESS = 0;
// include/clight.h:5-17
static const uint LIGHT_GEOM_RECT = 1, LIGHT_GEOM_DISC = 2, LIGHT_GEOM_SPHERE = 3, LIGHT_GEOM_DIRECT = 4, LIGHT_GEOM_POINT = 5, LIGHT_GEOM_ENV = 6;
static const uint LIGHT_DIST_LAMBERT = 0, LIGHT_DIST_OMNI = 1, LIGHT_DIST_SPOT = 2;
static const uint LIGHT_FLAG_POINT_AREA = 1, LIGHT_FLAG_PROJECTIVE = 2;
// integrator_pt.h:330-332, 406-408
static const uint INTEGRATOR_STUPID_PT = 0, INTEGRATOR_SHADOW_PT = 1, INTEGRATOR_MIS_PT = 2;
static const uint FB_COLOR = 0, FB_DIRECT = 1, FB_INDIRECT = 2;

static const float GEPSILON = 1e-5f, DEPSILON = 1e-20f;

// ---- include/crandom.h ----------------------------------------------------------------------------------------
struct RandomGen { uint sx, sy; };

static inline uint NextState(RandomGen* g)                           // crandom.h:17-23
{
  const uint x = g->sx * 17u + g->sy * 13123u;
  g->sx = (x << 13) ^ x;
  g->sy ^= (x << 7);
  return x;
}
static inline RandomGen RandomGenInit(int seed)                      // crandom.h:25-36 (int arithmetic wraps; done in uint here)
{
  const uint s = (uint)seed;
  RandomGen g;
  g.sx = (s * (s * s * 15731u + 74323u) + 871483u);
  g.sy = (s * (s * s * 13734u + 37828u) + 234234u);
  for (int i = 0; i < (seed % 7); i++) NextState(&g);
  return g;
}
static inline f4 rndFloat4(RandomGen* g)                             // crandom.h:43-55
{
  const uint x = NextState(g);
  const uint x1 = (x * (x * x * 15731u + 74323u) + 871483u);
  const uint y1 = (x * (x * x * 13734u + 37828u) + 234234u);
  const uint z1 = (x * (x * x * 11687u + 26461u) + 137589u);
  const uint w1 = (x * (x * x * 15707u + 789221u) + 1376312589u);
  const float scale = (1.0f / 4294967296.0f);
  return mk4((float)x1, (float)y1, (float)z1, (float)w1) * scale;
}
static inline float rndFloat1(RandomGen* g)                          // crandom.h:69-75
{
  const uint x = NextState(g);
  const uint tmp = (x * (x * x * 15731u + 74323u) + 871483u);
  return ((float)tmp) * (1.0f / 4294967296.0f);
}

// ---- include/cglobals.h helpers -------------------------------------------------------------------------------
struct MisData { float matSamplePdf, cosTheta, ior, dummy; };       // cglobals.h:292-311
static inline MisData makeInitialMisData() { MisData d = {1.0f, 1.0f, 1.0f, 0.0f}; return d; }

static inline f3 EyeRayDirNormalized(float x, float y, const m4& projInv)   // cglobals.h:49-55
{
  f4 pos = mk4(2.0f * x - 1.0f, 2.0f * y - 1.0f, 0.0f, 1.0f);
  pos = mul(projInv, pos);
  pos = pos / pos.w;
  return normalize(xyz(pos));
}

static inline void CoordinateSystemV2(f3 n, f3* s, f3* t)            // cglobals.h:120-132
{
  const float sign = n.z >= 0 ? 1.0f : -1.0f;
  const float a = -(1.0f / (sign + n.z));
  const float b = n.x * n.y * a;
  const float tmp = (n.z >= 0 ? n.x * n.x * a : -n.x * n.x * a);
  *s = mk3(tmp + 1.0f, n.z >= 0 ? b : -b, n.z >= 0 ? -n.x : n.x);
  *t = mk3(b, n.y * n.y * a + sign, -n.y);
}

static inline f3 MapSampleToCosineDistribution(float r1, float r2, f3 direction, f3 hit_norm, float power)   // cglobals.h:143-181
{
  if (power >= 1e6f) return direction;
  const float sin_phi = std::sin(kTWOPI * r1);
  const float cos_phi = std::cos(kTWOPI * r1);
  const float cos_theta = std::pow(1.0f - r2, 1.0f / (power + 1.0f));
  const float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
  const f3 deviation = mk3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
  f3 ny = direction, nx, nz;
  CoordinateSystemV2(ny, &nx, &nz);
  { const f3 temp = ny; ny = nz; nz = temp; }
  f3 res = nx * deviation.x + ny * deviation.y + nz * deviation.z;
  const float invSign = dot(direction, hit_norm) > 0.0f ? 1.0f : -1.0f;
  if (invSign * dot(res, hit_norm) < 0.0f)
    res = (-1.0f) * nx * deviation.x + ny * deviation.y - nz * deviation.z;
  return res;
}

static inline f2 MapSamplesToDisc(f2 xy)                             // cglobals.h:188-231
{
  const float x = xy.x, y = xy.y;
  float r = 0, phi = 0;
  if (x > y && x > -y)  { r = x;  phi = 0.25f * 3.141592654f * (y / x); }
  if (x < y && x > -y)  { r = y;  phi = 0.25f * 3.141592654f * (2.0f - x / y); }
  if (x < y && x < -y)  { r = -x; phi = 0.25f * 3.141592654f * (4.0f + y / x); }
  if (x > y && x < -y)  { r = -y; phi = 0.25f * 3.141592654f * (6 - x / y); }
  const float sin_phi = std::sin(phi), cos_phi = std::cos(phi);
  return mk2(r * sin_phi, r * cos_phi);
}

static inline float epsilonOfPos(f3 p)                               // cglobals.h:233
{
  return std::max(std::max(std::abs(p.x), std::max(std::abs(p.y), std::abs(p.z))), 2.0f * GEPSILON) * GEPSILON;
}
static inline f3 OffsRayPos(f3 hitPos, f3 surfaceNorm, f3 sampleDir)  // cglobals.h:242-247
{
  const float signOfNormal2 = dot(sampleDir, surfaceNorm) < 0.0f ? -1.0f : 1.0f;
  const float offsetEps = epsilonOfPos(hitPos);
  return hitPos + signOfNormal2 * offsetEps * surfaceNorm;
}
static inline void transform_ray3f(const m4& worldViewInv, f3* ray_pos, f3* ray_dir)   // cglobals.h:254-263
{
  const f3 pos = mul4x3(worldViewInv, *ray_pos);
  const f3 pos2 = mul4x3(worldViewInv, (*ray_pos) + 100.0f * (*ray_dir));
  const f3 diff = pos2 - pos;
  *ray_pos = pos;
  *ray_dir = normalize(diff);
}
static inline float PdfAtoW(float aPdfA, float aDist, float aCosThere) { return (aPdfA * aDist * aDist) / std::max(aCosThere, 1e-30f); }  // cglobals.h:265-268
static inline float maxcomp(f3 v) { return std::max(v.x, std::max(v.y, v.z)); }                                                       // cglobals.h:275
static inline float misHeuristicPower1(float p) { return std::isfinite(p) ? std::abs(p) : 0.0f; }                                   // cglobals.h:277
static inline float misWeightHeuristic(float a, float b)                                                                              // cglobals.h:278-282
{
  const float w = misHeuristicPower1(a) / std::max(misHeuristicPower1(a) + misHeuristicPower1(b), 1e-30f);
  return std::isfinite(w) ? w : 0.0f;
}
static inline f2 mulRows2x4(f4 row0, f4 row1, f2 v)                  // cglobals.h:315-321
{
  return mk2(row0.x * v.x + row0.y * v.y + row0.w, row1.x * v.x + row1.y * v.y + row1.w);
}
static inline f2 sphereMapTo2DTexCoord(f3 ray_dir, float* pSinTheta) // cglobals.h:335-362
{
  const float x = ray_dir.z, y = ray_dir.x, z = -ray_dir.y;
  float theta = std::acos(z);
  float phi = std::atan2(y, x);
  if (phi < 0.0f) phi += 2.0f * kPI;
  const float texX = clampf(phi * 0.5f * kINV_PI, 0.0f, 1.0f);
  const float texY = clampf(theta * kINV_PI, 0.0f, 1.0f);
  *pSinTheta = std::sqrt(1.0f - ray_dir.y * ray_dir.y);
  return mk2(texX, texY);
}

static inline f3 texCoord2DToSphereMap(f2 a_texCoord, float* pSinTheta) // cglobals.h:360-373, the reverse of sphereMapTo2DTexCoord
{
  const float phi = a_texCoord.x * 2.f * kPI, theta = a_texCoord.y * kPI;
  const float sinTheta = std::sin(theta);
  const float x = sinTheta * std::cos(phi), y = sinTheta * std::sin(phi), z = std::cos(theta);
  *pSinTheta = sinTheta;
  return mk3(y, -z, x);
}
// ---- include/clight.h:128-218: the piecewise-constant 2-D distribution of a sampled environment map -----------------
static inline int SelectIndexPropToOpt(float a_r, const float* a_accum, int a_offset, int N, float* pPDF)
{
  int leftBound = 0, rightBound = N - 2, counter = 0, currPos = -1;
  const int maxStep = 50;
  const float x = a_r * a_accum[a_offset + N - 1];
  while (rightBound - leftBound > 1 && counter < maxStep) {
    const int currSize = rightBound + leftBound;
    const int currPos1 = (currSize % 2 == 0) ? (currSize + 1) / 2 : (currSize + 0) / 2;
    const float a = a_accum[a_offset + currPos1 + 0], b = a_accum[a_offset + currPos1 + 1];
    if (a < x && x <= b) { currPos = currPos1; break; }
    else if (x <= a) rightBound = currPos1;
    else if (x > b) leftBound = currPos1;
    counter++;
  }
  if (currPos < 0) {
    const float a1 = a_accum[a_offset + leftBound + 0], b1 = a_accum[a_offset + leftBound + 1];
    const float a2 = a_accum[a_offset + rightBound + 0], b2 = a_accum[a_offset + rightBound + 1];
    if (a1 < x && x <= b1) currPos = leftBound;
    if (a2 < x && x <= b2) currPos = rightBound;
  }
  if (x == 0.0f) currPos = 0;
  else if (currPos < 0) currPos = (rightBound + leftBound + 1) / 2;
  *pPDF = (a_accum[a_offset + currPos + 1] - a_accum[a_offset + currPos]) / a_accum[a_offset + N - 1];
  return currPos;
}
static inline float evalMap2DPdf(f2 texCoordT, const float* intervals, int a_inOffs, int sizeX, int sizeY)
{
  const float fw = (float)sizeX, fh = (float)sizeY;
  if (texCoordT.x < 0.0f || texCoordT.x > 1.0f) texCoordT.x -= (float)((int)(texCoordT.x));
  if (texCoordT.y < 0.0f || texCoordT.x > 1.0f) texCoordT.y -= (float)((int)(texCoordT.y));     // (sic: .x in the second test, clight.h:199)
  int pixelX = (int)(fw * texCoordT.x - 0.5f), pixelY = (int)(fh * texCoordT.y - 0.5f);
  if (pixelX >= sizeX) pixelX = sizeX - 1;
  if (pixelY >= sizeY) pixelY = sizeY - 1;
  if (pixelX < 0) pixelX += sizeX;
  if (pixelY < 0) pixelY += sizeY;
  const int pixelOffset = pixelY * sizeX + pixelX, maxSize = sizeX * sizeY;
  const int offset0 = (pixelOffset + 0 < maxSize + 0) ? pixelOffset + 0 : maxSize - 1;
  const int offset1 = (pixelOffset + 1 < maxSize + 1) ? pixelOffset + 1 : maxSize;
  const float i0 = intervals[a_inOffs + offset0], i1 = intervals[a_inOffs + offset1];
  return (i1 - i0) * (fw * fh) / intervals[a_inOffs + sizeX * sizeY];
}

// ---- include/cmaterial.h --------------------------------------------------------------------------------------
struct BsdfSample { f4 val; f3 dir; float pdf; uint flags; float ior; };   // cmaterial.h:9-16
struct BsdfEval { f4 val; float pdf; };                                    // cmaterial.h:18-22

static inline float safe_sqrt(float v) { return std::sqrt(std::max(v, 0.0f)); }                      // :206-209
static inline f3 lambertSample(f2 rands, f3 v, f3 n) { return MapSampleToCosineDistribution(rands.x, rands.y, n, n, 1.0f); }  // :215-218
static inline float lambertEvalPDF(f3 l, f3 v, f3 n) { return std::abs(dot(l, n)) * kINV_PI; }        // :220-223
static inline float lambertEvalBSDF(f3 l, f3 v, f3 n) { return kINV_PI; }                             // :225-228

static inline float cosPhiPBRT(f3 w, float sintheta) { return sintheta == 0.0f ? 1.0f : clampf(w.x / sintheta, -1.0f, 1.0f); }   // :234-240
static inline float sinPhiPBRT(f3 w, float sintheta) { return sintheta == 0.0f ? 0.0f : clampf(w.y / sintheta, -1.0f, 1.0f); }   // :242-248

static inline float orennayarFunc(f3 a_l, f3 a_v, f3 a_n, float a_roughness)    // :254-306
{
  const float cosTheta_wi = dot(a_l, a_n), cosTheta_wo = dot(a_v, a_n);
  const float sinTheta_wi = safe_sqrt(1.0f - cosTheta_wi * cosTheta_wi);
  const float sinTheta_wo = safe_sqrt(1.0f - cosTheta_wo * cosTheta_wo);
  const float sigma = a_roughness * kPI * 0.5f;
  const float sigma2 = sigma * sigma;
  const float A = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
  const float B = 0.45f * sigma2 / (sigma2 + 0.09f);
  f3 nx, ny, nz = a_n;
  CoordinateSystemV2(nz, &nx, &ny);
  float maxcos = 0.0f;
  if (sinTheta_wi > 1e-4f && sinTheta_wo > 1e-4f) {
    const f3 wo = mk3(-dot(a_v, nx), -dot(a_v, ny), -dot(a_v, nz));
    const f3 wi = mk3(-dot(a_l, nx), -dot(a_l, ny), -dot(a_l, nz));
    const float sinphii = sinPhiPBRT(wi, sinTheta_wi), cosphii = cosPhiPBRT(wi, sinTheta_wi);
    const float sinphio = sinPhiPBRT(wo, sinTheta_wo), cosphio = cosPhiPBRT(wo, sinTheta_wo);
    const float dcos = cosphii * cosphio + sinphii * sinphio;
    maxcos = std::max(0.0f, dcos);
  }
  float sinalpha = 0.0f, tanbeta = 0.0f;
  if (std::abs(cosTheta_wi) > std::abs(cosTheta_wo)) { sinalpha = sinTheta_wo; tanbeta = sinTheta_wi / std::max(std::abs(cosTheta_wi), DEPSILON); }
  else                                               { sinalpha = sinTheta_wi; tanbeta = sinTheta_wo / std::max(std::abs(cosTheta_wo), DEPSILON); }
  return (A + B * maxcos * sinalpha * tanbeta);
}

static inline float GGX_Distribution(float cosThetaNH, float alpha)  // :322-328
{
  const float alpha2 = alpha * alpha;
  const float NH_sqr = clampf(cosThetaNH * cosThetaNH, 0.0f, 1.0f);
  const float den = NH_sqr * alpha2 + (1.0f - NH_sqr);
  return alpha2 / std::max((float)(kPI) * den * den, 1e-6f);
}
static inline float GGX_GeomShadMask(float cosThetaN, float alpha)   // :330-343
{
  const float cosTheta_sqr = clampf(cosThetaN * cosThetaN, 0.0f, 1.0f);
  const float tan2 = (1.0f - cosTheta_sqr) / std::max(cosTheta_sqr, 1e-6f);
  return 2.0f / (1.0f + safe_sqrt(1.0f + alpha * alpha * tan2));
}
static inline f3 ggxSample(f2 rands, f3 v, f3 n, float roughness)    // :347-362
{
  const float roughSqr = roughness * roughness;
  f3 nx, ny, nz = n;
  CoordinateSystemV2(nz, &nx, &ny);
  const f3 wo = mk3(dot(v, nx), dot(v, ny), dot(v, nz));
  const float phi = rands.x * kTWOPI;
  const float cosTheta = clampf(safe_sqrt((1.0f - rands.y) / (1.0f + roughSqr * roughSqr * rands.y - rands.y)), 0.0f, 1.0f);
  const float sinTheta = safe_sqrt(1.0f - cosTheta * cosTheta);
  const f3 wh = mk3(sinTheta * std::cos(phi), sinTheta * std::sin(phi), cosTheta);
  const f3 wi = 2.0f * dot(wo, wh) * wh - wo;
  return normalize(wi.x * nx + wi.y * ny + wi.z * nz);
}
static inline float ggxEvalPDF(f3 l, f3 v, f3 n, float roughness)    // :364-378
{
  const float dotNV = dot(n, v), dotNL = dot(n, l);
  if (dotNV < 1e-6f || dotNL < 1e-6f) return 1.0f;
  const float roughSqr = roughness * roughness;
  const f3 h = normalize(v + l);
  const float dotNH = dot(n, h), dotHV = dot(h, v);
  const float D = GGX_Distribution(dotNH, roughSqr);
  return D * dotNH / (4.0f * std::max(dotHV, 1e-6f));
}
static inline float ggxEvalBSDF(f3 l, f3 v, f3 n, float roughness)   // :380-397
{
  if (std::abs(dot(l, n)) < 1e-5f) return 0.0f;
  const float dotNV = dot(n, v), dotNL = dot(n, l);
  if (dotNV < 1e-6f || dotNL < 1e-6f) return 0.0f;
  const float roughSqr = roughness * roughness;
  const f3 h = normalize(v + l);
  const float dotNH = dot(n, h);
  const float D = GGX_Distribution(dotNH, roughSqr);
  const float G = GGX_GeomShadMask(dotNV, roughSqr) * GGX_GeomShadMask(dotNL, roughSqr);
  return (D * G / std::max(4.0f * dotNV * dotNL, 1e-6f));
}

// Trowbridge-Reitz (cmaterial.h:405-530)
static inline float Cos2Theta(f3 w) { return w.z * w.z; }
static inline float AbsCosTheta(f3 w) { return std::abs(w.z); }
static inline float Sin2Theta(f3 w) { return std::max(0.0f, 1.0f - Cos2Theta(w)); }
static inline float SinTheta(f3 w) { return safe_sqrt(Sin2Theta(w)); }
static inline float Tan2Theta(f3 w) { return Sin2Theta(w) / Cos2Theta(w); }
static inline float CosPhi(f3 w) { const float s = SinTheta(w); return (s == 0) ? 1 : clampf(w.x / s, -1.0f, 1.0f); }
static inline float SinPhi(f3 w) { const float s = SinTheta(w); return (s == 0) ? 0 : clampf(w.y / s, -1.0f, 1.0f); }
static inline f3 FaceForward(f3 v, f3 n2) { return (dot(v, n2) < 0.f) ? (-1.0f) * v : v; }
static inline f2 SampleUniformDiskPolar(f2 u) { const float r = safe_sqrt(u.x); const float th = kTWOPI * u.y; return mk2(r * std::cos(th), r * std::sin(th)); }
static inline float trD(f3 wm, f2 alpha)                             // :460-470
{
  const float tan2Theta = Tan2Theta(wm);
  if (std::isinf(tan2Theta)) return 0;
  const float cos4Theta = Cos2Theta(wm) * Cos2Theta(wm);
  if (cos4Theta < 1e-16f) return 0;
  const float e = tan2Theta * ((CosPhi(wm) / alpha.x) * (CosPhi(wm) / alpha.x) + (SinPhi(wm) / alpha.y) * (SinPhi(wm) / alpha.y));
  return 1.0f / (kPI * alpha.x * alpha.y * cos4Theta * (1 + e) * (1 + e));
}
static inline bool trEffectivelySmooth(f2 alpha) { return std::max(alpha.x, alpha.y) < 1e-3f; }     // :472-475
static inline float trLambda(f3 w, f2 alpha)                         // :477-484
{
  const float tan2Theta = Tan2Theta(w);
  if (std::isinf(tan2Theta)) return 0;
  const float alpha2 = (CosPhi(w) * alpha.x) * (CosPhi(w) * alpha.x) + (SinPhi(w) * alpha.y) * (SinPhi(w) * alpha.y);
  return (safe_sqrt(1.0f + alpha2 * tan2Theta) - 1.0f) / 2.0f;
}
static inline float trG1(f3 w, f2 alpha) { return 1.0f / (1.0f + trLambda(w, alpha)); }
static inline float trG(f3 wo, f3 wi, f2 alpha) { return 1.0f / (1.0f + trLambda(wo, alpha) + trLambda(wi, alpha)); }
static inline float trD2(f3 w, f3 wm, f2 alpha) { return trG1(w, alpha) / AbsCosTheta(w) * trD(wm, alpha) * std::abs(dot(w, wm)); }   // :496-499
static inline float trPDF(f3 w, f3 wm, f2 alpha) { return trD2(w, wm, alpha); }
static inline f3 trSample(f3 wo, f2 rands, f2 alpha)                 // :506-530
{
  f3 wh = normalize(mk3(alpha.x * wo.x, alpha.y * wo.y, wo.z));
  if (wh.z < 0) wh = (-1.0f) * wh;
  const f3 T1 = (wh.z < 0.99999f) ? normalize(cross(mk3(0, 0, 1), wh)) : mk3(1, 0, 0);
  const f3 T2 = cross(wh, T1);
  f2 p = SampleUniformDiskPolar(rands);
  const float h = safe_sqrt(1 - p.x * p.x);
  p.y = lerpf(h, p.y, (1 + wh.z) / 2);
  const float pz = safe_sqrt(1.0f - dot2(p, p));
  const f3 nh = p.x * T1 + p.y * T2 + pz * wh;
  return normalize(mk3(alpha.x * nh.x, alpha.y * nh.y, std::max(1e-6f, nh.z)));
}

static inline float FrDielectricPBRT(float cosThetaI, float etaI, float etaT)   // :536-561
{
  cosThetaI = clampf(cosThetaI, -1.0f, 1.0f);
  const bool entering = cosThetaI > 0.0f;
  if (!entering) { const float tmp = etaI; etaI = etaT; etaT = tmp; cosThetaI = std::abs(cosThetaI); }
  const float sinThetaI = safe_sqrt(1.0f - cosThetaI * cosThetaI);
  const float sinThetaT = etaI / etaT * sinThetaI;
  if (sinThetaT >= 1.0f) return 1.0f;
  const float cosThetaT = safe_sqrt(1.0f - sinThetaT * sinThetaT);
  const float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
  const float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
  return 0.5f * (Rparl * Rparl + Rperp * Rperp);
}
static inline f4 FrDielectricDetailedV2(float cos_theta_i, float eta)          // :646-683
{
  cos_theta_i = clampf(cos_theta_i, -1.0f, 1.0f);
  float eta_it = eta, eta_ti = 1.f / eta;
  if (cos_theta_i < 0.0f) { eta_it = eta_ti; eta_ti = eta; }
  const float cos_theta_t_sqr = -1.f * (-1.f * cos_theta_i * cos_theta_i + 1.f) * eta_ti * eta_ti + 1.f;
  const float cos_theta_i_abs = std::abs(cos_theta_i);
  const float cos_theta_t_abs = safe_sqrt(cos_theta_t_sqr);
  float r = 0.0f;
  if ((eta == 1.f) || (cos_theta_i_abs == 0.f)) r = (eta == 1.f) ? 0.f : 1.f;
  else {
    const float a_s = (-1.f * eta_it * cos_theta_t_abs + cos_theta_i_abs) / (eta_it * cos_theta_t_abs + cos_theta_i_abs);
    const float a_p = (-1.f * eta_it * cos_theta_i_abs + cos_theta_t_abs) / (eta_it * cos_theta_i_abs + cos_theta_t_abs);
    r = 0.5f * (a_s * a_s + a_p * a_p);
  }
  const float cos_theta_t = cos_theta_i >= 0 ? -cos_theta_t_abs : cos_theta_t_abs;
  return mk4(r, cos_theta_t, eta_it, eta_ti);
}
static inline float FrComplexConductor(float cosThetaI, cplx eta)    // :685-694
{
  const float sinThetaI = 1.0f - cosThetaI * cosThetaI;
  const cplx sinThetaT = rdiv(sinThetaI, eta * eta);
  const cplx cosThetaT = csqrt_(rsub(1.0f, sinThetaT));
  const cplx r_parl = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
  const cplx r_perp = rsub(cosThetaI, eta * cosThetaT) / radd(cosThetaI, eta * cosThetaT);
  return (cnorm(r_parl) + cnorm(r_perp)) / 2.0f;
}
static inline float fresnelSlick(float VdotH) { const float tmp = 1.0f - std::abs(VdotH); return (tmp * tmp) * (tmp * tmp) * tmp; }   // :705-709
static inline f4 hydraFresnelCond(f4 f0, float VdotH, float ior, float roughness)   // :711-717
{
  if (ior == 0.0f) return f0;
  return f0 + (splat4(1.0f) - f0) * fresnelSlick(VdotH);
}
static inline f3 refract_(f3 wi, float cos_theta_t, float eta_ti) { return mk3(-eta_ti * wi.x, -eta_ti * wi.y, cos_theta_t); }   // :917-920

// ---- include/cmat_gltf.h --------------------------------------------------------------------------------------
// The base colour is generic (`C`) so that the DR replay can push a dual number through the very same code.
template <class C>
struct BsdfSampleT { C val; f3 dir; float pdf; uint flags; float ior; };
template <class C>
struct BsdfEvalT { C val; float pdf; };

template <class C>
static inline void gltfSampleAndEval(const Material& m, f4 rands, f3 v, f3 n, f2 tc, C baseColor, f4 fourParams, BsdfSampleT<C>* pRes)   // cmat_gltf.h:6-90
{
  const uint cflags = m.cflags;
  const C metalCol = baseColor * m.colors[GLTF_COLOR_METAL];
  const f4 coatCol = m.colors[GLTF_COLOR_COAT];
  const float roughness = clampf(1.0f - m.data[GLTF_FLOAT_GLOSINESS] * fourParams.x, 0.0f, 1.0f);
  float metalness = m.data[GLTF_FLOAT_ALPHA] * fourParams.y;
  const float coatValue = m.data[GLTF_FLOAT_REFL_COAT] * fourParams.z;
  const float fresnelIOR = m.data[GLTF_FLOAT_IOR];
  if (cflags == GLTF_COMPONENT_METAL) metalness = 1.0f;

  f3 ggxDir; float ggxPdf, ggxVal;
  if (roughness == 0.0f) {
    const f3 pefReflDir = reflect((-1.0f) * v, n);
    const float cosThetaOut = dot(pefReflDir, n);
    ggxDir = pefReflDir;
    ggxVal = (cosThetaOut <= 1e-6f) ? 0.0f : (1.0f / std::max(cosThetaOut, 1e-6f));
    ggxPdf = 1.0f;
  } else {
    ggxDir = ggxSample(mk2(rands.x, rands.y), v, n, roughness);
    ggxPdf = ggxEvalPDF(ggxDir, v, n, roughness);
    ggxVal = ggxEvalBSDF(ggxDir, v, n, roughness);
  }
  const f3 lambertDir = lambertSample(mk2(rands.x, rands.y), v, n);
  const float lambertPdf = lambertEvalPDF(lambertDir, v, n);
  const float lambertVal = lambertEvalBSDF(lambertDir, v, n);

  float pdfSelect = 1.0f;
  if (rands.z < metalness) {
    pdfSelect *= metalness;
    const float VdotH = dot(v, normalize(v + ggxDir));
    pRes->dir = ggxDir;
    // hydraFresnelCond (cmaterial.h:711-717) on a generic colour
    C fr = metalCol;
    if (fresnelIOR != 0.0f) fr = metalCol + (C(splat4(1.0f)) - metalCol) * fresnelSlick(VdotH);
    pRes->val = fr * (ggxVal * metalness);
    pRes->pdf = ggxPdf;
    pRes->flags = (roughness == 0.0f) ? RAY_EVENT_S : RAY_FLAG_HAS_NON_SPEC;
  } else {
    pdfSelect *= 1.0f - metalness;
    const float f_i = FrDielectricPBRT(std::abs(dot(v, n)), 1.0f, fresnelIOR);
    const float prob_specular = 0.5f * coatValue;
    const float prob_diffuse = 1.0f - prob_specular;
    if (rands.w < prob_specular) {
      pdfSelect *= prob_specular;
      pRes->dir = ggxDir;
      pRes->val = C(ggxVal * coatCol * (1.0f - metalness) * f_i * coatValue);
      pRes->pdf = ggxPdf;
      pRes->flags = (roughness == 0.0f) ? RAY_EVENT_S : RAY_FLAG_HAS_NON_SPEC;
    } else {
      pdfSelect *= prob_diffuse;
      pRes->dir = lambertDir;
      pRes->val = baseColor * lambertVal * (1.0f - metalness);
      pRes->pdf = lambertPdf;
      pRes->flags = RAY_FLAG_HAS_NON_SPEC;
      if (coatValue > 0.0f && fresnelIOR > 0.0f) {
        const float m_fdr_int = m.data[GLTF_FLOAT_MI_FDR_INT];
        const float f_o = FrDielectricPBRT(std::abs(dot(lambertDir, n)), 1.0f, fresnelIOR);
        pRes->val = pRes->val * lerpf(1.0f, (1.0f - f_i) * (1.0f - f_o) / (fresnelIOR * fresnelIOR * (1.0f - m_fdr_int)), coatValue);
      }
    }
  }
  pRes->pdf *= pdfSelect;
}

template <class C>
static inline void gltfEval(const Material& m, f3 l, f3 v, f3 n, f2 tc, C baseColor, f4 fourParams, BsdfEvalT<C>* res)    // cmat_gltf.h:93-147
{
  const uint cflags = m.cflags;
  const C metalCol = baseColor * m.colors[GLTF_COLOR_METAL];
  const f4 coatCol = m.colors[GLTF_COLOR_COAT];
  const float roughness = clampf(1.0f - m.data[GLTF_FLOAT_GLOSINESS] * fourParams.x, 0.0f, 1.0f);
  float metalness = m.data[GLTF_FLOAT_ALPHA] * fourParams.y;
  const float coatValue = m.data[GLTF_FLOAT_REFL_COAT] * fourParams.z;
  const float fresnelIOR = m.data[GLTF_FLOAT_IOR];
  if (cflags == GLTF_COMPONENT_METAL) metalness = 1.0f;

  float ggxVal, ggxPdf, VdotH;
  if (roughness != 0.0f) {
    ggxVal = ggxEvalBSDF(l, v, n, roughness);
    ggxPdf = ggxEvalPDF(l, v, n, roughness);
    VdotH = dot(v, normalize(v + l));
  } else { ggxVal = 0.0f; ggxPdf = 0.0f; VdotH = dot(v, n); }

  float lambertVal = lambertEvalBSDF(l, v, n);
  const float lambertPdf = lambertEvalPDF(l, v, n);
  float f_i = 1.0f;
  if (coatValue > 0.0f && metalness < 1.0f && fresnelIOR > 0.0f) {
    f_i = FrDielectricPBRT(std::abs(dot(v, n)), 1.0f, fresnelIOR);
    const float f_o = FrDielectricPBRT(std::abs(dot(l, n)), 1.0f, fresnelIOR);
    const float m_fdr_int = m.data[GLTF_FLOAT_MI_FDR_INT];
    const float coeff = lerpf(1.0f, (1.f - f_i) * (1.f - f_o) / (fresnelIOR * fresnelIOR * (1.f - m_fdr_int)), coatValue);
    lambertVal *= coeff;
  }
  C fConductor = metalCol;
  if (fresnelIOR != 0.0f) fConductor = metalCol + (C(splat4(1.0f)) - metalCol) * fresnelSlick(VdotH);
  const C specularColor = fConductor * ggxVal;
  const float prob_specular = 0.5f * coatValue;
  const float prob_diffuse = 1.0f - prob_specular;
  const C dielectricVal = baseColor * lambertVal + C(ggxVal * coatCol * f_i * coatValue);
  const float dielectricPdf = lambertPdf * prob_diffuse + ggxPdf * prob_specular;
  res->val = specularColor * metalness + dielectricVal * (1.0f - metalness);
  res->pdf = metalness * ggxPdf + (1.0f - metalness) * dielectricPdf;
}

// ---- include/cmat_diffuse.h -----------------------------------------------------------------------------------
static inline void diffuseSampleAndEval(const Material& m, f4 reflSpec, f4 rands, f3 v, f3 n, f2 tc, BsdfSample* pRes)   // :8-24
{
  const f3 lambertDir = lambertSample(mk2(rands.x, rands.y), v, n);
  const float lambertPdf = lambertEvalPDF(lambertDir, v, n);
  const float lambertVal = lambertEvalBSDF(lambertDir, v, n);
  pRes->dir = lambertDir;
  pRes->val = lambertVal * reflSpec;
  pRes->pdf = lambertPdf;
  pRes->flags = RAY_FLAG_HAS_NON_SPEC;
  if ((m.cflags & GLTF_COMPONENT_ORENNAYAR) != 0)
    pRes->val = pRes->val * orennayarFunc(lambertDir, (-1.0f) * v, n, m.data[DIFFUSE_ROUGHNESS]);
}
static inline void diffuseEval(const Material& m, f4 reflSpec, f3 l, f3 v, f3 n, f2 tc, BsdfEval* res)   // :27-39
{
  float lambertVal = lambertEvalBSDF(l, v, n);
  const float lambertPdf = lambertEvalPDF(l, v, n);
  if ((m.cflags & GLTF_COMPONENT_ORENNAYAR) != 0)
    lambertVal *= orennayarFunc(l, v, n, m.data[DIFFUSE_ROUGHNESS]);
  res->val = lambertVal * reflSpec;
  res->pdf = lambertPdf;
}

// ---- include/cmat_conductor.h ---------------------------------------------------------------------------------
static inline void conductorSmoothSampleAndEval(const Material& m, f4 etaSpec, f4 kSpec, f4 rands, f3 v, f3 n, f2 tc, BsdfSample* pRes)   // :7-28
{
  const f4 rgb_reflectance = m.colors[CONDUCTOR_COLOR];
  const f3 pefReflDir = reflect((-1.0f) * v, n);
  const float cosThetaOut = dot(pefReflDir, n);
  float val[4];
  const float eta[4] = { etaSpec.x, etaSpec.y, etaSpec.z, etaSpec.w }, kk[4] = { kSpec.x, kSpec.y, kSpec.z, kSpec.w };
  for (int i = 0; i < 4; ++i) {
    val[i] = FrComplexConductor(cosThetaOut, cmk(eta[i], kk[i]));
    val[i] = (cosThetaOut <= 1e-6f) ? 0.0f : (val[i] / std::max(cosThetaOut, 1e-6f));
  }
  pRes->val = mk4(val[0], val[1], val[2], val[3]) * rgb_reflectance;
  pRes->dir = pefReflDir;
  pRes->pdf = 1.0f;
  pRes->flags = RAY_EVENT_S;
}
static inline float conductorRoughEvalInternal(f3 wo, f3 wi, f3 wm, f2 alpha, cplx ior)   // :42-58
{
  if (wo.z * wi.z < 0) return 0.0f;
  const float cosTheta_o = AbsCosTheta(wo), cosTheta_i = AbsCosTheta(wi);
  if (cosTheta_i == 0 || cosTheta_o == 0) return 0.0f;
  const float F = FrComplexConductor(std::abs(dot(wo, wm)), ior);
  return trD(wm, alpha) * F * trG(wo, wi, alpha) / (4.0f * cosTheta_i * cosTheta_o);
}
static inline void conductorRoughSampleAndEval(const Material& m, f4 etaSpec, f4 kSpec, f4 rands, f3 v, f3 n, f2 tc, f3 alpha_tex, BsdfSample* pRes)   // :61-100
{
  if (v.z == 0) return;
  const f4 rgb_reflectance = m.colors[CONDUCTOR_COLOR];
  const f2 alpha = mk2(std::min(m.data[CONDUCTOR_ROUGH_U], alpha_tex.x), std::min(m.data[CONDUCTOR_ROUGH_V], alpha_tex.y));
  f3 nx, ny, nz = n;
  CoordinateSystemV2(nz, &nx, &ny);
  const f3 wo = mk3(dot(v, nx), dot(v, ny), dot(v, nz));
  if (wo.z == 0) return;
  const f3 wm = trSample(wo, mk2(rands.x, rands.y), alpha);
  const f3 wi = reflect((-1.0f) * wo, wm);
  if (wo.z * wi.z < 0) return;
  const float eta[4] = { etaSpec.x, etaSpec.y, etaSpec.z, etaSpec.w }, kk[4] = { kSpec.x, kSpec.y, kSpec.z, kSpec.w };
  float val[4];
  for (int i = 0; i < 4; ++i) val[i] = conductorRoughEvalInternal(wo, wi, wm, alpha, cmk(eta[i], kk[i]));
  pRes->val = mk4(val[0], val[1], val[2], val[3]) * rgb_reflectance;
  pRes->dir = normalize(wi.x * nx + wi.y * ny + wi.z * nz);
  pRes->pdf = trPDF(wo, wm, alpha) / (4.0f * std::abs(dot(wo, wm)));
  pRes->flags = RAY_FLAG_HAS_NON_SPEC;
}
static inline void conductorRoughEval(const Material& m, f4 etaSpec, f4 kSpec, f3 l, f3 v, f3 n, f2 tc, f3 alpha_tex, BsdfEval* pRes)   // :103-137
{
  const f2 alpha = mk2(std::min(m.data[CONDUCTOR_ROUGH_U], alpha_tex.x), std::min(m.data[CONDUCTOR_ROUGH_V], alpha_tex.y));
  const f4 rgb_reflectance = m.colors[CONDUCTOR_COLOR];
  f3 nx, ny, nz = n;
  CoordinateSystemV2(nz, &nx, &ny);
  const f3 wo = mk3(dot(v, nx), dot(v, ny), dot(v, nz));
  const f3 wi = mk3(dot(l, nx), dot(l, ny), dot(l, nz));
  if (wo.z * wi.z < 0.0f) return;
  f3 wm = wo + wi;
  if (dot(wm, wm) == 0) return;
  wm = normalize(wm);
  const float eta[4] = { etaSpec.x, etaSpec.y, etaSpec.z, etaSpec.w }, kk[4] = { kSpec.x, kSpec.y, kSpec.z, kSpec.w };
  float val[4];
  for (int i = 0; i < 4; ++i) val[i] = conductorRoughEvalInternal(wo, wi, wm, alpha, cmk(eta[i], kk[i]));
  pRes->val = mk4(val[0], val[1], val[2], val[3]) * rgb_reflectance;
  wm = FaceForward(wm, mk3(0.0f, 0.0f, 1.0f));
  pRes->pdf = trPDF(wo, wm, alpha) / (4.0f * std::abs(dot(wo, wm)));
}

// ---- include/cmat_dielectric.h --------------------------------------------------------------------------------
static inline void dielectricSmoothSampleAndEval(const Material& m, f4 etaSpec, float _extIOR, f4 rands, f3 v, f3 n, f2 tc, BsdfSample* pRes)   // :8-56
{
  const float extIOR = m.data[DIELECTRIC_ETA_EXT];
  if ((pRes->flags & RAY_FLAG_HAS_INV_NORMAL) != 0) n = (-1.0f) * n;
  f3 s, t = n;
  CoordinateSystemV2(n, &s, &t);
  const f3 wi = mk3(dot(v, s), dot(v, t), dot(v, n));
  const float eta = etaSpec.x / extIOR;
  const f4 fr = FrDielectricDetailedV2(wi.z, eta);
  const float R = fr.x, cos_theta_t = fr.y, eta_ti = fr.w;
  const float T = 1 - R;
  if (rands.x < R) {
    const f3 wo = mk3(-wi.x, -wi.y, wi.z);
    pRes->val = splat4(R);
    pRes->pdf = R;
    pRes->dir = normalize(wo.x * s + wo.y * t + wo.z * n);
    pRes->flags |= RAY_EVENT_S;
    pRes->ior = _extIOR;
  } else {
    const f3 wo = refract_(wi, cos_theta_t, eta_ti);
    pRes->val = splat4((eta_ti * eta_ti) * T);
    pRes->pdf = T;
    pRes->dir = normalize(wo.x * s + wo.y * t + wo.z * n);
    pRes->flags |= (RAY_EVENT_S | RAY_EVENT_T);
    pRes->ior = (_extIOR == etaSpec.x) ? extIOR : etaSpec.x;
  }
  pRes->val = pRes->val / std::max(std::abs(dot(pRes->dir, n)), 1e-6f);
}

// ---- include/cmat_glass.h (the legacy Hydra glass: glassSampleAndEval :236-277, helpers :190-233; glassEval :281-287 returns zero) ----
static inline f3 reflect2(f3 dir, f3 n) { return normalize(dir - 2.0f * dot(dir, n) * n); }
static inline f3 refract2(f3 dir, f3 n, float relativeIor)
{
  const float cosi = dot(dir, n);
  const float eta = 1.0f / relativeIor;
  const float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
  if (k < 0) return reflect2(dir, n);
  return normalize(eta * dir - (eta * cosi + std::sqrt(k)) * n);
}
static inline float fresnel2(f3 v, f3 n, float ior)
{
  const float cosi = dot(v, n);
  const float sint = std::sqrt(1.0f - cosi * cosi) / ior;
  if (sint > 1.0f) return 1.0f;
  const float cost = std::sqrt(1.0f - sint * sint);
  const float Rp = (ior * cosi - cost) / (ior * cosi + cost);
  const float Rs = (cosi - ior * cost) / (cosi + ior * cost);
  return (Rp * Rp + Rs * Rs) * 0.5f;
}
static inline void glassSampleAndEval(const Material& m, f4 rands, f3 viewDir, f3 normal, BsdfSample* pRes, float* misPrevIor)
{
  const f4 colorReflect = m.colors[GLASS_COLOR_REFLECT], colorTransp = m.colors[GLASS_COLOR_TRANSP];
  const float ior = m.data[GLASS_FLOAT_IOR];
  const f3 rayDir = (-1.0f) * viewDir;
  float relativeIor = ior / *misPrevIor;
  if ((pRes->flags & RAY_FLAG_HAS_INV_NORMAL) != 0) { if (*misPrevIor == ior) relativeIor = 1.0f / ior; }
  const float fresnel = fresnel2(viewDir, normal, relativeIor);
  f3 dir;
  if (rands.w < fresnel) { dir = reflect2(rayDir, normal); pRes->val = colorReflect; pRes->flags |= RAY_EVENT_S; }
  else { dir = refract2(rayDir, normal, relativeIor); pRes->val = colorTransp; *misPrevIor = ior; pRes->flags |= (RAY_EVENT_S | RAY_EVENT_T); }
  const float cosThetaOut = std::abs(dot(dir, normal));
  pRes->val = pRes->val / std::max(cosThetaOut, 1e-6f);
  pRes->dir = dir;
  pRes->pdf = 1.0f;
}

// ---- Mitsuba-style GGX helpers (include/cmaterial.h:563-583, 746-905) and MAT_TYPE_PLASTIC (include/cmat_plastic.h) ---------------------
static const uint MAT_TYPE_PLASTIC = 5;                              // include/cmaterial.h:42; data: roughness, ior ratio, spec weight, reflectance (:133-136)
static const int  MI_ROUGH_TRANSMITTANCE_RES = 64;                   // include/cglobals.h:18
static const float EPSILON_32 = 5.960464477539063E-8f;               // include/cglobals.h:24
static inline float FrDielectric(float cosTheta_i, float eta)        // :563-583
{
  cosTheta_i = clampf(cosTheta_i, -1.0f, 1.0f);
  if (cosTheta_i < 0.0f) { eta = 1.0f / eta; cosTheta_i = -cosTheta_i; }
  const float sin2Theta_i = 1.0f - cosTheta_i * cosTheta_i;
  const float sin2Theta_t = sin2Theta_i / (eta * eta);
  if (sin2Theta_t >= 1.0f) return 1.f;
  const float cosTheta_t = safe_sqrt(1.0f - sin2Theta_t);
  const float r_parl = (eta * cosTheta_i - cosTheta_t) / (eta * cosTheta_i + cosTheta_t);
  const float r_perp = (cosTheta_i - eta * cosTheta_t) / (cosTheta_i + eta * cosTheta_t);
  return (r_parl * r_parl + r_perp * r_perp) / 2.0f;
}
static inline f2 square_to_uniform_disk_concentric(f2 s)             // :770-791
{
  const float x = 2.f * s.x - 1.f, y = 2.f * s.y - 1.f;
  float phi, r;
  if (x == 0 && y == 0) { r = phi = 0; }
  else if (x * x > y * y) { r = x; phi = (kPI / 4.f) * (y / x); }
  else { r = y; phi = (kPI / 2.f) - (x / y) * (kPI / 4.f); }
  return mk2(r * std::cos(phi), r * std::sin(phi));
}
static inline f3 square_to_cosine_hemisphere(f2 s)                   // :793-803
{
  const f2 p = square_to_uniform_disk_concentric(s);
  return mk3(p.x, p.y, safe_sqrt(1.f - (p.x * p.x + p.y * p.y)));
}
static inline float smith_g1(f3 v, f3 m, f2 alpha)                   // :813-833
{
  const float xy_alpha_2 = alpha.x * v.x * alpha.x * v.x + alpha.y * v.y * alpha.y * v.y, tan_theta_alpha_2 = xy_alpha_2 / (v.z * v.z);
  float result = 2.f / (1.f + safe_sqrt(1.f + tan_theta_alpha_2));
  if (xy_alpha_2 == 0.f) result = 1.f;
  if (v.z * dot(v, m) <= 0.f) result = 0.f;
  return result;
}
static inline float eval_microfacet_ggx(f3 m, f2 alpha)              // :840-857, type 1
{
  const float alpha_uv = alpha.x * alpha.y, cos_theta = m.z;
  const float ax = m.x / alpha.x, ay = m.y / alpha.y;
  const float q = ax * ax + ay * ay + m.z * m.z;
  const float result = 1.f / (kPI * alpha_uv * (q * q));
  return result * cos_theta > 1e-20f ? result : 0.f;
}
static inline f3 sample_visible_normal(f3 wi, f2 rands, f2 alpha)    // :859-900 (the pdf in .w is not used by the plastic)
{
  const f3 wi_p = normalize(mk3(alpha.x * wi.x, alpha.y * wi.y, wi.z));
  const float sin_theta2 = wi_p.x * wi_p.x + wi_p.y * wi_p.y;       // sincos_phi (:751-762)
  const float inv_sin_theta = 1.f / safe_sqrt(sin_theta2);
  float rx = wi_p.x * inv_sin_theta, ry = wi_p.y * inv_sin_theta;
  if (std::abs(sin_theta2) <= 4.f * EPSILON_32) { rx = 1.f; ry = 0.f; } else { rx = clampf(rx, -1.f, 1.f); ry = clampf(ry, -1.f, 1.f); }
  const float sin_phi = ry, cos_phi = rx, cos_theta = wi_p.z;
  f2 p = square_to_uniform_disk_concentric(rands);                   // sample_visible_11 (:859-874)
  const float s = 0.5f * (1.f + cos_theta);
  p.y = lerpf(safe_sqrt(1.f - p.x * p.x), p.y, s);
  const float x = p.x, y = p.y, z = safe_sqrt(1.f - (p.x * p.x + p.y * p.y));
  const float sin_theta_i = safe_sqrt(1.f - cos_theta * cos_theta);
  const float norm = 1.f / (sin_theta_i * y + cos_theta * z);
  const float slx = (cos_theta * y - sin_theta_i * z) * norm, sly = x * norm;
  const float sx = (cos_phi * slx - sin_phi * sly) * alpha.x, sy = (sin_phi * slx + cos_phi * sly) * alpha.y;
  return normalize(mk3(-sx, -sy, 1.0f));
}
static inline float plasticTransmittance(const float* transmittance, uint trOffset, float cos_theta)   // the lerp_gather written out at cmat_plastic.h:27-38
{
  float x = cos_theta;
  x *= float(MI_ROUGH_TRANSMITTANCE_RES - 1);
  const uint index = std::min(uint(x), uint(MI_ROUGH_TRANSMITTANCE_RES - 2));
  const float v0 = transmittance[trOffset + index], v1 = transmittance[trOffset + index + 1];
  return lerpf(v0, v1, x - float(index));
}
static inline void plasticSampleAndEval(const Material& m, f4 a_reflSpec, f4 rands, f3 v, f3 n, BsdfSample* pRes, const float* transmittance, uint trOffset)   // cmat_plastic.h:7-99
{
  const float alpha = m.data[0], eta = m.data[1], spec_weight = m.data[2], internal_refl = m.data[3];
  const uint nonlinear = m.nonlinear;
  const f2 alpha2 = mk2(alpha, alpha);
  f3 s = n, t = n;
  CoordinateSystemV2(n, &s, &t);
  const f3 wi = mk3(dot(v, s), dot(v, t), dot(v, n));
  if (wi.z <= 0) return;
  const float cos_theta_i = std::max(wi.z, EPSILON_32);
  const float t_i = plasticTransmittance(transmittance, trOffset, cos_theta_i);
  float prob_specular = (1.f - t_i) * spec_weight, prob_diffuse = t_i * (1.f - spec_weight);
  if (prob_diffuse != 0.0f && prob_specular != 0.0f) { prob_specular = prob_specular / (prob_specular + prob_diffuse); prob_diffuse = 1.f - prob_specular; }
  else { prob_diffuse = 1.0f; prob_specular = 0.0f; }
  const bool sample_specular = rands.z < prob_specular;
  f3 wo = mk3(0, 0, 0);
  if (sample_specular) {
    const f3 wm = sample_visible_normal(wi, mk2(rands.x, rands.y), alpha2);
    const f3 d = (-1.0f) * wi;
    wo = d - 2.0f * dot(d, wm) * wm;                                  // reflect((-1)*wi, wm)
  } else wo = square_to_cosine_hemisphere(mk2(rands.x, rands.y));
  if (cos_theta_i * wo.z <= 0) return;
  const float cos_theta_o = std::max(wo.z, EPSILON_32);
  const f3 H = normalize(wo + wi);
  const float D = eval_microfacet_ggx(H, alpha2);
  float pdf = D * smith_g1(wi, H, alpha2) / (4.f * cos_theta_i);
  pdf *= prob_specular;
  pdf += prob_diffuse * kINV_PI * cos_theta_o;
  const float F = FrDielectric(dot(wi, H), eta);
  const float G = smith_g1(wi, H, alpha2) * smith_g1(wo, H, alpha2);
  const float val = F * D * G / (4.f * cos_theta_i * cos_theta_o);
  const float t_o = plasticTransmittance(transmittance, trOffset, cos_theta_o);
  const f4 diffuse = a_reflSpec / (splat4(1.f) - (nonlinear > 0 ? (a_reflSpec * internal_refl) : splat4(internal_refl)));
  const float inv_eta_2 = 1.f / (eta * eta);
  pRes->dir = normalize(wo.x * s + wo.y * t + wo.z * n);
  pRes->val = splat4(val) + diffuse * (kINV_PI * inv_eta_2 * t_i * t_o);
  pRes->pdf = pdf;
  pRes->flags = RAY_FLAG_HAS_NON_SPEC;
}
static inline void plasticEval(const Material& m, f4 a_reflSpec, f3 l, f3 v, f3 n, BsdfEval* pRes, const float* transmittance, uint trOffset)   // cmat_plastic.h:102-191
{
  const float alpha = m.data[0], eta = m.data[1], spec_weight = m.data[2], internal_refl = m.data[3];
  const uint nonlinear = m.nonlinear;
  const f2 alpha2 = mk2(alpha, alpha);
  f3 s = n, t = n;
  CoordinateSystemV2(n, &s, &t);
  const f3 wo = mk3(dot(l, s), dot(l, t), dot(l, n)), wi = mk3(dot(v, s), dot(v, t), dot(v, n));
  if (wi.z * wo.z <= 0) return;
  const float cos_theta_i = std::max(wi.z, EPSILON_32), cos_theta_o = std::max(wo.z, EPSILON_32);
  const float t_i = plasticTransmittance(transmittance, trOffset, cos_theta_i);
  float prob_specular = (1.f - t_i) * spec_weight, prob_diffuse = t_i * (1.f - spec_weight);
  if (prob_diffuse != 0.0f && prob_specular != 0.0f) { prob_specular = prob_specular / (prob_specular + prob_diffuse); prob_diffuse = 1.f - prob_specular; }
  else { prob_diffuse = 1.0f; prob_specular = 0.0f; }
  const f3 H = normalize(wo + wi);
  const float D = eval_microfacet_ggx(H, alpha2);
  const float smith_g1_wi = smith_g1(wi, H, alpha2);
  float pdf = D * smith_g1_wi / (4.f * cos_theta_i);
  pdf *= prob_specular;
  pdf += prob_diffuse * kINV_PI * cos_theta_o;
  const float F = FrDielectric(dot(wi, H), eta);
  const float G = smith_g1(wo, H, alpha2) * smith_g1_wi;
  const float val = F * D * G / (4.f * cos_theta_i * cos_theta_o);
  const float t_o = plasticTransmittance(transmittance, trOffset, cos_theta_o);
  const f4 diffuse = a_reflSpec / (splat4(1.f) - (nonlinear > 0 ? (a_reflSpec * internal_refl) : splat4(internal_refl)));
  const float inv_eta_2 = 1.f / (eta * eta);
  pRes->val = splat4(val) + diffuse * (kINV_PI * inv_eta_2 * t_i * t_o);
  pRes->pdf = pdf;
}

// ---- include/cmat_film.h, include/airy_reflectance.h, include/cmaterial.h:957-1036 (MAT_TYPE_THIN_FILM) ----------------------------------
static const uint MAT_TYPE_THIN_FILM = 8;                                                              // include/cmaterial.h:45
static const uint FILM_ANGLE_RES = 180, FILM_LENGTH_RES = 94, FILM_THICKNESS_RES = 32;                 // include/cglobals.h:19-21
static const float FILM_LAMBDA_MIN = 360.0f, FILM_LAMBDA_MAX = 830.0f;                                 // include/cglobals.h:22-23
static const int FILM_ROUGH_U = 0, FILM_ROUGH_V = 1, FILM_PRECOMP_FLAG = 2, FILM_PRECOMP_OFFSET = 3, FILM_ETA_OFFSET = 4, FILM_K_OFFSET = 5,
                 FILM_ETA_SPECID_OFFSET = 6, FILM_K_SPECID_OFFSET = 7, FILM_ETA_EXT = 8, FILM_THICKNESS_OFFSET = 9, FILM_THICKNESS_MIN = 10,
                 FILM_THICKNESS_MAX = 11, FILM_THICKNESS_MAP = 12, FILM_THICKNESS = 13, FILM_LAYERS_COUNT = 14, FILM_TRANSPARENT = 15;   // include/cmaterial.h:164-179
static inline uint as_uint_(float f) { uint u; std::memcpy(&u, &f, 4); return u; }
static inline float sum4(f4 v) { return v.x + v.y + v.z + v.w; }                                       // cmaterial.h:955
struct FrReflRefr { float refl, refr; };                                                               // :962-965
static inline float getRefractionFactor(float cosThetaI, cplx cosThetaT, cplx iorI, cplx iorT)         // :967-975
{
  const cplx mult = cosThetaT * iorT;
  if (cosThetaI <= 1e-6f || mult.im > 1e-6f) return 0.0f;
  return mult.re / (iorI.re * cosThetaI);
}
static inline cplx FrComplexRefl(cplx cosThetaI, cplx cosThetaT, cplx iorI, cplx iorT, int polP)       // :995-1010 (polP: 0 = PolarizationS, 1 = PolarizationP)
{
  if (cnorm(cosThetaI) < 1e-6f) return cmk(-1.0f, 0.0f);
  if (!polP) return (iorI * cosThetaI - iorT * cosThetaT) / (iorI * cosThetaI + iorT * cosThetaT);
  return (iorT * cosThetaI - iorI * cosThetaT) / (iorT * cosThetaI + iorI * cosThetaT);
}
static inline cplx FrComplexRefr(cplx cosThetaI, cplx cosThetaT, cplx iorI, cplx iorT, int polP)       // :1012-1031
{
  if (cnorm(cosThetaI) < 1e-6f) return (cnorm(iorI - iorT) < 1e-6f) ? cmk(1.0f, 0.0f) : cmk(0.0f, 0.0f);
  if (!polP) return ((iorI * 2.0f) * cosThetaI) / (iorI * cosThetaI + iorT * cosThetaT);
  return ((iorI * 2.0f) * cosThetaI) / (iorT * cosThetaI + iorI * cosThetaT);
}
static inline cplx filmPhaseDiff(cplx cosTheta, cplx eta, float thickness, float lambda)               // :1033-1036
{ return (((eta * float(4 * M_PI)) * cosTheta) * thickness) / cmk(lambda, 0.0f); }
// the cosines of the refracted directions inside the film and the substrate (airy_reflectance.h:11-15, 39-43, 69-73)
static inline void filmCosines(float cosThetaI, cplx etaI, cplx etaF, cplx etaT, cplx* cosThetaF, cplx* cosThetaT)
{
  const cplx sinThetaI = cmk(1.0f - cosThetaI * cosThetaI, 0.0f);
  const cplx sinThetaF = (sinThetaI * (etaI.re * etaI.re)) / (etaF * etaF);
  *cosThetaF = csqrt_(rsub(1.0f, sinThetaF));
  const cplx sinThetaT = (sinThetaI * (etaI.re * etaI.re)) / (etaT * etaT);
  *cosThetaT = csqrt_(rsub(1.0f, sinThetaT));
}
static inline float FrFilmRefl(float cosThetaI, cplx etaI, cplx etaF, cplx etaT, float thickness, float lambda)   // airy_reflectance.h:9-33
{
  cplx cosThetaF, cosThetaT;
  filmCosines(cosThetaI, etaI, etaF, etaT, &cosThetaF, &cosThetaT);
  const cplx phaseDiff = filmPhaseDiff(cosThetaF, etaF, thickness, lambda);
  float result = 0;
  for (int p = 0; p <= 1; ++p) {
    const cplx FrReflI = FrComplexRefl(cmk(cosThetaI, 0.0f), cosThetaF, etaI, etaF, p);
    const cplx FrReflF = FrComplexRefl(cosThetaF, cosThetaT, etaF, etaT, p);
    cplx FrRefl = (FrReflF * std::exp(-phaseDiff.im)) * cmk(std::cos(phaseDiff.re), std::sin(phaseDiff.re));
    FrRefl = (FrReflI + FrRefl) / radd(1.0f, FrReflI * FrRefl);
    result += cnorm(FrRefl);
  }
  return result / 2;
}
static inline FrReflRefr FrFilm(float cosThetaI, cplx etaI, cplx etaF, cplx etaT, float thickness, float lambda)  // airy_reflectance.h:67-106
{
  cplx cosThetaF, cosThetaT;
  filmCosines(cosThetaI, etaI, etaF, etaT, &cosThetaF, &cosThetaT);
  const cplx phaseDiff = filmPhaseDiff(cosThetaF, etaF, thickness, lambda);
  FrReflRefr result = { 0, 0 };
  for (int p = 0; p <= 1; ++p) {
    const cplx FrReflI = FrComplexRefl(cmk(cosThetaI, 0.0f), cosThetaF, etaI, etaF, p);
    const cplx FrReflF = FrComplexRefl(cosThetaF, cosThetaT, etaF, etaT, p);
    const cplx FrRefrI = FrComplexRefr(cmk(cosThetaI, 0.0f), cosThetaF, etaI, etaF, p);
    const cplx FrRefrF = FrComplexRefr(cosThetaF, cosThetaT, etaF, etaT, p);
    const cplx exp_1 = cmk(std::cos(phaseDiff.re / 2), std::sin(phaseDiff.re / 2)) * std::exp(-phaseDiff.im / 2);
    const cplx exp_2 = exp_1 * exp_1;
    const cplx denom = radd(1.0f, (FrReflI * FrReflF) * exp_2);
    if (cnorm(denom) < 1e-6f) result.refl += 0.5f;
    else {
      result.refl += cnorm((FrReflI + FrReflF * exp_2) / denom) / 2;
      result.refr += cnorm(((FrRefrI * FrRefrF) * exp_1) / denom) / 2;
    }
  }
  result.refr *= getRefractionFactor(cosThetaI, cosThetaT, etaI, etaT);
  return result;
}
// The table look-ups of cmat_film.h (:41-143, 227-329, 461-535): the angle is the table's second axis; the first is the wavelength (spectral mode),
// the thickness (RGB mode with a thickness map) or absent. Reflectance at refl_offset, transmittance at refr_offset (units of FILM_ANGLE_RES rows).
static inline float filmThetaIndex(float cosThetaI)
{ return clampf(float(double(std::acos(cosThetaI) * 2.f) / M_PI), 0.f, 1.f) * float(FILM_ANGLE_RES - 1); }
static inline float filmLerp2D(const float* pre, uint base, uint rowLen, uint stride, uint ch, float x, float theta, uint xRes)
{
  const uint index1 = std::min(uint(x), uint(xRes - 2)), index2 = std::min(uint(theta), uint(FILM_ANGLE_RES - 2));
  const float alpha = x - float(index1), beta = theta - float(index2);
  const uint a = (base + index1 * rowLen + index2) * stride + ch, b = (base + (index1 + 1) * rowLen + index2) * stride + ch;
  const uint c = (base + index1 * rowLen + index2 + 1) * stride + ch, d = (base + (index1 + 1) * rowLen + index2 + 1) * stride + ch;
  const float v0 = lerpf(pre[a], pre[b], alpha), v1 = lerpf(pre[c], pre[d], alpha);
  return lerpf(v0, v1, beta);
}
static inline void filmReflTrans(const Material& m, float cosThetaI, float extIOR, cplx filmIOR, cplx intIOR, float thickness, float lambda, bool reversed,
                                 const float* pre, uint off, bool spectral_mode, bool precomputed, bool wantT, f4* R, f4* T)
{
  const uint refl_offset = reversed ? FILM_ANGLE_RES * 2 : 0, refr_offset = reversed ? FILM_ANGLE_RES * 3 : FILM_ANGLE_RES;
  *R = mk4(0, 0, 0, 0); *T = mk4(0, 0, 0, 0);
  if (spectral_mode) {
    if (precomputed) {
      const float w = clampf((lambda - FILM_LAMBDA_MIN) / (FILM_LAMBDA_MAX - FILM_LAMBDA_MIN), 0.f, 1.f) * float(FILM_LENGTH_RES - 1);
      const float theta = filmThetaIndex(cosThetaI);
      R->x = filmLerp2D(pre + off, refl_offset * FILM_LENGTH_RES, FILM_ANGLE_RES, 1, 0, w, theta, FILM_LENGTH_RES);
      if (wantT) T->x = filmLerp2D(pre + off, refr_offset * FILM_LENGTH_RES, FILM_ANGLE_RES, 1, 0, w, theta, FILM_LENGTH_RES);
    } else if (wantT) {
      const FrReflRefr r = !reversed ? FrFilm(cosThetaI, cmk(extIOR, 0.0f), filmIOR, intIOR, thickness, lambda) : FrFilm(cosThetaI, intIOR, filmIOR, cmk(extIOR, 0.0f), thickness, lambda);
      R->x = r.refl; T->x = r.refr;
    } else
      R->x = !reversed ? FrFilmRefl(cosThetaI, cmk(extIOR, 0.0f), filmIOR, intIOR, thickness, lambda) : FrFilmRefl(cosThetaI, intIOR, filmIOR, cmk(extIOR, 0.0f), thickness, lambda);
  } else {
    const float theta = filmThetaIndex(cosThetaI);
    float* r = &R->x; float* t = &T->x;
    if (as_uint_(m.data[FILM_THICKNESS_MAP]) == 1u) {
      const float tmin = m.data[FILM_THICKNESS_MIN], tmax = m.data[FILM_THICKNESS_MAX];
      const float tt = clampf((thickness - tmin) / (tmax - tmin), 0.f, 1.f) * float(FILM_THICKNESS_RES - 1);
      for (uint ch = 0; ch < 3; ch++) {
        r[ch] = filmLerp2D(pre + off, refl_offset * FILM_THICKNESS_RES, FILM_ANGLE_RES, 3, ch, tt, theta, FILM_THICKNESS_RES);
        if (wantT) t[ch] = filmLerp2D(pre + off, refr_offset * FILM_THICKNESS_RES, FILM_ANGLE_RES, 3, ch, tt, theta, FILM_THICKNESS_RES);
      }
    } else {
      const uint index = std::min(uint(theta), uint(FILM_ANGLE_RES - 2));
      const float alpha = theta - float(index);
      for (uint ch = 0; ch < 3; ch++) {
        r[ch] = lerpf(pre[off + (refl_offset + index) * 3 + ch], pre[off + (refl_offset + index + 1) * 3 + ch], alpha);
        if (wantT) t[ch] = lerpf(pre[off + (refr_offset + index) * 3 + ch], pre[off + (refr_offset + index + 1) * 3 + ch], alpha);
      }
    }
  }
}
static inline void filmSmoothSampleAndEval(const Material& m, float extIOR, cplx filmIOR, cplx intIOR, float thickness, f4 a_wavelengths, float _extIOR, f4 rands, f3 v, f3 n,
                                           BsdfSample* pRes, const float* pre, uint precompOffset, bool spectral_mode, bool precomputed)   // cmat_film.h:9-181
{
  const uint transparFlag = as_uint_(m.data[FILM_TRANSPARENT]);
  if ((pRes->flags & RAY_FLAG_HAS_INV_NORMAL) != 0) n = (-1.0f) * n;
  const bool reversed = dot(n, v) < 0.f && intIOR.im < 0.001f;
  f3 s, t = n;
  CoordinateSystemV2(n, &s, &t);
  const f3 wi = mk3(dot(v, s), dot(v, t), dot(v, n));
  const float cosThetaI = clampf(std::abs(wi.z), 0.0001f, 1.0f);
  const float ior = intIOR.re / extIOR;
  f4 R, T;
  filmReflTrans(m, cosThetaI, extIOR, filmIOR, intIOR, thickness, a_wavelengths.x, reversed, pre, precompOffset, spectral_mode, precomputed, true, &R, &T);
  if (intIOR.im > 0.001f || transparFlag == 0) {
    const f3 wo = mk3(-wi.x, -wi.y, wi.z);
    pRes->val = R; pRes->pdf = 1.f;
    pRes->dir = normalize(wo.x * s + wo.y * t + wo.z * n);
    pRes->flags |= RAY_EVENT_S; pRes->ior = _extIOR;
  } else if (rands.x * (sum4(R) + sum4(T)) < sum4(R)) {
    const f3 wo = mk3(-wi.x, -wi.y, wi.z);
    pRes->val = R; pRes->pdf = sum4(R) / (sum4(R) + sum4(T));
    pRes->dir = normalize(wo.x * s + wo.y * t + wo.z * n);
    pRes->flags |= RAY_EVENT_S; pRes->ior = _extIOR;
  } else {
    const f4 fr = FrDielectricDetailedV2(wi.z, ior);
    const f3 wo = refract_(wi, fr.y, fr.w);
    pRes->val = T; pRes->pdf = sum4(T) / (sum4(R) + sum4(T));
    pRes->dir = normalize(wo.x * s + wo.y * t + wo.z * n);
    pRes->flags |= (RAY_EVENT_S | RAY_EVENT_T);
    pRes->ior = (_extIOR == intIOR.re) ? extIOR : intIOR.re;
  }
  pRes->val = pRes->val / std::max(std::abs(dot(pRes->dir, n)), 1e-6f);
}
static inline float microfacet_G(f3 wi, f3 wo, f3 mm, f2 alpha) { return smith_g1(wi, mm, alpha) * smith_g1(wo, mm, alpha); }   // cmaterial.h:903-906
static inline void filmRoughSampleAndEval(const Material& m, float extIOR, cplx filmIOR, cplx intIOR, float thickness, f4 a_wavelengths, float _extIOR, f4 rands, f3 v, f3 n,
                                          f3 alpha_tex, BsdfSample* pRes, const float* pre, uint precompOffset, bool spectral_mode, bool precomputed)   // cmat_film.h:183-410
{
  const uint transparFlag = as_uint_(m.data[FILM_TRANSPARENT]);
  if ((pRes->flags & RAY_FLAG_HAS_INV_NORMAL) != 0) n = (-1.0f) * n;
  const bool reversed = dot(v, n) < 0.f && intIOR.im < 0.001f;
  const f2 alpha = mk2(std::min(m.data[FILM_ROUGH_V], alpha_tex.x), std::min(m.data[FILM_ROUGH_U], alpha_tex.y));
  f3 s, t = n;
  CoordinateSystemV2(n, &s, &t);
  f3 wi = mk3(dot(v, s), dot(v, t), dot(v, n));
  float ior = intIOR.re / extIOR;
  if (reversed) { wi = (-1.0f) * wi; ior = 1.f / ior; }
  const f3 wm = trSample(wi, mk2(rands.x, rands.y), alpha);
  const float cosThetaI = clampf(std::abs(dot(wi, wm)), 0.00001f, 1.0f);
  f4 R, T;
  filmReflTrans(m, cosThetaI, extIOR, filmIOR, intIOR, thickness, a_wavelengths.x, reversed, pre, precompOffset, spectral_mode, precomputed, true, &R, &T);
  const bool opaque = intIOR.im > 0.001f || transparFlag == 0;
  if (opaque || rands.w * (sum4(R) + sum4(T)) < sum4(R)) {
    f3 wo = reflect((-1.0f) * wi, wm);
    if (wi.z < 0.f || wo.z <= 0.f) return;
    const float cos_theta_i = std::max(wi.z, EPSILON_32), cos_theta_o = std::max(wo.z, EPSILON_32);
    pRes->pdf = trPDF(wi, wm, alpha) / (4.0f * std::abs(dot(wi, wm)));
    if (!opaque) pRes->pdf = pRes->pdf * sum4(R) / (sum4(R) + sum4(T));
    pRes->val = ((trD(wm, alpha) * microfacet_G(wi, wo, wm, alpha)) * R) / (4.0f * cos_theta_i * cos_theta_o);
    if (reversed) wo = (-1.0f) * wo;
    pRes->dir = normalize(wo.x * s + wo.y * t + wo.z * n);
    pRes->flags = RAY_FLAG_HAS_NON_SPEC;
    pRes->ior = _extIOR;
  } else {
    const f4 fr = FrDielectricDetailedV2(dot(wi, wm), ior);
    const float cosThetaT = fr.y, eta_it = fr.z, eta_ti = fr.w;
    f3 ws, wt;
    CoordinateSystemV2(wm, &ws, &wt);
    const f3 local_wi = mk3(dot(ws, wi), dot(wt, wi), dot(wm, wi));
    const f3 local_wo = refract_(local_wi, cosThetaT, eta_ti);
    f3 wo = local_wo.x * ws + local_wo.y * wt + local_wo.z * wm;
    if (wo.z > 0.f) return;
    const float cos_theta_i = std::max(wi.z, EPSILON_32), cos_theta_o = std::min(wo.z, -EPSILON_32);
    if (std::abs(eta_it - 1.f) <= 1e-6f) {
      pRes->pdf = trPDF(wi, wm, alpha) / (4.0f * std::abs(dot(wi, wm))) * sum4(T) / (sum4(R) + sum4(T));
      pRes->val = ((trD(wm, alpha) * microfacet_G(wi, wo, wm, alpha)) * T) / (4.0f * -cos_theta_i * cos_theta_o);
    } else {
      const float sq = dot(wo, wm) + dot(wi, wm) / eta_it;
      const float denom = sq * sq;
      const float dwm_dwi = std::abs(dot(wo, wm)) / denom;
      pRes->pdf = trPDF(wi, wm, alpha) * dwm_dwi * sum4(T) / (sum4(R) + sum4(T));
      pRes->val = ((trD(wm, alpha) * microfacet_G(wi, wo, wm, alpha)) * T) * std::abs(dot(wi, wm) * dot(wo, wm) / (cos_theta_i * cos_theta_o * denom));
    }
    if (reversed) wo = (-1.0f) * wo;
    pRes->dir = normalize(wo.x * s + wo.y * t + wo.z * n);
    pRes->flags = RAY_FLAG_HAS_NON_SPEC;
    pRes->ior = (_extIOR == intIOR.re) ? extIOR : intIOR.re;
  }
}
static inline void filmRoughEval(const Material& m, float extIOR, cplx filmIOR, cplx intIOR, float thickness, f4 a_wavelengths, f3 l, f3 v, f3 n, f3 alpha_tex, BsdfEval* pRes,
                                 const float* pre, uint precompOffset, bool spectral_mode, bool precomputed)   // cmat_film.h:413-544
{
  if (intIOR.im < 0.001f) return;                                   // a film on a dielectric evaluates to zero; `reversed` (:425-431) can then never hold
  const f2 alpha = mk2(std::min(m.data[FILM_ROUGH_V], alpha_tex.x), std::min(m.data[FILM_ROUGH_U], alpha_tex.y));
  f3 s, t = n;
  CoordinateSystemV2(n, &s, &t);
  const f3 wo = mk3(dot(l, s), dot(l, t), dot(l, n)), wi = mk3(dot(v, s), dot(v, t), dot(v, n));
  const f3 wm = normalize(wo + wi);
  if (wi.z * wo.z < 0.f) return;
  const float cosThetaI = clampf(std::abs(dot(wo, wm)), 0.00001f, 1.0f);
  f4 R, T;
  filmReflTrans(m, cosThetaI, extIOR, filmIOR, intIOR, thickness, a_wavelengths.x, false, pre, precompOffset, spectral_mode, precomputed, false, &R, &T);
  const float cos_theta_i = std::max(wi.z, EPSILON_32), cos_theta_o = std::max(wo.z, EPSILON_32);
  pRes->pdf = trPDF(wi, wm, alpha) / (4.0f * std::abs(dot(wi, wm)));
  pRes->val = ((trD(wm, alpha) * microfacet_G(wi, wo, wm, alpha)) * R) / (4.0f * cos_theta_i * cos_theta_o);
}

// ---- include/clight.h -----------------------------------------------------------------------------------------
struct LightSample { f3 pos, norm; float pdf; bool isOmni, hasIES; };   // :58-65

static inline LightSample areaLightSampleRev(const LightSource& L, f2 rands)   // :67-84
{
  f2 sampleOff = 2.0f * (mk2(-0.5f, -0.5f) + rands) * L.size;
  if (L.geomType == LIGHT_GEOM_DISC) {
    const float offsetX = rands.x * 2.0f - 1.0f, offsetY = rands.y * 2.0f - 1.0f;
    sampleOff = MapSamplesToDisc(mk2(offsetX, offsetY)) * L.size.x;
  }
  const f3 samplePos = mul3x3(L.matrix, mk3(sampleOff.x, 0.0f, sampleOff.y)) + xyz(L.pos) + epsilonOfPos(xyz(L.pos)) * xyz(L.norm);
  LightSample r;
  r.pos = samplePos; r.norm = xyz(L.norm); r.isOmni = false; r.hasIES = (L.iesId != uint(-1)); r.pdf = 1.0f;
  return r;
}
static inline LightSample sphereLightSampleRev(const LightSource& L, f2 rands)   // :86-103
{
  const float theta = 2.0f * kPI * rands.x;
  const float phi = std::acos(1.0f - 2.0f * rands.y);
  const float x = std::sin(phi) * std::cos(theta), y = std::sin(phi) * std::sin(theta), z = std::cos(phi);
  const f3 lcenter = xyz(L.pos);
  const float lradius = L.size.x;
  const f3 samplePos = lcenter + (lradius * 1.000001f) * mk3(x, y, z);
  LightSample r;
  r.pos = samplePos; r.norm = normalize(samplePos - lcenter); r.isOmni = false; r.hasIES = (L.iesId != uint(-1)); r.pdf = 1.0f;
  return r;
}
static inline LightSample directLightSampleRev(const LightSource& L, f2 rands, f3 illuminationPoint)   // :105-115
{
  const f3 norm = xyz(L.norm);
  LightSample r;
  r.pos = illuminationPoint - norm * 100000.0f; r.norm = norm; r.isOmni = false; r.hasIES = false; r.pdf = 1.0f;
  return r;
}
static inline LightSample pointLightSampleRev(const LightSource& L)   // :117-126
{
  LightSample r;
  r.pos = xyz(L.pos); r.norm = xyz(L.norm); r.isOmni = (L.distType == LIGHT_DIST_OMNI); r.hasIES = (L.iesId != uint(-1)); r.pdf = 1.0f;
  return r;
}
static inline float mylocalsmoothstep(float edge0, float edge1, float x)   // :220-225
{
  const float tVal = (x - edge0) / (edge1 - edge0);
  const float t = std::min(std::max(tVal, 0.0f), 1.0f);
  return t * t * (3.0f - 2.0f * t);
}

} // namespace orc
