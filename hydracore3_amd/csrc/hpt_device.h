// gfx950 device functions of the path-tracing core: RNG, sampling, BSDFs, lights, texture fetch and BVH2 traversal.
//
// Written for wave64 / CDNA4: everything is inlined into one persistent kernel (hpt_kernels.hip), per-lane state
// lives in VGPRs, the traversal stack lives in LDS ([depth][lane] so that a push/pop is one conflict-free
// ds_write/ds_read_b32 per wave), nodes are fetched as 64-byte lines, triangles as 3 x 16 bytes.
//
// Results must equal the reference's CPU integrator at fixed seeds, so the arithmetic follows the reference's
// formulas in the reference's evaluation order (cited per function), is compiled with -ffp-contract=off, uses
// IEEE division / sqrt, and min/max follow std::min/std::max (not fminf/fmaxf) wherever a value is consumed.
// Only the ray/box test uses fmin/fmax: it merely has to be conservative.
#pragma once
#include <hip/hip_runtime.h>
#include "hpt_types.h"

namespace hpt {

#define HPT_DEV __device__ __forceinline__

// ---- flags / enums (include/cglobals.h:9-16, include/cmaterial.h:26-56, include/clight.h:5-17, integrator_pt.h:330-332,406-408)
enum : uint {
  RAY_FLAG_IS_DEAD = 0x80000000u, RAY_FLAG_OUT_OF_SCENE = 0x40000000u, RAY_FLAG_HIT_LIGHT = 0x20000000u,
  RAY_FLAG_HAS_NON_SPEC = 0x10000000u, RAY_FLAG_HAS_INV_NORMAL = 0x08000000u, RAY_FLAG_WAVES_DIVERGED = 0x04000000u,
  RAY_FLAG_PRIME_RAY_MISS = 0x02000000u, RAY_FLAG_FIRST_NON_SPEC = 0x01000000u };
enum : uint { GLTF_COMPONENT_METAL = 4, GLTF_COMPONENT_ORENNAYAR = 16, FLAG_NMAP_INVERT_X = 32, FLAG_NMAP_INVERT_Y = 64, FLAG_NMAP_SWAP_XY = 128, FLAG_FOUR_TEXTURES = 256, FLAG_PACK_FOUR_PARAMS_IN_TEXTURE = 512 };
enum : uint { MAT_TYPE_BLEND = 6, BLEND_STACK_SIZE = 4 };   // include/cmaterial.h:43,155 (data[0] = weight, datai[0..1] = children, texid[0] = mask); integrator_pt.h:599
enum : uint { MAT_TYPE_GLASS = 2 };    // include/cmaterial.h:39; colours: 0 reflect, 1 transparency; data[2] = IOR (:85-92)
enum : uint { MAT_TYPE_GLTF = 1, MAT_TYPE_CONDUCTOR = 3, MAT_TYPE_DIFFUSE = 4, MAT_TYPE_DIELECTRIC = 7, MAT_TYPE_LIGHT_SOURCE = 0xEFFFFFFFu };
enum : uint { RAY_EVENT_S = 1, RAY_EVENT_T = 8 };
enum : uint { LIGHT_GEOM_RECT = 1, LIGHT_GEOM_DISC = 2, LIGHT_GEOM_SPHERE = 3, LIGHT_GEOM_DIRECT = 4, LIGHT_GEOM_POINT = 5, LIGHT_GEOM_ENV = 6 };
enum : uint { LIGHT_DIST_LAMBERT = 0, LIGHT_DIST_OMNI = 1, LIGHT_DIST_SPOT = 2, LIGHT_FLAG_POINT_AREA = 1, LIGHT_FLAG_PROJECTIVE = 2 };
enum : uint { INTEGRATOR_STUPID_PT = 0, INTEGRATOR_SHADOW_PT = 1, INTEGRATOR_MIS_PT = 2, FB_COLOR = 0, FB_DIRECT = 1, FB_INDIRECT = 2 };
// Material::colors / Material::data slots (include/cmaterial.h:67-147)
enum { GLTF_COLOR_BASE = 0, GLTF_COLOR_COAT = 1, GLTF_COLOR_METAL = 2 };
enum { GLTF_FLOAT_MI_FDR_INT = 0, GLTF_FLOAT_ALPHA = 3, GLTF_FLOAT_GLOSINESS = 4, GLTF_FLOAT_IOR = 5, GLTF_FLOAT_REFL_COAT = 7 };

#define HPT_PI      3.14159265358979323846f
#define HPT_TWOPI   6.28318530717958647692f
#define HPT_INV_PI  0.31830988618379067154f
#define HPT_FLT_MAX 3.402823466e+38f

// ---- small vector helpers ------------------------------------------------------------------------------------------
struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

HPT_DEV V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
HPT_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
HPT_DEV V3 v3s(float a) { return v3(a, a, a); }
HPT_DEV V4 v4(float x, float y, float z, float w) { V4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
HPT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
HPT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
HPT_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
HPT_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
HPT_DEV V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
HPT_DEV V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
HPT_DEV V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
HPT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
HPT_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
HPT_DEV float length(V3 a) { return __builtin_sqrtf(dot(a, a)); }
HPT_DEV V3 normalize(V3 a) { const float lenInv = 1.0f / length(a); return a * lenInv; }
HPT_DEV V3 reflect(V3 i, V3 n) { return i - 2.0f * dot(n, i) * n; }
// std::min / std::max semantics (NaN behaviour included), as the reference's host build evaluates them
HPT_DEV float smax(float a, float b) { return (a < b) ? b : a; }
HPT_DEV float smin(float a, float b) { return (b < a) ? b : a; }
HPT_DEV float clampf(float x, float lo, float hi) { return smin(smax(x, lo), hi); }
HPT_DEV float lerpf(float a, float b, float t) { return a + t * (b - a); }
HPT_DEV float absf(float a) { return __builtin_fabsf(a); }
HPT_DEV float sqrtf_(float a) { return __builtin_sqrtf(a); }
HPT_DEV bool finitef(float a) { return absf(a) <= HPT_FLT_MAX; }     // false for NaN and +-inf

// column-major 4x4 (LiteMath float4x4): element (row,col) = m[col*4+row]
HPT_DEV V4 mul4x4(const float* m, V4 v)
{
  V4 r;
  r.x = v.x * m[0] + v.y * m[4] + v.z * m[8]  + v.w * m[12];
  r.y = v.x * m[1] + v.y * m[5] + v.z * m[9]  + v.w * m[13];
  r.z = v.x * m[2] + v.y * m[6] + v.z * m[10] + v.w * m[14];
  r.w = v.x * m[3] + v.y * m[7] + v.z * m[11] + v.w * m[15];
  return r;
}
HPT_DEV V3 mul4x3(const float* m, V3 p)
{
  return v3(m[0] * p.x + m[4] * p.y + m[8]  * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9]  * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
HPT_DEV V3 mul3x3(const float* m, V3 v)
{
  return v3(m[0] * v.x + m[4] * v.y + m[8]  * v.z,
            m[1] * v.x + m[5] * v.y + m[9]  * v.z,
            m[2] * v.x + m[6] * v.y + m[10] * v.z);
}

// ---- RNG: include/crandom.h:11-75 (pure uint32 arithmetic, one state step per draw) --------------------------------
struct Rng { uint sx, sy; };

// Pool state (hpt_decl.h: WfPool's per-slot arrays, the generators, the ray queues) is read once and written once per round: it goes through the caches as NON-TEMPORAL traffic, so that what L2 keeps
// from one pass to the next is the tree's nodes (measured on the 1 M-triangle interior: streaming schedule 308 -> 318 Mpaths/s).
#ifndef HPT_POOL_NT
#define HPT_POOL_NT 1
#endif
typedef float nt_f4 __attribute__((ext_vector_type(4)));
HPT_DEV float4 ldP(const float4* p) { if (HPT_POOL_NT) { const nt_f4 v = __builtin_nontemporal_load((const nt_f4*)p); return make_float4(v.x, v.y, v.z, v.w); } return *p; }
HPT_DEV uint   ldP(const uint* p)   { if (HPT_POOL_NT) return __builtin_nontemporal_load(p); return *p; }
HPT_DEV float  ldP(const float* p)  { if (HPT_POOL_NT) return __builtin_nontemporal_load(p); return *p; }
HPT_DEV void   stP(float4* p, const float4 v) { if (HPT_POOL_NT) { nt_f4 x; x.x = v.x; x.y = v.y; x.z = v.z; x.w = v.w; __builtin_nontemporal_store(x, (nt_f4*)p); } else *p = v; }
HPT_DEV void   stP(uint* p, const uint v)     { if (HPT_POOL_NT) __builtin_nontemporal_store(v, p); else *p = v; }
HPT_DEV void   stP(float* p, const float v)   { if (HPT_POOL_NT) __builtin_nontemporal_store(v, p); else *p = v; }
typedef uint nt_u2 __attribute__((ext_vector_type(2)));
HPT_DEV Rng    ldP(const Rng* p)    { if (HPT_POOL_NT) { const nt_u2 v = __builtin_nontemporal_load((const nt_u2*)p); Rng g; g.sx = v.x; g.sy = v.y; return g; } return *p; }
HPT_DEV void   stP(Rng* p, const Rng g)       { if (HPT_POOL_NT) { nt_u2 v; v.x = g.sx; v.y = g.sy; __builtin_nontemporal_store(v, (nt_u2*)p); } else *p = g; }
HPT_DEV uint2  ldP(const uint2* p)  { if (HPT_POOL_NT) { const nt_u2 v = __builtin_nontemporal_load((const nt_u2*)p); return make_uint2(v.x, v.y); } return *p; }
HPT_DEV void   stP(uint2* p, const uint2 g)   { if (HPT_POOL_NT) { nt_u2 v; v.x = g.x; v.y = g.y; __builtin_nontemporal_store(v, (nt_u2*)p); } else *p = g; }

HPT_DEV uint rng_next(Rng& g)
{
  const uint x = g.sx * 17u + g.sy * 13123u;
  g.sx = (x << 13) ^ x;
  g.sy ^= (x << 7);
  return x;
}
HPT_DEV Rng rng_init(uint seed)                                       // RandomGenInit (crandom.h:25-36); seed = tid >= 0
{
  Rng g;
  g.sx = (seed * (seed * seed * 15731u + 74323u) + 871483u);
  g.sy = (seed * (seed * seed * 13734u + 37828u) + 234234u);
  const uint warm = seed % 7u;
  for (uint i = 0; i < warm; i++) rng_next(g);
  return g;
}
HPT_DEV V4 rng_float4(Rng& g)
{
  const uint x = rng_next(g);
  const uint x1 = (x * (x * x * 15731u + 74323u) + 871483u);
  const uint y1 = (x * (x * x * 13734u + 37828u) + 234234u);
  const uint z1 = (x * (x * x * 11687u + 26461u) + 137589u);
  const uint w1 = (x * (x * x * 15707u + 789221u) + 1376312589u);
  const float scale = (1.0f / 4294967296.0f);
  return v4((float)x1 * scale, (float)y1 * scale, (float)z1 * scale, (float)w1 * scale);
}
HPT_DEV float rng_float1(Rng& g)
{
  const uint x = rng_next(g);
  const uint t = (x * (x * x * 15731u + 74323u) + 871483u);
  return ((float)t) * (1.0f / 4294967296.0f);
}

// ---- include/cglobals.h ---------------------------------------------------------------------------------------------
HPT_DEV void coordinateSystemV2(V3 n, V3& s, V3& t)                    // :120-132
{
  const float sign = n.z >= 0 ? 1.0f : -1.0f;
  const float a = -(1.0f / (sign + n.z));
  const float b = n.x * n.y * a;
  const float tmp = (n.z >= 0 ? n.x * n.x * a : -n.x * n.x * a);
  s = v3(tmp + 1.0f, n.z >= 0 ? b : -b, n.z >= 0 ? -n.x : n.x);
  t = v3(b, n.y * n.y * a + sign, -n.y);
}

HPT_DEV V3 mapSampleToCosineDistribution(float r1, float r2, V3 direction, V3 hit_norm, float power)   // :143-181
{
  if (power >= 1e6f) return direction;
  const float sin_phi = sinf(HPT_TWOPI * r1);
  const float cos_phi = cosf(HPT_TWOPI * r1);
  // pow(x, 1 / (power + 1)): every caller on the path passes power = 1 (Lambert), where the exponent is exactly 0.5 and the correctly rounded
  // square root is the same function to the last bit or one off it - the distance device powf and glibc powf keep from each other anyway -
  // at a tenth of the instructions
  const float cos_theta = (power == 1.0f) ? sqrtf_(1.0f - r2) : powf(1.0f - r2, 1.0f / (power + 1.0f));
  const float sin_theta = sqrtf_(1.0f - cos_theta * cos_theta);
  const V3 dev = v3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
  V3 nx, nz;
  coordinateSystemV2(direction, nx, nz);
  const V3 ny = nz;                 // the reference swaps ny and nz after building the frame
  const V3 nz2 = direction;
  V3 res = nx * dev.x + ny * dev.y + nz2 * dev.z;
  const float invSign = dot(direction, hit_norm) > 0.0f ? 1.0f : -1.0f;
  if (invSign * dot(res, hit_norm) < 0.0f)
    res = (-1.0f) * nx * dev.x + ny * dev.y - nz2 * dev.z;
  return res;
}

HPT_DEV V2 mapSamplesToDisc(V2 xy)                                     // :188-231
{
  const float x = xy.x, y = xy.y;
  float r = 0, phi = 0;
  if (x > y && x > -y)  { r = x;  phi = 0.25f * 3.141592654f * (y / x); }
  if (x < y && x > -y)  { r = y;  phi = 0.25f * 3.141592654f * (2.0f - x / y); }
  if (x < y && x < -y)  { r = -x; phi = 0.25f * 3.141592654f * (4.0f + y / x); }
  if (x > y && x < -y)  { r = -y; phi = 0.25f * 3.141592654f * (6 - x / y); }
  const float sin_phi = sinf(phi), cos_phi = cosf(phi);
  return v2(r * sin_phi, r * cos_phi);
}

HPT_DEV float epsilonOfPos(V3 p) { return smax(smax(absf(p.x), smax(absf(p.y), absf(p.z))), 2.0f * 1e-5f) * 1e-5f; }   // :233
HPT_DEV V3 offsRayPos(V3 hitPos, V3 surfaceNorm, V3 sampleDir)         // :242-247
{
  const float signOfNormal2 = dot(sampleDir, surfaceNorm) < 0.0f ? -1.0f : 1.0f;
  const float offsetEps = epsilonOfPos(hitPos);
  return hitPos + signOfNormal2 * offsetEps * surfaceNorm;
}
HPT_DEV float pdfAtoW(float aPdfA, float aDist, float aCosThere) { return (aPdfA * aDist * aDist) / smax(aCosThere, 1e-30f); }   // :265-268
HPT_DEV float maxcomp(V3 v) { return smax(v.x, smax(v.y, v.z)); }       // :275
HPT_DEV float misPower1(float p) { return finitef(p) ? absf(p) : 0.0f; }   // :277
HPT_DEV float misWeightHeuristic(float a, float b)                      // :278-282
{
  const float w = misPower1(a) / smax(misPower1(a) + misPower1(b), 1e-30f);
  return finitef(w) ? w : 0.0f;
}
HPT_DEV V2 mulRows2x4(const float* row0, const float* row1, V2 v)       // :315-321
{
  return v2(row0[0] * v.x + row0[1] * v.y + row0[3], row1[0] * v.x + row1[1] * v.y + row1[3]);
}
HPT_DEV V2 sphereMapTo2DTexCoord(V3 ray_dir)                            // :335-362
{
  const float x = ray_dir.z, y = ray_dir.x, z = -ray_dir.y;
  const float theta = acosf(z);
  float phi = atan2f(y, x);
  if (phi < 0.0f) phi += 2.0f * HPT_PI;
  return v2(clampf(phi * 0.5f * HPT_INV_PI, 0.0f, 1.0f), clampf(theta * HPT_INV_PI, 0.0f, 1.0f));
}

// ---- textures: LiteImage bilinear sampler as restated in-tree by Tex2DFetchAD (diff_render/integrator_dr.cpp:60-161) --
struct Taps { int off[4]; float w[4]; float fx, fy; uint base, ch; };   // fx, fy: the fractions the weights are products of; base / ch: first float of a parameter texture in a_data, its channels (texFetchAD)

HPT_DEV int wrapi(int p, int n) { const int r = p % n; return r < 0 ? r + n : r; }

HPT_DEV Taps bilinearTaps(uint w, uint h, uint addrU, uint addrV, V2 uv)
{
  float ffx = uv.x * float(w) - 0.5f;
  float ffy = uv.y * float(h) - 0.5f;
  if (addrU == 2 && ffx < 0) ffx = 0.0f;
  if (addrV == 2 && ffy < 0) ffy = 0.0f;
  const int px = (int)ffx, py = (int)ffy;
  const float fx = absf(ffx - (float)px), fy = absf(ffy - (float)py);
  const float fx1 = 1.0f - fx, fy1 = 1.0f - fy;
  const int sx = (ffx > 0.0f) ? 1 : -1, sy = (ffy > 0.0f) ? 1 : -1;
  const bool pw = (w & (w - 1u)) == 0u, ph = (h & (h - 1u)) == 0u;
  int x0, x1, y0, y1;
  if (pw && ph) { x0 = px & ((int)w - 1); x1 = (px + sx) & ((int)w - 1); y0 = py & ((int)h - 1); y1 = (py + sy) & ((int)h - 1); }   // == wrapi for a power of two (two's complement)
  else { x0 = wrapi(px, (int)w); x1 = wrapi(px + sx, (int)w); y0 = wrapi(py, (int)h); y1 = wrapi(py + sy, (int)h); }
  Taps r;
  r.off[0] = y0 * (int)w + x0; r.off[1] = y0 * (int)w + x1; r.off[2] = y1 * (int)w + x0; r.off[3] = y1 * (int)w + x1;
  r.w[0] = fx1 * fy1; r.w[1] = fx * fy1; r.w[2] = fx1 * fy; r.w[3] = fx * fy;
  r.fx = fx; r.fy = fy; r.base = 0u; r.ch = 0u;
  return r;
}

HPT_DEV V4 texel(const TexRec& t, int off)
{
  if (t.format == 0) {
    const uint v = ((const uint*)t.data)[off];
    const float s = 1.0f / 255.0f;
    return v4(float(v & 0xFF) * s, float((v >> 8) & 0xFF) * s, float((v >> 16) & 0xFF) * s, float(v >> 24) * s);
  }
  if (t.format == 1) { const float4 q = ((const float4*)t.data)[off]; return v4(q.x, q.y, q.z, q.w); }
  const float v = ((const float*)t.data)[off];
  return v4(v, v, v, v);
}

HPT_DEV V4 texSample(const TexRec* texs, uint texId, V2 uv)
{
  const TexRec t = texs[texId];
  V4 res;
  if (t.filter == 0) {
    int px = (int)floorf(uv.x * float(t.w)), py = (int)floorf(uv.y * float(t.h));
    px = (t.addrU == 2) ? min(max(px, 0), (int)t.w - 1) : wrapi(px, (int)t.w);
    py = (t.addrV == 2) ? min(max(py, 0), (int)t.h - 1) : wrapi(py, (int)t.h);
    res = texel(t, py * (int)t.w + px);
  } else {
    const Taps k = bilinearTaps(t.w, t.h, t.addrU, t.addrV, uv);
    const V4 a = texel(t, k.off[0]), b = texel(t, k.off[1]), c = texel(t, k.off[2]), d = texel(t, k.off[3]);
    res.x = a.x * k.w[0] + b.x * k.w[1] + c.x * k.w[2] + d.x * k.w[3];
    res.y = a.y * k.w[0] + b.y * k.w[1] + c.y * k.w[2] + d.y * k.w[3];
    res.z = a.z * k.w[0] + b.z * k.w[1] + c.z * k.w[2] + d.z * k.w[3];
    res.w = a.w * k.w[0] + b.w * k.w[1] + c.w * k.w[2] + d.w * k.w[3];
  }
  if (t.flags & 1u) { res.x = powf(res.x, 2.2f); res.y = powf(res.y, 2.2f); res.z = powf(res.z, 2.2f); }
  return res;
}

// ---- include/cmaterial.h ------------------------------------------------------------------------------------------------
HPT_DEV float safe_sqrt(float v) { return sqrtf_(smax(v, 0.0f)); }
HPT_DEV float cosPhiPBRT(V3 w, float sintheta) { return sintheta == 0.0f ? 1.0f : clampf(w.x / sintheta, -1.0f, 1.0f); }
HPT_DEV float sinPhiPBRT(V3 w, float sintheta) { return sintheta == 0.0f ? 0.0f : clampf(w.y / sintheta, -1.0f, 1.0f); }

HPT_DEV float orennayarFunc(V3 a_l, V3 a_v, V3 a_n, float a_roughness)     // :254-306
{
  const float cosTheta_wi = dot(a_l, a_n), cosTheta_wo = dot(a_v, a_n);
  const float sinTheta_wi = safe_sqrt(1.0f - cosTheta_wi * cosTheta_wi);
  const float sinTheta_wo = safe_sqrt(1.0f - cosTheta_wo * cosTheta_wo);
  const float sigma = a_roughness * HPT_PI * 0.5f;
  const float sigma2 = sigma * sigma;
  const float A = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
  const float B = 0.45f * sigma2 / (sigma2 + 0.09f);
  V3 nx, ny;
  coordinateSystemV2(a_n, nx, ny);
  float maxcos = 0.0f;
  if (sinTheta_wi > 1e-4f && sinTheta_wo > 1e-4f) {
    const V3 wo = v3(-dot(a_v, nx), -dot(a_v, ny), -dot(a_v, a_n));
    const V3 wi = v3(-dot(a_l, nx), -dot(a_l, ny), -dot(a_l, a_n));
    const float sinphii = sinPhiPBRT(wi, sinTheta_wi), cosphii = cosPhiPBRT(wi, sinTheta_wi);
    const float sinphio = sinPhiPBRT(wo, sinTheta_wo), cosphio = cosPhiPBRT(wo, sinTheta_wo);
    const float dcos = cosphii * cosphio + sinphii * sinphio;
    maxcos = smax(0.0f, dcos);
  }
  float sinalpha, tanbeta;
  if (absf(cosTheta_wi) > absf(cosTheta_wo)) { sinalpha = sinTheta_wo; tanbeta = sinTheta_wi / smax(absf(cosTheta_wi), 1e-20f); }
  else                                       { sinalpha = sinTheta_wi; tanbeta = sinTheta_wo / smax(absf(cosTheta_wo), 1e-20f); }
  return (A + B * maxcos * sinalpha * tanbeta);
}

HPT_DEV float ggxDistribution(float cosThetaNH, float alpha)             // :322-328
{
  const float alpha2 = alpha * alpha;
  const float NH_sqr = clampf(cosThetaNH * cosThetaNH, 0.0f, 1.0f);
  const float den = NH_sqr * alpha2 + (1.0f - NH_sqr);
  return alpha2 / smax(HPT_PI * den * den, 1e-6f);
}
HPT_DEV float ggxGeomShadMask(float cosThetaN, float alpha)              // :330-343
{
  const float c2 = clampf(cosThetaN * cosThetaN, 0.0f, 1.0f);
  const float tan2 = (1.0f - c2) / smax(c2, 1e-6f);
  return 2.0f / (1.0f + safe_sqrt(1.0f + alpha * alpha * tan2));
}
HPT_DEV V3 ggxSample(V2 rands, V3 v, V3 n, float roughness)              // :347-362
{
  const float roughSqr = roughness * roughness;
  V3 nx, ny;
  coordinateSystemV2(n, nx, ny);
  const V3 wo = v3(dot(v, nx), dot(v, ny), dot(v, n));
  const float phi = rands.x * HPT_TWOPI;
  const float cosTheta = clampf(safe_sqrt((1.0f - rands.y) / (1.0f + roughSqr * roughSqr * rands.y - rands.y)), 0.0f, 1.0f);
  const float sinTheta = safe_sqrt(1.0f - cosTheta * cosTheta);
  const V3 wh = v3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
  const V3 wi = 2.0f * dot(wo, wh) * wh - wo;
  return normalize(wi.x * nx + wi.y * ny + wi.z * n);
}
HPT_DEV float ggxEvalPDF(V3 l, V3 v, V3 n, float roughness)              // :364-378
{
  const float dotNV = dot(n, v), dotNL = dot(n, l);
  if (dotNV < 1e-6f || dotNL < 1e-6f) return 1.0f;
  const float roughSqr = roughness * roughness;
  const V3 h = normalize(v + l);
  const float dotNH = dot(n, h), dotHV = dot(h, v);
  const float D = ggxDistribution(dotNH, roughSqr);
  return D * dotNH / (4.0f * smax(dotHV, 1e-6f));
}
HPT_DEV float ggxEvalBSDF(V3 l, V3 v, V3 n, float roughness)             // :380-397
{
  if (absf(dot(l, n)) < 1e-5f) return 0.0f;
  const float dotNV = dot(n, v), dotNL = dot(n, l);
  if (dotNV < 1e-6f || dotNL < 1e-6f) return 0.0f;
  const float roughSqr = roughness * roughness;
  const V3 h = normalize(v + l);
  const float dotNH = dot(n, h);
  const float D = ggxDistribution(dotNH, roughSqr);
  const float G = ggxGeomShadMask(dotNV, roughSqr) * ggxGeomShadMask(dotNL, roughSqr);
  return (D * G / smax(4.0f * dotNV * dotNL, 1e-6f));
}

// Trowbridge-Reitz (cmaterial.h:405-530)
HPT_DEV float cos2Theta(V3 w) { return w.z * w.z; }
HPT_DEV float sin2Theta(V3 w) { return smax(0.0f, 1.0f - cos2Theta(w)); }
HPT_DEV float sinTheta_(V3 w) { return safe_sqrt(sin2Theta(w)); }
HPT_DEV float tan2Theta(V3 w) { return sin2Theta(w) / cos2Theta(w); }
HPT_DEV float cosPhi(V3 w) { const float s = sinTheta_(w); return (s == 0) ? 1 : clampf(w.x / s, -1.0f, 1.0f); }
HPT_DEV float sinPhi(V3 w) { const float s = sinTheta_(w); return (s == 0) ? 0 : clampf(w.y / s, -1.0f, 1.0f); }
HPT_DEV bool isinf_(float a) { return absf(a) > HPT_FLT_MAX; }
HPT_DEV float trD(V3 wm, V2 alpha)                                       // :460-470
{
  const float t2 = tan2Theta(wm);
  if (isinf_(t2)) return 0;
  const float cos4Theta = cos2Theta(wm) * cos2Theta(wm);
  if (cos4Theta < 1e-16f) return 0;
  const float e = t2 * ((cosPhi(wm) / alpha.x) * (cosPhi(wm) / alpha.x) + (sinPhi(wm) / alpha.y) * (sinPhi(wm) / alpha.y));
  return 1.0f / (HPT_PI * alpha.x * alpha.y * cos4Theta * (1 + e) * (1 + e));
}
HPT_DEV float trLambda(V3 w, V2 alpha)                                   // :477-484
{
  const float t2 = tan2Theta(w);
  if (isinf_(t2)) return 0;
  const float alpha2 = (cosPhi(w) * alpha.x) * (cosPhi(w) * alpha.x) + (sinPhi(w) * alpha.y) * (sinPhi(w) * alpha.y);
  return (safe_sqrt(1.0f + alpha2 * t2) - 1.0f) / 2.0f;
}
HPT_DEV float trG1(V3 w, V2 alpha) { return 1.0f / (1.0f + trLambda(w, alpha)); }
HPT_DEV float trG(V3 wo, V3 wi, V2 alpha) { return 1.0f / (1.0f + trLambda(wo, alpha) + trLambda(wi, alpha)); }
HPT_DEV float trPDF(V3 w, V3 wm, V2 alpha) { return trG1(w, alpha) / absf(w.z) * trD(wm, alpha) * absf(dot(w, wm)); }   // :496-504
HPT_DEV V3 trSample(V3 wo, V2 rands, V2 alpha)                            // :506-530
{
  V3 wh = normalize(v3(alpha.x * wo.x, alpha.y * wo.y, wo.z));
  if (wh.z < 0) wh = (-1.0f) * wh;
  const V3 T1 = (wh.z < 0.99999f) ? normalize(cross(v3(0, 0, 1), wh)) : v3(1, 0, 0);
  const V3 T2 = cross(wh, T1);
  const float r = safe_sqrt(rands.x);
  const float th = HPT_TWOPI * rands.y;
  V2 p = v2(r * cosf(th), r * sinf(th));
  const float h = safe_sqrt(1 - p.x * p.x);
  p.y = lerpf(h, p.y, (1 + wh.z) / 2);
  const float pz = safe_sqrt(1.0f - (p.x * p.x + p.y * p.y));
  const V3 nh = p.x * T1 + p.y * T2 + pz * wh;
  return normalize(v3(alpha.x * nh.x, alpha.y * nh.y, smax(1e-6f, nh.z)));
}

HPT_DEV float frDielectricPBRT(float cosThetaI, float etaI, float etaT)  // :536-561
{
  cosThetaI = clampf(cosThetaI, -1.0f, 1.0f);
  const bool entering = cosThetaI > 0.0f;
  if (!entering) { const float tmp = etaI; etaI = etaT; etaT = tmp; cosThetaI = absf(cosThetaI); }
  const float sinThetaI = safe_sqrt(1.0f - cosThetaI * cosThetaI);
  const float sinThetaT = etaI / etaT * sinThetaI;
  if (sinThetaT >= 1.0f) return 1.0f;
  const float cosThetaT = safe_sqrt(1.0f - sinThetaT * sinThetaT);
  const float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
  const float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
  return 0.5f * (Rparl * Rparl + Rperp * Rperp);
}

// complex helpers for FrComplexConductor (cmaterial.h:685-694); LiteMath's complex is absent: pbrt-v4 formulation
struct Cx { float re, im; };
HPT_DEV Cx cx(float re, float im) { Cx r; r.re = re; r.im = im; return r; }
HPT_DEV Cx operator+(Cx a, Cx b) { return cx(a.re + b.re, a.im + b.im); }
HPT_DEV Cx operator-(Cx a, Cx b) { return cx(a.re - b.re, a.im - b.im); }
HPT_DEV Cx operator*(Cx a, Cx b) { return cx(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
HPT_DEV Cx operator*(Cx a, float s) { return cx(a.re * s, a.im * s); }
HPT_DEV Cx operator/(Cx a, Cx b)
{
  const float scale = 1.0f / (b.re * b.re + b.im * b.im);
  return cx(scale * (a.re * b.re + a.im * b.im), scale * (a.im * b.re - a.re * b.im));
}
HPT_DEV float cnorm(Cx a) { return a.re * a.re + a.im * a.im; }
HPT_DEV Cx csqrt_(Cx z)
{
  const float n = sqrtf_(cnorm(z));
  if (n == 0.0f) return cx(0.0f, 0.0f);
  const float t1 = sqrtf_(0.5f * (n + absf(z.re)));
  const float t2 = 0.5f * z.im / t1;
  if (z.re >= 0.0f) return cx(t1, t2);
  return cx(absf(t2), __builtin_copysignf(t1, z.im));
}
HPT_DEV float frComplexConductor(float cosThetaI, Cx eta)
{
  const float sinThetaI = 1.0f - cosThetaI * cosThetaI;
  const Cx sinThetaT = cx(sinThetaI, 0.0f) / (eta * eta);
  const Cx cosThetaT = csqrt_(cx(1.0f - sinThetaT.re, -sinThetaT.im));
  const Cx r_parl = (eta * cosThetaI - cosThetaT) / (eta * cosThetaI + cosThetaT);
  const Cx ect = eta * cosThetaT;
  const Cx r_perp = cx(cosThetaI - ect.re, -ect.im) / cx(cosThetaI + ect.re, ect.im);
  return (cnorm(r_parl) + cnorm(r_perp)) / 2.0f;
}
HPT_DEV float fresnelSlick(float VdotH) { const float t = 1.0f - absf(VdotH); return (t * t) * (t * t) * t; }   // :705-709

HPT_DEV V4 frDielectricDetailedV2(float cos_theta_i, float eta)          // :646-683
{
  cos_theta_i = clampf(cos_theta_i, -1.0f, 1.0f);
  float eta_it = eta, eta_ti = 1.f / eta;
  if (cos_theta_i < 0.0f) { eta_it = eta_ti; eta_ti = eta; }
  const float cos_theta_t_sqr = -1.f * (-1.f * cos_theta_i * cos_theta_i + 1.f) * eta_ti * eta_ti + 1.f;
  const float cos_theta_i_abs = absf(cos_theta_i);
  const float cos_theta_t_abs = safe_sqrt(cos_theta_t_sqr);
  float r;
  if ((eta == 1.f) || (cos_theta_i_abs == 0.f)) r = (eta == 1.f) ? 0.f : 1.f;
  else {
    const float a_s = (-1.f * eta_it * cos_theta_t_abs + cos_theta_i_abs) / (eta_it * cos_theta_t_abs + cos_theta_i_abs);
    const float a_p = (-1.f * eta_it * cos_theta_i_abs + cos_theta_t_abs) / (eta_it * cos_theta_i_abs + cos_theta_t_abs);
    r = 0.5f * (a_s * a_s + a_p * a_p);
  }
  const float cos_theta_t = cos_theta_i >= 0 ? -cos_theta_t_abs : cos_theta_t_abs;
  return v4(r, cos_theta_t, eta_it, eta_ti);
}

// ---- BSDF results ---------------------------------------------------------------------------------------------------
// val / dir / pdf / flags / ior as in BsdfSample (cmaterial.h:9-16); dval = d val / d baseColor per channel, filled only
// by the differentiable path (gltf is the only differentiable material, diff_render/integrator_dr.cpp:461-612).
struct BsdfS { V3 val; V3 dir; float pdf; uint flags; float ior; V3 dval; };
struct BsdfE { V3 val; float pdf; V3 dval; };

HPT_DEV V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }

// include/cmat_gltf.h:6-90
// (Mcol, coatCol: the material's metal and coat colours - the spectral kernel runs the routine a second time on their fourth components)
HPT_DEV void gltfSampleAndEvalC(const MaterialRec& m, const V3 Mcol, const V3 coatCol, V4 rands, V3 v, V3 n, V3 baseColor, V3 fourParams, BsdfS& r)
{
  const uint cflags = m.cflags;
  const V3 metalCol = baseColor * Mcol;
  const float roughness = clampf(1.0f - m.data[GLTF_FLOAT_GLOSINESS] * fourParams.x, 0.0f, 1.0f);
  float metalness = m.data[GLTF_FLOAT_ALPHA] * fourParams.y;
  const float coatValue = m.data[GLTF_FLOAT_REFL_COAT] * fourParams.z;
  const float fresnelIOR = m.data[GLTF_FLOAT_IOR];
  if (cflags == GLTF_COMPONENT_METAL) metalness = 1.0f;

  V3 ggxDir; float ggxPdf, ggxVal;
  if (roughness == 0.0f) {
    const V3 pefReflDir = reflect((-1.0f) * v, n);
    const float cosThetaOut = dot(pefReflDir, n);
    ggxDir = pefReflDir;
    ggxVal = (cosThetaOut <= 1e-6f) ? 0.0f : (1.0f / smax(cosThetaOut, 1e-6f));
    ggxPdf = 1.0f;
  } else {
    ggxDir = ggxSample(v2(rands.x, rands.y), v, n, roughness);
    ggxPdf = ggxEvalPDF(ggxDir, v, n, roughness);
    ggxVal = ggxEvalBSDF(ggxDir, v, n, roughness);
  }
  const V3 lambertDir = mapSampleToCosineDistribution(rands.x, rands.y, n, n, 1.0f);
  const float lambertPdf = absf(dot(lambertDir, n)) * HPT_INV_PI;
  const float lambertVal = HPT_INV_PI;

  float pdfSelect = 1.0f;
  if (rands.z < metalness) {
    pdfSelect *= metalness;
    const float VdotH = dot(v, normalize(v + ggxDir));
    r.dir = ggxDir;
    V3 fr = metalCol, dfr = Mcol;                              // hydraFresnelCond (cmaterial.h:711-717)
    if (fresnelIOR != 0.0f) { const float s = fresnelSlick(VdotH); fr = metalCol + (v3s(1.0f) - metalCol) * s; dfr = Mcol * (1.0f - s); }
    r.val = fr * (ggxVal * metalness);
    r.dval = dfr * (ggxVal * metalness);
    r.pdf = ggxPdf;
    r.flags = (roughness == 0.0f) ? RAY_EVENT_S : RAY_FLAG_HAS_NON_SPEC;
  } else {
    pdfSelect *= 1.0f - metalness;
    const float f_i = frDielectricPBRT(absf(dot(v, n)), 1.0f, fresnelIOR);
    const float prob_specular = 0.5f * coatValue;
    const float prob_diffuse = 1.0f - prob_specular;
    if (rands.w < prob_specular) {
      pdfSelect *= prob_specular;
      r.dir = ggxDir;
      r.val = ggxVal * coatCol * (1.0f - metalness) * f_i * coatValue;
      r.dval = v3s(0.0f);
      r.pdf = ggxPdf;
      r.flags = (roughness == 0.0f) ? RAY_EVENT_S : RAY_FLAG_HAS_NON_SPEC;
    } else {
      pdfSelect *= prob_diffuse;
      r.dir = lambertDir;
      r.val = baseColor * lambertVal * (1.0f - metalness);
      r.dval = v3s(lambertVal * (1.0f - metalness));
      r.pdf = lambertPdf;
      r.flags = RAY_FLAG_HAS_NON_SPEC;
      if (coatValue > 0.0f && fresnelIOR > 0.0f) {
        const float m_fdr_int = m.data[GLTF_FLOAT_MI_FDR_INT];
        const float f_o = frDielectricPBRT(absf(dot(lambertDir, n)), 1.0f, fresnelIOR);
        const float k = lerpf(1.0f, (1.0f - f_i) * (1.0f - f_o) / (fresnelIOR * fresnelIOR * (1.0f - m_fdr_int)), coatValue);
        r.val = r.val * k;
        r.dval = r.dval * k;
      }
    }
  }
  r.pdf *= pdfSelect;
}
HPT_DEV void gltfSampleAndEval(const MaterialRec& m, V4 rands, V3 v, V3 n, V3 baseColor, V3 fourParams, BsdfS& r)
{ gltfSampleAndEvalC(m, ld3(m.colors[GLTF_COLOR_METAL]), ld3(m.colors[GLTF_COLOR_COAT]), rands, v, n, baseColor, fourParams, r); }

// include/cmat_gltf.h:93-147
HPT_DEV void gltfEvalC(const MaterialRec& m, const V3 Mcol, const V3 coatCol, V3 l, V3 v, V3 n, V3 baseColor, V3 fourParams, BsdfE& res)
{
  const uint cflags = m.cflags;
  const V3 metalCol = baseColor * Mcol;
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

  float lambertVal = HPT_INV_PI;
  const float lambertPdf = absf(dot(l, n)) * HPT_INV_PI;
  float f_i = 1.0f;
  if (coatValue > 0.0f && metalness < 1.0f && fresnelIOR > 0.0f) {
    f_i = frDielectricPBRT(absf(dot(v, n)), 1.0f, fresnelIOR);
    const float f_o = frDielectricPBRT(absf(dot(l, n)), 1.0f, fresnelIOR);
    const float m_fdr_int = m.data[GLTF_FLOAT_MI_FDR_INT];
    const float coeff = lerpf(1.0f, (1.f - f_i) * (1.f - f_o) / (fresnelIOR * fresnelIOR * (1.f - m_fdr_int)), coatValue);
    lambertVal *= coeff;
  }
  V3 fConductor = metalCol, dfc = Mcol;
  if (fresnelIOR != 0.0f) { const float s = fresnelSlick(VdotH); fConductor = metalCol + (v3s(1.0f) - metalCol) * s; dfc = Mcol * (1.0f - s); }
  const V3 specularColor = fConductor * ggxVal;
  const float prob_specular = 0.5f * coatValue;
  const float prob_diffuse = 1.0f - prob_specular;
  const V3 dielectricVal = baseColor * lambertVal + ggxVal * coatCol * f_i * coatValue;
  const float dielectricPdf = lambertPdf * prob_diffuse + ggxPdf * prob_specular;
  res.val = specularColor * metalness + dielectricVal * (1.0f - metalness);
  res.dval = dfc * (ggxVal * metalness) + v3s(lambertVal * (1.0f - metalness));
  res.pdf = metalness * ggxPdf + (1.0f - metalness) * dielectricPdf;
}
HPT_DEV void gltfEval(const MaterialRec& m, V3 l, V3 v, V3 n, V3 baseColor, V3 fourParams, BsdfE& res)
{ gltfEvalC(m, ld3(m.colors[GLTF_COLOR_METAL]), ld3(m.colors[GLTF_COLOR_COAT]), l, v, n, baseColor, fourParams, res); }

// include/cmat_diffuse.h:8-39
HPT_DEV void diffuseSampleAndEval(const MaterialRec& m, V3 reflSpec, V4 rands, V3 v, V3 n, BsdfS& r)
{
  const V3 lambertDir = mapSampleToCosineDistribution(rands.x, rands.y, n, n, 1.0f);
  r.dir = lambertDir;
  r.val = HPT_INV_PI * reflSpec;
  r.pdf = absf(dot(lambertDir, n)) * HPT_INV_PI;
  r.flags = RAY_FLAG_HAS_NON_SPEC;
  if ((m.cflags & GLTF_COMPONENT_ORENNAYAR) != 0) r.val = r.val * orennayarFunc(lambertDir, (-1.0f) * v, n, m.data[0]);
}
HPT_DEV void diffuseEval(const MaterialRec& m, V3 reflSpec, V3 l, V3 v, V3 n, BsdfE& res)
{
  float lambertVal = HPT_INV_PI;
  if ((m.cflags & GLTF_COMPONENT_ORENNAYAR) != 0) lambertVal *= orennayarFunc(l, v, n, m.data[0]);
  res.val = lambertVal * reflSpec;
  res.pdf = absf(dot(l, n)) * HPT_INV_PI;
}

// include/cmat_conductor.h (RGB mode: eta / k are the same scalar in every channel, integrator_spectrum.cpp:25-29)
HPT_DEV void conductorSmoothSampleAndEval(const MaterialRec& m, float eta, float k, V3 v, V3 n, BsdfS& r)   // :7-28
{
  const V3 pefReflDir = reflect((-1.0f) * v, n);
  const float cosThetaOut = dot(pefReflDir, n);
  float val = frComplexConductor(cosThetaOut, cx(eta, k));
  val = (cosThetaOut <= 1e-6f) ? 0.0f : (val / smax(cosThetaOut, 1e-6f));
  r.val = v3s(val) * ld3(m.colors[0]);
  r.dir = pefReflDir;
  r.pdf = 1.0f;
  r.flags = RAY_EVENT_S;
}
HPT_DEV float conductorRoughEvalInternal(V3 wo, V3 wi, V3 wm, V2 alpha, Cx ior)   // :42-58
{
  if (wo.z * wi.z < 0) return 0.0f;
  const float cosTheta_o = absf(wo.z), cosTheta_i = absf(wi.z);
  if (cosTheta_i == 0 || cosTheta_o == 0) return 0.0f;
  const float F = frComplexConductor(absf(dot(wo, wm)), ior);
  return trD(wm, alpha) * F * trG(wo, wi, alpha) / (4.0f * cosTheta_i * cosTheta_o);
}
HPT_DEV void conductorRoughSampleAndEval(const MaterialRec& m, float eta, float k, V4 rands, V3 v, V3 n, V3 alpha_tex, BsdfS& r)   // :61-100
{
  if (v.z == 0) return;
  const V2 alpha = v2(smin(m.data[0], alpha_tex.x), smin(m.data[1], alpha_tex.y));
  V3 nx, ny;
  coordinateSystemV2(n, nx, ny);
  const V3 wo = v3(dot(v, nx), dot(v, ny), dot(v, n));
  if (wo.z == 0) return;
  const V3 wm = trSample(wo, v2(rands.x, rands.y), alpha);
  const V3 wi = reflect((-1.0f) * wo, wm);
  if (wo.z * wi.z < 0) return;
  const float val = conductorRoughEvalInternal(wo, wi, wm, alpha, cx(eta, k));
  r.val = v3s(val) * ld3(m.colors[0]);
  r.dir = normalize(wi.x * nx + wi.y * ny + wi.z * n);
  r.pdf = trPDF(wo, wm, alpha) / (4.0f * absf(dot(wo, wm)));
  r.flags = RAY_FLAG_HAS_NON_SPEC;
}
HPT_DEV void conductorRoughEval(const MaterialRec& m, float eta, float k, V3 l, V3 v, V3 n, V3 alpha_tex, BsdfE& res)   // :103-137
{
  const V2 alpha = v2(smin(m.data[0], alpha_tex.x), smin(m.data[1], alpha_tex.y));
  V3 nx, ny;
  coordinateSystemV2(n, nx, ny);
  const V3 wo = v3(dot(v, nx), dot(v, ny), dot(v, n));
  const V3 wi = v3(dot(l, nx), dot(l, ny), dot(l, n));
  if (wo.z * wi.z < 0.0f) return;
  V3 wm = wo + wi;
  if (dot(wm, wm) == 0) return;
  wm = normalize(wm);
  const float val = conductorRoughEvalInternal(wo, wi, wm, alpha, cx(eta, k));
  res.val = v3s(val) * ld3(m.colors[0]);
  if (dot(wm, v3(0.0f, 0.0f, 1.0f)) < 0.f) wm = (-1.0f) * wm;       // FaceForward
  res.pdf = trPDF(wo, wm, alpha) / (4.0f * absf(dot(wo, wm)));
}

// include/cmat_dielectric.h:8-56
HPT_DEV void dielectricSmoothSampleAndEval(const MaterialRec& m, float etaInt, float _extIOR, V4 rands, V3 v, V3 n, BsdfS& r)
{
  const float extIOR = m.data[0];
  if ((r.flags & RAY_FLAG_HAS_INV_NORMAL) != 0) n = (-1.0f) * n;
  V3 s, t;
  coordinateSystemV2(n, s, t);
  const V3 wi = v3(dot(v, s), dot(v, t), dot(v, n));
  const float eta = etaInt / extIOR;
  const V4 fr = frDielectricDetailedV2(wi.z, eta);
  const float R = fr.x, cos_theta_t = fr.y, eta_ti = fr.w;
  const float T = 1 - R;
  if (rands.x < R) {
    const V3 wo = v3(-wi.x, -wi.y, wi.z);
    r.val = v3s(R);
    r.pdf = R;
    r.dir = normalize(wo.x * s + wo.y * t + wo.z * n);
    r.flags |= RAY_EVENT_S;
    r.ior = _extIOR;
  } else {
    const V3 wo = v3(-eta_ti * wi.x, -eta_ti * wi.y, cos_theta_t);   // refract (cmaterial.h:917-920)
    r.val = v3s((eta_ti * eta_ti) * T);
    r.pdf = T;
    r.dir = normalize(wo.x * s + wo.y * t + wo.z * n);
    r.flags |= (RAY_EVENT_S | RAY_EVENT_T);
    r.ior = (_extIOR == etaInt) ? extIOR : etaInt;
  }
  r.val = r.val / smax(absf(dot(r.dir, n)), 1e-6f);
}

// include/cmat_glass.h:190-277 - the legacy Hydra glass (specular reflection / refraction chosen by the Fresnel term; glassEval is zero)
HPT_DEV V3 reflect2(V3 dir, V3 n) { return normalize(dir - 2.0f * dot(dir, n) * n); }
HPT_DEV V3 refract2(V3 dir, V3 n, float relativeIor)
{
  const float cosi = dot(dir, n);
  const float eta = 1.0f / relativeIor;
  const float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
  if (k < 0) return reflect2(dir, n);
  return normalize(eta * dir - (eta * cosi + sqrtf_(k)) * n);
}
HPT_DEV float fresnel2(V3 v, V3 n, float ior)
{
  const float cosi = dot(v, n);
  const float sint = sqrtf_(1.0f - cosi * cosi) / ior;
  if (sint > 1.0f) return 1.0f;
  const float cost = sqrtf_(1.0f - sint * sint);
  const float Rp = (ior * cosi - cost) / (ior * cosi + cost);
  const float Rs = (cosi - ior * cost) / (cosi + ior * cost);
  return (Rp * Rp + Rs * Rs) * 0.5f;
}
HPT_DEV void glassSampleAndEval(const MaterialRec& m, V4 rands, V3 viewDir, V3 normal, BsdfS& r, float& misPrevIor)
{
  const V3 colorReflect = ld3(m.colors[0]), colorTransp = ld3(m.colors[1]);
  const float ior = m.data[2];
  const V3 rayDir = (-1.0f) * viewDir;
  float relativeIor = ior / misPrevIor;
  if ((r.flags & RAY_FLAG_HAS_INV_NORMAL) != 0) { if (misPrevIor == ior) relativeIor = 1.0f / ior; }
  const float fresnel = fresnel2(viewDir, normal, relativeIor);
  V3 dir;
  if (rands.w < fresnel) { dir = reflect2(rayDir, normal); r.val = colorReflect; r.flags |= RAY_EVENT_S; }
  else { dir = refract2(rayDir, normal, relativeIor); r.val = colorTransp; misPrevIor = ior; r.flags |= (RAY_EVENT_S | RAY_EVENT_T); }
  const float cosThetaOut = absf(dot(dir, normal));
  r.val = r.val / smax(cosThetaOut, 1e-6f);
  r.dir = dir;
  r.pdf = 1.0f;
}

// ---- Mitsuba-style GGX helpers (include/cmaterial.h:563-583, 746-905) and MAT_TYPE_PLASTIC (include/cmat_plastic.h) ---------------------
enum : uint { MAT_TYPE_PLASTIC = 5 };                              // include/cmaterial.h:42; data: roughness, ior ratio, spec weight, reflectance (:133-136)
enum : int { MI_ROUGH_TRANSMITTANCE_RES = 64 };                   // include/cglobals.h:18
#define HPT_EPSILON_32 5.960464477539063E-8f               // include/cglobals.h:24
HPT_DEV float FrDielectric(float cosTheta_i, float eta)        // :563-583
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
HPT_DEV V2 square_to_uniform_disk_concentric(V2 s)             // :770-791
{
  const float x = 2.f * s.x - 1.f, y = 2.f * s.y - 1.f;
  float phi, r;
  if (x == 0 && y == 0) { r = phi = 0; }
  else if (x * x > y * y) { r = x; phi = (HPT_PI / 4.f) * (y / x); }
  else { r = y; phi = (HPT_PI / 2.f) - (x / y) * (HPT_PI / 4.f); }
  return v2(r * cosf(phi), r * sinf(phi));
}
HPT_DEV V3 square_to_cosine_hemisphere(V2 s)                   // :793-803
{
  const V2 p = square_to_uniform_disk_concentric(s);
  return v3(p.x, p.y, safe_sqrt(1.f - (p.x * p.x + p.y * p.y)));
}
HPT_DEV float smith_g1(V3 v, V3 m, V2 alpha)                   // :813-833
{
  const float xy_alpha_2 = alpha.x * v.x * alpha.x * v.x + alpha.y * v.y * alpha.y * v.y, tan_theta_alpha_2 = xy_alpha_2 / (v.z * v.z);
  float result = 2.f / (1.f + safe_sqrt(1.f + tan_theta_alpha_2));
  if (xy_alpha_2 == 0.f) result = 1.f;
  if (v.z * dot(v, m) <= 0.f) result = 0.f;
  return result;
}
HPT_DEV float eval_microfacet_ggx(V3 m, V2 alpha)              // :840-857, type 1
{
  const float alpha_uv = alpha.x * alpha.y, cos_theta = m.z;
  const float ax = m.x / alpha.x, ay = m.y / alpha.y;
  const float q = ax * ax + ay * ay + m.z * m.z;
  const float result = 1.f / (HPT_PI * alpha_uv * (q * q));
  return result * cos_theta > 1e-20f ? result : 0.f;
}
HPT_DEV V3 sample_visible_normal(V3 wi, V2 rands, V2 alpha)    // :859-900 (the pdf in .w is not used by the plastic)
{
  const V3 wi_p = normalize(v3(alpha.x * wi.x, alpha.y * wi.y, wi.z));
  const float sin_theta2 = wi_p.x * wi_p.x + wi_p.y * wi_p.y;       // sincos_phi (:751-762)
  const float inv_sin_theta = 1.f / safe_sqrt(sin_theta2);
  float rx = wi_p.x * inv_sin_theta, ry = wi_p.y * inv_sin_theta;
  if (absf(sin_theta2) <= 4.f * HPT_EPSILON_32) { rx = 1.f; ry = 0.f; } else { rx = clampf(rx, -1.f, 1.f); ry = clampf(ry, -1.f, 1.f); }
  const float sin_phi = ry, cos_phi = rx, cos_theta = wi_p.z;
  V2 p = square_to_uniform_disk_concentric(rands);                   // sample_visible_11 (:859-874)
  const float s = 0.5f * (1.f + cos_theta);
  p.y = lerpf(safe_sqrt(1.f - p.x * p.x), p.y, s);
  const float x = p.x, y = p.y, z = safe_sqrt(1.f - (p.x * p.x + p.y * p.y));
  const float sin_theta_i = safe_sqrt(1.f - cos_theta * cos_theta);
  const float norm = 1.f / (sin_theta_i * y + cos_theta * z);
  const float slx = (cos_theta * y - sin_theta_i * z) * norm, sly = x * norm;
  const float sx = (cos_phi * slx - sin_phi * sly) * alpha.x, sy = (sin_phi * slx + cos_phi * sly) * alpha.y;
  return normalize(v3(-sx, -sy, 1.0f));
}
HPT_DEV float plasticTransmittance(const float* transmittance, uint trOffset, float cos_theta)   // the lerp_gather written out at cmat_plastic.h:27-38
{
  float x = cos_theta;
  x *= float(MI_ROUGH_TRANSMITTANCE_RES - 1);
  const uint index = min(uint(x), uint(MI_ROUGH_TRANSMITTANCE_RES - 2));
  const float v0 = transmittance[trOffset + index], v1 = transmittance[trOffset + index + 1];
  return lerpf(v0, v1, x - float(index));
}
HPT_DEV void plasticSampleAndEval(const MaterialRec& m, V3 a_reflSpec, V4 rands, V3 v, V3 n, BsdfS& r, const float* transmittance, uint trOffset)   // cmat_plastic.h:7-99
{
  const float alpha = m.data[0], eta = m.data[1], spec_weight = m.data[2], internal_refl = m.data[3];
  const uint nonlinear = m.nonlinear;
  const V2 alpha2 = v2(alpha, alpha);
  V3 s = n, t = n;
  coordinateSystemV2(n, s, t);
  const V3 wi = v3(dot(v, s), dot(v, t), dot(v, n));
  if (wi.z <= 0) return;
  const float cos_theta_i = smax(wi.z, HPT_EPSILON_32);
  const float t_i = plasticTransmittance(transmittance, trOffset, cos_theta_i);
  float prob_specular = (1.f - t_i) * spec_weight, prob_diffuse = t_i * (1.f - spec_weight);
  if (prob_diffuse != 0.0f && prob_specular != 0.0f) { prob_specular = prob_specular / (prob_specular + prob_diffuse); prob_diffuse = 1.f - prob_specular; }
  else { prob_diffuse = 1.0f; prob_specular = 0.0f; }
  const bool sample_specular = rands.z < prob_specular;
  V3 wo = v3(0, 0, 0);
  if (sample_specular) {
    const V3 wm = sample_visible_normal(wi, v2(rands.x, rands.y), alpha2);
    const V3 d = (-1.0f) * wi;
    wo = d - 2.0f * dot(d, wm) * wm;                                  // reflect((-1)*wi, wm)
  } else wo = square_to_cosine_hemisphere(v2(rands.x, rands.y));
  if (cos_theta_i * wo.z <= 0) return;
  const float cos_theta_o = smax(wo.z, HPT_EPSILON_32);
  const V3 H = normalize(wo + wi);
  const float D = eval_microfacet_ggx(H, alpha2);
  float pdf = D * smith_g1(wi, H, alpha2) / (4.f * cos_theta_i);
  pdf *= prob_specular;
  pdf += prob_diffuse * HPT_INV_PI * cos_theta_o;
  const float F = FrDielectric(dot(wi, H), eta);
  const float G = smith_g1(wi, H, alpha2) * smith_g1(wo, H, alpha2);
  const float val = F * D * G / (4.f * cos_theta_i * cos_theta_o);
  const float t_o = plasticTransmittance(transmittance, trOffset, cos_theta_o);
  const V3 diffuse = a_reflSpec / (v3(1.f, 1.f, 1.f) - (nonlinear > 0 ? (a_reflSpec * internal_refl) : v3(internal_refl, internal_refl, internal_refl)));
  const float inv_eta_2 = 1.f / (eta * eta);
  r.dir = normalize(wo.x * s + wo.y * t + wo.z * n);
  r.val = v3(val, val, val) + diffuse * (HPT_INV_PI * inv_eta_2 * t_i * t_o);
  r.pdf = pdf;
  r.flags = RAY_FLAG_HAS_NON_SPEC;
}
HPT_DEV void plasticEval(const MaterialRec& m, V3 a_reflSpec, V3 l, V3 v, V3 n, BsdfE& r, const float* transmittance, uint trOffset)   // cmat_plastic.h:102-191
{
  const float alpha = m.data[0], eta = m.data[1], spec_weight = m.data[2], internal_refl = m.data[3];
  const uint nonlinear = m.nonlinear;
  const V2 alpha2 = v2(alpha, alpha);
  V3 s = n, t = n;
  coordinateSystemV2(n, s, t);
  const V3 wo = v3(dot(l, s), dot(l, t), dot(l, n)), wi = v3(dot(v, s), dot(v, t), dot(v, n));
  if (wi.z * wo.z <= 0) return;
  const float cos_theta_i = smax(wi.z, HPT_EPSILON_32), cos_theta_o = smax(wo.z, HPT_EPSILON_32);
  const float t_i = plasticTransmittance(transmittance, trOffset, cos_theta_i);
  float prob_specular = (1.f - t_i) * spec_weight, prob_diffuse = t_i * (1.f - spec_weight);
  if (prob_diffuse != 0.0f && prob_specular != 0.0f) { prob_specular = prob_specular / (prob_specular + prob_diffuse); prob_diffuse = 1.f - prob_specular; }
  else { prob_diffuse = 1.0f; prob_specular = 0.0f; }
  const V3 H = normalize(wo + wi);
  const float D = eval_microfacet_ggx(H, alpha2);
  const float smith_g1_wi = smith_g1(wi, H, alpha2);
  float pdf = D * smith_g1_wi / (4.f * cos_theta_i);
  pdf *= prob_specular;
  pdf += prob_diffuse * HPT_INV_PI * cos_theta_o;
  const float F = FrDielectric(dot(wi, H), eta);
  const float G = smith_g1(wo, H, alpha2) * smith_g1_wi;
  const float val = F * D * G / (4.f * cos_theta_i * cos_theta_o);
  const float t_o = plasticTransmittance(transmittance, trOffset, cos_theta_o);
  const V3 diffuse = a_reflSpec / (v3(1.f, 1.f, 1.f) - (nonlinear > 0 ? (a_reflSpec * internal_refl) : v3(internal_refl, internal_refl, internal_refl)));
  const float inv_eta_2 = 1.f / (eta * eta);
  r.val = v3(val, val, val) + diffuse * (HPT_INV_PI * inv_eta_2 * t_i * t_o);
  r.pdf = pdf;
}

// ---- lights: include/clight.h:58-126, integrator_pt_lgt.cpp:21-173 ------------------------------------------------------
struct LightSam { V3 pos, norm; float pdf; bool isOmni, hasIES; };

HPT_DEV LightSam lightSampleRev(const LightRec& L, V3 rands, V3 illuminationPoint)
{
  LightSam r;
  r.pdf = 1.0f; r.isOmni = false; r.hasIES = (L.iesId != 0xFFFFFFFFu);
  const uint g = L.geomType;
  if (g == LIGHT_GEOM_DIRECT) {
    const V3 norm = ld3(L.norm);
    r.pos = illuminationPoint - norm * 100000.0f; r.norm = norm; r.hasIES = false;
  } else if (g == LIGHT_GEOM_SPHERE) {
    const float theta = 2.0f * HPT_PI * rands.x;
    const float phi = acosf(1.0f - 2.0f * rands.y);
    const float x = sinf(phi) * cosf(theta), y = sinf(phi) * sinf(theta), z = cosf(phi);
    const V3 lcenter = ld3(L.pos);
    const V3 samplePos = lcenter + (L.size[0] * 1.000001f) * v3(x, y, z);
    r.pos = samplePos; r.norm = normalize(samplePos - lcenter);
  } else if (g == LIGHT_GEOM_POINT) {
    r.pos = ld3(L.pos); r.norm = ld3(L.norm); r.isOmni = (L.distType == LIGHT_DIST_OMNI);
  } else {                                                   // rect / disc (clight.h:67-84)
    V2 off = v2((2.0f * (-0.5f + rands.x)) * L.size[0], (2.0f * (-0.5f + rands.y)) * L.size[1]);
    if (g == LIGHT_GEOM_DISC) {
      const V2 d = mapSamplesToDisc(v2(rands.x * 2.0f - 1.0f, rands.y * 2.0f - 1.0f));
      off = v2(d.x * L.size[0], d.y * L.size[0]);
    }
    const V3 lp = ld3(L.pos);
    r.pos = mul3x3(L.matrix, v3(off.x, 0.0f, off.y)) + lp + epsilonOfPos(lp) * ld3(L.norm);
    r.norm = ld3(L.norm);
  }
  return r;
}

HPT_DEV float lightEvalPDF(const LightRec& L, V3 illuminationPoint, V3 ray_dir, V3 lpos, V3 lnorm, float a_envPdf)   // :71-107
{
  const uint g = L.geomType;
  if (g == LIGHT_GEOM_ENV) return a_envPdf;
  const float hitDist = length(illuminationPoint - lpos);
  const float cosValTmp = dot(ray_dir, -1.0f * lnorm);
  float cosVal = 1.0f;
  if (g == LIGHT_GEOM_SPHERE) { const V3 dirToV = normalize(lpos - illuminationPoint); cosVal = absf(dot(dirToV, lnorm)); }
  else if (g == LIGHT_GEOM_POINT) { if (L.distType == LIGHT_DIST_LAMBERT) cosVal = smax(cosValTmp, 0.0f); }
  else cosVal = (L.iesId == 0xFFFFFFFFu) ? smax(cosValTmp, 0.0f) : absf(cosValTmp);
  return pdfAtoW(L.pdfA, hitDist, cosVal);
}

HPT_DEV V3 lightIntensity(const DevScene& S, const LightRec& L, V3 a_rayPos, V3 a_rayDir)   // :109-173 (RGB mode)
{
  V3 lightColor = ld3(L.intensity);
  lightColor = lightColor * L.mult;
  if (L.iesId != 0xFFFFFFFFu) {
    if ((L.flags & LIGHT_FLAG_POINT_AREA) != 0) a_rayDir = normalize(ld3(L.pos) - a_rayPos);
    const V4 dt = mul4x4(L.iesMatrix, v4(a_rayDir.x, a_rayDir.y, a_rayDir.z, 0.0f));
    const V2 tc = sphereMapTo2DTexCoord((-1.0f) * v3(dt.x, dt.y, dt.z));
    const V4 texColor = texSample(S.textures, L.iesId, tc);
    lightColor = lightColor * v3(texColor.x, texColor.y, texColor.z);
  }
  if (L.distType == LIGHT_DIST_SPOT) {
    const float cos_theta = smax(-dot(a_rayDir, ld3(L.norm)), 0.0f);
    const float tVal = (cos_theta - L.lightCos2) / (L.lightCos1 - L.lightCos2);    // mylocalsmoothstep (clight.h:220-225)
    const float t = smin(smax(tVal, 0.0f), 1.0f);
    lightColor = lightColor * (t * t * (3.0f - 2.0f * t));
    if ((L.flags & LIGHT_FLAG_PROJECTIVE) != 0 && L.texId != 0xFFFFFFFFu) {       // :153-161: a slide projector - the texture through the light's view-projection
      const V4 clip = mul4x4(L.iesMatrix, v4(a_rayPos.x, a_rayPos.y, a_rayPos.z, 1.0f));
      const V3 ndc = v3(clip.x, clip.y, clip.z) / clip.w;
      const V4 texColor = texSample(S.textures, L.texId, v2(ndc.x * 0.5f + 0.5f, ndc.y * 0.5f + 0.5f));
      lightColor = lightColor * v3(texColor.x, texColor.y, texColor.z);
    }
  }
  else if (L.texId != 0xFFFFFFFFu) {                                    // :163-170: the environment map seen along the shadow ray
    const V2 tc = mulRows2x4(L.samplerRow0, L.samplerRow1, sphereMapTo2DTexCoord(a_rayDir));
    const V4 texColor = texSample(S.textures, L.texId, tc);
    lightColor = lightColor * v3(texColor.x, texColor.y, texColor.z);
  }
  return lightColor;
}

// ---- sampled environment map: include/clight.h:128-218, integrator_pt_lgt.cpp:30-55, 175-236, cglobals.h:360-373 -----------------------------
HPT_DEV int selectIndexPropToOpt(float a_r, const float* a_accum, int N, float& pdf)
{
  int leftBound = 0, rightBound = N - 2, counter = 0, currPos = -1;
  const float x = a_r * a_accum[N - 1];
  while (rightBound - leftBound > 1 && counter < 50) {
    const int currSize = rightBound + leftBound;
    const int currPos1 = (currSize % 2 == 0) ? (currSize + 1) / 2 : (currSize + 0) / 2;
    const float a = a_accum[currPos1 + 0], b = a_accum[currPos1 + 1];
    if (a < x && x <= b) { currPos = currPos1; break; }
    else if (x <= a) rightBound = currPos1;
    else if (x > b) leftBound = currPos1;
    counter++;
  }
  if (currPos < 0) {
    const float a1 = a_accum[leftBound + 0], b1 = a_accum[leftBound + 1], a2 = a_accum[rightBound + 0], b2 = a_accum[rightBound + 1];
    if (a1 < x && x <= b1) currPos = leftBound;
    if (a2 < x && x <= b2) currPos = rightBound;
  }
  if (x == 0.0f) currPos = 0;
  else if (currPos < 0) currPos = (rightBound + leftBound + 1) / 2;
  pdf = (a_accum[currPos + 1] - a_accum[currPos]) / a_accum[N - 1];
  return currPos;
}
HPT_DEV float evalMap2DPdf(V2 tc, const float* intervals, int sizeX, int sizeY)
{
  const float fw = (float)sizeX, fh = (float)sizeY;
  if (tc.x < 0.0f || tc.x > 1.0f) tc.x -= (float)((int)(tc.x));
  if (tc.y < 0.0f || tc.x > 1.0f) tc.y -= (float)((int)(tc.y));                 // (sic: .x in the second test, clight.h:199)
  int pixelX = (int)(fw * tc.x - 0.5f), pixelY = (int)(fh * tc.y - 0.5f);
  if (pixelX >= sizeX) pixelX = sizeX - 1;
  if (pixelY >= sizeY) pixelY = sizeY - 1;
  if (pixelX < 0) pixelX += sizeX;
  if (pixelY < 0) pixelY += sizeY;
  const int pixelOffset = pixelY * sizeX + pixelX, maxSize = sizeX * sizeY;
  const int offset0 = (pixelOffset + 0 < maxSize + 0) ? pixelOffset + 0 : maxSize - 1;
  const int offset1 = (pixelOffset + 1 < maxSize + 1) ? pixelOffset + 1 : maxSize;
  return (intervals[offset1] - intervals[offset0]) * (fw * fh) / intervals[sizeX * sizeY];
}
// LightSampleRev, LIGHT_GEOM_ENV: SampleMap2D over the pdf table, inverse sampler transform, lat-long to direction
HPT_DEV LightSam envLightSampleRev(const DevScene& S, const LightRec& L, V3 rands, V3 illuminationPoint)
{
  const int sizeX = (int)L.pdfTableSizeX, sizeY = (int)L.pdfTableSizeY;
  const float fw = (float)sizeX, fh = (float)sizeY, fN = fw * fh;
  float pdf = 1.0f;
  int pixelOffset = selectIndexPropToOpt(rands.z, S.arrays1f + L.pdfTableOffset, sizeX * sizeY + 1, pdf);
  if (pixelOffset >= sizeX * sizeY) pixelOffset = sizeX * sizeY - 1;
  const int yPos = pixelOffset / sizeX, xPos = pixelOffset - yPos * sizeX;
  const float texX = (1.0f / fw) * (((float)(xPos) + 0.5f) + (rands.x * 2.0f - 1.0f) * 0.5f);
  const float texY = (1.0f / fh) * (((float)(yPos) + 0.5f) + (rands.y * 2.0f - 1.0f) * 0.5f);
  const float mapPdf = pdf * fN;
  const V2 tcT = mulRows2x4(L.samplerRow0Inv, L.samplerRow1Inv, v2(texX, texY));
  const float phi = tcT.x * 2.f * HPT_PI, theta = tcT.y * HPT_PI;                // texCoord2DToSphereMap
  const float sinTheta = sinf(theta);
  const float x = sinTheta * cosf(phi), y = sinTheta * sinf(phi), z = cosf(theta);
  const V3 sampleDir = v3(y, -z, x);
  LightSam r;
  r.hasIES = false; r.isOmni = true; r.norm = sampleDir;
  r.pos = illuminationPoint + sampleDir * 1000.0f;
  r.pdf = (mapPdf * 1.0f) / (2.f * HPT_PI * HPT_PI * smax(absf(sinTheta), 1e-20f));
  return r;
}
// kernel_HitEnvironment (integrator_pt.cpp:550-595) + EnvironmentColor (integrator_pt_lgt.cpp:175-210): what a ray that left the scene sees
HPT_DEV V3 environmentRadiance(const DevScene& S, V3 rdir, float misPdf, uint flags, uint XY)
{
  V3 color = ld3(S.envColor);
  if (S.envTexId == 0xFFFFFFFFu && S.envCamBackId == 0xFFFFFFFFu) return color;   // (the plain-colour case: nothing else can apply)
  float envPdf = 1.0f;
  if (S.envTexId != 0xFFFFFFFFu) {
    const float sinTheta = sqrtf_(1.0f - rdir.y * rdir.y);
    const V2 tcT = mulRows2x4(S.envSamRow0, S.envSamRow1, sphereMapTo2DTexCoord(rdir));
    if (sinTheta != 0.f && S.envEnableSam != 0 && S.integratorType == INTEGRATOR_MIS_PT && S.envLightId != 0xFFFFFFFFu) {
      const LightRec& L = S.lights[S.envLightId];
      const float mapPdf = evalMap2DPdf(tcT, S.arrays1f + L.pdfTableOffset, (int)L.pdfTableSizeX, (int)L.pdfTableSizeY);
      envPdf = (mapPdf * 1.0f) / (2.f * HPT_PI * HPT_PI * smax(absf(sinTheta), 1e-20f));
    }
    const V4 t = texSample(S.textures, S.envTexId, tcT);
    color = color * v3(t.x, t.y, t.z);
  }
  const bool isSpec = misPdf < 0.0f, exitZero = (flags & RAY_FLAG_PRIME_RAY_MISS) != 0;
  if (S.integratorType == INTEGRATOR_MIS_PT && S.envEnableSam != 0 && !isSpec && !exitZero)
    color = color * misWeightHeuristic(misPdf, (1.0f / float(S.numLights)) * envPdf);
  else if (S.integratorType == INTEGRATOR_SHADOW_PT && S.envEnableSam != 0) color = v3(0, 0, 0);
  if (exitZero && S.envCamBackId != 0xFFFFFFFFu) {                               // the camera back plate
    const uint x = XY & 0x0000FFFFu, y = (XY & 0xFFFF0000u) >> 16;
    const V4 t = texSample(S.textures, S.envCamBackId, v2((float(x) + 0.5f) / float(S.winWidth), (float(y) + 0.5f) / float(S.winHeight)));
    color = v3(t.x, t.y, t.z);
  }
  return color;
}

// RemapMaterialId (integrator_pt_mat.cpp:530-573)
HPT_DEV uint remapMaterialId(const DevScene& S, uint a_mId, uint a_instId)
{
  const int remapListId = S.remapInst[2 * a_instId + 0];
  if (remapListId == -1) return a_mId;
  const int r_offset = S.allRemapLists[S.allRemapListsSize + remapListId];
  const int r_size = S.allRemapLists[S.allRemapListsSize + remapListId + 1] - r_offset;
  uint res = a_mId;
  int low = 0, high = r_size - 1;
  while (low <= high) {
    const int mid = low + ((high - low) / 2);
    const int idRemapFrom = S.allRemapLists[r_offset + mid * 2 + 0];
    if (uint(idRemapFrom) >= a_mId) high = mid - 1; else low = mid + 1;
  }
  if (high + 1 < r_size) {
    const int idRemapFrom = S.allRemapLists[r_offset + (high + 1) * 2 + 0];
    const int idRemapTo = S.allRemapLists[r_offset + (high + 1) * 2 + 1];
    res = (uint(idRemapFrom) == a_mId) ? uint(idRemapTo) : a_mId;
  }
  return res;
}

// ---- BVH2 traversal: replaces ISceneObject::RayQuery_NearestHit / RayQuery_AnyHit (CrossRT.h:148-176) ---------------------
// Semantics of the Embree backend (EmbreeRT.cpp:310-484): two-level scene, ray taken into object space with the inverse
// instance matrix, t shared between spaces, no back-face culling, hit iff tnear <= t <= tfar.
// Closest hit = min t with ties broken by (instId, primId): independent of tree shape and traversal order.
struct HitRec { float t; uint prim, inst; float u, v; uint slot = 0xFFFFFFFFu; };   // inst == 0xFFFFFFFF: miss; slot: the hit's triangle record (single-level layout), for DevScene::shadeTris

struct TravStats { uint nodes, tris, insts, waveNodeIters, waveTriIters; };   // wave*: counted by the first active lane of each trip
HPT_DEV bool firstActiveLane() { const uint l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); return l == (uint)__builtin_amdgcn_readfirstlane((int)l); }

// Loop shape ("while-while"): every lane first walks inner nodes in a tight loop until it holds a leaf reference, and only
// then do the lanes of the wave handle their leaves together. Mixing the three node kinds in one loop body makes a wave
// pay for the inner-node code, the triangle code and the instance code on every step, whichever its lanes need.
HPT_DEV V3 rcp3(V3 d) { return v3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)); }   // box test only
// The ray's side of the box tests, once per ray (and per instance entered): the reciprocal direction kept FINITE - a zero component gives +-1e30,
// the slab then reads "inside for every t" or "never" as with the infinite one, without the inf - inf of the folded form plane * id - origin * id -
// and origin * that. Against (plane - origin) * id the distances move by a few ulp of (coordinate * id); the boxes carry a relative 1e-5 of padding
// (Aabb::pad) and the tests their own widening for exactly that. Boxes only cull: hits are decided by the exact triangle test.
HPT_DEV void slabRay(const V3 o, const V3 d, V3& id, V3& oid)
{
  const V3 r = rcp3(d);
  id = v3(fminf(fmaxf(r.x, -1.0e30f), 1.0e30f), fminf(fmaxf(r.y, -1.0e30f), 1.0e30f), fminf(fmaxf(r.z, -1.0e30f), 1.0e30f));
  oid = o * id;
}

// Traversal stack: the first LDS_STACK entries of a lane live in LDS ([depth][lane]: a push or pop is one conflict-free
// ds_write/ds_read_b32 per wave); deeper entries - rare, a push only happens when both children are hit - go to a per-lane
// slice of an HBM scratch buffer ([depth][global lane], coalesced). LDS use is therefore independent of the tree depth.
// Voted exit of the inner-node loop (S.nodeMin): waiting for the LAST lane to reach a leaf costs the big scenes dearly (1M triangles:
// 107 -> 134 Mpaths/s with 16 in the megakernel, 159 -> 188 in the wavefront trace kernel), but on the Cornell box, where a ray sees 7
// nodes and 3 triangles, the extra trips through the leaf code cost more than they save (1859 -> 1584): the host sets it per scene.
#ifndef HPT_LDS_STACK
#define HPT_LDS_STACK 16
#endif
static const int LDS_STACK = HPT_LDS_STACK;
struct TravStack { uint* lds; uint* ovf; uint ovfStride; };
HPT_DEV void stkPush(const TravStack& k, int sp, uint v) { if (sp < LDS_STACK) k.lds[sp * 256] = v; else k.ovf[(size_t)(sp - LDS_STACK) * k.ovfStride] = v; }
HPT_DEV uint stkPop(const TravStack& k, int sp)
{
  // The LDS read is unconditional (clamped slot) and the HBM read conditional. The empty asm makes the LDS value opaque: without it
  // the compiler folds the two reads into ONE flat_load_dword with a selected address, and every pop - which sits on the critical
  // path to the next node fetch - takes the flat path (vmcnt + lgkmcnt wait) instead of a ds_read_b32.
  uint v = k.lds[(sp < LDS_STACK ? sp : LDS_STACK - 1) * 256];
  asm volatile("" : "+v"(v));
  if (sp >= LDS_STACK) v = k.ovf[(size_t)(sp - LDS_STACK) * k.ovfStride];
  return v;
}

// Slab test of both children of a node against a ray (slabRay: finite reciprocal direction id, oid = origin * id; interval [tnear, best]).
// One fma per plane. (Round 2 tried the fma form with the raw reciprocal and dropped it after 8 parity failures: with id = +-inf every plane of that
// axis reads inf - inf; the finite reciprocal is what makes it work.) The node stores (lo, hi) pairs per axis, so each axis of each child CAN be one
// packed fma (HPT_PACKED_SLABS: 6 instructions instead of 12) - packed math measured slower in round 2, off by default. Boxes were padded by the
// builder; the interval is widened a little more so that rounding (and the 1-ulp reciprocal) can only make the test more conservative than the exact triangle test
// (tests/cpp/slab_fold_test.cpp).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef HPT_FLAT_WIDE
#define HPT_FLAT_WIDE 0      // 1: the megakernel's single-level traversal walks the 4-wide compressed tree on EVERY scene that has one (default: heavy scenes only, DevScene::megaWide)
#endif
#ifndef HPT_PACKED_SLABS
#define HPT_PACKED_SLABS 0   // measured: v_pk_add/mul_f32 (12 instead of 24 instructions) is NOT faster here: Cornell 1805 vs 1820, 1M triangles 200 vs 206
#endif
HPT_DEV void nodeSlabs(const float4 q0, const float4 q1, const float4 q2, const V3 oid, const V3 id, const float tnear, const float best,
                       bool& h0, bool& h1, float& t0n, float& t1n)
{
  // plane distance = plane * id - origin * id (slabRay: id finite, oid = origin * id): one fma per plane
#if HPT_PACKED_SLABS
  const f32x2 ox = {-oid.x, -oid.x}, oy = {-oid.y, -oid.y}, oz = {-oid.z, -oid.z}, ix = {id.x, id.x}, iy = {id.y, id.y}, iz = {id.z, id.z};
  const f32x2 ax = __builtin_elementwise_fma(f32x2{q0.x, q0.y}, ix, ox), ay = __builtin_elementwise_fma(f32x2{q0.z, q0.w}, iy, oy), az = __builtin_elementwise_fma(f32x2{q1.x, q1.y}, iz, oz);
  const f32x2 bx = __builtin_elementwise_fma(f32x2{q1.z, q1.w}, ix, ox), by = __builtin_elementwise_fma(f32x2{q2.x, q2.y}, iy, oy), bz = __builtin_elementwise_fma(f32x2{q2.z, q2.w}, iz, oz);
#else
  const f32x2 ax = {__builtin_fmaf(q0.x, id.x, -oid.x), __builtin_fmaf(q0.y, id.x, -oid.x)}, ay = {__builtin_fmaf(q0.z, id.y, -oid.y), __builtin_fmaf(q0.w, id.y, -oid.y)}, az = {__builtin_fmaf(q1.x, id.z, -oid.z), __builtin_fmaf(q1.y, id.z, -oid.z)};
  const f32x2 bx = {__builtin_fmaf(q1.z, id.x, -oid.x), __builtin_fmaf(q1.w, id.x, -oid.x)}, by = {__builtin_fmaf(q2.x, id.y, -oid.y), __builtin_fmaf(q2.y, id.y, -oid.y)}, bz = {__builtin_fmaf(q2.z, id.z, -oid.z), __builtin_fmaf(q2.w, id.z, -oid.z)};
#endif
  t0n = fmaxf(fmaxf(fminf(ax.x, ax.y), fminf(ay.x, ay.y)), fmaxf(fminf(az.x, az.y), tnear));
  const float t0f = fminf(fminf(fmaxf(ax.x, ax.y), fmaxf(ay.x, ay.y)), fminf(fmaxf(az.x, az.y), best));
  t1n = fmaxf(fmaxf(fminf(bx.x, bx.y), fminf(by.x, by.y)), fmaxf(fminf(bz.x, bz.y), tnear));
  const float t1f = fminf(fminf(fmaxf(bx.x, bx.y), fmaxf(by.x, by.y)), fminf(fmaxf(bz.x, bz.y), best));
  h0 = (t0n * 0.999999f <= t0f * 1.000001f);
  h1 = (t1n * 0.999999f <= t1f * 1.000001f);
}

// Moeller-Trumbore on a 48-byte record (v0, e1, e2): exact (IEEE) arithmetic, inclusive [tnear, current best], no back-face culling; at equal
// distance the lower (instId, primId) wins, so the closest hit does not depend on tree shape or traversal order. ONE definition for the
// megakernel's two traversals, the wavefront trace kernel and the ray-query kernel. Returns true when the hit record was updated.
HPT_DEV bool triangleTest(const float4 a, const float4 b, const float4 c, const V3 o, const V3 d, const float tnear, const uint inst,
                          float& bestT, uint& bestPrim, uint& bestInst, float& bestU, float& bestV, bool& found)
{
  const V3 e1 = v3(b.x, b.y, b.z), e2 = v3(c.x, c.y, c.z);
  const V3 pvec = cross(d, e2);
  const float det = dot(e1, pvec);
  const float inv = 1.0f / det;
  const V3 tvec = o - v3(a.x, a.y, a.z);
  const float uu = dot(tvec, pvec) * inv;
  const V3 qvec = cross(tvec, e1);
  const float vv = dot(d, qvec) * inv;
  const float tt = dot(e2, qvec) * inv;
  const uint prim = __float_as_uint(a.w);
  bool ok = (det != 0.0f) && (uu >= 0.0f) && (vv >= 0.0f) && (uu + vv <= 1.0f) && (tt >= tnear) && (tt <= bestT);
  if (ok && found && tt == bestT) ok = (inst != bestInst) ? (inst < bestInst) : (prim < bestPrim);
  if (ok) { bestT = tt; bestPrim = prim; bestInst = inst; bestU = uu; bestV = vv; found = true; }
  return ok;
}
// world -> object space of an instance (EmbreeRT.cpp:242-292 semantics: t is shared between the two spaces)
HPT_DEV void toObjectSpace(const BvhInst* insts, const uint inst, const V3 wo, const V3 wd, V3& o, V3& d)
{
  const float4* ip = (const float4*)(insts + inst);
  const float4 r0 = ip[0], r1 = ip[1], r2 = ip[2];
  o = v3(r0.x * wo.x + r0.y * wo.y + r0.z * wo.z + r0.w, r1.x * wo.x + r1.y * wo.y + r1.z * wo.z + r1.w, r2.x * wo.x + r2.y * wo.y + r2.z * wo.z + r2.w);
  d = v3(r0.x * wd.x + r0.y * wd.y + r0.z * wd.z, r1.x * wd.x + r1.y * wd.y + r1.z * wd.z, r2.x * wd.x + r2.y * wd.y + r2.z * wd.z);
}

// A moving instance (AddInstanceMotion, EmbreeRT.cpp:264-292): the object->world matrix is interpolated linearly between its two keys at the
// ray's time and inverted for this ray (Embree's motion-blurred instances do the same: lerp of local2world, then its inverse).
// m = 24 floats: rows of the 3x4 matrix at time 0, then at time 1. Cofactor inverse, the translation subtracted first.
HPT_DEV void toObjectSpaceMotion(const float* m, const float time, const V3 wo, const V3 wd, V3& o, V3& d)
{
  // In double: a float cofactor inverse leaves an error of the size of the ray-origin offsets (5e-6 * |p|), and rays leaving the moving
  // surface then re-hit it or not depending on the last bit of their direction (measured: one path in ~15 % of the fuzz scenes ended
  // differently from the checker's, against ~0.5 % with static instances, whose inverse the host computes in double). FP64 is cheap on
  // this chip and only rays entering a MOVING instance pay for it.
  double a[12];
  const double t = (double)time;
  for (int k = 0; k < 12; k++) a[k] = (double)m[k] + t * ((double)m[12 + k] - (double)m[k]);
  const double c00 = a[5] * a[10] - a[6] * a[9], c01 = a[2] * a[9] - a[1] * a[10], c02 = a[1] * a[6] - a[2] * a[5];
  const double c10 = a[6] * a[8] - a[4] * a[10], c11 = a[0] * a[10] - a[2] * a[8], c12 = a[2] * a[4] - a[0] * a[6];
  const double c20 = a[4] * a[9] - a[5] * a[8],  c21 = a[1] * a[8] - a[0] * a[9],  c22 = a[0] * a[5] - a[1] * a[4];
  const double det = a[0] * c00 + a[1] * c10 + a[2] * c20;
  const double id = 1.0 / det;
  const double px = (double)wo.x - a[3], py = (double)wo.y - a[7], pz = (double)wo.z - a[11];
  const double dx = (double)wd.x, dy = (double)wd.y, dz = (double)wd.z;
  o = v3((float)((c00 * px + c01 * py + c02 * pz) * id), (float)((c10 * px + c11 * py + c12 * pz) * id), (float)((c20 * px + c21 * py + c22 * pz) * id));
  d = v3((float)((c00 * dx + c01 * dy + c02 * dz) * id), (float)((c10 * dx + c11 * dy + c12 * dz) * id), (float)((c20 * dx + c21 * dy + c22 * dz) * id));
}

template <bool ANY, bool STATS, bool DEEP, bool MOTION = false>
HPT_DEV bool traceRay(const DevScene& S, const V3 wo, const V3 wd, float tnear, float tfar, HitRec& hit, const TravStack& stk, TravStats& st, const float time = 0.0f)
{
  hit.t = tfar; hit.prim = 0xFFFFFFFFu; hit.inst = 0xFFFFFFFFu; hit.u = 0.0f; hit.v = 0.0f;
  bool found = false;
  uint cur = S.rootRef;
  if (cur == REF_NONE) return false;

  V3 o = wo, d = wd;                                         // current-space ray; the world-space one is the caller's
  V3 id, oid; slabRay(o, d, id, oid);
  uint curInst = 0xFFFFFFFFu;
  int sp = 0;

  // DEEP = false: the whole stack fits the LDS part (host checked the tree depth), no overflow test on push / pop
#define HPT_PUSH(v) do { if (DEEP) stkPush(stk, sp, (v)); else stk.lds[sp * 256] = (v); sp++; } while (0)
#define HPT_POP()   do { sp--; cur = DEEP ? stkPop(stk, sp) : stk.lds[sp * 256]; } while (0)

  while (true) {
    // ---- (a) inner nodes: one 64-byte line holds both child boxes ---------------------------------------------------
    // Two copies of the loop: without the vote (small scenes - even a never-taken scalar test per step cost the Cornell box 4 %)
    // and with it (heavy scenes: leave when only a few lanes of the wave are still walking inner nodes, serve the leaves first).
#define HPT_NODE_STEP()                                                                                 \
      const float4* np = (const float4*)(S.nodes + cur);                                                \
      const float4 q0 = np[0], q1 = np[1], q2 = np[2];                                                  \
      const uint4  q3 = ((const uint4*)np)[3];                                                          \
      if (STATS) { st.nodes++; if (firstActiveLane()) st.waveNodeIters++; }                             \
      bool h0, h1; float t0n, t1n;                                                                      \
      nodeSlabs(q0, q1, q2, oid, id, tnear, hit.t, h0, h1, t0n, t1n);                                   \
      if (h0 && h1) {                                                                                   \
        const bool firstIs0 = t0n <= t1n;                                                               \
        HPT_PUSH(firstIs0 ? q3.y : q3.x);                                                               \
        cur = firstIs0 ? q3.x : q3.y;                                                                   \
      } else if (h0) cur = q3.x;                                                                        \
      else if (h1) cur = q3.y;                                                                          \
      else if (sp > 0) HPT_POP();                                                                       \
      else cur = REF_NONE;
    if (S.nodeMin == 0u) {
      while ((cur & REF_LEAF) == 0u) { HPT_NODE_STEP() }
    } else {
      while ((cur & REF_LEAF) == 0u) {
        HPT_NODE_STEP()
        if ((uint)__popcll(__ballot((cur & REF_LEAF) == 0u)) < S.nodeMin) break;
      }
    }
    if (cur == REF_NONE) break;
    if ((cur & REF_LEAF) == 0u) continue;

    // ---- (b) leaves --------------------------------------------------------------------------------------------------
    const uint cnt = (cur >> 28) & 7u;
    if (cnt >= 1u && cnt <= 4u) {
      // triangles: Moeller-Trumbore on (v0, e1, e2), 3 x 16-byte loads each; exact (IEEE) arithmetic
      const uint first = cur & 0x0FFFFFFFu;
      for (uint k = 0; k < cnt; k++) {
        const float4* tp = (const float4*)(S.tris + first + k);
        const float4 a = tp[0], b = tp[1], c = tp[2];
        if (STATS) { st.tris++; if (firstActiveLane()) st.waveTriIters++; }
        if (triangleTest(a, b, c, o, d, tnear, curInst, hit.t, hit.prim, hit.inst, hit.u, hit.v, found) && ANY) return true;
      }
      if (sp > 0) HPT_POP(); else break;
    } else if (cnt == 0u) {
      // instance leaf: enter object space (EmbreeRT.cpp:242-292 semantics: t is shared between the two spaces)
      const uint inst = cur & 0x0FFFFFFFu;
      const uint root = S.insts[inst].root;
      if (STATS) st.insts++;
      if (MOTION && S.insts[inst].pad0 != 0u) toObjectSpaceMotion(S.instMotion + 24u * inst, time, wo, wd, o, d);
      else toObjectSpace(S.insts, inst, wo, wd, o, d);
      slabRay(o, d, id, oid);
      curInst = inst;
      HPT_PUSH(REF_RESTORE);
      cur = root;
    } else {
      // marker: back to world space
      o = wo; d = wd; slabRay(o, d, id, oid); curInst = 0xFFFFFFFFu;
      if (sp > 0) HPT_POP(); else break;
    }
  }
#undef HPT_PUSH
#undef HPT_POP
  return found;
}

// One visit of a 4-wide compressed node (BvhNode4, hpt_types.h): decode the four child boxes (v_cvt_f32_ubyteN + one fma per bound), slab-test
// them like nodeSlabs does (same widening, so the test stays conservative with respect to the exact triangle test), sort the children that
// were hit by entry distance with a five-exchange network on (distance bits | child index, reference) pairs, continue with the nearest and
// push the others farthest first. Traversal ORDER never changes a result: the closest hit is min t with ties broken by (instId, primId).
HPT_DEV float ubyteToFloat(uint w, int k) { return (float)((w >> (8 * k)) & 0xFFu); }      // v_cvt_f32_ubyte<k>
// Per node the decode and the slab test are folded: plane distance = q * (scale * id) + (base * id - org * id) - one cvt and one fma per plane -
// and the ray's octant says which of a child's two planes per axis is the near one, so no min / max pairs are spent on finding out. Against
// decode-then-test the distances move by a few ulp of (coordinate * id); the boxes carry a relative 1e-5 of padding (Aabb::pad) for exactly that.
template <bool DEEP>
HPT_DEV void wideNodeStep(const DevScene& S, const TravStack& stk, const V3 oid, const V3 id, const float best, uint& cur, int& sp, const float tnear = 0.0f)
{
  const uint4* np = (const uint4*)(S.nodes4 + cur);
  const uint4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
  const float sx = __uint_as_float((w0.w & 0xFFu) << 23) * id.x, sy = __uint_as_float(((w0.w >> 8) & 0xFFu) << 23) * id.y, sz = __uint_as_float(((w0.w >> 16) & 0xFFu) << 23) * id.z;
  const float bx = __builtin_fmaf(__uint_as_float(w0.x), id.x, -oid.x), by = __builtin_fmaf(__uint_as_float(w0.y), id.y, -oid.y), bz = __builtin_fmaf(__uint_as_float(w0.z), id.z, -oid.z);
  const bool negX = id.x < 0.0f, negY = id.y < 0.0f, negZ = id.z < 0.0f;
  const uint nX = negX ? w1.w : w1.x, fX = negX ? w1.x : w1.w, nY = negY ? w2.x : w1.y, fY = negY ? w1.y : w2.x, nZ = negZ ? w2.y : w1.z, fZ = negZ ? w1.z : w2.y;
  const float bestW = best * 1.0000021f;                                                       // (tn * 0.999999 <= tf * 1.000001 of nodeSlabs, as one factor on the far side)
  uint key[4], ref[4] = { w3.x, w3.y, w3.z, w3.w };
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const float tnx = __builtin_fmaf(ubyteToFloat(nX, c), sx, bx), tny = __builtin_fmaf(ubyteToFloat(nY, c), sy, by), tnz = __builtin_fmaf(ubyteToFloat(nZ, c), sz, bz);
    const float tfx = __builtin_fmaf(ubyteToFloat(fX, c), sx, bx), tfy = __builtin_fmaf(ubyteToFloat(fY, c), sy, by), tfz = __builtin_fmaf(ubyteToFloat(fZ, c), sz, bz);
    const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tnear));
    const float tf = fminf(fminf(fminf(tfx, tfy), tfz) * 1.0000021f, bestW);
    const bool hit = (tn <= tf) & (((w0.w >> (24 + c)) & 1u) != 0u);
    key[c] = hit ? (((__float_as_uint(tn) & 0x7FFFFFFCu)) | (uint)c) : 0xFFFFFFFFu;        // tn >= 0: its bit pattern orders like the value
  }
#define HPT_CE(a, b) do { const bool sw = key[b] < key[a]; const uint ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b], ra = sw ? ref[b] : ref[a], rb = sw ? ref[a] : ref[b]; \
                          key[a] = ka; key[b] = kb; ref[a] = ra; ref[b] = rb; } while (0)
  HPT_CE(0, 1); HPT_CE(2, 3); HPT_CE(0, 2); HPT_CE(1, 3); HPT_CE(1, 2);
#undef HPT_CE
  if (key[3] != 0xFFFFFFFFu) { if (DEEP) stkPush(stk, sp, ref[3]); else stk.lds[sp * 256] = ref[3]; sp++; }
  if (key[2] != 0xFFFFFFFFu) { if (DEEP) stkPush(stk, sp, ref[2]); else stk.lds[sp * 256] = ref[2]; sp++; }
  if (key[1] != 0xFFFFFFFFu) { if (DEEP) stkPush(stk, sp, ref[1]); else stk.lds[sp * 256] = ref[1]; sp++; }
  if (key[0] != 0xFFFFFFFFu) cur = ref[0];
  else if (sp > 0) { sp--; cur = DEEP ? stkPop(stk, sp) : stk.lds[sp * 256]; }
  else cur = REF_NONE;
}

// ---- single-level variant ---------------------------------------------------------------------------------------------------
// For static scenes whose instanced triangle count fits the budget (hpt_host.hip: FLAT_TRI_BUDGET) the host builds ONE BVH2 over
// all instanced triangles with WORLD-space boxes: no TLAS/BLAS box overlap, no instance enter / leave trips through the loop.
// The triangle test itself still runs in the instance's OBJECT space - the ray is taken there with the same world->object rows the
// two-level path uses, cached while consecutive triangles belong to the same instance - so every hit (t, u, v, ids) is bit-identical
// to the two-level / Embree semantics. Boxes only cull.
template <bool ANY, bool STATS, bool DEEP, bool MOTION = false>
HPT_DEV bool traceRayFlat(const DevScene& S, const V3 wo, const V3 wd, float tnear, float tfar, HitRec& hit, const TravStack& stk, TravStats& st, const float time = 0.0f)
{
  hit.t = tfar; hit.prim = 0xFFFFFFFFu; hit.inst = 0xFFFFFFFFu; hit.u = 0.0f; hit.v = 0.0f;
  bool found = false;
  uint cur = S.rootRef;
  if (cur == REF_NONE) return false;
  V3 id, oid; slabRay(wo, wd, id, oid);                       // the boxes of this layout are in world space
  V3 o = wo, d = wd;                                          // object-space ray of instance `curInst`
  uint curInst = 0xFFFFFFFFu;
  int sp = 0;
#define HPT_PUSH(v) do { if (DEEP) stkPush(stk, sp, (v)); else stk.lds[sp * 256] = (v); sp++; } while (0)
#define HPT_POP()   do { sp--; cur = DEEP ? stkPop(stk, sp) : stk.lds[sp * 256]; } while (0)
  const bool wide = (((HPT_FLAT_WIDE || S.megaWide != 0u) && !STATS) || (STATS && S.statsWide != 0u)) && !MOTION && S.nodes4 != nullptr;   // wave-uniform: the 4-wide compressed tree of the same scene
  if (wide) cur = S.root4;
  while (true) {
    if (wide) {
      while ((cur & REF_LEAF) == 0u) {
        if (STATS) { st.nodes++; if (firstActiveLane()) st.waveNodeIters++; }
        wideNodeStep<DEEP>(S, stk, oid, id, hit.t, cur, sp, tnear);
        if (S.nodeMin4 != 0u && (uint)__popcll(__ballot((cur & REF_LEAF) == 0u)) < S.nodeMin4) break;
      }
    } else if (S.nodeMin == 0u) {
      while ((cur & REF_LEAF) == 0u) { HPT_NODE_STEP() }
    } else {
      while ((cur & REF_LEAF) == 0u) {
        HPT_NODE_STEP()
        if ((uint)__popcll(__ballot((cur & REF_LEAF) == 0u)) < S.nodeMin) break;
      }
    }
    if (cur == REF_NONE) break;
    if ((cur & REF_LEAF) == 0u) continue;
    {
      const uint cnt = (cur >> 28) & 7u;
      const uint first = cur & 0x0FFFFFFFu;
      for (uint k = 0; k < cnt; k++) {
        const float4* tp = (const float4*)(S.tris + first + k);
        const float4 a = tp[0], b = tp[1], c = tp[2];
        if (STATS) { st.tris++; if (firstActiveLane()) st.waveTriIters++; }
        const uint inst = __float_as_uint(b.w);
        if (inst != curInst) {                                // world -> object space of this triangle's instance
          if (STATS) st.insts++;
          // a moving instance: its world boxes span both keys (host), its triangles are met in the object space of the ray's own time
          if (MOTION && S.insts[inst].pad0 != 0u) toObjectSpaceMotion(S.instMotion + 24u * inst, time, wo, wd, o, d);
          else toObjectSpace(S.insts, inst, wo, wd, o, d);
          curInst = inst;
        }
        const bool upd = triangleTest(a, b, c, o, d, tnear, inst, hit.t, hit.prim, hit.inst, hit.u, hit.v, found);
        if (upd) hit.slot = first + k;
        if (upd && ANY) return true;
      }
      if (sp > 0) HPT_POP(); else break;
    }
  }
#undef HPT_PUSH
#undef HPT_POP
#undef HPT_NODE_STEP
  return found;
}

// ---- sweep: no tree at all ----------------------------------------------------------------------------------------------------------------
// Scenes of a few dozen triangles (the Cornell-box class, hpt_host.hip: SWEEP_MAX_TRIS). A BVH walk keeps a third of a wave's lanes busy
// there (per-ray trip counts differ, measured 0.31 in the node loop and 0.33 in the triangle loop) and waits on a dependent load every step.
// Here the WAVE walks the scene instead: instance by instance, triangle by triangle, the same for all 64 lanes, the records fetched with
// scalar loads through the constant address space (s_load_dwordx4: SGPR operands, no vector memory instruction, no stack, no LDS) and
// every lane tests its own ray against the wave's triangle - 100 % of the lanes on every instruction. The ray is taken to each
// instance's object space with the rows the two-level path uses and the triangle test is the shared one, so every hit (t, u, v, ids) is
// bit-identical to the other layouts: the closest hit does not depend on the order of the tests (ties go to the lower (instId, primId)).
// DevScene::sweepInsts: per instance {world->object rows, first triangle record, geomId, 0, triangle count}; DevScene::sweepTris: the triangle
// records per MESH in primitive order, padded to an even count with a degenerate record (det = 0: never hit). The two-level structure stays valid beside them (the wavefront schedule and forced layouts use it).
typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef uint  u32x4n __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const f32x4n cfloat4;          // uniform address + constant address space = scalar load
typedef __attribute__((address_space(4))) const u32x4n cuint4;
HPT_DEV float4 ldc4(const cfloat4* p) { const f32x4n v = *p; return make_float4(v.x, v.y, v.z, v.w); }
// The sweep visits (instance, primitive) pairs in increasing order (DevScene::sweepTris holds each instance's records sorted by primitive
// id), so the tie rule of triangleTest - at equal distance the lower (instId, primId) wins - reduces to "the first one found stays".
HPT_DEV void triangleTestInOrder(const float4 a, const float4 b, const float4 c, const V3 o, const V3 d, const float tnear, const uint inst,
                                 float& bestT, uint& bestPrim, uint& bestInst, float& bestU, float& bestV, bool& found)
{
  const V3 e1 = v3(b.x, b.y, b.z), e2 = v3(c.x, c.y, c.z);
  const V3 pvec = cross(d, e2);
  const float det = dot(e1, pvec);
  const float inv = 1.0f / det;
  const V3 tvec = o - v3(a.x, a.y, a.z);
  const float uu = dot(tvec, pvec) * inv;
  const V3 qvec = cross(tvec, e1);
  const float vv = dot(d, qvec) * inv;
  const float tt = dot(e2, qvec) * inv;
  // (non-short-circuit on purpose: the conditions become lane masks combined on the scalar unit instead of selects of 0 / 1 in VGPRs)
  const bool closer = (tt < bestT) | (!found & (tt == bestT));              // = found ? tt < bestT : tt <= bestT
  const bool ok = (det != 0.0f) & (uu >= 0.0f) & (vv >= 0.0f) & (uu + vv <= 1.0f) & (tt >= tnear) & closer;
  if (ok) { bestT = tt; bestPrim = __float_as_uint(a.w); bestInst = inst; bestU = uu; bestV = vv; found = true; }
}
// the same test for occlusion queries: any triangle with t in [tnear, tfar] ends the search, so nothing but the flag is kept
HPT_DEV bool triangleOccludes(const float4 a, const float4 b, const float4 c, const V3 o, const V3 d, const float tnear, const float tfar)
{
  const V3 e1 = v3(b.x, b.y, b.z), e2 = v3(c.x, c.y, c.z);
  const V3 pvec = cross(d, e2);
  const float det = dot(e1, pvec);
  const float inv = 1.0f / det;
  const V3 tvec = o - v3(a.x, a.y, a.z);
  const float uu = dot(tvec, pvec) * inv;
  const V3 qvec = cross(tvec, e1);
  const float vv = dot(d, qvec) * inv;
  const float tt = dot(e2, qvec) * inv;
  return (det != 0.0f) & (uu >= 0.0f) & (vv >= 0.0f) & (uu + vv <= 1.0f) & (tt >= tnear) & (tt <= tfar);
}

template <bool ANY, bool STATS>
HPT_DEV bool traceSweep(const DevScene& S, const V3 wo, const V3 wd, const float tnear, const float tfar, HitRec& hit, TravStats& st)
{
  hit.t = tfar; hit.prim = 0xFFFFFFFFu; hit.inst = 0xFFFFFFFFu; hit.u = 0.0f; hit.v = 0.0f;
  bool found = false;
  const cfloat4* insts = (const cfloat4*)S.sweepInsts;
  const cfloat4* tris = (const cfloat4*)S.sweepTris;
  const cfloat4* boxes = (const cfloat4*)S.sweepBoxes;
  const V3 id = rcp3(wd);
  const uint ni = S.numInsts;
  for (uint i = 0; i < ni; i++) {
    const float4 r0 = ldc4(insts + 4u * i + 0u), r1 = ldc4(insts + 4u * i + 1u), r2 = ldc4(insts + 4u * i + 2u);
    const u32x4n r3 = ((const cuint4*)insts)[4u * i + 3u];                 // {first triangle record, geomId, 0, number of record PAIRS}
    {
      // Wave-uniform skip: when NO ray of the wave can reach the instance's (padded) world box within its interval - [tnear, closest hit so far],
      // or [tnear, tfar] for a shadow ray that has no occluder yet - none of its triangles can change a result, and the whole wave steps over
      // them (the eye rays of a tile looking past the boxes of the Cornell scene, rays whose hit is already nearer than the box). Boxes only
      // cull, the slab test is nodeSlabs' (widened: conservative against the exact triangle test), so hits stay bit-identical.
      const float4 blo = ldc4(boxes + 2u * i), bhi = ldc4(boxes + 2u * i + 1u);
      const float lim = ANY ? (found ? -1.0f : tfar) : hit.t;
      const float ax0 = (blo.x - wo.x) * id.x, ax1 = (bhi.x - wo.x) * id.x, ay0 = (blo.y - wo.y) * id.y, ay1 = (bhi.y - wo.y) * id.y, az0 = (blo.z - wo.z) * id.z, az1 = (bhi.z - wo.z) * id.z;
      const float tn = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fmaxf(fminf(az0, az1), tnear));
      const float tf = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fminf(fmaxf(az0, az1), lim));
      if (__ballot(tn * 0.999999f <= tf * 1.000001f) == 0ull) continue;
    }
    // toObjectSpace (same expressions, same order)
    const V3 o = v3(r0.x * wo.x + r0.y * wo.y + r0.z * wo.z + r0.w, r1.x * wo.x + r1.y * wo.y + r1.z * wo.z + r1.w, r2.x * wo.x + r2.y * wo.y + r2.z * wo.z + r2.w);
    const V3 d = v3(r0.x * wd.x + r0.y * wd.y + r0.z * wd.z, r1.x * wd.x + r1.y * wd.y + r1.z * wd.z, r2.x * wd.x + r2.y * wd.y + r2.z * wd.z);
    if (STATS) st.insts++;
    const uint first = r3.x, pairs = r3.w;                                 // records come in pairs (the host pads an odd mesh with a record that cannot be hit)
    const cfloat4* tp = tris + 3u * first;
    for (uint k = 0; k < pairs; k++, tp += 6) {
      // six scalar loads, one wait: two triangles per trip
      const float4 a0 = ldc4(tp), b0 = ldc4(tp + 1), c0 = ldc4(tp + 2), a1 = ldc4(tp + 3), b1 = ldc4(tp + 4), c1 = ldc4(tp + 5);
      if (STATS) { st.tris += 2; if (firstActiveLane()) st.waveTriIters += 2; }
      if (ANY) {
        found |= triangleOccludes(a0, b0, c0, o, d, tnear, tfar);
        found |= triangleOccludes(a1, b1, c1, o, d, tnear, tfar);
        if (__ballot(!found) == 0ull) return true;                         // every lane of the wave that traces a ray has its occluder
        continue;
      }
      triangleTestInOrder(a0, b0, c0, o, d, tnear, i, hit.t, hit.prim, hit.inst, hit.u, hit.v, found);
      triangleTestInOrder(a1, b1, c1, o, d, tnear, i, hit.t, hit.prim, hit.inst, hit.u, hit.v, found);
      if (ANY && __ballot(!found) == 0ull) return true;                    // every lane of the wave that traces a ray has its occluder
    }
  }
  return found;
}

// dispatch on the scene's acceleration-structure layout (compile-time: each kernel variant is built for one layout)
template <bool ANY, bool STATS, bool DEEP, bool FLAT, bool MOTION = false, bool SWEEP = false>
HPT_DEV bool traceAny(const DevScene& S, const V3 wo, const V3 wd, float tnear, float tfar, HitRec& hit, const TravStack& stk, TravStats& st, const float time = 0.0f)
{
  if (SWEEP) return traceSweep<ANY, STATS>(S, wo, wd, tnear, tfar, hit, st);
  if (FLAT) return traceRayFlat<ANY, STATS, DEEP, MOTION>(S, wo, wd, tnear, tfar, hit, stk, st, time);
  return traceRay<ANY, STATS, DEEP, MOTION>(S, wo, wd, tnear, tfar, hit, stk, st, time);
}

} // namespace hpt
