// ORACLE -- TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the shipped product path.
//
// Minimal float vector / matrix helpers for the CPU restatement of HydraCore3's path tracer.
// The reference uses LiteMath (external/LiteMath, an un-vendored and therefore ABSENT submodule); the
// conventions below are inferred from the reference's call sites:
//   * float4x4 is column-major (m_col[4]); M*v is matrix x column vector          (integrator_pt_scene.cpp:443-448)
//   * mul4x3(M,p) = affine point transform, mul3x3(M,v) = upper-left 3x3 times v  (include/cglobals.h:254-263)
//   * normalize(v) = v * (1 / sqrt(dot(v,v))) - one reciprocal, three products, the form of LiteMath's generated header (`float lenInv = float(1)/length(a);
//     return a*lenInv;`); LiteMath is absent from the tree, so this is from memory of the library, unpinned like the rest. dot sums x,y,z left to right
//   * reflect(i,n) = i - 2*dot(n,i)*n (GLSL semantics)                            (include/cmat_gltf.h:26)
//   * complex{re,im}, complex_norm = re^2 + im^2                                  (include/cmaterial.h:685-694)
// PARITY UNPINNED: LiteMath itself cannot be consulted, these are textbook definitions.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>

namespace orc {

typedef unsigned int uint;

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

static inline f2 mk2(float x, float y) { f2 r = {x, y}; return r; }
static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f4 mk4(float x, float y, float z, float w) { f4 r = {x, y, z, w}; return r; }
static inline f4 splat4(float a) { return mk4(a, a, a, a); }
static inline f3 xyz(f4 a) { return mk3(a.x, a.y, a.z); }
static inline f4 xyzw(f3 a, float w) { return mk4(a.x, a.y, a.z, w); }

static inline f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
static inline f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
static inline f3 neg(f3 a) { return mk3(-a.x, -a.y, -a.z); }

static inline f4 operator+(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline f4 operator-(f4 a, f4 b) { return mk4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline f4 operator*(f4 a, f4 b) { return mk4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline f4 operator*(f4 a, float s) { return mk4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline f4 operator*(float s, f4 a) { return mk4(s * a.x, s * a.y, s * a.z, s * a.w); }
static inline f4 operator/(f4 a, float s) { return mk4(a.x / s, a.y / s, a.z / s, a.w / s); }
static inline f4 operator/(f4 a, f4 b) { return mk4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }

static inline f2 operator+(f2 a, f2 b) { return mk2(a.x + b.x, a.y + b.y); }
static inline f2 operator*(f2 a, f2 b) { return mk2(a.x * b.x, a.y * b.y); }
static inline f2 operator*(float s, f2 a) { return mk2(s * a.x, s * a.y); }
static inline f2 operator*(f2 a, float s) { return mk2(a.x * s, a.y * s); }

static inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float dot2(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
static inline f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float length(f3 a) { return std::sqrt(dot(a, a)); }
static inline f3 normalize(f3 a) { const float lenInv = 1.0f / length(a); return a * lenInv; }
static inline f3 reflect(f3 i, f3 n) { return i - 2.0f * dot(n, i) * n; }
static inline float clampf(float x, float lo, float hi) { return std::min(std::max(x, lo), hi); }
static inline float lerpf(float a, float b, float t) { return a + t * (b - a); }
static inline f3 lerp3(f3 a, f3 b, float t) { return a + t * (b - a); }

static inline uint as_uint(float f) { uint u; std::memcpy(&u, &f, 4); return u; }
static inline float as_float(uint u) { float f; std::memcpy(&f, &u, 4); return f; }

static const float kPI = 3.14159265358979323846f;
static const float kTWOPI = 6.28318530717958647692f;
static const float kINV_PI = 0.31830988618379067154f;

// column-major 4x4: c[col][row]
struct m4 { float c[4][4]; };

static inline float at(const m4& m, int row, int col) { return m.c[col][row]; }

static inline f4 mul(const m4& m, f4 v)
{
  f4 r;
  r.x = v.x * m.c[0][0] + v.y * m.c[1][0] + v.z * m.c[2][0] + v.w * m.c[3][0];
  r.y = v.x * m.c[0][1] + v.y * m.c[1][1] + v.z * m.c[2][1] + v.w * m.c[3][1];
  r.z = v.x * m.c[0][2] + v.y * m.c[1][2] + v.z * m.c[2][2] + v.w * m.c[3][2];
  r.w = v.x * m.c[0][3] + v.y * m.c[1][3] + v.z * m.c[2][3] + v.w * m.c[3][3];
  return r;
}
static inline f3 mul4x3(const m4& m, f3 p)
{
  f3 r;
  r.x = m.c[0][0] * p.x + m.c[1][0] * p.y + m.c[2][0] * p.z + m.c[3][0];
  r.y = m.c[0][1] * p.x + m.c[1][1] * p.y + m.c[2][1] * p.z + m.c[3][1];
  r.z = m.c[0][2] * p.x + m.c[1][2] * p.y + m.c[2][2] * p.z + m.c[3][2];
  return r;
}
static inline f3 mul3x3(const m4& m, f3 v)
{
  f3 r;
  r.x = m.c[0][0] * v.x + m.c[1][0] * v.y + m.c[2][0] * v.z;
  r.y = m.c[0][1] * v.x + m.c[1][1] * v.y + m.c[2][1] * v.z;
  r.z = m.c[0][2] * v.x + m.c[1][2] * v.y + m.c[2][2] * v.z;
  return r;
}

// affine inverse (upper 3x3 + translation) evaluated in double, rounded once to float.
// Embree computes world->local for an instance the same way (inverse of the AffineSpace).
static inline m4 affine_inverse(const m4& m)
{
  double a00 = at(m,0,0), a01 = at(m,0,1), a02 = at(m,0,2);
  double a10 = at(m,1,0), a11 = at(m,1,1), a12 = at(m,1,2);
  double a20 = at(m,2,0), a21 = at(m,2,1), a22 = at(m,2,2);
  double tx = at(m,0,3), ty = at(m,1,3), tz = at(m,2,3);
  double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  double det = a00 * c00 + a01 * c01 + a02 * c02;
  double id = 1.0 / det;
  double i00 = c00 * id, i01 = (a02 * a21 - a01 * a22) * id, i02 = (a01 * a12 - a02 * a11) * id;
  double i10 = c01 * id, i11 = (a00 * a22 - a02 * a20) * id, i12 = (a02 * a10 - a00 * a12) * id;
  double i20 = c02 * id, i21 = (a01 * a20 - a00 * a21) * id, i22 = (a00 * a11 - a01 * a10) * id;
  m4 r;
  r.c[0][0] = (float)i00; r.c[1][0] = (float)i01; r.c[2][0] = (float)i02; r.c[3][0] = (float)(-(i00 * tx + i01 * ty + i02 * tz));
  r.c[0][1] = (float)i10; r.c[1][1] = (float)i11; r.c[2][1] = (float)i12; r.c[3][1] = (float)(-(i10 * tx + i11 * ty + i12 * tz));
  r.c[0][2] = (float)i20; r.c[1][2] = (float)i21; r.c[2][2] = (float)i22; r.c[3][2] = (float)(-(i20 * tx + i21 * ty + i22 * tz));
  r.c[0][3] = 0.0f; r.c[1][3] = 0.0f; r.c[2][3] = 0.0f; r.c[3][3] = 1.0f;
  return r;
}

// complex numbers as used by FrComplexConductor (include/cmaterial.h:685-694)
struct cplx { float re, im; };
static inline cplx cmk(float re, float im) { cplx r = {re, im}; return r; }
static inline cplx operator+(cplx a, cplx b) { return cmk(a.re + b.re, a.im + b.im); }
static inline cplx operator-(cplx a, cplx b) { return cmk(a.re - b.re, a.im - b.im); }
static inline cplx operator*(cplx a, cplx b) { return cmk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline cplx operator*(cplx a, float s) { return cmk(a.re * s, a.im * s); }
static inline cplx operator/(cplx a, cplx b)
{
  const float scale = 1.0f / (b.re * b.re + b.im * b.im);
  return cmk(scale * (a.re * b.re + a.im * b.im), scale * (a.im * b.re - a.re * b.im));
}
static inline cplx rsub(float s, cplx a) { return cmk(s - a.re, -a.im); }   // s - a
static inline cplx radd(float s, cplx a) { return cmk(s + a.re, a.im); }    // s + a
static inline cplx rdiv(float s, cplx a) { return cmk(s, 0.0f) / a; }       // s / a
static inline float cnorm(cplx a) { return a.re * a.re + a.im * a.im; }
// principal square root, trig-free form (same formulation pbrt-v4 uses, which cmaterial.h's
// Trowbridge-Reitz code is taken from)
static inline cplx csqrt_(cplx z)
{
  const float n = std::sqrt(cnorm(z));
  if (n == 0.0f) return cmk(0.0f, 0.0f);
  const float t1 = std::sqrt(0.5f * (n + std::abs(z.re)));
  const float t2 = 0.5f * z.im / t1;
  if (z.re >= 0.0f) return cmk(t1, t2);
  return cmk(std::abs(t2), std::copysign(t1, z.im));
}

} // namespace orc
