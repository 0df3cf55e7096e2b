// Host-side precomputation for MAT_TYPE_PLASTIC: mi::fresnel_coat_precompute (mi_materials.cpp:377-451 with eval_transmittance :242-285,
// eval_reflectance :288-329, gauss_legendre :170-228, legendre_pd :134-168) over the Mitsuba-style GGX helpers of include/cmaterial.h
// (:206-209, 619-643, 746-905).  LoadPlasticMaterial (integrator_pt_scene_mat.cpp:675-757) stores the 64-entry transmittance table in
// m_arrays1f and the two scalars in Material::data.  RGB mode only.  Plain C++17, no device code: both scene loaders call the one
// implementation (the Python one through hpt_plastic_precompute), so their tables are identical.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <utility>
#include <vector>

namespace hydra_hip {
namespace plastic {

static const int   TRANSMITTANCE_RES = 64;            // MI_ROUGH_TRANSMITTANCE_RES (include/cglobals.h:18)
static const float EPSILON_32 = 5.960464477539063E-8f;
static const float kPi = 3.14159265358979323846f;

struct F3 { float x, y, z; };
static inline float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline F3 normalize3(F3 a) { const float l = std::sqrt(dot3(a, a)); return F3{ a.x / l, a.y / l, a.z / l }; }
static inline float safeSqrt(float v) { return std::sqrt(std::max(v, 0.0f)); }
static inline float lerpf(float a, float b, float t) { return a + t * (b - a); }
static inline float clampf(float v, float lo, float hi) { return std::min(std::max(v, lo), hi); }

static inline void diskConcentric(float sx, float sy, float& px, float& py)       // square_to_uniform_disk_concentric
{
  const float x = 2.f * sx - 1.f, y = 2.f * sy - 1.f;
  float phi, r;
  if (x == 0 && y == 0) { r = phi = 0; }
  else if (x * x > y * y) { r = x; phi = (kPi / 4.f) * (y / x); }
  else { r = y; phi = (kPi / 2.f) - (x / y) * (kPi / 4.f); }
  px = r * std::cos(phi); py = r * std::sin(phi);
}
static inline float smithG1(F3 v, F3 m, float ax, float ay)
{
  const float xy_alpha_2 = ax * v.x * ax * v.x + ay * v.y * ay * v.y, tan_theta_alpha_2 = xy_alpha_2 / (v.z * v.z);
  float result = 2.f / (1.f + safeSqrt(1.f + tan_theta_alpha_2));
  if (xy_alpha_2 == 0.f) result = 1.f;
  if (v.z * dot3(v, m) <= 0.f) result = 0.f;
  return result;
}
static inline F3 sampleVisibleNormal(F3 wi, float r0, float r1, float ax, float ay)
{
  const F3 wi_p = normalize3(F3{ ax * wi.x, ay * wi.y, wi.z });
  const float sin_theta2 = wi_p.x * wi_p.x + wi_p.y * wi_p.y;                      // sincos_phi
  const float inv_sin_theta = 1.f / safeSqrt(sin_theta2);
  float rx = wi_p.x * inv_sin_theta, ry = wi_p.y * inv_sin_theta;
  if (std::abs(sin_theta2) <= 4.f * EPSILON_32) { rx = 1.f; ry = 0.f; } else { rx = clampf(rx, -1.f, 1.f); ry = clampf(ry, -1.f, 1.f); }
  const float sin_phi = ry, cos_phi = rx;
  const float cos_theta = wi_p.z;
  float px, py; diskConcentric(r0, r1, px, py);                                     // sample_visible_11
  const float s = 0.5f * (1.f + cos_theta);
  py = lerpf(safeSqrt(1.f - px * px), py, s);
  const float x = px, y = py, z = safeSqrt(1.f - (px * px + py * py));
  const float sin_theta_i = safeSqrt(1.f - cos_theta * cos_theta);
  const float norm = 1.f / (sin_theta_i * y + cos_theta * z);
  float slx = (cos_theta * y - sin_theta_i * z) * norm, sly = x * norm;
  const float sx2 = (cos_phi * slx - sin_phi * sly) * ax, sy2 = (sin_phi * slx + cos_phi * sly) * ay;
  return normalize3(F3{ -sx2, -sy2, 1.0f });
}
// FrDielectricDetailed (:619-643): {r, cos_theta_t, eta_it, eta_ti}
static inline void frDielectricDetailed(float cosTheta_i, float eta, float& r, float& cosTheta_t, float& eta_ti)
{
  cosTheta_i = clampf(cosTheta_i, -1.0f, 1.0f);
  if (cosTheta_i < 0.0f) { eta = 1.0f / eta; cosTheta_i = -cosTheta_i; }
  const float sin2Theta_i = 1.0f - cosTheta_i * cosTheta_i, sin2Theta_t = sin2Theta_i / (eta * eta);
  cosTheta_t = safeSqrt(1.0f - sin2Theta_t);
  const float r_parl = (eta * cosTheta_i - cosTheta_t) / (eta * cosTheta_i + cosTheta_t);
  const float r_perp = (cosTheta_i - eta * cosTheta_t) / (cosTheta_i + eta * cosTheta_t);
  r = (r_parl * r_parl + r_perp * r_perp) / 2.0f;
  cosTheta_t = cosTheta_i >= 0 ? -cosTheta_t : cosTheta_t;
  eta_ti = 1.f / eta;
}

template <class V> static inline std::pair<V, V> legendrePd(int l, V x)
{
  V l_cur = V(0), d_cur = V(0);
  if (l > 1) {
    V l_p_pred = V(1), l_pred = x, d_p_pred = V(0), d_pred = V(1);
    V k0 = V(3), k1 = V(2), k2 = V(1);
    for (int ki = 2; ki <= l; ++ki) {
      l_cur = (k0 * x * l_pred - k2 * l_p_pred) / k1;
      d_cur = d_p_pred + k0 * l_pred;
      l_p_pred = l_pred; l_pred = l_cur;
      d_p_pred = d_pred; d_pred = d_cur;
      k2 = k1; k0 += V(2); k1 += V(1);
    }
  } else if (l == 0) { l_cur = V(1); d_cur = V(0); }
  else { l_cur = x; d_cur = V(1); }
  return { l_cur, d_cur };
}
static inline void gaussLegendre(int n, std::vector<float>& nodes, std::vector<float>& weights)
{
  nodes.assign((size_t)n, 0.0f); weights.assign((size_t)n, 0.0f);
  n--;
  if (n == 0) { nodes[0] = 0.0f; weights[0] = 2.0f; }
  else if (n == 1) { nodes[0] = -std::sqrt(1.0f / 3.0f); nodes[1] = -nodes[0]; weights[0] = weights[1] = 1.0f; }
  const int m = (n + 1) / 2;
  for (int i = 0; i < m; ++i) {
    double x = -std::cos((double)(2 * i + 1) / (double)(2 * n + 2) * 3.14159265358979323846);
    for (int it = 0; it < 20; ++it) {                                               // Newton on P_{n+1}
      const std::pair<double, double> L = legendrePd<double>(n + 1, x);
      const double step = L.first / L.second;
      x -= step;
      if (std::abs(step) <= 4 * std::abs(x) * 0x1p-53) break;
    }
    const std::pair<double, double> L = legendrePd<double>(n + 1, x);
    weights[(size_t)i] = weights[(size_t)(n - i)] = (float)(2 / ((1 - x * x) * (L.second * L.second)));
    nodes[(size_t)i] = (float)x; nodes[(size_t)(n - i)] = (float)-x;
  }
  if ((n % 2) == 0) {
    const std::pair<double, double> L = legendrePd<double>(n + 1, 0.0);
    weights[(size_t)(n / 2)] = (float)(2.0 / (L.second * L.second));
    nodes[(size_t)(n / 2)] = 0.0f;
  }
}

// transmit == true: eval_transmittance(alpha, eta); false: eval_reflectance(alpha, eta)
static inline void evalTable(bool transmit, float alpha, float eta, const float* mu, const float* oneMinusMuSqr, float* result)
{
  const int grid = eta > 1.0f ? 32 : 128;
  std::vector<float> nodes, weights; gaussLegendre(grid, nodes, weights);
  for (int i = 0; i < TRANSMITTANCE_RES; ++i) {
    const F3 wi = { oneMinusMuSqr[i], 0.0f, mu[i] };
    float acc = 0.f;
    for (int j = 0; j < grid * grid; ++j) {
      const int ix = j % grid, iy = j / grid;
      const float nx = nodes[(size_t)ix] * 0.5f + 0.5f, ny = nodes[(size_t)iy] * 0.5f + 0.5f;
      const F3 normal = sampleVisibleNormal(wi, nx, ny, alpha, alpha);
      float f, cos_theta_t, eta_ti; frDielectricDetailed(dot3(wi, normal), eta, f, cos_theta_t, eta_ti);
      float smith;
      if (transmit) {
        const float k = dot3(wi, normal) * eta_ti + cos_theta_t;                    // mi::refract
        const F3 wo = { normal.x * k - wi.x * eta_ti, normal.y * k - wi.y * eta_ti, normal.z * k - wi.z * eta_ti };
        smith = smithG1(wo, normal, alpha, alpha) * (1.f - f);
        if (wo.z * wi.z >= 0.f) smith = 0.f;
      } else {
        const float d2 = 2.f * dot3(wi, normal);                                    // mi::reflect
        const F3 wo = { normal.x * d2 - wi.x, normal.y * d2 - wi.y, normal.z * d2 - wi.z };
        smith = smithG1(wo, normal, alpha, alpha) * f;
        if (wo.z <= 0.f) smith = 0.f;
        if (wi.z <= 0.f) smith = 0.f;
      }
      acc += smith * weights[(size_t)ix] * weights[(size_t)iy] * 0.25f;
    }
    result[i] = acc;
  }
}

struct CoatPrecomputed { float transmittance[TRANSMITTANCE_RES]; float internalReflectance, specularSamplingWeight; };

static inline CoatPrecomputed fresnelCoatPrecompute(float alpha, float intIor, float extIor, const float* diffuse4, const float* specular4)
{
  CoatPrecomputed res;
  const float eta = intIor / extIor;
  float d_mean = 0.f, s_mean = 0.f;
  for (int i = 0; i < 3; ++i) { d_mean += diffuse4[i]; s_mean += specular4[i]; }
  d_mean /= 3; s_mean /= 3;
  res.specularSamplingWeight = s_mean / (d_mean + s_mean);
  float mu[TRANSMITTANCE_RES], om[TRANSMITTANCE_RES], refl[TRANSMITTANCE_RES];
  const float delta = (1.0f - 0.0f) / (TRANSMITTANCE_RES - 1);
  for (int i = 0; i < TRANSMITTANCE_RES; ++i) { mu[i] = std::max(float(i) * delta + 0.0f, 1e-6f); om[i] = std::sqrt(1.0f - mu[i] * mu[i]); }
  evalTable(true, alpha, eta, mu, om, res.transmittance);
  evalTable(false, alpha, 1.f / eta, mu, om, refl);
  float ir = 0.0f;
  for (int i = 0; i < TRANSMITTANCE_RES; ++i) ir += refl[i] * mu[i];
  res.internalReflectance = (ir / TRANSMITTANCE_RES) * 2.f;
  return res;
}

} // namespace plastic
} // namespace hydra_hip
