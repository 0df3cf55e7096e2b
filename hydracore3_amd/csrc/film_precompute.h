// Host-side precomputation for MAT_TYPE_THIN_FILM: what LoadThinFilmMaterial (integrator_pt_scene_mat.cpp:1020-1193) appends to
// m_precomp_thin_films - precomputeThinFilmSpectral (:791-890), precomputeThinFilmRGB (:892-1018) over FrFilm / multFrFilm / multFrFilm_r
// (include/airy_reflectance.h:67-209) and the complex Fresnel amplitudes of include/cmaterial.h:957-1036.  Plain C++17, no device code: both
// scene loaders call the one implementation (the Python one through hpt_film_precompute), so their tables are identical.
//
// One material's table = four blocks {reflectance from outside, transmittance from outside, reflectance from inside, transmittance from inside}:
//   spectral rendering : FILM_LENGTH_RES x FILM_ANGLE_RES floats per block ([wavelength][angle])
//   RGB rendering      : thicknessRes x FILM_ANGLE_RES x 3 floats per block ([thickness][angle][rgb]); thicknessRes = FILM_THICKNESS_RES with a
//                        thickness map (or a single layer), else 1
// RGB: the reference integrates each angle's 94-point spectrum against its `spectral` library (external/spectral: spectre2xyz under CIE D65, 1 nm,
// then xyz2rgb with clamping). Restated here over the observer table the loader holds (m_cie_xyz) and the CIE D65 illuminant at 10 nm (the
// standard's 5 nm table is its linear interpolation); the normalisation is the Y integral of that pair, so a unit spectrum maps to Y = 1.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace hydra_hip {
namespace film {

static const uint32_t ANGLE_RES = 180, LENGTH_RES = 94, THICKNESS_RES = 32;   // include/cglobals.h:19-21
static const float LAMBDA_MIN = 360.0f, LAMBDA_MAX = 830.0f;                  // :22-23
static const uint32_t MAX_LAYERS = 64;

struct Cx { float re, im; };
static inline Cx cx(float re, float im) { Cx r = { re, im }; return r; }
static inline Cx operator+(Cx a, Cx b) { return cx(a.re + b.re, a.im + b.im); }
static inline Cx operator-(Cx a, Cx b) { return cx(a.re - b.re, a.im - b.im); }
static inline Cx operator*(Cx a, Cx b) { return cx(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline Cx operator*(Cx a, float s) { return cx(a.re * s, a.im * s); }
static inline Cx operator/(Cx a, Cx b) { const float scale = 1.0f / (b.re * b.re + b.im * b.im); return cx(scale * (a.re * b.re + a.im * b.im), scale * (a.im * b.re - a.re * b.im)); }
static inline Cx rsub(float s, Cx a) { return cx(s - a.re, -a.im); }
static inline Cx radd(float s, Cx a) { return cx(s + a.re, a.im); }
static inline float cnorm(Cx a) { return a.re * a.re + a.im * a.im; }
static inline Cx csqrt_(Cx z)
{
  const float n = std::sqrt(cnorm(z));
  if (n == 0.0f) return cx(0.0f, 0.0f);
  const float t1 = std::sqrt(0.5f * (n + std::abs(z.re)));
  const float t2 = 0.5f * z.im / t1;
  if (z.re >= 0.0f) return cx(t1, t2);
  return cx(std::abs(t2), std::copysign(t1, z.im));
}
struct ReflRefr { float refl, refr; };
static inline float refractionFactor(float cosThetaI, Cx cosThetaT, Cx iorI, Cx iorT)            // getRefractionFactor (cmaterial.h:967-975)
{
  const Cx mult = cosThetaT * iorT;
  if (cosThetaI <= 1e-6f || mult.im > 1e-6f) return 0.0f;
  return mult.re / (iorI.re * cosThetaI);
}
static inline Cx complexRefl(Cx cosThetaI, Cx cosThetaT, Cx iorI, Cx iorT, int polP)             // FrComplexRefl (:995-1010)
{
  if (cnorm(cosThetaI) < 1e-6f) return cx(-1.0f, 0.0f);
  if (!polP) return (iorI * cosThetaI - iorT * cosThetaT) / (iorI * cosThetaI + iorT * cosThetaT);
  return (iorT * cosThetaI - iorI * cosThetaT) / (iorT * cosThetaI + iorI * cosThetaT);
}
static inline Cx complexRefr(Cx cosThetaI, Cx cosThetaT, Cx iorI, Cx iorT, int polP)             // FrComplexRefr (:1012-1031)
{
  if (cnorm(cosThetaI) < 1e-6f) return (cnorm(iorI - iorT) < 1e-6f) ? cx(1.0f, 0.0f) : cx(0.0f, 0.0f);
  if (!polP) return ((iorI * 2.0f) * cosThetaI) / (iorI * cosThetaI + iorT * cosThetaT);
  return ((iorI * 2.0f) * cosThetaI) / (iorT * cosThetaI + iorI * cosThetaT);
}
static inline Cx phaseDiff(Cx cosTheta, Cx eta, float thickness, float lambda)                   // filmPhaseDiff (:1033-1036)
{ return (((eta * 12.566370614359172f) * cosTheta) * thickness) / cx(lambda, 0.0f); }
static inline Cx phaseExp(Cx pd) { return cx(std::cos(pd.re / 2.f), std::sin(pd.re / 2.f)) * std::exp(-pd.im / 2.f); }

static inline ReflRefr frFilm(float cosThetaI, Cx etaI, Cx etaF, Cx etaT, float thickness, float lambda)   // FrFilm (airy_reflectance.h:67-106)
{
  const Cx sinThetaI = cx(1.0f - cosThetaI * cosThetaI, 0.0f);
  const Cx cosThetaF = csqrt_(rsub(1.0f, (sinThetaI * (etaI.re * etaI.re)) / (etaF * etaF)));
  const Cx cosThetaT = csqrt_(rsub(1.0f, (sinThetaI * (etaI.re * etaI.re)) / (etaT * etaT)));
  const Cx pd = phaseDiff(cosThetaF, etaF, thickness, lambda);
  ReflRefr result = { 0, 0 };
  for (int p = 0; p <= 1; ++p) {
    const Cx ReflI = complexRefl(cx(cosThetaI, 0.0f), cosThetaF, etaI, etaF, p), ReflF = complexRefl(cosThetaF, cosThetaT, etaF, etaT, p);
    const Cx RefrI = complexRefr(cx(cosThetaI, 0.0f), cosThetaF, etaI, etaF, p), RefrF = complexRefr(cosThetaF, cosThetaT, etaF, etaT, p);
    const Cx exp_1 = phaseExp(pd), exp_2 = exp_1 * exp_1;
    const Cx denom = radd(1.0f, (ReflI * ReflF) * exp_2);
    if (cnorm(denom) < 1e-6f) result.refl += 0.5f;
    else { result.refl += cnorm((ReflI + ReflF * exp_2) / denom) / 2; result.refr += cnorm(((RefrI * RefrF) * exp_1) / denom) / 2; }
  }
  result.refr *= refractionFactor(cosThetaI, cosThetaT, etaI, etaT);
  return result;
}
// calculateMultFrFilmForward / Backward (airy_reflectance.h:108-164): the stack folded interface by interface, from the far side
static inline ReflRefr multFold(const Cx* cosTheta, const Cx* ior, const Cx* pd, uint32_t layers, int p, bool backward)
{
  Cx Refl, Refr;
  if (!backward) {
    Refl = complexRefl(cosTheta[layers - 1], cosTheta[layers], ior[layers - 1], ior[layers], p);
    Refr = complexRefr(cosTheta[layers - 1], cosTheta[layers], ior[layers - 1], ior[layers], p);
  } else {
    Refl = complexRefl(cosTheta[1], cosTheta[0], ior[1], ior[0], p);
    Refr = complexRefr(cosTheta[1], cosTheta[0], ior[1], ior[0], p);
  }
  for (int k = 0; k + 1 < (int)layers; ++k) {
    const int i = backward ? k + 1 : (int)layers - 2 - k;
    const Cx ReflI = backward ? complexRefl(cosTheta[i + 1], cosTheta[i], ior[i + 1], ior[i], p) : complexRefl(cosTheta[i], cosTheta[i + 1], ior[i], ior[i + 1], p);
    const Cx RefrI = backward ? complexRefr(cosTheta[i + 1], cosTheta[i], ior[i + 1], ior[i], p) : complexRefr(cosTheta[i], cosTheta[i + 1], ior[i], ior[i + 1], p);
    const Cx exp_1 = phaseExp(pd[backward ? i - 1 : i]);
    Refr = (RefrI * Refr) * exp_1;
    Refl = (Refl * exp_1) * exp_1;
    const Cx denom = radd(1.0f, ReflI * Refl);
    if (cnorm(denom) < 1e-6f) { Refr = cx(0.0f, 0.0f); Refl = cx(1.0f, 0.0f); }
    else { Refr = Refr / denom; Refl = (ReflI + Refl) / denom; }
  }
  ReflRefr r = { cnorm(Refl), cnorm(Refr) };
  return r;
}
// multFrFilm / multFrFilm_r (airy_reflectance.h:166-209): ior[0] = outside medium, ior[1 .. layers - 1] = films, ior[layers] = substrate
static inline ReflRefr multFrFilm(float cosThetaI, const Cx* ior, const float* thickness, uint32_t layers, float lambda, bool reversed)
{
  Cx cosTheta[MAX_LAYERS + 2], pd[MAX_LAYERS + 1];
  const float sinThetaI = 1.0f - cosThetaI * cosThetaI;
  if (!reversed) {
    cosTheta[0] = cx(cosThetaI, 0.0f);
    for (uint32_t i = 1; i <= layers; ++i) {
      cosTheta[i] = csqrt_(rsub(1.0f, cx(sinThetaI * ior[0].re * ior[0].re, 0.0f) / (ior[i] * ior[i])));      // sinThetaI * re * re / (ior * ior): left to right
      if (i < layers) pd[i - 1] = phaseDiff(cosTheta[i], ior[i], thickness[i - 1], lambda);
    }
  } else {
    cosTheta[layers] = cx(cosThetaI, 0.0f);
    for (int i = (int)layers - 1; i >= 0; --i) {
      cosTheta[i] = csqrt_(rsub(1.0f, cx(sinThetaI * ior[layers].re * ior[layers].re, 0.0f) / (ior[i] * ior[i])));
      if (i > 0) pd[i - 1] = phaseDiff(cosTheta[i], ior[i], thickness[i - 1], lambda);
    }
  }
  const ReflRefr P = multFold(cosTheta, ior, pd, layers, 1, reversed), S = multFold(cosTheta, ior, pd, layers, 0, reversed);
  ReflRefr r = { (P.refl + S.refl) / 2.f, (P.refr + S.refr) / 2.f };
  r.refr *= !reversed ? refractionFactor(cosThetaI, cosTheta[layers], ior[0], ior[layers]) : refractionFactor(cosThetaI, cosTheta[0], ior[layers], ior[0]);
  return r;
}

// SampleUniformSpectrum at one wavelength (spectrum.h:106-126)
static inline float sampleUniformSpectrum(const float* vals, uint32_t offset, float w)
{
  const int WAVESN = int(LAMBDA_MAX - LAMBDA_MIN);
  const int i1 = (int)std::min(std::max(w - LAMBDA_MIN, 0.0f), float(WAVESN - 1)), i2 = std::min(i1 + 1, WAVESN - 1);
  const float y1 = vals[offset + (uint32_t)i1], y2 = vals[offset + (uint32_t)i2];
  return y1 + (w - (LAMBDA_MIN + float(i1))) * (y2 - y1);
}

// CIE standard illuminant D65, 300 .. 830 nm at 10 nm (relative spectral power, 100 at 560 nm)
static const float D65_10NM[54] = {
  0.0341f, 3.2945f, 20.236f, 37.0535f, 39.9488f, 44.9117f, 46.6383f, 52.0891f, 49.9755f, 54.6482f, 82.7549f, 91.486f, 93.4318f, 86.6823f, 104.865f, 117.008f,
  117.812f, 114.861f, 115.923f, 108.811f, 109.354f, 107.802f, 104.79f, 107.689f, 104.405f, 104.046f, 100.0f, 96.3342f, 95.788f, 88.6856f, 90.0062f, 89.5991f,
  87.6987f, 83.2886f, 83.6992f, 80.0268f, 80.2146f, 82.2778f, 78.2842f, 69.7213f, 71.6091f, 74.349f, 61.604f, 69.8856f, 75.087f, 63.5927f, 46.4182f, 66.8054f,
  63.3828f, 64.304f, 59.4519f, 51.959f, 57.4406f, 60.3125f };
static inline float d65(int lambda) { const int k = (lambda - 300) / 10; if (lambda < 300 || lambda > 830) return 0.0f; if (k >= 53) return D65_10NM[53]; const float t = float(lambda - 300 - 10 * k) / 10.0f; return D65_10NM[k] + (D65_10NM[k + 1] - D65_10NM[k]) * t; }

struct Params
{
  int spectralMode; float extIOR; uint32_t layers;            // layers: FILM_LAYERS_COUNT (the films plus the substrate)
  const float* eta; const float* k;                            // per layer (m_films_eta_k_vec at FILM_ETA_OFFSET / FILM_K_OFFSET)
  const uint32_t* etaSpecId; const uint32_t* kSpecId;          // per layer (m_films_spec_id_vec), 0xFFFFFFFF: the constant
  const float* thickness;                                      // per film (m_films_thickness_vec at FILM_THICKNESS_OFFSET)
  int thicknessMap; float thicknessMin, thicknessMax;
  const float* specValues; const uint32_t* specOffsetSz; uint32_t numSpectra;
  const float* cieXYZ;                                         // float4 x 471 (RGB rendering only)
};
static inline uint32_t thicknessRes(const Params& p) { return (p.thicknessMap || p.layers == 1) ? THICKNESS_RES : 1u; }   // (integrator_pt_scene_mat.cpp:1163)
static inline size_t tableSize(const Params& p) { return p.spectralMode ? size_t(4) * ANGLE_RES * LENGTH_RES : size_t(4) * ANGLE_RES * 3 * thicknessRes(p); }
// whether LoadThinFilmMaterial precomputes at all (:1147): always in RGB; in spectral mode unless one film with a thickness map
static inline bool precomputed(const Params& p) { return p.spectralMode == 0 || !p.thicknessMap || p.layers > 2; }

static inline void layerIors(const Params& p, float wavelength, Cx* ior)
{
  ior[0] = cx(p.extIOR, 0.f);
  for (uint32_t l = 0; l < p.layers; ++l) {
    float eta = p.eta[l], k = p.k[l];
    if (p.etaSpecId[l] < 0xFFFFFFFFu && p.etaSpecId[l] < p.numSpectra && p.specValues) eta = sampleUniformSpectrum(p.specValues, p.specOffsetSz[2 * p.etaSpecId[l]], wavelength);
    if (p.kSpecId[l] < 0xFFFFFFFFu && p.kSpecId[l] < p.numSpectra && p.specValues) k = sampleUniformSpectrum(p.specValues, p.specOffsetSz[2 * p.kSpecId[l]], wavelength);
    ior[l + 1] = cx(eta, k);
  }
}
static inline void bothWays(const Params& p, const Cx* ior, float cosTheta, float thickness0, float wavelength, ReflRefr& fwd, ReflRefr& bwd)
{
  if (p.layers == 2) { fwd = frFilm(cosTheta, ior[0], ior[1], ior[2], thickness0, wavelength); bwd = frFilm(cosTheta, ior[2], ior[1], ior[0], thickness0, wavelength); }
  else { fwd = multFrFilm(cosTheta, ior, p.thickness, p.layers, wavelength, false); bwd = multFrFilm(cosTheta, ior, p.thickness, p.layers, wavelength, true); }
}
static inline float angleCos(int j) { const float theta = float(M_PI / 2 / float(ANGLE_RES - 1) * j); return std::min(std::max(std::cos(theta), 1e-3f), 1.f); }

// out: tableSize(p) floats; returns false for parameters the reference's loader cannot have produced
static inline bool precompute(const Params& p, float* out)
{
  if (p.layers < 1 || p.layers > MAX_LAYERS || !p.eta || !p.k || !p.etaSpecId || !p.kSpecId || !out) return false;
  if (p.layers >= 2 && !p.thickness) return false;
  if (!p.spectralMode && !p.cieXYZ) return false;
  Cx ior[MAX_LAYERS + 2];
  if (p.spectralMode) {                                                                 // precomputeThinFilmSpectral (:791-890)
    const size_t blk = size_t(ANGLE_RES) * LENGTH_RES;
    for (uint32_t i = 0; i < LENGTH_RES; ++i) {
      const float wavelength = (LAMBDA_MAX - LAMBDA_MIN - 1) / (LENGTH_RES - 1) * i + LAMBDA_MIN;   // (the reference's "- 1": 360 .. 829)
      layerIors(p, wavelength, ior);
      for (uint32_t j = 0; j < ANGLE_RES; ++j) {
        ReflRefr f, b;
        bothWays(p, ior, angleCos((int)j), p.thickness ? p.thickness[0] : 0.0f, wavelength, f, b);
        out[0 * blk + i * ANGLE_RES + j] = f.refl; out[1 * blk + i * ANGLE_RES + j] = f.refr;
        out[2 * blk + i * ANGLE_RES + j] = b.refl; out[3 * blk + i * ANGLE_RES + j] = b.refr;
      }
    }
    return true;
  }
  // precomputeThinFilmRGB (:892-1018)
  const uint32_t tres = thicknessRes(p);
  const size_t blk = size_t(ANGLE_RES) * 3 * tres;
  double yint = 0.0;
  for (int lambda = 360; lambda <= 830; ++lambda) yint += double(p.cieXYZ[4 * (lambda - 360) + 1]) * double(d65(lambda));
  std::vector<float> keys(LENGTH_RES), spec(size_t(4) * ANGLE_RES * LENGTH_RES);
  for (uint32_t t = 0; t < tres; ++t) {
    const float thickness = tres == 1 ? (p.thickness ? p.thickness[0] : 0.0f) : (p.thicknessMax - p.thicknessMin) / (tres - 1) * t + p.thicknessMin;
    for (uint32_t i = 0; i < LENGTH_RES; ++i) {
      const float wavelength = (LAMBDA_MAX - LAMBDA_MIN) / (LENGTH_RES - 1) * i + LAMBDA_MIN;
      keys[i] = wavelength;
      layerIors(p, wavelength, ior);
      for (uint32_t j = 0; j < ANGLE_RES; ++j) {
        ReflRefr f, b;
        bothWays(p, ior, angleCos((int)j), thickness, wavelength, f, b);
        float* s = &spec[(size_t(j) * LENGTH_RES + i) * 4];
        s[0] = f.refl; s[1] = f.refr; s[2] = b.refl; s[3] = b.refr;
      }
    }
    for (uint32_t j = 0; j < ANGLE_RES; ++j) {
      double xyz[4][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
      uint32_t hi = 0;                                                                  // BasicSpectrum::get_or_interpolate: first key >= lambda
      for (int lambda = 360; lambda <= 830; ++lambda) {
        const float w = float(lambda);
        while (hi < LENGTH_RES && keys[hi] < w) hi++;
        float v[4] = { 0, 0, 0, 0 };
        if (hi < LENGTH_RES) {
          const float* sb = &spec[(size_t(j) * LENGTH_RES + hi) * 4];
          if (keys[hi] == w) { for (int q = 0; q < 4; q++) v[q] = sb[q]; }
          else if (hi > 0) { const float* sa = sb - 4; const float a = keys[hi - 1], b2 = keys[hi]; for (int q = 0; q < 4; q++) v[q] = sa[q] + (sb[q] - sa[q]) * (w - a) / (b2 - a); }
        }
        const float light = d65(lambda);
        const float* c = p.cieXYZ + 4 * (lambda - 360);
        for (int q = 0; q < 4; q++) { const double val = double(v[q] * light); xyz[q][0] += c[0] * val; xyz[q][1] += c[1] * val; xyz[q][2] += c[2] * val; }
      }
      for (int q = 0; q < 4; q++) {
        const float X = float(xyz[q][0] / yint), Y = float(xyz[q][1] / yint), Z = float(xyz[q][2] / yint);
        const float rgb[3] = { 3.2404542f * X - 1.5371385f * Y - 0.4985314f * Z, -0.9692660f * X + 1.8760108f * Y + 0.0415560f * Z, 0.0556434f * X - 0.2040259f * Y + 1.0572252f * Z };
        for (int ch = 0; ch < 3; ch++) out[q * blk + (size_t(ANGLE_RES) * t + j) * 3 + ch] = std::min(std::max(rgb[ch], 0.0f), 1.0f);   // xyz2rgb clamps
      }
    }
  }
  return true;
}

} // namespace film
} // namespace hydra_hip
