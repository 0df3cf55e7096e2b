// MAT_TYPE_THIN_FILM on the device: include/cmat_film.h (filmSmoothSampleAndEval :9-181, filmRoughSampleAndEval :183-410, filmRoughEval :413-544),
// include/airy_reflectance.h (FrFilm :67-106, FrFilmRefl :9-33), the complex Fresnel amplitudes of include/cmaterial.h:957-1036 and the set-up both
// dispatch branches share (integrator_pt_mat.cpp:197-249, 422-470; SampleFilmsSpectrum, integrator_spectrum.cpp:46-65).
//
// RGB rendering always reads the loader's tables (m_precomp_thin_films: reflectance and transmittance over the angle, or over thickness x angle
// with a thickness map, three channels, from outside and from inside); spectral rendering reads its own tables over wavelength x angle, or -
// one film on a substrate with a thickness map - evaluates the Airy summation per vertex. A film always marks the path RAY_FLAG_WAVES_DIVERGED:
// in spectral mode only the first wavelength is computed and only it reaches the image.
//
// Compiled into the FILM kernel variants only (pathTraceKernel MODE 4 / 5 / 6, the spectral kernel): scenes without films run the same
// kernels as before.
#pragma once
#include "hpt_device.h"

namespace hpt {

enum : uint { MAT_TYPE_THIN_FILM = 8 };                                                              // include/cmaterial.h:45
enum : uint { FILM_ANGLE_RES = 180, FILM_LENGTH_RES = 94, FILM_THICKNESS_RES = 32 };                 // include/cglobals.h:19-21
enum { FILM_ROUGH_U = 0, FILM_ROUGH_V = 1, FILM_PRECOMP_FLAG = 2, FILM_PRECOMP_OFFSET = 3, FILM_ETA_OFFSET = 4, FILM_K_OFFSET = 5, FILM_ETA_SPECID_OFFSET = 6,
       FILM_K_SPECID_OFFSET = 7, FILM_ETA_EXT = 8, FILM_THICKNESS_OFFSET = 9, FILM_THICKNESS_MIN = 10, FILM_THICKNESS_MAX = 11, FILM_THICKNESS_MAP = 12,
       FILM_THICKNESS = 13, FILM_LAYERS_COUNT = 14, FILM_TRANSPARENT = 15 };                         // include/cmaterial.h:164-179
static constexpr float LAMBDA_MIN = 360.0f, LAMBDA_MAX = 830.0f;                                     // include/cglobals.h:22-23

// SampleUniformSpectrum (spectrum.h:106-126): 1 nm table starting at LAMBDA_MIN, linear between neighbours
HPT_DEV float sampleUniformSpectrum1(const float* vals, uint offset, float w)
{
  const int WAVESN = int(LAMBDA_MAX - LAMBDA_MIN);
  const int i1 = (int)smin(smax(w - LAMBDA_MIN, 0.0f), float(WAVESN - 1));
  const int i2 = min(i1 + 1, WAVESN - 1);
  const float x1 = LAMBDA_MIN + float(i1);
  const float y1 = vals[offset + (uint)i1], y2 = vals[offset + (uint)i2];
  return y1 + (w - x1) * (y2 - y1);
}

HPT_DEV Cx rsubc(float s, Cx a) { return cx(s - a.re, -a.im); }     // s - a
HPT_DEV Cx raddc(float s, Cx a) { return cx(s + a.re, a.im); }      // s + a
HPT_DEV float getRefractionFactor(float cosThetaI, Cx cosThetaT, Cx iorI, Cx iorT)                   // cmaterial.h:967-975
{
  const Cx mult = cosThetaT * iorT;
  if (cosThetaI <= 1e-6f || mult.im > 1e-6f) return 0.0f;
  return mult.re / (iorI.re * cosThetaI);
}
HPT_DEV Cx frComplexRefl(Cx cosThetaI, Cx cosThetaT, Cx iorI, Cx iorT, int polP)                     // :995-1010 (polP: 0 = PolarizationS, 1 = PolarizationP)
{
  if (cnorm(cosThetaI) < 1e-6f) return cx(-1.0f, 0.0f);
  if (!polP) return (iorI * cosThetaI - iorT * cosThetaT) / (iorI * cosThetaI + iorT * cosThetaT);
  return (iorT * cosThetaI - iorI * cosThetaT) / (iorT * cosThetaI + iorI * cosThetaT);
}
HPT_DEV Cx frComplexRefr(Cx cosThetaI, Cx cosThetaT, Cx iorI, Cx iorT, int polP)                     // :1012-1031
{
  if (cnorm(cosThetaI) < 1e-6f) return (cnorm(iorI - iorT) < 1e-6f) ? cx(1.0f, 0.0f) : cx(0.0f, 0.0f);
  if (!polP) return ((iorI * 2.0f) * cosThetaI) / (iorI * cosThetaI + iorT * cosThetaT);
  return ((iorI * 2.0f) * cosThetaI) / (iorT * cosThetaI + iorI * cosThetaT);
}
HPT_DEV Cx filmPhaseDiff(Cx cosTheta, Cx eta, float thickness, float lambda)                         // :1033-1036
{ return (((eta * 12.566370614359172f) * cosTheta) * thickness) / cx(lambda, 0.0f); }
HPT_DEV void filmCosines(float cosThetaI, Cx etaI, Cx etaF, Cx etaT, Cx& cosThetaF, Cx& cosThetaT)   // airy_reflectance.h:11-15
{
  const Cx sinThetaI = cx(1.0f - cosThetaI * cosThetaI, 0.0f);
  const Cx sinThetaF = (sinThetaI * (etaI.re * etaI.re)) / (etaF * etaF);
  cosThetaF = csqrt_(rsubc(1.0f, sinThetaF));
  const Cx sinThetaT = (sinThetaI * (etaI.re * etaI.re)) / (etaT * etaT);
  cosThetaT = csqrt_(rsubc(1.0f, sinThetaT));
}
// FrFilm (airy_reflectance.h:67-106) with wantT, FrFilmRefl (:9-33) without: reflectance (and transmittance) of one film between two media
HPT_DEV void frFilm(float cosThetaI, Cx etaI, Cx etaF, Cx etaT, float thickness, float lambda, bool wantT, float& refl, float& refr)
{
  Cx cosThetaF, cosThetaT;
  filmCosines(cosThetaI, etaI, etaF, etaT, cosThetaF, cosThetaT);
  const Cx phaseDiff = filmPhaseDiff(cosThetaF, etaF, thickness, lambda);
  refl = 0.0f; refr = 0.0f;
  for (int p = 0; p <= 1; ++p) {
    const Cx FrReflI = frComplexRefl(cx(cosThetaI, 0.0f), cosThetaF, etaI, etaF, p);
    const Cx FrReflF = frComplexRefl(cosThetaF, cosThetaT, etaF, etaT, p);
    if (wantT) {
      const Cx FrRefrI = frComplexRefr(cx(cosThetaI, 0.0f), cosThetaF, etaI, etaF, p);
      const Cx FrRefrF = frComplexRefr(cosThetaF, cosThetaT, etaF, etaT, p);
      const Cx exp_1 = cx(cosf(phaseDiff.re / 2), sinf(phaseDiff.re / 2)) * expf(-phaseDiff.im / 2);
      const Cx exp_2 = exp_1 * exp_1;
      const Cx denom = raddc(1.0f, (FrReflI * FrReflF) * exp_2);
      if (cnorm(denom) < 1e-6f) refl += 0.5f;
      else {
        refl += cnorm((FrReflI + FrReflF * exp_2) / denom) / 2;
        refr += cnorm(((FrRefrI * FrRefrF) * exp_1) / denom) / 2;
      }
    } else {
      Cx FrRefl = (FrReflF * expf(-phaseDiff.im)) * cx(cosf(phaseDiff.re), sinf(phaseDiff.re));
      FrRefl = (FrReflI + FrRefl) / raddc(1.0f, FrReflI * FrRefl);
      refl += cnorm(FrRefl);
    }
  }
  if (wantT) refr *= getRefractionFactor(cosThetaI, cosThetaT, etaI, etaT); else refl = refl / 2;
}

// the position on the tables' angle axis (the reference divides by the double M_PI)
HPT_DEV float filmThetaIndex(float cosThetaI)
{ return clampf(float(double(acosf(cosThetaI) * 2.f) / 3.14159265358979323846), 0.f, 1.f) * float(FILM_ANGLE_RES - 1); }
HPT_DEV float filmLerp2D(const float* pre, uint base, uint stride, uint ch, float x, float theta, uint xRes)
{
  const uint index1 = min(uint(x), uint(xRes - 2)), index2 = min(uint(theta), uint(FILM_ANGLE_RES - 2));
  const float alpha = x - float(index1), beta = theta - float(index2);
  const uint a = (base + index1 * FILM_ANGLE_RES + index2) * stride + ch, b = (base + (index1 + 1) * FILM_ANGLE_RES + index2) * stride + ch;
  const float v0 = lerpf(pre[a], pre[b], alpha), v1 = lerpf(pre[a + stride], pre[b + stride], alpha);
  return lerpf(v0, v1, beta);
}

// what both branches set up before the BSDF call
struct FilmArgs { float extIOR, thickness, lambda; Cx intIOR, filmIOR; bool spectral, precomp; const float* pre; };
HPT_DEV float sampleFilmsSpectrum(const DevScene& S, const MaterialRec& m, float wavelength, int paramId, int paramSpecId, uint layer)   // integrator_spectrum.cpp:46-65
{
  float res = S.filmsEtaK[__float_as_uint(m.data[paramId]) + layer];
  const uint specId = S.filmsSpecId[__float_as_uint(m.data[paramSpecId]) + layer];
  if (specId < 0xFFFFFFFFu) res = sampleUniformSpectrum1(S.specValues, S.specOffsetSz[2u * specId], wavelength);   // (the CPU reference samples the spectrum in RGB mode too)
  return res;
}
// the substrate's index: at the path's first wavelength, or at 525 nm in RGB mode (integrator_pt_mat.cpp:209-213)
HPT_DEV Cx filmIntIOR(const DevScene& S, const MaterialRec& m, float wave0)
{
  const uint layers = __float_as_uint(m.data[FILM_LAYERS_COUNT]);
  const float waveSample = wave0 > 0.0f ? wave0 : 525.f;
  return cx(sampleFilmsSpectrum(S, m, waveSample, FILM_ETA_OFFSET, FILM_ETA_SPECID_OFFSET, layers - 1), sampleFilmsSpectrum(S, m, waveSample, FILM_K_OFFSET, FILM_K_SPECID_OFFSET, layers - 1));
}
HPT_DEV FilmArgs filmArgs(const DevScene& S, const MaterialRec& m, V2 uv, float wave0, Cx intIOR)
{
  FilmArgs a;
  a.spectral = wave0 > 0.0f;
  a.lambda = wave0;                                                 // wavelengths_spec[0]; the RGB triples of :208 / :436 are never read
  a.extIOR = m.data[FILM_ETA_EXT];
  a.intIOR = intIOR;
  a.precomp = __float_as_uint(m.data[FILM_PRECOMP_FLAG]) > 0u;
  a.filmIOR = cx(1.0f, 0.0f);
  if (!a.precomp) a.filmIOR = cx(sampleFilmsSpectrum(S, m, wave0, FILM_ETA_OFFSET, FILM_ETA_SPECID_OFFSET, 0), sampleFilmsSpectrum(S, m, wave0, FILM_K_OFFSET, FILM_K_SPECID_OFFSET, 0));   // (only the Airy summation reads it)
  if (__float_as_uint(m.data[FILM_THICKNESS_MAP]) > 0u) {
    const V4 tv = texSample(S.textures, m.texid[2], mulRows2x4(m.row0[2], m.row1[2], uv));
    const float tmax = m.data[FILM_THICKNESS_MAX], tmin = m.data[FILM_THICKNESS_MIN];
    a.thickness = (tmax - tmin) * tv.x + tmin;
  } else a.thickness = m.data[FILM_THICKNESS];
  a.pre = S.precompThinFilms + (a.precomp ? __float_as_uint(m.data[FILM_PRECOMP_OFFSET]) : 0u);
  return a;
}
HPT_DEV FilmArgs filmArgs(const DevScene& S, const MaterialRec& m, V2 uv, float wave0) { return filmArgs(S, m, uv, wave0, filmIntIOR(S, m, wave0)); }
// MaterialEval's film branch (integrator_pt_mat.cpp:422-470): rough films only, and filmRoughEval returns zero over a dielectric - decided from
// the substrate's index alone, before the thickness map and the tables are touched
HPT_DEV void filmEvalBranch(const DevScene& S, const MaterialRec& m, V2 uv, float wave0, V3 l, V3 v, V3 n, V3 alpha_tex, BsdfE& res);
// reflectance / transmittance at an angle (cmat_film.h:41-143, 227-329, 461-535); spectral mode fills .x only
HPT_DEV void filmReflTrans(const MaterialRec& m, const FilmArgs& a, float cosThetaI, bool reversed, bool wantT, V3& R, V3& T)
{
  const uint refl_offset = reversed ? FILM_ANGLE_RES * 2 : 0, refr_offset = reversed ? FILM_ANGLE_RES * 3 : FILM_ANGLE_RES;
  R = v3(0, 0, 0); T = v3(0, 0, 0);
  if (a.spectral) {
    if (a.precomp) {
      const float w = clampf((a.lambda - LAMBDA_MIN) / (LAMBDA_MAX - LAMBDA_MIN), 0.f, 1.f) * float(FILM_LENGTH_RES - 1);
      const float theta = filmThetaIndex(cosThetaI);
      R.x = filmLerp2D(a.pre, refl_offset * FILM_LENGTH_RES, 1, 0, w, theta, FILM_LENGTH_RES);
      if (wantT) T.x = filmLerp2D(a.pre, refr_offset * FILM_LENGTH_RES, 1, 0, w, theta, FILM_LENGTH_RES);
    } else if (!reversed) frFilm(cosThetaI, cx(a.extIOR, 0.0f), a.filmIOR, a.intIOR, a.thickness, a.lambda, wantT, R.x, T.x);
    else                  frFilm(cosThetaI, a.intIOR, a.filmIOR, cx(a.extIOR, 0.0f), a.thickness, a.lambda, wantT, R.x, T.x);
  } else {
    const float theta = filmThetaIndex(cosThetaI);
    if (__float_as_uint(m.data[FILM_THICKNESS_MAP]) == 1u) {
      const float tmin = m.data[FILM_THICKNESS_MIN], tmax = m.data[FILM_THICKNESS_MAX];
      const float tt = clampf((a.thickness - tmin) / (tmax - tmin), 0.f, 1.f) * float(FILM_THICKNESS_RES - 1);
      R = v3(filmLerp2D(a.pre, refl_offset * FILM_THICKNESS_RES, 3, 0, tt, theta, FILM_THICKNESS_RES), filmLerp2D(a.pre, refl_offset * FILM_THICKNESS_RES, 3, 1, tt, theta, FILM_THICKNESS_RES),
             filmLerp2D(a.pre, refl_offset * FILM_THICKNESS_RES, 3, 2, tt, theta, FILM_THICKNESS_RES));
      if (wantT) T = v3(filmLerp2D(a.pre, refr_offset * FILM_THICKNESS_RES, 3, 0, tt, theta, FILM_THICKNESS_RES), filmLerp2D(a.pre, refr_offset * FILM_THICKNESS_RES, 3, 1, tt, theta, FILM_THICKNESS_RES),
                        filmLerp2D(a.pre, refr_offset * FILM_THICKNESS_RES, 3, 2, tt, theta, FILM_THICKNESS_RES));
    } else {
      const uint index = min(uint(theta), uint(FILM_ANGLE_RES - 2));
      const float alpha = theta - float(index);
      const float* r0 = a.pre + (refl_offset + index) * 3; const float* t0 = a.pre + (refr_offset + index) * 3;
      R = v3(lerpf(r0[0], r0[3], alpha), lerpf(r0[1], r0[4], alpha), lerpf(r0[2], r0[5], alpha));
      if (wantT) T = v3(lerpf(t0[0], t0[3], alpha), lerpf(t0[1], t0[4], alpha), lerpf(t0[2], t0[5], alpha));
    }
  }
}
HPT_DEV float sum3(V3 v) { return v.x + v.y + v.z; }                  // sum(float4) over (r, g, b, 0): ((x + y) + z) + 0
HPT_DEV float microfacet_G(V3 wi, V3 wo, V3 mm, V2 alpha) { return smith_g1(wi, mm, alpha) * smith_g1(wo, mm, alpha); }   // cmaterial.h:903-906

// filmSmoothSampleAndEval (cmat_film.h:9-181); n: the geometric normal; prevIor: a_misPrev->ior
HPT_DEV void filmSmoothSampleAndEval(const MaterialRec& m, const FilmArgs& a, float prevIor, V4 rands, V3 v, V3 n, BsdfS& r)
{
  const uint transparFlag = __float_as_uint(m.data[FILM_TRANSPARENT]);
  if ((r.flags & RAY_FLAG_HAS_INV_NORMAL) != 0) n = (-1.0f) * n;
  const bool reversed = dot(n, v) < 0.f && a.intIOR.im < 0.001f;
  V3 s, t;
  coordinateSystemV2(n, s, t);
  const V3 wi = v3(dot(v, s), dot(v, t), dot(v, n));
  const float cosThetaI = clampf(absf(wi.z), 0.0001f, 1.0f);
  const float ior = a.intIOR.re / a.extIOR;
  V3 R, T;
  filmReflTrans(m, a, cosThetaI, reversed, !(a.intIOR.im > 0.001f || transparFlag == 0), R, T);    // (an opaque film never reads its transmittance)
  const float sR = sum3(R), sT = sum3(T);
  if (a.intIOR.im > 0.001f || transparFlag == 0) {
    const V3 wo = v3(-wi.x, -wi.y, wi.z);
    r.val = R; r.pdf = 1.f; r.dir = normalize(wo.x * s + wo.y * t + wo.z * n); r.flags |= RAY_EVENT_S; r.ior = prevIor;
  } else if (rands.x * (sR + sT) < sR) {
    const V3 wo = v3(-wi.x, -wi.y, wi.z);
    r.val = R; r.pdf = sR / (sR + sT); r.dir = normalize(wo.x * s + wo.y * t + wo.z * n); r.flags |= RAY_EVENT_S; r.ior = prevIor;
  } else {
    const V4 fr = frDielectricDetailedV2(wi.z, ior);
    const V3 wo = v3(-fr.w * wi.x, -fr.w * wi.y, fr.y);                 // refract (cmaterial.h:917-920)
    r.val = T; r.pdf = sT / (sR + sT); r.dir = normalize(wo.x * s + wo.y * t + wo.z * n); r.flags |= (RAY_EVENT_S | RAY_EVENT_T);
    r.ior = (prevIor == a.intIOR.re) ? a.extIOR : a.intIOR.re;
  }
  r.val = r.val / smax(absf(dot(r.dir, n)), 1e-6f);
}
// filmRoughSampleAndEval (cmat_film.h:183-410)
HPT_DEV void filmRoughSampleAndEval(const MaterialRec& m, const FilmArgs& a, float prevIor, V4 rands, V3 v, V3 n, V3 alpha_tex, BsdfS& r)
{
  const uint transparFlag = __float_as_uint(m.data[FILM_TRANSPARENT]);
  if ((r.flags & RAY_FLAG_HAS_INV_NORMAL) != 0) n = (-1.0f) * n;
  const bool reversed = dot(v, n) < 0.f && a.intIOR.im < 0.001f;
  const V2 alpha = v2(smin(m.data[FILM_ROUGH_V], alpha_tex.x), smin(m.data[FILM_ROUGH_U], alpha_tex.y));
  V3 s, t;
  coordinateSystemV2(n, s, t);
  V3 wi = v3(dot(v, s), dot(v, t), dot(v, n));
  float ior = a.intIOR.re / a.extIOR;
  if (reversed) { wi = (-1.0f) * wi; ior = 1.f / ior; }
  const V3 wm = trSample(wi, v2(rands.x, rands.y), alpha);
  const float cosThetaI = clampf(absf(dot(wi, wm)), 0.00001f, 1.0f);
  V3 R, T;
  const bool opaque = a.intIOR.im > 0.001f || transparFlag == 0;
  filmReflTrans(m, a, cosThetaI, reversed, !opaque, R, T);          // (an opaque film never reads its transmittance)
  const float sR = sum3(R), sT = sum3(T);
  if (opaque || rands.w * (sR + sT) < sR) {
    V3 wo = reflect((-1.0f) * wi, wm);
    if (wi.z < 0.f || wo.z <= 0.f) return;
    const float cos_theta_i = smax(wi.z, HPT_EPSILON_32), cos_theta_o = smax(wo.z, HPT_EPSILON_32);
    r.pdf = trPDF(wi, wm, alpha) / (4.0f * absf(dot(wi, wm)));
    if (!opaque) r.pdf = r.pdf * sR / (sR + sT);
    r.val = ((trD(wm, alpha) * microfacet_G(wi, wo, wm, alpha)) * R) / (4.0f * cos_theta_i * cos_theta_o);
    if (reversed) wo = (-1.0f) * wo;
    r.dir = normalize(wo.x * s + wo.y * t + wo.z * n);
    r.flags = RAY_FLAG_HAS_NON_SPEC;
    r.ior = prevIor;
  } else {
    const V4 fr = frDielectricDetailedV2(dot(wi, wm), ior);
    const float cosThetaT = fr.y, eta_it = fr.z, eta_ti = fr.w;
    V3 ws, wt;
    coordinateSystemV2(wm, ws, wt);
    const V3 local_wi = v3(dot(ws, wi), dot(wt, wi), dot(wm, wi));
    const V3 local_wo = v3(-eta_ti * local_wi.x, -eta_ti * local_wi.y, cosThetaT);
    V3 wo = local_wo.x * ws + local_wo.y * wt + local_wo.z * wm;
    if (wo.z > 0.f) return;
    const float cos_theta_i = smax(wi.z, HPT_EPSILON_32), cos_theta_o = smin(wo.z, -HPT_EPSILON_32);
    if (absf(eta_it - 1.f) <= 1e-6f) {
      r.pdf = trPDF(wi, wm, alpha) / (4.0f * absf(dot(wi, wm))) * sT / (sR + sT);
      r.val = ((trD(wm, alpha) * microfacet_G(wi, wo, wm, alpha)) * T) / (4.0f * -cos_theta_i * cos_theta_o);
    } else {
      const float sq = dot(wo, wm) + dot(wi, wm) / eta_it;
      const float denom = sq * sq;
      const float dwm_dwi = absf(dot(wo, wm)) / denom;
      r.pdf = trPDF(wi, wm, alpha) * dwm_dwi * sT / (sR + sT);
      r.val = ((trD(wm, alpha) * microfacet_G(wi, wo, wm, alpha)) * T) * absf(dot(wi, wm) * dot(wo, wm) / (cos_theta_i * cos_theta_o * denom));
    }
    if (reversed) wo = (-1.0f) * wo;
    r.dir = normalize(wo.x * s + wo.y * t + wo.z * n);
    r.flags = RAY_FLAG_HAS_NON_SPEC;
    r.ior = (prevIor == a.intIOR.re) ? a.extIOR : a.intIOR.re;
  }
}
// filmRoughEval (cmat_film.h:413-544): zero for a film on a dielectric
HPT_DEV void filmRoughEval(const MaterialRec& m, const FilmArgs& a, V3 l, V3 v, V3 n, V3 alpha_tex, BsdfE& res)
{
  if (a.intIOR.im < 0.001f) return;
  const V2 alpha = v2(smin(m.data[FILM_ROUGH_V], alpha_tex.x), smin(m.data[FILM_ROUGH_U], alpha_tex.y));
  V3 s, t;
  coordinateSystemV2(n, s, t);
  const V3 wo = v3(dot(l, s), dot(l, t), dot(l, n)), wi = v3(dot(v, s), dot(v, t), dot(v, n));
  const V3 wm = normalize(wo + wi);
  if (wi.z * wo.z < 0.f) return;
  const float cosThetaI = clampf(absf(dot(wo, wm)), 0.00001f, 1.0f);
  V3 R, T;
  filmReflTrans(m, a, cosThetaI, false, false, R, T);
  const float cos_theta_i = smax(wi.z, HPT_EPSILON_32), cos_theta_o = smax(wo.z, HPT_EPSILON_32);
  res.pdf = trPDF(wi, wm, alpha) / (4.0f * absf(dot(wi, wm)));
  res.val = ((trD(wm, alpha) * microfacet_G(wi, wo, wm, alpha)) * R) / (4.0f * cos_theta_i * cos_theta_o);
}
HPT_DEV void filmEvalBranch(const DevScene& S, const MaterialRec& m, V2 uv, float wave0, V3 l, V3 v, V3 n, V3 alpha_tex, BsdfE& res)
{
  if (smax(m.data[FILM_ROUGH_V], m.data[FILM_ROUGH_U]) < 1e-3f) return;                              // trEffectivelySmooth
  const Cx intIOR = filmIntIOR(S, m, wave0);
  if (intIOR.im < 0.001f) return;
  const FilmArgs fa = filmArgs(S, m, uv, wave0, intIOR);
  filmRoughEval(m, fa, l, v, n, alpha_tex, res);
}

} // namespace hpt
