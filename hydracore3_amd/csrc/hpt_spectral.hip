// Spectral rendering (m_spectral_mode != 0, KSPEC_SPECTRAL_RENDERING): the same path tracer carrying four wavelengths per path.
//
// What changes against the RGB path (integrator_pt.cpp:117-118, 145-146, 618-622; integrator_spectrum.cpp; spectrum.h):
//   * kernel_InitEyeRay2 draws one more generator step per path (GetRandomNumbersSpec, after the lens numbers) and places four
//     wavelengths in [LAMBDA_MIN, LAMBDA_MAX] with SampleWavelengths (one uniform offset, three rotations by a quarter of the range);
//   * radiance and throughput are float4 - one value per wavelength - and every colour-valued quantity is looked up per wavelength:
//     SampleMatColorSpectrumTexture / SampleMatParamSpectrum / LightIntensity read m_spec_values (every spectrum resampled at 1 nm,
//     spectrum.cpp: ResampleUniform) through m_spec_offset_sz[spdid] with SampleUniformSpectrum's linear interpolation;
//   * kernel_ContributeToImage turns the four samples into RGB with the CIE 1931 observer (SpectrumToXYZ over m_cie_xyz, XYZToRGB),
//     or adds the first sample (channels == 1).
// The tables (m_spec_values, m_spec_offset_sz, m_cie_xyz) come in through the C ABI with the other scene vectors: the host owns them.
//
// Scope of this kernel: PathTraceBlock with everything the RGB kernels hold - every material type (gltf and the legacy glass carry their colours
// as four samples as they are; thin films: hpt_film.h), blends, normal maps, every light type with an intensity spectrum, sampled environment
// maps and back plates, a sky spectrum, moving instances, the lens stack. Three forms share the per-vertex function shadeVertexSpec and render
// bit-identical frames: pathTraceSpectralKernel (one thread per pixel, in-place path regeneration, no work queue: light scenes, input-ray batches,
// moving instances), pathTraceBlockSpectralKernel (block-local ray repacking: scenes whose rays walk a real tree) and wfShadeSpecKernel (the shade
// half of the wavefront schedule: heavy scenes in calls of 2^19 pixels or more). With more than four channels the output is the reference's stack
// of wavelength layers.
#include <hip/hip_runtime.h>
#include "hpt_decl.h"
#include "hpt_block_trace.h"
#ifndef HPT_SPEC_FILM
#define HPT_SPEC_FILM 1      // 0: an A/B build without the thin-film branches (what they cost the scenes that have none)
#endif

namespace hpt {

HPT_DEV V4 operator+(V4 a, V4 b) { return v4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
HPT_DEV V4 operator*(V4 a, V4 b) { return v4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
HPT_DEV V4 operator*(V4 a, float s) { return v4(a.x * s, a.y * s, a.z * s, a.w * s); }
HPT_DEV V4 operator*(float s, V4 a) { return v4(s * a.x, s * a.y, s * a.z, s * a.w); }
HPT_DEV V4 operator/(V4 a, float s) { return v4(a.x / s, a.y / s, a.z / s, a.w / s); }
HPT_DEV V4 v4s(float a) { return v4(a, a, a, a); }
HPT_DEV V4 ld4(const float* p) { return v4(p[0], p[1], p[2], p[3]); }
HPT_DEV float comp(const V4& a, int i) { return i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : a.w; }

// SampleWavelengths (spectrum.h:58-75)
HPT_DEV V4 sampleWavelengths(float u, float a, float b)
{
  float r[4];
  r[0] = lerpf(a, b, u);
  const float delta = (b - a) / 4.0f;
  for (int i = 1; i < 4; i++) { r[i] = r[i - 1] + delta; if (r[i] > b) r[i] = a + (r[i] - b); }
  return v4(r[0], r[1], r[2], r[3]);
}
// SampleUniformSpectrum (spectrum.h:106-126) over four wavelengths (the scalar form lives in hpt_film.h)
HPT_DEV V4 sampleUniformSpectrum(const float* vals, uint offset, V4 w)
{ return v4(sampleUniformSpectrum1(vals, offset, w.x), sampleUniformSpectrum1(vals, offset, w.y), sampleUniformSpectrum1(vals, offset, w.z), sampleUniformSpectrum1(vals, offset, w.w)); }

// BinarySearchU2 (spectrum.h:42-55) with its probes kept inside the table: the reference shrinks the range by `last - half + 1` (pbrt's FindInterval:
// `size - (half + 1)`), so for wavelengths in the upper part of a spectrum its probes run past the end; such a probe is answered "greater" here,
// which ends the loop, and the clamp returns the interval the wavelength lies in - the reference's result wherever its own reads stay inside.
HPT_DEV uint binarySearchU2(const uint* array2, uint a_offset, uint array_sz, float val)
{
  int last = int(array_sz) - 2, first = 1;
  while (last > 0) {
    const int half = last >> 1, middle = first + half;
    const bool predResult = middle < int(array_sz) && float(array2[2u * (a_offset + (uint)middle) + 1u]) <= val;
    first = predResult ? middle + 1 : first;
    last = predResult ? last - half + 1 : half;
  }
  return (uint)min(max(first - 1, 0), int(array_sz) - 2);
}
// SampleMatColorSpectrumTexture (integrator_spectrum.cpp:128-180): the material's colour as a spectrum - tabulated, or (KSPEC_SPD_TEX) one texture
// per wavelength band, linear between the two around each wavelength, zero outside the bands
HPT_DEV V4 matColorSpectrum(const DevScene& S, const MaterialRec& m, V4 waves, int paramId, int specSlot, V2 uv)
{
  const uint specId = m.spdid[specSlot];
  if (specId >= 0xFFFFFFFFu) return ld4(m.colors[paramId]);
  const uint texSize = S.specTexOffsetSz[2u * specId + 1u];
  if (texSize == 0u) return sampleUniformSpectrum(S.specValues, S.specOffsetSz[2u * specId], waves);
  const uint texOffset = S.specTexOffsetSz[2u * specId];
  const uint* tw = S.specTexIdsWavelengths;
  const V2 tcT = mulRows2x4(m.row0[0], m.row1[0], uv);
  float r[4];
  for (int i = 0; i < 4; i++) {
    const float w = comp(waves, i);
    if (w < float(tw[2u * texOffset + 1u]) || w > float(tw[2u * (texOffset + texSize - 1u) + 1u])) { r[i] = 0.0f; continue; }
    const uint o = binarySearchU2(tw, texOffset, texSize, w);
    const uint y0 = tw[2u * (texOffset + o) + 1u], y1 = tw[2u * (texOffset + o + 1u) + 1u];
    const V4 c1 = texSample(S.textures, tw[2u * (texOffset + o)], tcT), c2 = texSample(S.textures, tw[2u * (texOffset + o + 1u)], tcT);
    r[i] = lerpf(c1.x, c2.x, (w - float(y0)) / float(y1 - y0));
  }
  return v4(r[0], r[1], r[2], r[3]);
}
// SampleMatParamSpectrum (:25-44)
HPT_DEV V4 matParamSpectrum(const DevScene& S, const MaterialRec& m, V4 waves, int paramId, int specSlot)
{
  const uint specId = m.spdid[specSlot];
  if (specId < 0xFFFFFFFFu) return sampleUniformSpectrum(S.specValues, S.specOffsetSz[2u * specId], waves);
  return v4s(m.data[paramId]);
}
// LightIntensity with the light's spectrum (integrator_pt_lgt.cpp:109-173): the scalar factors are those of the RGB path
template <int SCOPE>
HPT_DEV V4 lightIntensitySpec(const DevScene& S, const LightRec& L, V4 waves, V3 a_rayPos, V3 a_rayDir)
{
  V4 lightColor = ld4(L.intensity);
  if (L.specId < 0xFFFFFFFFu) lightColor = sampleUniformSpectrum(S.specValues, S.specOffsetSz[2u * L.specId], waves);
  lightColor = lightColor * L.mult;
  if (L.iesId != 0xFFFFFFFFu) {
    if ((L.flags & LIGHT_FLAG_POINT_AREA) != 0) a_rayDir = normalize(ld3(L.pos) - a_rayPos);
    const V4 dt = mul4x4(L.iesMatrix, v4(a_rayDir.x, a_rayDir.y, a_rayDir.z, 0.0f));
    const V2 tc = sphereMapTo2DTexCoord((-1.0f) * v3(dt.x, dt.y, dt.z));
    lightColor = lightColor * texSample(S.textures, L.iesId, tc);
  }
  if (L.distType == LIGHT_DIST_SPOT) {
    const float cos_theta = smax(-dot(a_rayDir, ld3(L.norm)), 0.0f);
    const float tVal = (cos_theta - L.lightCos2) / (L.lightCos1 - L.lightCos2);
    const float t = smin(smax(tVal, 0.0f), 1.0f);
    lightColor = lightColor * (t * t * (3.0f - 2.0f * t));
    if ((L.flags & LIGHT_FLAG_PROJECTIVE) != 0 && L.texId != 0xFFFFFFFFu) {
      const V4 clip = mul4x4(L.iesMatrix, v4(a_rayPos.x, a_rayPos.y, a_rayPos.z, 1.0f));
      const V3 ndc = v3(clip.x, clip.y, clip.z) / clip.w;
      lightColor = lightColor * texSample(S.textures, L.texId, v2(ndc.x * 0.5f + 0.5f, ndc.y * 0.5f + 0.5f));
    }
  }
  else if (SCOPE >= 2 && L.texId != 0xFFFFFFFFu)                                   // :163-170: the environment map seen along the shadow ray (all four components of the texel)
    lightColor = lightColor * texSample(S.textures, L.texId, mulRows2x4(L.samplerRow0, L.samplerRow1, sphereMapTo2DTexCoord(a_rayDir)));
  return lightColor;
}
// kernel_HitEnvironment + EnvironmentColor (integrator_pt.cpp:550-595, integrator_pt_lgt.cpp:175-210) on four samples: the environment's
// spectrum or m_envColor, times the map's texel, the MIS weight against the map's pdf table, the camera back plate (hpt_device.h: environmentRadiance)
HPT_DEV V4 environmentRadianceSpec(const DevScene& S, V3 rdir, V4 waves, float misPdf, uint flags, uint XY)
{
  V4 color = ld4(S.envColor);
  if (S.envSpecId != 0xFFFFFFFFu) color = sampleUniformSpectrum(S.specValues, S.specOffsetSz[2u * S.envSpecId], waves) * (S.envSpecMult / 106.856895f);
  if (S.envTexId == 0xFFFFFFFFu && S.envCamBackId == 0xFFFFFFFFu) return color;
  float envPdf = 1.0f;
  if (S.envTexId != 0xFFFFFFFFu) {
    const float sinTheta = sqrtf_(1.0f - rdir.y * rdir.y);
    const V2 tcT = mulRows2x4(S.envSamRow0, S.envSamRow1, sphereMapTo2DTexCoord(rdir));
    if (sinTheta != 0.f && S.envEnableSam != 0 && S.integratorType == INTEGRATOR_MIS_PT && S.envLightId != 0xFFFFFFFFu) {
      const LightRec& L = S.lights[S.envLightId];
      const float mapPdf = evalMap2DPdf(tcT, S.arrays1f + L.pdfTableOffset, (int)L.pdfTableSizeX, (int)L.pdfTableSizeY);
      envPdf = (mapPdf * 1.0f) / (2.f * HPT_PI * HPT_PI * smax(absf(sinTheta), 1e-20f));
    }
    color = color * texSample(S.textures, S.envTexId, tcT);
  }
  const bool isSpec = misPdf < 0.0f, exitZero = (flags & RAY_FLAG_PRIME_RAY_MISS) != 0;
  if (S.integratorType == INTEGRATOR_MIS_PT && S.envEnableSam != 0 && !isSpec && !exitZero)
    color = color * misWeightHeuristic(misPdf, (1.0f / float(S.numLights)) * envPdf);
  else if (S.integratorType == INTEGRATOR_SHADOW_PT && S.envEnableSam != 0) color = v4s(0.0f);
  if (exitZero && S.envCamBackId != 0xFFFFFFFFu) {
    const uint x = XY & 0x0000FFFFu, y = (XY & 0xFFFF0000u) >> 16;
    color = texSample(S.textures, S.envCamBackId, v2((float(x) + 0.5f) / float(S.winWidth), (float(y) + 0.5f) / float(S.winHeight)));
  }
  return color;
}

// SpectrumToXYZ + XYZToRGB (spectrum.h:151-214); terminateWaves: the path met a dispersive surface (RAY_FLAG_WAVES_DIVERGED)
HPT_DEV V3 spectrumToRGB(const DevScene& S, V4 spec, V4 lambda, bool terminateWaves)
{
  const float pdf0 = 1.0f / (LAMBDA_MAX - LAMBDA_MIN);
  const float CIE_Y_integral = 106.856895f;
  float X = 0.0f, Y = 0.0f, Z = 0.0f;
  float xs[4], ys[4], zs[4];
  for (int i = 0; i < 4; i++) {
    // a path that met a dispersive surface keeps its first wavelength only (terminate_waves: pdf[0] / 4, the others 0 and their samples dropped)
    const float pdf = terminateWaves ? (i == 0 ? pdf0 / 4.0f : 0.0f) : pdf0;
    const float s = (pdf != 0.0f) ? comp(spec, i) / pdf : 0.0f;
    const uint offset = (uint)(floorf(comp(lambda, i) + 0.5f) - LAMBDA_MIN);
    float cx = 0.0f, cy = 0.0f, cz = 0.0f;
    if (offset < 471u && offset < S.numCieXYZ) { const float4 c = S.cieXYZ[offset]; cx = c.x; cy = c.y; cz = c.z; }
    xs[i] = cx * s; ys[i] = cy * s; zs[i] = cz * s;
  }
  X = (((xs[0] + xs[1]) + xs[2]) + xs[3]) / 4.0f;                           // SpectrumAverage: left-to-right sum / SPECTRUM_SAMPLE_SZ
  Y = (((ys[0] + ys[1]) + ys[2]) + ys[3]) / 4.0f;
  Z = (((zs[0] + zs[1]) + zs[2]) + zs[3]) / 4.0f;
  const float x = X / CIE_Y_integral, y = Y / CIE_Y_integral, z = Z / CIE_Y_integral;
  return v3(+3.240479f * x - 1.537150f * y - 0.498535f * z, -0.969256f * x + 1.875991f * y + 0.041556f * z, +0.055648f * x - 0.204043f * y + 1.057311f * z);
}
// SpectralCamRespoceToRGB (integrator_spectrum.cpp:68-124)
HPT_DEV V3 spectralCamResponseToRGB(const DevScene& S, V4 spec, V4 waves, uint rayFlags)
{
  if (S.camResponseSpectrumId[0] < 0) return spectrumToRGB(S, spec, waves, (rayFlags & RAY_FLAG_WAVES_DIVERGED) != 0u);
  V4 rX = v4s(1.0f), rY, rZ;
  rX = sampleUniformSpectrum(S.specValues, S.specOffsetSz[2 * S.camResponseSpectrumId[0]], waves);
  rY = S.camResponseSpectrumId[1] >= 0 ? sampleUniformSpectrum(S.specValues, S.specOffsetSz[2 * S.camResponseSpectrumId[1]], waves) : rX;
  rZ = S.camResponseSpectrumId[2] >= 0 ? sampleUniformSpectrum(S.specValues, S.specOffsetSz[2 * S.camResponseSpectrumId[2]], waves) : rY;
  V3 xyz = v3(0, 0, 0);
  for (int i = 0; i < 4; i++) { xyz.x += comp(spec, i) * comp(rX, i); xyz.y += comp(spec, i) * comp(rY, i); xyz.z += comp(spec, i) * comp(rZ, i); }
  if (S.camResponseType == 0u)                                              // CAM_RESPONCE_XYZ = 0, CAM_RESPONCE_RGB = 1 (integrator_pt.h:531-532)
    return v3(+3.240479f * xyz.x - 1.537150f * xyz.y - 0.498535f * xyz.z, -0.969256f * xyz.x + 1.875991f * xyz.y + 0.041556f * xyz.z, +0.055648f * xyz.x - 0.204043f * xyz.y + 1.057311f * xyz.z);
  return xyz;
}

struct SpecEval { V4 val; float pdf; };
struct SpecSample { V4 val; V3 dir; float pdf; uint flags; float ior; };

// MaterialEval, spectral (integrator_pt_mat.cpp:405-420, 471-482 with cmat_conductor.h:103-137, cmat_diffuse.h:27-39)
// the "four scalar parameters" of a gltf material (integrator_pt_mat.cpp:151-167; hpt_shade.h: leafTextures)
HPT_DEV V3 fourParamsSpec(const DevScene& S, const MaterialRec& m, V2 uv) { V3 t3, four; leafTextures(S, m, uv, t3, four); return four; }
// n: the shading normal (the leaf's normal map applied), gn: the geometric one (films); SCOPE: see pathTraceSpectralKernel
template <int SCOPE>
HPT_DEV SpecEval materialEvalSpec(const DevScene& S, const MaterialRec& m, V4 waves, V3 l, V3 v, V3 n, V3 gn, V4 texColor, V2 uv)
{
  const V3 texColor3 = v3(texColor.x, texColor.y, texColor.z);
  SpecEval r; r.val = v4s(0.0f); r.pdf = 0.0f;
  if (SCOPE >= 1 && m.mtype == MAT_TYPE_GLTF) {
    // gltfEval on float4 (integrator_pt_mat.cpp:385-394): the base colour times the texel is taken as four spectral samples as it is (the
    // reference's loader warns about such colours); nothing but the colours depends on the channel, so the four wavelengths are two passes
    // through the RGB routine - (x, y, z), then w in every slot
    const V4 base = ld4(m.colors[GLTF_COLOR_BASE]) * texColor, mc = ld4(m.colors[GLTF_COLOR_METAL]), cc = ld4(m.colors[GLTF_COLOR_COAT]);
    const V3 four = fourParamsSpec(S, m, uv);
    BsdfE a, b; a.val = v3(0, 0, 0); a.pdf = 0.0f; a.dval = v3(0, 0, 0); b = a;
    gltfEvalC(m, v3(mc.x, mc.y, mc.z), v3(cc.x, cc.y, cc.z), l, v, n, v3(base.x, base.y, base.z), four, a);
    gltfEvalC(m, v3s(mc.w), v3s(cc.w), l, v, n, v3s(base.w), four, b);
    r.val = v4(a.val.x, a.val.y, a.val.z, b.val.x); r.pdf = a.pdf;
    return r;
  }
  if (m.mtype == MAT_TYPE_DIFFUSE) {
    float lambertVal = HPT_INV_PI;
    if ((m.cflags & GLTF_COMPONENT_ORENNAYAR) != 0) lambertVal *= orennayarFunc(l, v, n, m.data[0]);
    r.val = lambertVal * matColorSpectrum(S, m, waves, 0, 0, uv);               // (not multiplied by the texture in spectral mode, :259-260)
    r.pdf = absf(dot(l, n)) * HPT_INV_PI;
  } else if (m.mtype == MAT_TYPE_PLASTIC) {
    // plasticEval on float4 (cmat_plastic.h:102-191): the reflectance enters channel by channel and nothing else depends on it, so the four
    // wavelengths are two passes through the RGB routine - (x, y, z), then w in every slot - with the same arithmetic per channel
    const V4 refl = matColorSpectrum(S, m, waves, 0, 0, uv);                     // PLASTIC_COLOR; not multiplied by the texture in spectral mode (integrator_pt_mat.cpp:490-493)
    BsdfE a, b; a.val = v3(0, 0, 0); a.pdf = 0.0f; a.dval = v3(0, 0, 0); b = a;
    plasticEval(m, v3(refl.x, refl.y, refl.z), l, v, n, a, S.arrays1f, m.datai[0]);
    plasticEval(m, v3(refl.w, refl.w, refl.w), l, v, n, b, S.arrays1f, m.datai[0]);
    r.val = v4(a.val.x, a.val.y, a.val.z, b.val.x); r.pdf = a.pdf;
  } else if (m.mtype == MAT_TYPE_CONDUCTOR) {
    if (!(smax(m.data[1], m.data[0]) < 1e-3f)) {                            // trEffectivelySmooth: the smooth conductor evaluates to zero
      const V4 etaSpec = matParamSpectrum(S, m, waves, 2, 0), kSpec = matParamSpectrum(S, m, waves, 3, 1);
      const V2 alpha = v2(smin(m.data[0], texColor3.x), smin(m.data[1], texColor3.y));
      V3 nx, ny;
      coordinateSystemV2(n, nx, ny);
      const V3 wo = v3(dot(v, nx), dot(v, ny), dot(v, n)), wi = v3(dot(l, nx), dot(l, ny), dot(l, n));
      if (wo.z * wi.z < 0.0f) return r;
      V3 wm = wo + wi;
      if (dot(wm, wm) == 0) return r;
      wm = normalize(wm);
      r.val = v4(conductorRoughEvalInternal(wo, wi, wm, alpha, cx(etaSpec.x, kSpec.x)), conductorRoughEvalInternal(wo, wi, wm, alpha, cx(etaSpec.y, kSpec.y)),
                 conductorRoughEvalInternal(wo, wi, wm, alpha, cx(etaSpec.z, kSpec.z)), conductorRoughEvalInternal(wo, wi, wm, alpha, cx(etaSpec.w, kSpec.w))) * ld4(m.colors[0]);
      if (dot(wm, v3(0.0f, 0.0f, 1.0f)) < 0.f) wm = (-1.0f) * wm;
      r.pdf = trPDF(wo, wm, alpha) / (4.0f * absf(dot(wo, wm)));
    }
  } else if (HPT_SPEC_FILM && m.mtype == MAT_TYPE_THIN_FILM) {                               // integrator_pt_mat.cpp:422-470: rough films only, the first wavelength only
    BsdfE e; e.val = v3(0, 0, 0); e.pdf = 0.0f; e.dval = v3(0, 0, 0);
    filmEvalBranch(S, m, uv, waves.x, l, v, gn, texColor3, e);
    r.val = v4(e.val.x, 0.0f, 0.0f, 0.0f); r.pdf = e.pdf;
  }
  return r;
}
// MaterialEval over a blend tree (integrator_pt_mat.cpp:316-333, 511-527; hpt_shade.h: blendTreeEval), spectral: a plain material is a tree of one
// leaf. Each leaf fetches its own texel, bends the normal by its own map and scales by cos(shade) / cos(geom) (:341-355).
HPT_DEV SpecEval materialEvalTreeSpec(const DevScene& S, uint rootId, V4 waves, V3 l, V3 v, V3 gn, V3 tan, V2 uv)
{
  SpecEval res; res.val = v4s(0.0f); res.pdf = 0.0f;
  uint stackId[BLEND_STACK_SIZE]; float stackW[BLEND_STACK_SIZE];
  uint curId = rootId; float curW = 1.0f;
  stackId[0] = curId; stackW[0] = curW;
  int top = 0; bool needPop = false;
  do {
    if (needPop) { top--; const int t = top > 0 ? top : 0; curId = stackId[t]; curW = stackW[t]; } else needPop = true;
    const MaterialRec& m = S.materials[curId];
    const V4 texColor = texSample(S.textures, m.texid[0], mulRows2x4(m.row0[0], m.row1[0], uv));
    if (m.mtype == MAT_TYPE_BLEND) {                                        // BlendEval: first child next (no pop), second child waits on the stack
      const float w = m.data[0] * texColor.x;
      const uint id1 = m.datai[0], id2 = m.datai[1];
      const float w1 = curW * (1.0f - w), w2 = curW * w;
      curId = id1; curW = w1; needPop = false;
      if (top + 1 <= (int)BLEND_STACK_SIZE) { stackId[top] = id2; stackW[top] = w2; top++; }
      continue;
    }
    V3 n = gn; float bm = 1.0f;
    if (m.texid[1] != 0xFFFFFFFFu) { n = bumpNormal(S, m, gn, tan, uv); bm = bumpCosMult(l, gn, n); }
    const SpecEval cv = materialEvalSpec<2>(S, m, waves, l, v, n, gn, texColor, uv);
    res.val = res.val + cv.val * (curW * bm); res.pdf += cv.pdf * curW;
  } while (top > 0);
  return res;
}
// MaterialSampleAndEval, spectral (integrator_pt_mat.cpp:184-196, 252-263 with cmat_conductor.h:7-100, cmat_diffuse.h:8-24)
// n: the shading normal, gn: the geometric one (legacy glass and films sample about it); pdf0: what the blend descent left in the sample's pdf
template <int SCOPE>
HPT_DEV SpecSample materialSampleSpec(const DevScene& S, const MaterialRec& m, V4 waves, V4 rands, V3 v, V3 n, V3 gn, V4 texColor, uint flags0, float prevIor, V2 uv, float pdf0)
{
  const V3 texColor3 = v3(texColor.x, texColor.y, texColor.z);
  SpecSample r; r.val = v4s(0.0f); r.pdf = pdf0; r.dir = v3(0, 1, 0); r.flags = flags0; r.ior = 1.0f;
  if (SCOPE >= 2 && m.mtype == MAT_TYPE_GLASS) {                                  // glassSampleAndEval on float4 (cmat_glass.h:236-277): the geometric normal (integrator_pt_mat.cpp:178-183)
    const V4 colorReflect = ld4(m.colors[0]), colorTransp = ld4(m.colors[1]);
    const float ior = m.data[2];
    const V3 rayDir = (-1.0f) * v;
    float relativeIor = ior / prevIor;
    if ((r.flags & RAY_FLAG_HAS_INV_NORMAL) != 0) { if (prevIor == ior) relativeIor = 1.0f / ior; }
    const float fresnel = fresnel2(v, gn, relativeIor);
    V3 dir; r.ior = prevIor;
    if (rands.w < fresnel) { dir = reflect2(rayDir, gn); r.val = colorReflect; r.flags |= RAY_EVENT_S; }
    else { dir = refract2(rayDir, gn, relativeIor); r.val = colorTransp; r.ior = ior; r.flags |= (RAY_EVENT_S | RAY_EVENT_T); }
    r.val = r.val / smax(absf(dot(dir, gn)), 1e-6f);
    r.dir = dir; r.pdf = 1.0f;
    return r;
  }
  if (SCOPE >= 1 && m.mtype == MAT_TYPE_GLTF) {                                   // gltfSampleAndEval on float4 (integrator_pt_mat.cpp:170-176), as in materialEvalSpec
    const V4 base = ld4(m.colors[GLTF_COLOR_BASE]) * texColor, mc = ld4(m.colors[GLTF_COLOR_METAL]), cc = ld4(m.colors[GLTF_COLOR_COAT]);
    const V3 four = fourParamsSpec(S, m, uv);
    BsdfS a; a.val = v3(0, 0, 0); a.dval = v3(0, 0, 0); a.pdf = pdf0; a.dir = v3(0, 1, 0); a.flags = flags0; a.ior = 1.0f;
    BsdfS b = a;
    gltfSampleAndEvalC(m, v3(mc.x, mc.y, mc.z), v3(cc.x, cc.y, cc.z), rands, v, n, v3(base.x, base.y, base.z), four, a);
    gltfSampleAndEvalC(m, v3s(mc.w), v3s(cc.w), rands, v, n, v3s(base.w), four, b);
    r.val = v4(a.val.x, a.val.y, a.val.z, b.val.x); r.dir = a.dir; r.pdf = a.pdf; r.flags = a.flags;
    return r;
  }
  if (m.mtype == MAT_TYPE_DIFFUSE) {
    const V3 lambertDir = mapSampleToCosineDistribution(rands.x, rands.y, n, n, 1.0f);
    r.dir = lambertDir;
    r.val = HPT_INV_PI * matColorSpectrum(S, m, waves, 0, 0, uv);
    r.pdf = absf(dot(lambertDir, n)) * HPT_INV_PI;
    r.flags = RAY_FLAG_HAS_NON_SPEC;
    if ((m.cflags & GLTF_COMPONENT_ORENNAYAR) != 0) r.val = r.val * orennayarFunc(lambertDir, (-1.0f) * v, n, m.data[0]);
  } else if (m.mtype == MAT_TYPE_DIELECTRIC) {
    // dielectricSmoothSampleAndEval (cmat_dielectric.h:8-56): the IOR of the FIRST wavelength decides the direction; a surface whose IOR is a
    // spectrum marks the path (RAY_FLAG_WAVES_DIVERGED): only that wavelength reaches the image (integrator_pt_mat.cpp:277-287)
    const V4 etaSpec = matParamSpectrum(S, m, waves, 1, 0);                  // DIELECTRIC_ETA_INT
    BsdfS a; a.val = v3(0, 0, 0); a.dval = v3(0, 0, 0); a.pdf = pdf0; a.dir = v3(0, 1, 0); a.flags = flags0; a.ior = 1.0f;
    dielectricSmoothSampleAndEval(m, etaSpec.x, prevIor, rands, v, n, a);
    r.val = v4s(a.val.x); r.dir = a.dir; r.pdf = a.pdf; r.flags = a.flags | ((m.spdid[0] < 0xFFFFFFFFu) ? RAY_FLAG_WAVES_DIVERGED : 0u); r.ior = a.ior;
  } else if (m.mtype == MAT_TYPE_PLASTIC) {                                  // plasticSampleAndEval on float4 (cmat_plastic.h:7-99), as in materialEvalSpec
    const V4 refl = matColorSpectrum(S, m, waves, 0, 0, uv);
    BsdfS a; a.val = v3(0, 0, 0); a.dval = v3(0, 0, 0); a.pdf = pdf0; a.dir = v3(0, 1, 0); a.flags = flags0; a.ior = 1.0f;
    BsdfS b = a;
    plasticSampleAndEval(m, v3(refl.x, refl.y, refl.z), rands, v, n, a, S.arrays1f, m.datai[0]);
    plasticSampleAndEval(m, v3(refl.w, refl.w, refl.w), rands, v, n, b, S.arrays1f, m.datai[0]);
    r.val = v4(a.val.x, a.val.y, a.val.z, b.val.x); r.dir = a.dir; r.pdf = a.pdf; r.flags = a.flags;
  } else if (m.mtype == MAT_TYPE_CONDUCTOR) {
    const V4 etaSpec = matParamSpectrum(S, m, waves, 2, 0), kSpec = matParamSpectrum(S, m, waves, 3, 1);
    if (smax(m.data[1], m.data[0]) < 1e-3f) {
      const V3 pefReflDir = reflect((-1.0f) * v, n);
      const float cosThetaOut = dot(pefReflDir, n);
      float val[4];
      for (int i = 0; i < 4; i++) {
        val[i] = frComplexConductor(cosThetaOut, cx(comp(etaSpec, i), comp(kSpec, i)));
        val[i] = (cosThetaOut <= 1e-6f) ? 0.0f : (val[i] / smax(cosThetaOut, 1e-6f));
      }
      r.val = v4(val[0], val[1], val[2], val[3]) * ld4(m.colors[0]);
      r.dir = pefReflDir; r.pdf = 1.0f; r.flags = RAY_EVENT_S;
    } else {
      if (v.z == 0) return r;
      const V2 alpha = v2(smin(m.data[0], texColor3.x), smin(m.data[1], texColor3.y));
      V3 nx, ny;
      coordinateSystemV2(n, nx, ny);
      const V3 wo = v3(dot(v, nx), dot(v, ny), dot(v, n));
      if (wo.z == 0) return r;
      const V3 wm = trSample(wo, v2(rands.x, rands.y), alpha);
      const V3 wi = reflect((-1.0f) * wo, wm);
      if (wo.z * wi.z < 0) return r;
      r.val = v4(conductorRoughEvalInternal(wo, wi, wm, alpha, cx(etaSpec.x, kSpec.x)), conductorRoughEvalInternal(wo, wi, wm, alpha, cx(etaSpec.y, kSpec.y)),
                 conductorRoughEvalInternal(wo, wi, wm, alpha, cx(etaSpec.z, kSpec.z)), conductorRoughEvalInternal(wo, wi, wm, alpha, cx(etaSpec.w, kSpec.w))) * ld4(m.colors[0]);
      r.dir = normalize(wi.x * nx + wi.y * ny + wi.z * n);
      r.pdf = trPDF(wo, wm, alpha) / (4.0f * absf(dot(wo, wm)));
      r.flags = RAY_FLAG_HAS_NON_SPEC;
    }
  } else if (HPT_SPEC_FILM && m.mtype == MAT_TYPE_THIN_FILM) {                               // integrator_pt_mat.cpp:197-249
    const FilmArgs fa = filmArgs(S, m, uv, waves.x);
    BsdfS a; a.val = v3(0, 0, 0); a.dval = v3(0, 0, 0); a.pdf = pdf0; a.dir = v3(0, 1, 0); a.flags = flags0; a.ior = 1.0f;
    if (smax(m.data[1], m.data[0]) < 1e-3f) filmSmoothSampleAndEval(m, fa, prevIor, rands, v, gn, a);
    else                                    filmRoughSampleAndEval(m, fa, prevIor, rands, v, gn, texColor3, a);
    r.val = v4(a.val.x, 0.0f, 0.0f, 0.0f); r.dir = a.dir; r.pdf = a.pdf; r.flags = a.flags | RAY_FLAG_WAVES_DIVERGED; r.ior = a.ior;
  }
  return r;
}

// One path vertex of the spectral integrator: kernel_GetRayColor's surface fetch + kernel_SampleLightSource + kernel_NextBounce on four wavelengths
// (integrator_pt.cpp:238-548 with m_spectral_mode != 0), shared by the one-thread-per-pixel kernel below and by the block-local schedule
// (pathTraceBlockSpectralKernel). As shadeVertex for the RGB kernels: the light sample's BSDF is evaluated before the shadow ray is traced,
// the caller traces it (wantShadow) and adds `contrib` when it comes back unoccluded. A miss only sets the OUT_OF_SCENE flags.
template <int SCOPE, bool MOTION>
HPT_DEV void shadeVertexSpec(const DevScene& S, const HitRec& hit, V3& rpos, V3& rdir, const V4 waves, V4& accum, V4& thr, float& misPdf, float& misIor, uint& flags,
                             const uint bounce, Rng& gen, const bool naive, bool& wantShadow, V3& shPos, V3& shDir, float& shFar, V4& contrib, const float pathTime)
{
  if (hit.inst == 0xFFFFFFFFu) {
    flags |= (bounce == 0) ? (RAY_FLAG_PRIME_RAY_MISS | RAY_FLAG_IS_DEAD | RAY_FLAG_OUT_OF_SCENE) : (RAY_FLAG_IS_DEAD | RAY_FLAG_OUT_OF_SCENE);
  } else {
    // -- surface attributes (integrator_pt.cpp:238-311), as in shadeVertex --
    const uint instId = hit.inst;
    const uint geomId = S.insts[instId].geomId;
    const uint triOffset = S.matVertOffset[2 * geomId + 0], vertOffset = S.matVertOffset[2 * geomId + 1];
    const V3 hitPos = rpos + hit.t * (1.f - 1e-6f) * rdir;
    const float uvx = hit.v, uvy = hit.u;
    const uint A = S.triIndices[(triOffset + hit.prim) * 3 + 0], B = S.triIndices[(triOffset + hit.prim) * 3 + 1], C = S.triIndices[(triOffset + hit.prim) * 3 + 2];
    const float4 nA = ((const float4*)S.vData8f)[2 * (A + vertOffset)], nB = ((const float4*)S.vData8f)[2 * (B + vertOffset)], nC = ((const float4*)S.vData8f)[2 * (C + vertOffset)];
    const float tyA = S.vData8f[8 * (A + vertOffset) + 7], tyB = S.vData8f[8 * (B + vertOffset) + 7], tyC = S.vData8f[8 * (C + vertOffset) + 7];
    const float wA = 1.0f - uvx - uvy;
    const V3 nrmO = v3(wA * nA.x + uvy * nB.x + uvx * nC.x, wA * nA.y + uvy * nB.y + uvx * nC.y, wA * nA.z + uvy * nB.z + uvx * nC.z);
    const V2 uv = v2(wA * nA.w + uvy * nB.w + uvx * nC.w, wA * tyA + uvy * tyB + uvx * tyC);
    const float* nm = S.normMat + 12 * instId;
    V3 hitNorm = v3(nm[0] * nrmO.x + nm[1] * nrmO.y + nm[2] * nrmO.z, nm[4] * nrmO.x + nm[5] * nrmO.y + nm[6] * nrmO.z, nm[8] * nrmO.x + nm[9] * nrmO.y + nm[10] * nrmO.z);
    if (MOTION && (S.motion & 2u) == 0u) {                             // integrator_pt.cpp:285-292, as in shadeVertex
      const float* nm2 = S.normMat2 + 12 * instId;
      const V3 n2 = v3(nm2[0] * hitNorm.x + nm2[1] * hitNorm.y + nm2[2] * hitNorm.z, nm2[4] * hitNorm.x + nm2[5] * hitNorm.y + nm2[6] * hitNorm.z, nm2[8] * hitNorm.x + nm2[9] * hitNorm.y + nm2[10] * hitNorm.z);
      hitNorm = hitNorm + pathTime * (n2 - hitNorm);
    }
    hitNorm = normalize(hitNorm);
    const float flipNorm = dot(rdir, hitNorm) > 0.001f ? -1.0f : 1.0f;
    hitNorm = flipNorm * hitNorm;
    if (flipNorm < 0.0f) flags |= RAY_FLAG_HAS_INV_NORMAL; else flags &= ~RAY_FLAG_HAS_INV_NORMAL;
    const uint matId = remapMaterialId(S, S.matIdByPrimId[triOffset + hit.prim], instId) & 0x00FFFFFFu;
    const MaterialRec& m = S.materials[matId];
    const uint mtype = m.mtype;
    const V3 vdir = (-1.0f) * rdir;
    V4 texColor = v4(1, 1, 1, 1);
    if (mtype != MAT_TYPE_LIGHT_SOURCE) texColor = texSample(S.textures, m.texid[0], mulRows2x4(m.row0[0], m.row1[0], uv));
    V3 hitTang = v3(0, 0, 0);                                          // only materials with a normal map (or a blend that may hold one) read it
    if (SCOPE >= 2 && (mtype == MAT_TYPE_BLEND || (mtype != MAT_TYPE_LIGHT_SOURCE && m.texid[1] != 0xFFFFFFFFu))) hitTang = hitTangent(S, A, B, C, vertOffset, wA, uvx, uvy, nm, flipNorm, (MOTION && (S.motion & 2u) == 0u) ? S.normMat2 + 12 * instId : nullptr, pathTime);

    // -- kernel_SampleLightSource (integrator_pt.cpp:350-424) --
    V4 shade = v4s(0.0f);
    if (!naive) {
      const float rndId = rng_float1(gen);
      const V4 r4 = rng_float4(gen);
      const int nLights = (int)S.numLights;
      const int lightId = min((int)floorf(rndId * float(nLights)), nLights - 1);
      if (lightId >= 0 && mtype != MAT_TYPE_LIGHT_SOURCE) {
        const LightRec& L = S.lights[lightId];
        const LightSam ls = (SCOPE >= 2 && L.geomType == LIGHT_GEOM_ENV) ? envLightSampleRev(S, L, v3(r4.x, r4.y, r4.z), hitPos) : lightSampleRev(L, v3(r4.x, r4.y, r4.z), hitPos);
        const V3 dlt = hitPos - ls.pos;
        const float hitDist = sqrtf_(dot(dlt, dlt));
        const V3 shadowRayDir = normalize(ls.pos - hitPos);
        const V3 shadowRayPos = hitPos + hitNorm * smax(maxcomp(hitPos), 1.0f) * 5e-6f;
        const bool inIllumArea = (dot(shadowRayDir, ls.norm) < 0.0f) || ls.isOmni || ls.hasIES;
        if (inIllumArea) {
          const SpecEval bv = SCOPE >= 2 ? materialEvalTreeSpec(S, matId, waves, shadowRayDir, vdir, hitNorm, hitTang, uv) : materialEvalSpec<SCOPE>(S, m, waves, shadowRayDir, vdir, hitNorm, hitNorm, texColor, uv);
          const float cosThetaOut = smax(dot(shadowRayDir, hitNorm), 0.0f);
          float lgtPdfW = (1.0f / float(nLights)) * lightEvalPDF(L, shadowRayPos, shadowRayDir, ls.pos, ls.norm, ls.pdf);
          float misWeight = (S.integratorType == INTEGRATOR_MIS_PT) ? misWeightHeuristic(lgtPdfW, bv.pdf) : 1.0f;
          if (L.geomType == LIGHT_GEOM_DIRECT) { misWeight = 1.0f; lgtPdfW = 1.0f; }
          else if (L.geomType == LIGHT_GEOM_POINT) misWeight = 1.0f;
          const bool isDirectLight = (flags & RAY_FLAG_HAS_NON_SPEC) == 0;
          if ((S.renderLayer == FB_DIRECT && !isDirectLight) || (S.renderLayer == FB_INDIRECT && isDirectLight)) misWeight = 0.0f;
          const V4 lightColor = lightIntensitySpec<SCOPE>(S, L, waves, shadowRayPos, shadowRayDir);
          shade = ((lightColor * bv.val) / lgtPdfW) * cosThetaOut * misWeight;
          wantShadow = true; shPos = shadowRayPos; shDir = shadowRayDir; shFar = hitDist * 0.9995f;
        }
      }
    }
    // -- kernel_NextBounce (integrator_pt.cpp:426-548) --
    if (mtype == MAT_TYPE_LIGHT_SOURCE) {
      const V4 tc = texSample(S.textures, m.texid[0], mulRows2x4(m.row0[0], m.row1[0], uv));
      const uint lightId = (uint)S.remapInst[2 * instId + 1];
      V4 lightInt = ld4(m.colors[0]) * tc;
      float misWeight = 1.0f;
      if (lightId != 0xFFFFFFFFu) {
        const LightRec& L = S.lights[lightId];
        const float lightCos = dot(rdir, ld3(L.norm));
        const float atten = (lightCos < 0.0f || L.geomType == LIGHT_GEOM_SPHERE) ? 1.0f : 0.0f;
        lightInt = lightIntensitySpec<SCOPE>(S, L, waves, rpos, rdir) * atten;
      }
      if (S.integratorType == INTEGRATOR_MIS_PT) {
        if (bounce > 0 && lightId != 0xFFFFFFFFu) {
          const float lgtPdf = (1.0f / float(S.numLights)) * lightEvalPDF(S.lights[lightId], rpos, rdir, hitPos, hitNorm, 1.0f);
          misWeight = misWeightHeuristic(misPdf, lgtPdf);
          if (misPdf <= 0.0f) misWeight = 1.0f;
        }
      } else if (S.integratorType == INTEGRATOR_SHADOW_PT && (flags & RAY_FLAG_HAS_NON_SPEC) != 0) misWeight = 0.0f;
      const bool isDirectLight = (flags & RAY_FLAG_HAS_NON_SPEC) == 0, isFirstNonSpec = (flags & RAY_FLAG_FIRST_NON_SPEC) != 0;
      if (S.renderLayer == FB_INDIRECT && (isDirectLight || isFirstNonSpec)) misWeight = 0.0f;
      accum = accum + thr * lightInt * misWeight;
      flags |= (RAY_FLAG_IS_DEAD | RAY_FLAG_HIT_LIGHT);
      wantShadow = false;
    } else {
      // blend descent (BlendSampleAndEval, integrator_pt_mat.cpp:23-54, 123-130): one generator step per layer BEFORE the float4
      const MaterialRec* lm = &m; uint lt = mtype; V4 ltexColor = texColor; float pdf0 = 1.0f;
      if (SCOPE >= 2 && mtype == MAT_TYPE_BLEND) {
        while (lt == MAT_TYPE_BLEND) {
          const V4 wd = texSample(S.textures, lm->texid[0], mulRows2x4(lm->row0[0], lm->row1[0], uv));
          const float weight = lm->data[0] * wd.x;
          const float select = rng_float1(gen);                        // GetRandomNumbersMatB (integrator_pt.cpp:37)
          if (select < weight) { pdf0 *= weight; lm = &S.materials[lm->datai[1]]; }
          else                 { pdf0 *= 1.0f - weight; lm = &S.materials[lm->datai[0]]; }
          lt = lm->mtype;
        }
        ltexColor = texSample(S.textures, lm->texid[0], mulRows2x4(lm->row0[0], lm->row1[0], uv));
      }
      const MaterialRec& ml = *lm;
      const bool leafBump = SCOPE >= 2 && ml.texid[1] != 0xFFFFFFFFu;                // the leaf's normal map bends the shading normal (:131-139)
      const V3 sNorm = leafBump ? bumpNormal(S, ml, hitNorm, hitTang, uv) : hitNorm;
      const V4 rands = rng_float4(gen);                                // GetRandomNumbersMats
      SpecSample ms = materialSampleSpec<SCOPE>(S, ml, waves, rands, vdir, sNorm, hitNorm, ltexColor, (flags & 0xFF000000u) | matId, misIor, uv, pdf0);
      if (lt == MAT_TYPE_DIELECTRIC || lt == MAT_TYPE_THIN_FILM || lt == MAT_TYPE_GLASS) misIor = ms.ior;
      if (leafBump) {                                                  // the caller multiplies by the cosine to the geometric normal (:298-303)
        const float c1 = absf(dot(ms.dir, hitNorm)), c2 = absf(dot(ms.dir, sNorm));
        ms.val = ms.val * (c2 / smax(c1, 1e-10f));
      }
      const float invPdf = 1.0f / smax(ms.pdf, 1e-20f);
      const V4 bxdfVal = ms.val * invPdf;
      const float cosTheta = absf(dot(ms.dir, hitNorm));
      misPdf = (ms.flags & RAY_EVENT_S) != 0 ? -1.0f : ms.pdf;
      if (S.integratorType == INTEGRATOR_STUPID_PT) { thr = thr * (cosTheta * bxdfVal); wantShadow = false; }
      else { contrib = thr * shade; thr = thr * cosTheta * bxdfVal; }
      V3 hp = hitPos;
      if ((ms.flags & RAY_EVENT_T) != 0) hp = hp + hit.t * rdir * 2.0f * 1e-6f;
      rpos = offsRayPos(hp, hitNorm, ms.dir);
      rdir = ms.dir;
      uint nextFlags = ((flags & ~RAY_FLAG_FIRST_NON_SPEC) | ms.flags);
      if (S.renderLayer == FB_DIRECT && (flags & RAY_FLAG_HAS_NON_SPEC) != 0) nextFlags |= RAY_FLAG_IS_DEAD;
      else if ((flags & RAY_FLAG_HAS_NON_SPEC) == 0 && (nextFlags & RAY_FLAG_HAS_NON_SPEC) != 0) nextFlags |= RAY_FLAG_FIRST_NON_SPEC;
      flags = nextFlags;
    }
  }
}

// One thread per pixel of the call, its passes one after the other (the pixel's generator continues from pass to pass as in the RGB kernels).
// SCOPE: what the scene needs (the host looks at the materials a hit can reach, the lights, the camera): 0 = what the reference's spectral fixtures
// use (diffuse, conductor, plastic, dielectric, thin films, analytic lights, a sky spectrum); 1 = + gltf surfaces (legacy hydra_material scenes);
// 2 = + legacy glass, blends, normal maps, environment maps, the lens stack. The code of a wider scope costs registers whether it runs or not:
// test_spectral 446 Mpaths/s in scope 0 against 322 with everything compiled in; the Cornell box under spectral mode 2153 in scope 1 against
// 1488 in scope 2 (scopes 0 / 1 are built for 4 waves per SIMD, scope 2 for 3 - and once more for 4, template value 3, for scenes with few
// material types: legacy_materials 1283 -> 1425, while typed_materials loses 6 % there).
template <bool DEEP, bool FLAT, bool SWEEP, bool MOTION, int SCOPE_>
__global__ void __launch_bounds__(256, SCOPE_ == 2 ? HPT_SPEC_WIDE_WAVES : HPT_SPEC_WAVES) pathTraceSpectralKernel(const DevScene S, const Job job)
{
  constexpr int SCOPE = SCOPE_ == 3 ? 2 : SCOPE_;                            // (3 = scope 2 built for 4 waves per SIMD: scenes with few material types)
  __shared__ uint stackMem[LDS_STACK * 256];
  const uint glane = blockIdx.x * 256u + threadIdx.x;
  TravStack stk; stk.lds = &stackMem[threadIdx.x]; stk.ovf = job.stackOverflow + glane; stk.ovfStride = job.gridLanes;
  TravStats st; st.nodes = st.tris = st.insts = st.waveNodeIters = st.waveTriIters = 0;
  const uint k = glane;
  bool valid = k < job.tidCount;
  uint tid = 0;
  if (valid) { tid = job.tidBegin + (k / job.tidChunk) * job.tidChunk * job.tidStride + (k % job.tidChunk); valid = tid < job.tidEnd; }
  Rng gen; gen.sx = gen.sy = 0;
  uint XY = 0, pixel = 0;
  float pix[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
  const bool inRays = job.inRayPos != nullptr;                               // PathTraceFromInputRays (integrator_pt.cpp:159-199, 659-676, 761-798): the caller's rays and wavelengths, raw samples out
  if (valid) {
    XY = inRays ? (tid < job.packedCount ? job.packedXY[tid] : 0u) : job.packedXY[tid]; gen = job.gens[tid];
    pixel = inRays ? tid : ((XY & 0xFFFF0000u) >> 16) * (uint)S.winWidth + (XY & 0x0000FFFFu);
    if (job.channels == 1) pix[0] = job.outColor[pixel];
    else if (job.channels <= 4) { const float* o = job.outColor + (size_t)pixel * job.channels; pix[0] = o[0]; pix[1] = o[1]; pix[2] = o[2]; if (inRays && job.channels == 4) pix[3] = o[3]; }
  }
  // In-place regeneration, as in the RGB megakernel but without its work queue: a lane whose path has ended starts its pixel's next pass at
  // the top of the loop instead of waiting for the longest path of the wave, so every trip traces a ray for (nearly) all lanes.
  V3 rpos = v3(0, 0, 0), rdir = v3(0, 0, 1);
  V4 waves = v4s(0.0f), accum = v4s(0.0f), thr = v4s(1.0f);
  float misPdf = 1.0f, misIor = 1.0f;                                        // MisData: matSamplePdf, ior (the medium the ray travels in)
  float pathTime = 0.0f;                                                     // motion blur: the path's time (MOTION variants only)
  uint flags = 0, bounce = 0, passesLeft = valid ? job.passNum : 0u;
  bool alive = false;
  while (true) {
    if (!alive && passesLeft != 0u) {
      passesLeft--;
      if (inRays) {                                                          // kernel_InitEyeRayFromInput: no generator step; every sample at the ray's own wavelength
        const float4 ip = job.inRayPos[tid], id4 = job.inRayDir[tid];
        const V3 org = v3(ip.x, ip.y, ip.z), dir = v3(id4.x, id4.y, id4.z);
        const V3 p1 = mul4x3(S.worldViewInv, org), p2 = mul4x3(S.worldViewInv, org + 100.0f * dir);   // transform_ray3f (cglobals.h:254-263)
        rpos = p1; rdir = normalize(p2 - p1);
        if (MOTION) pathTime = id4.w;
        waves = v4s(ip.w);
      } else {
        const V4 lens = rng_float4(gen);                                     // GetRandomNumbersLens, then GetRandomNumbersSpec (integrator_pt.cpp:114-118)
        cameraRay<(SCOPE >= 2)>(S, XY & 0x0000FFFFu, (XY & 0xFFFF0000u) >> 16, lens, rpos, rdir);
        if (MOTION) pathTime = rng_float1(gen);                              // GetRandomNumbersTime, before the wavelength (integrator_pt.cpp:114-118)
        waves = sampleWavelengths(rng_float1(gen), LAMBDA_MIN, LAMBDA_MAX);
      }
      accum = v4s(0.0f); thr = v4s(1.0f); misPdf = 1.0f; misIor = 1.0f; flags = 0; bounce = 0;
      alive = true;
    }
    if (!__any(alive)) break;
    const bool naive = job.naive != 0u;                                      // NaivePathTrace (integrator_pt.cpp:681-717): no light sampling, one more bounce
    const uint maxBounce = naive ? S.traceDepth + 1u : S.traceDepth;
    if (maxBounce != 0u) {                                                   // (depth 0: the path is its camera ray and ends below)
      HitRec hit; hit.inst = 0xFFFFFFFFu; hit.prim = 0; hit.t = 0; hit.u = hit.v = 0;
      if (alive) traceAny<false, false, DEEP, FLAT, MOTION, SWEEP>(S, rpos, rdir, 0.0f, HPT_FLT_MAX, hit, stk, st, pathTime);
      bool wantShadow = false;
      V3 shPos = v3(0, 0, 0), shDir = v3(0, 0, 1); float shFar = 0.0f;
      V4 contrib = v4s(0.0f);
      if (alive) {
        shadeVertexSpec<SCOPE, MOTION>(S, hit, rpos, rdir, waves, accum, thr, misPdf, misIor, flags, bounce, gen, naive, wantShadow, shPos, shDir, shFar, contrib, pathTime);
      }
      if (wantShadow) {
        HitRec sh;
        const bool occluded = traceAny<true, false, DEEP, FLAT, MOTION, SWEEP>(S, shPos, shDir, 0.0f, shFar, sh, stk, st, pathTime);
        if (!occluded) accum = accum + contrib;
      }
      bounce++;
    }
    if (alive && ((flags & RAY_FLAG_IS_DEAD) != 0 || bounce >= maxBounce)) {
      alive = false;
      if ((flags & RAY_FLAG_OUT_OF_SCENE) != 0) {                            // kernel_HitEnvironment
        V4 env = ld4(S.envColor);
        if (SCOPE >= 2) env = environmentRadianceSpec(S, rdir, waves, misPdf, flags, XY);
        else if (S.envSpecId != 0xFFFFFFFFu) env = sampleUniformSpectrum(S.specValues, S.specOffsetSz[2u * S.envSpecId], waves) * (S.envSpecMult / 106.856895f);
        if (S.integratorType == INTEGRATOR_STUPID_PT) accum = thr * env; else accum = accum + thr * env;
      }
      // kernel_ContributeToImage (integrator_pt.cpp:598-657), spectral; input rays: kernel_CopyColorToOutput, the raw samples (:659-676)
      if (inRays) { pix[0] += accum.x; pix[1] += accum.y; pix[2] += accum.z; pix[3] += accum.w; }
      else if (job.channels == 1) pix[0] += accum.x * S.exposureMult;
      else if (job.channels > 4) {                                           // "always spectral rendering": one layer of W x H per wavelength bin
        const V4 color = accum * S.exposureMult;
        for (int i = 0; i < 4; i++) {
          const float t = (comp(waves, i) - LAMBDA_MIN) / (LAMBDA_MAX - LAMBDA_MIN);
          const int channelId = min(int(float(job.channels) * t), int(job.channels) - 1);
          job.outColor[(size_t)channelId * (size_t)(S.winWidth * S.winHeight) + pixel] += comp(color, i);   // the pixel is this thread's alone
        }
      }
      else { const V3 rgb = spectralCamResponseToRGB(S, accum, waves, flags); pix[0] += S.exposureMult * rgb.x; pix[1] += S.exposureMult * rgb.y; pix[2] += S.exposureMult * rgb.z; }
    }
  }
  if (valid) {
    if (job.channels == 1) job.outColor[pixel] = pix[0];
    else if (job.channels <= 4) { float* o = job.outColor + (size_t)pixel * job.channels; o[0] = pix[0]; o[1] = pix[1]; o[2] = pix[2]; if (inRays && job.channels == 4) o[3] = pix[3]; }
    job.gens[tid] = gen;
  }
}

// ---- spectral rendering under the block-local schedule (hpt_block.hip's structure on four wavelengths) ------------------------------------------
// The kernel above is one thread per pixel without a work queue; its lanes keep their ray until the slowest lane of the wave is through. For
// scenes whose rays walk a real tree - the reference's 16 396-triangle spectral fixture, a 10^6-triangle interior under m_spectral_mode = 1 -
// this variant is the RGB block-local megakernel with V4 radiance: persistent blocks pulling pixels from the work queue, the lanes' next
// closest-hit and shadow rays pooled in LDS and drained by the block's four waves with ray replacement (blockTracePhase), the 4-wide compressed
// tree on heavy scenes. The per-vertex arithmetic is shadeVertexSpec, the same function the kernel above calls, in the same order, so both
// render bit-identical frames (tests/test_gpu_spectral.py). PathTraceFromInputRays and moving instances stay with the kernel above.
template <bool DEEP, bool FLAT, bool WIDE, int SCOPE_>
__global__ void __launch_bounds__(256, SCOPE_ == 2 ? HPT_SPEC_WIDE_WAVES : HPT_SPEC_WAVES) pathTraceBlockSpectralKernel(const DevScene S, const Job job, uint refillBelow, uint nodeMin)
{
  constexpr int SCOPE = SCOPE_ == 3 ? 2 : SCOPE_;
  __shared__ uint stackMem[LDS_STACK * 256];
  __shared__ uint pool[8 * BW_POOL];
  __shared__ float coldPix[3 * 256];
  __shared__ uint  coldU[3 * 256];
  __shared__ uint  poolTail, poolHead;
  const uint glane = blockIdx.x * 256u + threadIdx.x;
  const uint lane = threadIdx.x & 63u;
  TravStack stk; stk.lds = &stackMem[threadIdx.x]; stk.ovf = job.stackOverflow + glane; stk.ovfStride = job.gridLanes;
#define PIX(k)      coldPix[(k) * 256 + threadIdx.x]
#define PIX_XY      coldU[0 * 256 + threadIdx.x]
#define PIX_TID     coldU[1 * 256 + threadIdx.x]
#define PIX_PASSES  coldU[2 * 256 + threadIdx.x]
  bool havePixel = false, alive = false, drained = false, pend = false, ending = false;
  uint bounce = 0, flags = 0, myNear = 0, myShad = 0;
  Rng  gen; gen.sx = gen.sy = 0;
  V3   rpos = v3(0, 0, 0), rdir = v3(0, 0, 1);
  V4   waves = v4s(0.0f), accum = v4s(0.0f), thr = v4s(1.0f), contrib = v4s(0.0f);
  float misPdf = 1.0f, misIor = 1.0f;
  const bool naive = job.naive != 0u;
  const uint maxBounce = naive ? S.traceDepth + 1u : S.traceDepth;
  HitRec hit; hit.inst = 0xFFFFFFFFu; hit.prim = 0; hit.t = 0; hit.u = hit.v = 0;
  bool occluded = false;
  if (threadIdx.x == 0u) { poolTail = 0u; poolHead = 0u; }
  __syncthreads();

  while (true) {
    bool wantShadow = false;
    V3 shPos = v3(0, 0, 0), shDir = v3(0, 0, 1); float shFar = 0.0f;
    if (pend) { if (!occluded) accum = accum + contrib; pend = false; }     // the shadow ray of the last round, added in the plain kernel's order
    bool finalize = ending;
    ending = false;
    if (alive) {
      if (maxBounce != 0u) {
        shadeVertexSpec<SCOPE, false>(S, hit, rpos, rdir, waves, accum, thr, misPdf, misIor, flags, bounce, gen, naive, wantShadow, shPos, shDir, shFar, contrib, 0.0f);
        bounce++;                                                            // (the plain kernel counts every trip, a miss included)
      }
      if ((flags & RAY_FLAG_IS_DEAD) != 0 || bounce >= maxBounce) {
        alive = false;
        if (wantShadow) ending = true; else finalize = true;
      }
    }
    if (finalize) {
      if ((flags & RAY_FLAG_OUT_OF_SCENE) != 0) {                            // kernel_HitEnvironment, after the last shadow contribution as in the plain kernel
        V4 env = ld4(S.envColor);
        if (SCOPE >= 2) env = environmentRadianceSpec(S, rdir, waves, misPdf, flags, PIX_XY);
        else if (S.envSpecId != 0xFFFFFFFFu) env = sampleUniformSpectrum(S.specValues, S.specOffsetSz[2u * S.envSpecId], waves) * (S.envSpecMult / 106.856895f);
        if (S.integratorType == INTEGRATOR_STUPID_PT) accum = thr * env; else accum = accum + thr * env;
      }
      // kernel_ContributeToImage (integrator_pt.cpp:598-657), spectral
      if (job.channels == 1) PIX(0) += accum.x * S.exposureMult;
      else if (job.channels > 4) {
        const uint XY = PIX_XY;
        const uint pixel = ((XY & 0xFFFF0000u) >> 16) * (uint)S.winWidth + (XY & 0x0000FFFFu);
        const V4 color = accum * S.exposureMult;
        for (int i = 0; i < 4; i++) {
          const float t = (comp(waves, i) - LAMBDA_MIN) / (LAMBDA_MAX - LAMBDA_MIN);
          const int channelId = min(int(float(job.channels) * t), int(job.channels) - 1);
          job.outColor[(size_t)channelId * (size_t)(S.winWidth * S.winHeight) + pixel] += comp(color, i);   // the pixel is this lane's alone
        }
      }
      else { const V3 rgb = spectralCamResponseToRGB(S, accum, waves, flags); PIX(0) += S.exposureMult * rgb.x; PIX(1) += S.exposureMult * rgb.y; PIX(2) += S.exposureMult * rgb.z; }
    }
    const bool idle = !alive && !ending;
    if (idle && havePixel && PIX_PASSES == 0u) {                             // a finished pixel goes back to HBM
      const uint XY = PIX_XY;
      const uint pixel = ((XY & 0xFFFF0000u) >> 16) * (uint)S.winWidth + (XY & 0x0000FFFFu);
      if (job.channels == 1) job.outColor[pixel] = PIX(0);
      else if (job.channels <= 4) { float* o = job.outColor + (size_t)pixel * job.channels; o[0] = PIX(0); o[1] = PIX(1); o[2] = PIX(2); }
      job.gens[PIX_TID] = gen;
      havePixel = false;
    }
    {                                                                        // work queue: ballot the lanes without a pixel, one atomic per wave
      const bool need = idle && !havePixel && !drained;
      const unsigned long long mask = __ballot(need);
      if (mask != 0ull) {
        uint base = 0;
        if (need && mbcnt64(mask) == 0u) base = atomicAdd(job.queue, (uint)__popcll(mask));
        base = __shfl(base, (int)(__ffsll((long long)mask) - 1));
        if (need) {
          const uint k = base + mbcnt64(mask);
          const uint tid = job.tidBegin + (k / job.tidChunk) * job.tidChunk * job.tidStride + (k % job.tidChunk);
          if (k < job.tidCount && tid < job.tidEnd) {
            const uint XY = job.packedXY[tid];
            gen = job.gens[tid];
            const uint pixel = ((XY & 0xFFFF0000u) >> 16) * (uint)S.winWidth + (XY & 0x0000FFFFu);
            PIX(0) = 0.0f; PIX(1) = 0.0f; PIX(2) = 0.0f;
            if (job.channels == 1) PIX(0) = job.outColor[pixel];
            else if (job.channels <= 4) { const float* o = job.outColor + (size_t)pixel * job.channels; PIX(0) = o[0]; PIX(1) = o[1]; PIX(2) = o[2]; }
            PIX_XY = XY; PIX_TID = tid; PIX_PASSES = job.passNum;
            havePixel = true;
          } else drained = true;
        }
      }
    }
    if (idle && havePixel) {                                                 // next pass of the pixel: GetRandomNumbersLens, then GetRandomNumbersSpec (integrator_pt.cpp:114-118)
      PIX_PASSES = PIX_PASSES - 1u;
      const V4 lens = rng_float4(gen);
      const uint XY = PIX_XY;
      cameraRay<(SCOPE >= 2)>(S, XY & 0x0000FFFFu, (XY & 0xFFFF0000u) >> 16, lens, rpos, rdir);
      waves = sampleWavelengths(rng_float1(gen), LAMBDA_MIN, LAMBDA_MAX);
      accum = v4s(0.0f); thr = v4s(1.0f); misPdf = 1.0f; misIor = 1.0f; flags = 0; bounce = 0;
      alive = true;
    }
    {                                                                        // append this lane's rays to the block's pool
      const bool qNear = alive && maxBounce != 0u;
      const unsigned long long mn = __ballot(qNear), ms = __ballot(wantShadow);
      const uint cn = (uint)__popcll(mn), cs = (uint)__popcll(ms);
      uint base = 0;
      if (lane == 0u && cn + cs != 0u) base = atomicAdd(&poolTail, cn + cs);
      base = __shfl(base, 0);
      if (qNear) {
        const uint e = base + mbcnt64(mn);
        pool[0 * BW_POOL + e] = __float_as_uint(rpos.x); pool[1 * BW_POOL + e] = __float_as_uint(rpos.y); pool[2 * BW_POOL + e] = __float_as_uint(rpos.z);
        pool[3 * BW_POOL + e] = __float_as_uint(HPT_FLT_MAX);
        pool[4 * BW_POOL + e] = __float_as_uint(rdir.x); pool[5 * BW_POOL + e] = __float_as_uint(rdir.y); pool[6 * BW_POOL + e] = __float_as_uint(rdir.z);
        pool[7 * BW_POOL + e] = 0u;
        myNear = e;
      }
      if (wantShadow) {
        const uint e = base + cn + mbcnt64(ms);
        pool[0 * BW_POOL + e] = __float_as_uint(shPos.x); pool[1 * BW_POOL + e] = __float_as_uint(shPos.y); pool[2 * BW_POOL + e] = __float_as_uint(shPos.z);
        pool[3 * BW_POOL + e] = __float_as_uint(shFar);
        pool[4 * BW_POOL + e] = __float_as_uint(shDir.x); pool[5 * BW_POOL + e] = __float_as_uint(shDir.y); pool[6 * BW_POOL + e] = __float_as_uint(shDir.z);
        pool[7 * BW_POOL + e] = 1u;
        myShad = e; pend = true;
      }
    }
    __syncthreads();
    const uint total = poolTail;
    if (total == 0u) break;
    blockTracePhase<DEEP, FLAT, WIDE>(S, stk, pool, &poolHead, total, refillBelow, nodeMin, lane);
    __syncthreads();
    if (alive) {
      hit.t = __uint_as_float(pool[0 * BW_POOL + myNear]); hit.u = __uint_as_float(pool[1 * BW_POOL + myNear]); hit.v = __uint_as_float(pool[2 * BW_POOL + myNear]);
      hit.prim = pool[3 * BW_POOL + myNear]; hit.inst = pool[4 * BW_POOL + myNear];
    }
    if (pend) occluded = pool[0 * BW_POOL + myShad] != 0u;
    if (threadIdx.x == 0u) { poolTail = 0u; poolHead = 0u; }
    __syncthreads();
  }
#undef PIX
#undef PIX_XY
#undef PIX_TID
#undef PIX_PASSES
}

// ---- wavefront schedule ---------------------------------------------------------------------------------------------------------------------------
// Heavy scenes in calls with enough pixels (hpt_host.hip: useWavefront's rule): the shade half of a bounce as wfShadeKernel has it for RGB - one lane per
// pool slot, the previous shadow ray folded in first, shadeVertexSpec for the closest hit, path ends and regeneration in place, the slot's rays appended to
// the queue the shared trace kernel (hpt_wavefront.hip: wfTraceKernel, the 4-wide tree with ray replacement) drains. Four samples per radiance-valued word:
// WfPool::thr / acc / contrib are used in full, the wavelengths and (flags, bounce) have arrays of their own. Same arithmetic in the same order as the
// plain kernel: bit-identical frames (tests/test_gpu_spectral.py). PathTraceBlock only (no naive / input-ray / moving-instance variant: those keep their kernels).
template <int SCOPE_>
__global__ void __launch_bounds__(256, SCOPE_ == 2 ? HPT_SPEC_WIDE_WAVES : HPT_SPEC_WAVES) wfShadeSpecKernel(const DevScene S, const WfPool P, const WfJob job)
{
  constexpr int SCOPE = SCOPE_ == 3 ? 2 : SCOPE_;
  const uint s = blockIdx.x * 256u + threadIdx.x;
  uint* ctr = P.ctr + WF_CTR_WORDS * (job.iter & 1u);
  if (s <= WF_RANGES) P.ctr[WF_CTR_WORDS * ((job.iter + 1u) & 1u) + 32u * s] = 0u;   // counters of the NEXT round (its trace pass is long done)

  bool valid = s < job.itemCount;
  uint tid = 0;
  if (valid) {
    const uint k = job.itemBase + s;
    tid = job.tidBegin + (k / job.tidChunk) * job.tidChunk * job.tidStride + (k % job.tidChunk);
    valid = tid < job.tidEnd;
  }
  if (valid && ldP(&P.inflight[s]) != 0u) valid = false;       // a suspended ray of this pixel is still being traced: the pixel sits this round out
  uint st = valid ? ldP(&P.status[s]) : 0u;
  uint passes = st >> 8;
  bool alive = (st & WF_ALIVE) != 0u, pend = (st & WF_PEND) != 0u, ending = (st & WF_ENDING) != 0u;
  const bool active = valid && (alive || pend || ending || passes != 0u);
  bool wantShadow = false;

  if (active) {
    Rng gen = ldP(&job.gens[tid]);
    const uint XY = job.packedXY[tid];
    V4 accum = v4s(0.0f), thr = v4s(1.0f), waves = v4s(0.0f), contrib = v4s(0.0f);
    V3 rpos = v3(0, 0, 0), rdir = v3(0, 0, 1);
    float misPdf = 1.0f, misIor = 1.0f; uint flags = 0, bounce = 0;
    if (alive || ending) {
      const float4 a = ldP(&P.acc[s]), w = ldP(&P.waves[s]); const uint2 f = ldP(&P.fb[s]);
      accum = v4(a.x, a.y, a.z, a.w); waves = v4(w.x, w.y, w.z, w.w); flags = f.x; bounce = f.y;
    }
    // the shadow ray traced since the last visit: add the candidate contribution in the plain kernel's order
    if (pend) {
      if (ldP(&P.occl[s]) == 0u) { const float4 c = ldP(&P.contrib[s]); accum = accum + v4(c.x, c.y, c.z, c.w); }
      pend = false;
    }
    bool finalize = ending;                                                // path ended last time, only its shadow ray was outstanding
    ending = false;
    V3 shPos = v3(0, 0, 0), shDir = v3(0, 0, 1); float shFar = 0.0f;
    if (alive) {
      const float4 ro = ldP(&P.rayO[s]), rd = ldP(&P.rayD[s]), t4 = ldP(&P.thr[s]), h4 = ldP(&P.hit[s]);
      rpos = v3(ro.x, ro.y, ro.z); misPdf = ro.w; rdir = v3(rd.x, rd.y, rd.z); misIor = rd.w;
      thr = v4(t4.x, t4.y, t4.z, t4.w);
      HitRec hit; hit.t = h4.x; hit.u = h4.y; hit.v = h4.z; hit.prim = __float_as_uint(h4.w); hit.inst = ldP(&P.hitInst[s]);   // (the host hands the trace pass a scene without shading records: primitive ids come back)
      shadeVertexSpec<SCOPE, false>(S, hit, rpos, rdir, waves, accum, thr, misPdf, misIor, flags, bounce, gen, false, wantShadow, shPos, shDir, shFar, contrib, 0.0f);
      bounce++;                                                              // (the plain kernel counts every trip, a miss included)
      if ((flags & RAY_FLAG_IS_DEAD) != 0 || bounce >= S.traceDepth) {
        alive = false;
        if (wantShadow) ending = true; else finalize = true;
      }
    }
    if (finalize) {
      if ((flags & RAY_FLAG_OUT_OF_SCENE) != 0) {                            // kernel_HitEnvironment (a ray that left the scene sampled no light: never a slot whose shadow ray was outstanding)
        V4 env = ld4(S.envColor);
        if (SCOPE >= 2) env = environmentRadianceSpec(S, rdir, waves, misPdf, flags, XY);
        else if (S.envSpecId != 0xFFFFFFFFu) env = sampleUniformSpectrum(S.specValues, S.specOffsetSz[2u * S.envSpecId], waves) * (S.envSpecMult / 106.856895f);
        if (S.integratorType == INTEGRATOR_STUPID_PT) accum = thr * env; else accum = accum + thr * env;
      }
      // kernel_ContributeToImage (integrator_pt.cpp:598-657), spectral: the pixel is this slot's alone
      const uint pixel = ((XY & 0xFFFF0000u) >> 16) * (uint)S.winWidth + (XY & 0x0000FFFFu);
      if (job.channels == 1) job.outColor[pixel] += accum.x * S.exposureMult;
      else if (job.channels > 4) {
        const V4 color = accum * S.exposureMult;
        for (int i = 0; i < 4; i++) {
          const float t = (comp(waves, i) - LAMBDA_MIN) / (LAMBDA_MAX - LAMBDA_MIN);
          const int channelId = min(int(float(job.channels) * t), int(job.channels) - 1);
          job.outColor[(size_t)channelId * (size_t)(S.winWidth * S.winHeight) + pixel] += comp(color, i);
        }
      }
      else { const V3 rgb = spectralCamResponseToRGB(S, accum, waves, flags); float* o = job.outColor + (size_t)pixel * job.channels; o[0] += S.exposureMult * rgb.x; o[1] += S.exposureMult * rgb.y; o[2] += S.exposureMult * rgb.z; }
    }
    if (!alive && !ending && passes != 0u) {                                 // next pass of the pixel: GetRandomNumbersLens, then GetRandomNumbersSpec (integrator_pt.cpp:114-118)
      passes--;
      const V4 lens = rng_float4(gen);
      cameraRay<(SCOPE >= 2)>(S, XY & 0x0000FFFFu, (XY & 0xFFFF0000u) >> 16, lens, rpos, rdir);
      waves = sampleWavelengths(rng_float1(gen), LAMBDA_MIN, LAMBDA_MAX);
      accum = v4s(0.0f); thr = v4s(1.0f); misPdf = 1.0f; misIor = 1.0f; flags = 0; bounce = 0;
      alive = true;
    }
    stP(&job.gens[tid], gen);
    if (alive) {
      stP(&P.rayO[s], make_float4(rpos.x, rpos.y, rpos.z, misPdf));
      stP(&P.rayD[s], make_float4(rdir.x, rdir.y, rdir.z, misIor));
      stP(&P.thr[s], make_float4(thr.x, thr.y, thr.z, thr.w));
    }
    if (alive || ending) { stP(&P.acc[s], make_float4(accum.x, accum.y, accum.z, accum.w)); stP(&P.waves[s], make_float4(waves.x, waves.y, waves.z, waves.w)); stP(&P.fb[s], make_uint2(flags, bounce)); }
    if (wantShadow) {
      stP(&P.shO[s], make_float4(shPos.x, shPos.y, shPos.z, shFar));
      stP(&P.shD[s], make_float4(shDir.x, shDir.y, shDir.z, 0.0f));
      stP(&P.contrib[s], make_float4(contrib.x, contrib.y, contrib.z, contrib.w));
    }
    stP(&P.status[s], (passes << 8) | (alive ? WF_ALIVE : 0u) | (wantShadow ? WF_PEND : 0u) | (ending ? WF_ENDING : 0u));
  }
  const bool qNear = active && alive, qShad = active && wantShadow;
  uint kn, ks;
  blockAppend(&ctr[0], qNear, qShad, kn, ks);
  uint* rayQ = P.rayQ[job.iter & 1u];
  if (qNear) stP(&rayQ[kn], s);
  if (qShad) stP(&rayQ[ks], s | 0x80000000u);
}

// one translation unit per scope (-DHPT_SPEC_INST=1 / 2 / 3 / 4: scope 0 / 1 / 2 / 2 at 4 waves; 0: all), see __graft_entry__.build
#ifndef HPT_SPEC_INST
#define HPT_SPEC_INST 0
#endif
#define HPT_SPEC_BLOCK(SCOPE) \
  template __global__ void pathTraceBlockSpectralKernel<false, false, false, SCOPE>(const DevScene, const Job, uint, uint); \
  template __global__ void pathTraceBlockSpectralKernel<true,  false, false, SCOPE>(const DevScene, const Job, uint, uint); \
  template __global__ void pathTraceBlockSpectralKernel<false, true,  false, SCOPE>(const DevScene, const Job, uint, uint); \
  template __global__ void pathTraceBlockSpectralKernel<true,  true,  false, SCOPE>(const DevScene, const Job, uint, uint); \
  template __global__ void pathTraceBlockSpectralKernel<false, true,  true,  SCOPE>(const DevScene, const Job, uint, uint); \
  template __global__ void pathTraceBlockSpectralKernel<true,  true,  true,  SCOPE>(const DevScene, const Job, uint, uint);
#define HPT_SPEC(SCOPE) \
  template __global__ void pathTraceSpectralKernel<false, false, false, false, SCOPE>(const DevScene, const Job); \
  template __global__ void pathTraceSpectralKernel<true,  false, false, false, SCOPE>(const DevScene, const Job); \
  template __global__ void pathTraceSpectralKernel<false, true,  false, false, SCOPE>(const DevScene, const Job); \
  template __global__ void pathTraceSpectralKernel<true,  true,  false, false, SCOPE>(const DevScene, const Job); \
  template __global__ void pathTraceSpectralKernel<false, false, true,  false, SCOPE>(const DevScene, const Job); \
  template __global__ void pathTraceSpectralKernel<false, false, false, true,  SCOPE>(const DevScene, const Job);   /* moving instances */ \
  template __global__ void pathTraceSpectralKernel<true,  false, false, true,  SCOPE>(const DevScene, const Job); \
  template __global__ void pathTraceSpectralKernel<false, true,  false, true,  SCOPE>(const DevScene, const Job); \
  template __global__ void pathTraceSpectralKernel<true,  true,  false, true,  SCOPE>(const DevScene, const Job);
#if HPT_SPEC_INST == 0 || HPT_SPEC_INST == 1
HPT_SPEC(0)
HPT_SPEC_BLOCK(0)
#endif
#if HPT_SPEC_INST == 0 || HPT_SPEC_INST == 2
HPT_SPEC(1)
HPT_SPEC_BLOCK(1)
template __global__ void wfShadeSpecKernel<1>(const DevScene, const WfPool, const WfJob);
#endif
#if HPT_SPEC_INST == 0 || HPT_SPEC_INST == 3
HPT_SPEC(2)
HPT_SPEC_BLOCK(2)
template __global__ void wfShadeSpecKernel<2>(const DevScene, const WfPool, const WfJob);
#endif
#if HPT_SPEC_INST == 0 || HPT_SPEC_INST == 4
HPT_SPEC(3)
HPT_SPEC_BLOCK(3)
template __global__ void wfShadeSpecKernel<3>(const DevScene, const WfPool, const WfJob);
#endif

} // namespace hpt
