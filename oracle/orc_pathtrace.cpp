// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).  PARITY UNPINNED: the
// reference has no golden vectors for this path and cannot be compiled in this image (LiteMath, Embree,
// Enzyme absent); this file restates its algorithm and is pinned only by closed-form checks in tests/.
//
// CPU restatement of
//   Integrator::PathTraceBlock / NaivePathTraceBlock      integrator_pt_host.cpp:39-73
//   Integrator::PathTrace / NaivePathTrace + kernel_*     integrator_pt.cpp:13-157, 214-657, 681-759
//   MaterialSampleAndEval / MaterialEval / RemapMaterialId integrator_pt_mat.cpp:109-306, 308-573
//   LightSampleRev / LightEvalPDF / LightIntensity        integrator_pt_lgt.cpp:21-173
//   kernel_PackXY                                         integrator_rt.cpp:13-31
//   IntegratorDR::PathTraceDR / PathTraceReplay / PixelLossPT / Tex2DFetchAD / PutDiffTex2D
//                                                         diff_render/integrator_dr.cpp:33-161, 461-612, 790-1218
//   AdamOptimizer<float>::step                            diff_render/adam.h:43-62
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include "orc_shade.h"
#include <omp.h>
#include <cstdio>
#include <cstdlib>
#include <string>

namespace orc {

struct Params
{
  m4 projInv, worldViewInv;
  int winStartX, winStartY, winWidth, winHeight, fbWidth, fbHeight;
  uint traceDepth, integratorType, renderLayer, tileSize, spectralMode;
  float exposureMult, camLensRadius, camTargetDist;
  f4 camRespoceRGB, envColor;
  uint envTexId, envLightId, envCamBackId, envEnableSam;
  f4 envSamRow0, envSamRow1;
  uint envSpecId; float envSpecMult;           // m_envSpecId, m_envSpecMult
};

// IntegratorDR::TexInfo (diff_render/integrator_dr.h:56-64)
struct TexInfo { size_t offset; int width, height, channels; };

// what the Record* hooks capture (diff_render/integrator_dr2.cpp:22-80, integrator_dr.h:93-136)
static const int MAX_REC_BOUNCE = 16;
struct Record
{
  f4 lens;                       // pixel offsets / lens randoms
  orc_hit hit[MAX_REC_BOUNCE];
  int inShadow[MAX_REC_BOUNCE];  // NOTE inverted naming in the reference: 0 means "needShade"
  f4 lgt[MAX_REC_BOUNCE], mat[MAX_REC_BOUNCE];
  bool enabled;
};

// ---- forward-mode dual colour for the DR replay ------------------------------------------------------------------
// value + d(value)/d(texColor fetched at bounce b), per channel (channels never mix in the gltf/lambert path).
static const int MAXB = MAX_REC_BOUNCE;
struct AF4
{
  f4 v; f4 d[MAXB];
  AF4() { v = splat4(0); for (int i = 0; i < MAXB; i++) d[i] = splat4(0); }
  AF4(f4 a) { v = a; for (int i = 0; i < MAXB; i++) d[i] = splat4(0); }
};
static inline AF4 operator*(const AF4& a, f4 s) { AF4 r; r.v = a.v * s; for (int i = 0; i < MAXB; i++) r.d[i] = a.d[i] * s; return r; }
static inline AF4 operator*(const AF4& a, float s) { AF4 r; r.v = a.v * s; for (int i = 0; i < MAXB; i++) r.d[i] = a.d[i] * s; return r; }
static inline AF4 operator+(const AF4& a, const AF4& b) { AF4 r; r.v = a.v + b.v; for (int i = 0; i < MAXB; i++) r.d[i] = a.d[i] + b.d[i]; return r; }
static inline AF4 operator-(const AF4& a, const AF4& b) { AF4 r; r.v = a.v - b.v; for (int i = 0; i < MAXB; i++) r.d[i] = a.d[i] - b.d[i]; return r; }
static inline AF4 operator/(const AF4& a, float s) { AF4 r; r.v = a.v / s; for (int i = 0; i < MAXB; i++) r.d[i] = a.d[i] / s; return r; }
static inline AF4 mulAA(const AF4& a, const AF4& b) { AF4 r; r.v = a.v * b.v; for (int i = 0; i < MAXB; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }

struct Ctx
{
  Scene  sc;
  Params p;
  std::vector<uint>      packedXY;
  std::vector<RandomGen> randomGens;
  std::vector<TexInfo>   texAddressTable;   // DR
  size_t                 gradSize = 0;
  bool                   drSkipNonFinite = false;   // orc_set_option("dr_skip_nonfinite"): mirror of the product's option, not reference behaviour

  // ------------------------------------------------------------------------------------------------------------
  // integrator_rt.cpp:13-31
  void PackXY()
  {
    const int W = p.winWidth, H = p.winHeight;
    packedXY.assign((size_t)W * H, 0u);
    const uint ts = p.tileSize;
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        uint offset = (uint)y * (uint)W + (uint)x;
        if (ts != 1) {
          const uint inX = (uint)x % ts, inY = (uint)y % ts;
          const uint localIndex = inY * ts + inX;
          const uint wBlocks = (uint)W / ts;
          const uint blockX = (uint)x / ts, blockY = (uint)y / ts;
          offset = (blockX + blockY * wBlocks) * ts * ts + localIndex;
        }
        packedXY[offset] = (((uint)y << 16) & 0xFFFF0000u) | ((uint)x & 0x0000FFFFu);
      }
  }

  // integrator_pt_mat.cpp:530-573
  uint RemapMaterialId(uint a_mId, int a_instId) const
  {
    const int remapListId = sc.remapInst[2 * a_instId + 0];
    if (remapListId == -1) return a_mId;
    const int r_offset = sc.allRemapLists[sc.allRemapListsSize + remapListId];
    const int r_size = sc.allRemapLists[sc.allRemapListsSize + remapListId + 1] - r_offset;
    uint res = a_mId;
    int low = 0, high = r_size - 1;
    while (low <= high) {
      const int mid = low + ((high - low) / 2);
      const int idRemapFrom = sc.allRemapLists[r_offset + mid * 2 + 0];
      if (uint(idRemapFrom) >= a_mId) high = mid - 1; else low = mid + 1;
    }
    if (high + 1 < r_size) {
      const int idRemapFrom = sc.allRemapLists[r_offset + (high + 1) * 2 + 0];
      const int idRemapTo = sc.allRemapLists[r_offset + (high + 1) * 2 + 1];
      res = (uint(idRemapFrom) == a_mId) ? uint(idRemapTo) : a_mId;
    }
    return res;
  }

  // ---- lights (integrator_pt_lgt.cpp) ------------------------------------------------------------------------------
  LightSample LightSampleRev(int a_lightId, f3 rands, f3 illiminationPoint) const   // :21-58
  {
    const LightSource& L = sc.lights[a_lightId];
    const f2 rands2 = mk2(rands.x, rands.y);
    switch (L.geomType) {
      case LIGHT_GEOM_DIRECT: return directLightSampleRev(L, rands2, illiminationPoint);
      case LIGHT_GEOM_SPHERE: return sphereLightSampleRev(L, rands2);
      case LIGHT_GEOM_POINT:  return pointLightSampleRev(L);
      case LIGHT_GEOM_ENV: {                                          // :30-55: SampleMap2D (:212-236) over the pdf table in m_arrays1f
        const int sizeX = (int)L.pdfTableSizeX, sizeY = (int)L.pdfTableSizeY;
        const float fw = (float)sizeX, fh = (float)sizeY, fN = fw * fh;
        float pdf = 1.0f;
        int pixelOffset = SelectIndexPropToOpt(rands.z, sc.arrays1f.data(), (int)L.pdfTableOffset, sizeX * sizeY + 1, &pdf);
        if (pixelOffset >= sizeX * sizeY) pixelOffset = sizeX * sizeY - 1;
        const int yPos = pixelOffset / sizeX, xPos = pixelOffset - yPos * sizeX;
        const float texX = (1.0f / fw) * (((float)(xPos) + 0.5f) + (rands.x * 2.0f - 1.0f) * 0.5f);
        const float texY = (1.0f / fh) * (((float)(yPos) + 0.5f) + (rands.y * 2.0f - 1.0f) * 0.5f);
        const float mapPdf = pdf * fN;
        const f2 texCoordT = mulRows2x4(L.samplerRow0Inv, L.samplerRow1Inv, mk2(texX, texY));
        float sintheta = 0.0f;
        const f3 sampleDir = texCoord2DToSphereMap(texCoordT, &sintheta);
        LightSample res;
        res.hasIES = false; res.isOmni = true; res.norm = sampleDir;
        res.pos = illiminationPoint + sampleDir * 1000.0f;
        res.pdf = (mapPdf * 1.0f) / (2.f * kPI * kPI * std::max(std::abs(sintheta), 1e-20f));
        return res;
      }
      default:                return areaLightSampleRev(L, rands2);
    }
  }
  float LightPdfSelectRev(int) const { return 1.0f / float(sc.lights.size()); }   // :60-63

  float LightEvalPDF(int a_lightId, f3 illuminationPoint, f3 ray_dir, f3 lpos, f3 lnorm, float a_envPdf) const   // :71-107
  {
    const LightSource& L = sc.lights[a_lightId];
    const uint gtype = L.geomType;
    if (gtype == LIGHT_GEOM_ENV) return a_envPdf;
    const float hitDist = length(illuminationPoint - lpos);
    const float cosValTmp = dot(ray_dir, -1.0f * lnorm);
    float cosVal = 1.0f;
    switch (gtype) {
      case LIGHT_GEOM_SPHERE: { const f3 dirToV = normalize(lpos - illuminationPoint); cosVal = std::abs(dot(dirToV, lnorm)); } break;
      case LIGHT_GEOM_POINT:  { if (L.distType == LIGHT_DIST_LAMBERT) cosVal = std::max(cosValTmp, 0.0f); } break;
      default: cosVal = (L.iesId == uint(-1)) ? std::max(cosValTmp, 0.0f) : std::abs(cosValTmp); break;
    }
    return PdfAtoW(L.pdfA, hitDist, cosVal);
  }

  // ---- spectral rendering (m_spectral_mode != 0): spectrum.h + integrator_spectrum.cpp --------------------------------------
  // Restated for what the reference's own spectral fixture exercises - diffuse and conductor BSDFs, emissive surfaces, analytic lights - and plastic.
  // The other BSDFs keep their RGB parameters here; spectral textures (lambda_ref_ids), thin films and the environment spectrum are not restated.
  static constexpr float LAMBDA_MIN = 360.0f, LAMBDA_MAX = 830.0f;     // include/cglobals.h:22-23
  static f4 SampleWavelengths(float u, float a, float b)               // spectrum.h:58-75
  {
    float res[4];
    res[0] = lerpf(a, b, u);
    const float delta = (b - a) / 4.0f;
    for (int i = 1; i < 4; ++i) { res[i] = res[i - 1] + delta; if (res[i] > b) res[i] = a + (res[i] - b); }
    return mk4(res[0], res[1], res[2], res[3]);
  }
  f4 SampleUniformSpectrum(uint a_offset, f4 a_wavelengths) const      // spectrum.h:106-126
  {
    const int WAVESN = int(LAMBDA_MAX - LAMBDA_MIN);
    const float* w = &a_wavelengths.x;
    float r[4];
    for (int i = 0; i < 4; i++) {
      const int index1 = int(std::min(std::max(w[i] - LAMBDA_MIN, 0.0f), float(WAVESN - 1)));
      const int index2 = std::min(index1 + 1, WAVESN - 1);
      const float x1 = LAMBDA_MIN + float(index1);
      const float y1 = sc.specValues[a_offset + (uint)index1], y2 = sc.specValues[a_offset + (uint)index2];
      r[i] = y1 + (w[i] - x1) * (y2 - y1);
    }
    return mk4(r[0], r[1], r[2], r[3]);
  }
  // BinarySearchU2 (spectrum.h:42-55). The reference shrinks its range by `last - half + 1` where pbrt's FindInterval has `size - (half + 1)`: for
  // wavelengths in the upper part of a spectrum its probes leave the table. A probe past the end is answered "greater" here (what a read of
  // large garbage gives): the loop then ends, and the clamp at the bottom returns the interval the wavelength lies in - the reference's result
  // wherever its own reads stay inside the table.
  static uint BinarySearchU2(const uint* array2, int a_offset, uint array_sz, float val)
  {
    int last = int(array_sz) - 2, first = 1;
    while (last > 0) {
      const int half = last >> 1, middle = first + half;
      const bool predResult = middle < int(array_sz) && float(array2[2 * (a_offset + middle) + 1]) <= val;
      first = predResult ? middle + 1 : first;
      last = predResult ? last - half + 1 : half;
    }
    return uint(std::min(std::max(first - 1, 0), int(array_sz) - 2));
  }
  f4 SampleMatColorSpectrumTexture(uint matId, f4 a_wavelengths, uint paramId, uint paramSpecId, f2 texCoords) const   // integrator_spectrum.cpp:128-180
  {
    f4 res = sc.materials[matId].colors[paramId];
    if (a_wavelengths.x == 0.0f) return res;
    const uint specId = sc.materials[matId].spdid[paramSpecId];
    if (specId >= 0xFFFFFFFFu) return res;
    const uint tex_size = 2 * (size_t)specId + 1 < sc.specTexOffsetSz.size() ? sc.specTexOffsetSz[2 * specId + 1] : 0u;
    if (tex_size == 0u) return SampleUniformSpectrum(sc.specOffsetSz[2 * specId], a_wavelengths);
    // a spectrum given by textures (KSPEC_SPD_TEX): one texture per wavelength band, linear between the two around the wavelength; every
    // component of the tabulated lookup is overwritten (its table offset is 0xFFFFFFFF in the reference: not read here)
    const uint tex_offset = sc.specTexOffsetSz[2 * specId];
    const uint* tw = sc.specTexIdsWavelengths.data();
    const Material& m = sc.materials[matId];
    float* r = &res.x; const float* w = &a_wavelengths.x;
    for (int i = 0; i < 4; ++i) {
      if (w[i] < float(tw[2 * tex_offset + 1]) || w[i] > float(tw[2 * (tex_offset + tex_size - 1) + 1])) { r[i] = 0.0f; continue; }
      const uint o = BinarySearchU2(tw, int(tex_offset), tex_size, w[i]);
      const uint texID1 = tw[2 * (tex_offset + o)], texID2 = tw[2 * (tex_offset + o + 1)];
      const f2 texCoordT = mulRows2x4(m.row0[0], m.row1[0], texCoords);
      const f4 c1 = sc.tex_sample(texID1, texCoordT), c2 = sc.tex_sample(texID2, texCoordT);
      const float t = (w[i] - float(tw[2 * (tex_offset + o) + 1])) / float(tw[2 * (tex_offset + o + 1) + 1] - tw[2 * (tex_offset + o) + 1]);
      r[i] = lerpf(c1.x, c2.x, t);
    }
    return res;
  }
  f4 SampleMatParamSpectrum(uint matId, f4 a_wavelengths, uint paramId, uint paramSpecId) const          // integrator_spectrum.cpp:25-44
  {
    f4 res = splat4(sc.materials[matId].data[paramId]);
    if (a_wavelengths.x == 0.0f) return res;
    const uint specId = sc.materials[matId].spdid[paramSpecId];
    if (specId < 0xFFFFFFFFu) res = SampleUniformSpectrum(sc.specOffsetSz[2 * specId], a_wavelengths);
    return res;
  }
  f4 SampleFilmsSpectrum(uint matId, f4 a_wavelengths, uint paramId, uint paramSpecId, uint layer) const   // integrator_spectrum.cpp:46-65
  {
    f4 res = splat4(sc.filmsEtaK[as_uint_(sc.materials[matId].data[paramId]) + layer]);
    const uint specId = sc.filmsSpecId[as_uint_(sc.materials[matId].data[paramSpecId]) + layer];
    if (specId < 0xFFFFFFFFu) res = SampleUniformSpectrum(sc.specOffsetSz[2 * specId], a_wavelengths);   // (KSPEC_SPECTRAL_RENDERING != 0 holds on the CPU: in RGB mode too)
    return res;
  }
  // what both film branches of integrator_pt_mat.cpp set up before the BSDF call (:203-237, 431-463)
  struct FilmArgs { float extIOR, thickness; cplx intIOR, filmIOR; f4 wavelengths_spec; bool spectral_mode, precomp_flag; uint precomp_offset; };
  FilmArgs filmArgs(uint matId, f2 tc, f4 wavelengths, bool evalBranch) const
  {
    const Material& m = sc.materials[matId];
    FilmArgs a;
    const uint layers = as_uint_(m.data[FILM_LAYERS_COUNT]);
    a.spectral_mode = wavelengths.x > 0.0f;
    a.wavelengths_spec = a.spectral_mode ? mk4(wavelengths.x, 0, 0, 0) : (evalBranch ? mk4(700.f, 525.f, 450.f, 0.0f) : mk4(645.f, 525.f, 445.f, 0.0f));
    const f4 wavelengths_sample = a.spectral_mode ? mk4(wavelengths.x, 0, 0, 0) : mk4(525.f, 0, 0, 0);
    a.extIOR = m.data[FILM_ETA_EXT];
    a.intIOR = cmk(SampleFilmsSpectrum(matId, wavelengths_sample, FILM_ETA_OFFSET, FILM_ETA_SPECID_OFFSET, layers - 1).x,
                   SampleFilmsSpectrum(matId, wavelengths_sample, FILM_K_OFFSET, FILM_K_SPECID_OFFSET, layers - 1).x);
    a.filmIOR = cmk(SampleFilmsSpectrum(matId, wavelengths, FILM_ETA_OFFSET, FILM_ETA_SPECID_OFFSET, 0).x,
                    SampleFilmsSpectrum(matId, wavelengths, FILM_K_OFFSET, FILM_K_SPECID_OFFSET, 0).x);
    if (as_uint_(m.data[FILM_THICKNESS_MAP]) > 0u) {
      const f4 thickness_val = sc.tex_sample(m.texid[2], mulRows2x4(m.row0[2], m.row1[2], tc));
      const float thickness_max = m.data[FILM_THICKNESS_MAX], thickness_min = m.data[FILM_THICKNESS_MIN];
      a.thickness = (thickness_max - thickness_min) * thickness_val.x + thickness_min;
    } else a.thickness = m.data[FILM_THICKNESS];
    a.precomp_flag = as_uint_(m.data[FILM_PRECOMP_FLAG]) > 0u;
    a.precomp_offset = a.precomp_flag ? as_uint_(m.data[FILM_PRECOMP_OFFSET]) : 0;
    return a;
  }
  f3 SpectrumToXYZ(f4 spec4, f4 lambda4, bool terminate_waves) const   // spectrum.h:151-203
  {
    float pdf[4] = { 1.0f / (LAMBDA_MAX - LAMBDA_MIN), 1.0f / (LAMBDA_MAX - LAMBDA_MIN), 1.0f / (LAMBDA_MAX - LAMBDA_MIN), 1.0f / (LAMBDA_MAX - LAMBDA_MIN) };
    const float CIE_Y_integral = 106.856895f;
    const uint nCIESamples = 471;
    float spec[4] = { spec4.x, spec4.y, spec4.z, spec4.w };
    const float* lambda = &lambda4.x;
    if (terminate_waves) { pdf[0] /= 4.0f; for (int i = 1; i < 4; ++i) pdf[i] = 0.0f; }
    for (int i = 0; i < 4; ++i) spec[i] = (pdf[i] != 0) ? spec[i] / pdf[i] : 0.0f;
    float X[4], Y[4], Z[4];
    for (int i = 0; i < 4; ++i) {
      const uint offset = uint(float(std::floor(lambda[i] + 0.5f)) - LAMBDA_MIN);
      const bool out = offset >= nCIESamples || offset >= sc.cieXYZ.size();
      const f4 XYZ = out ? mk4(0, 0, 0, 0) : sc.cieXYZ[offset];
      X[i] = XYZ.x; Y[i] = XYZ.y; Z[i] = XYZ.z;
    }
    for (int i = 0; i < 4; ++i) { X[i] *= spec[i]; Y[i] *= spec[i]; Z[i] *= spec[i]; }
    auto average = [](const float* v) { float sum = v[0]; for (int i = 1; i < 4; ++i) sum += v[i]; return sum / 4.0f; };   // SpectrumAverage (:143-149)
    return mk3(average(X) / CIE_Y_integral, average(Y) / CIE_Y_integral, average(Z) / CIE_Y_integral);
  }
  static f3 XYZToRGB(f3 xyz)                                           // spectrum.h:206-214
  {
    return mk3(+3.240479f * xyz.x - 1.537150f * xyz.y - 0.498535f * xyz.z,
               -0.969256f * xyz.x + 1.875991f * xyz.y + 0.041556f * xyz.z,
               +0.055648f * xyz.x - 0.204043f * xyz.y + 1.057311f * xyz.z);
  }
  f3 SpectralCamRespoceToRGB(f4 specSamples, f4 waves, uint rayFlags) const   // integrator_spectrum.cpp:68-124
  {
    if (sc.camResponseSpectrumId[0] < 0) return XYZToRGB(SpectrumToXYZ(specSamples, waves, (rayFlags & RAY_FLAG_WAVES_DIVERGED) != 0));
    f4 responceX, responceY, responceZ;
    responceX = SampleUniformSpectrum(sc.specOffsetSz[2 * sc.camResponseSpectrumId[0]], waves);
    responceY = sc.camResponseSpectrumId[1] >= 0 ? SampleUniformSpectrum(sc.specOffsetSz[2 * sc.camResponseSpectrumId[1]], waves) : responceX;
    responceZ = sc.camResponseSpectrumId[2] >= 0 ? SampleUniformSpectrum(sc.specOffsetSz[2 * sc.camResponseSpectrumId[2]], waves) : responceY;
    f3 xyz = mk3(0, 0, 0);
    const float* sp = &specSamples.x; const float* rx = &responceX.x; const float* ry = &responceY.x; const float* rz = &responceZ.x;
    for (int i = 0; i < 4; ++i) { xyz.x += sp[i] * rx[i]; xyz.y += sp[i] * ry[i]; xyz.z += sp[i] * rz[i]; }
    return sc.camResponseType == 0u ? XYZToRGB(xyz) : xyz;             // CAM_RESPONCE_XYZ = 0, CAM_RESPONCE_RGB = 1 (integrator_pt.h:531-532)
  }

  f4 LightIntensity(uint a_lightId, f3 a_rayPos, f3 a_rayDir, f4 a_wavelengths = f4{0, 0, 0, 0}) const   // :109-173
  {
    const LightSource& L = sc.lights[a_lightId];
    f4 lightColor = L.intensity;
    if (p.spectralMode != 0 && L.specId < 0xFFFFFFFFu) lightColor = SampleUniformSpectrum(sc.specOffsetSz[2 * L.specId], a_wavelengths);   // :115-123
    lightColor = lightColor * L.mult;
    const uint iesId = L.iesId;
    if (iesId != uint(-1)) {
      if ((L.flags & LIGHT_FLAG_POINT_AREA) != 0) a_rayDir = normalize(xyz(L.pos) - a_rayPos);
      const f3 dirTrans = xyz(mul(L.iesMatrix, xyzw(a_rayDir, 0.0f)));
      float sintheta = 0.0f;
      const f2 texCoord = sphereMapTo2DTexCoord((-1.0f) * dirTrans, &sintheta);
      const f4 texColor = sc.tex_sample(iesId, texCoord);
      lightColor = lightColor * texColor;
    }
    if (L.distType == LIGHT_DIST_SPOT) {
      const float cos1 = L.lightCos1, cos2 = L.lightCos2;
      const f3 norm = xyz(L.norm);
      const float cos_theta = std::max(-dot(a_rayDir, norm), 0.0f);
      lightColor = lightColor * mylocalsmoothstep(cos2, cos1, cos_theta);
      if ((L.flags & LIGHT_FLAG_PROJECTIVE) != 0 && L.texId != uint(-1)) {          // :153-161: a slide projector - the texture through the light's view-projection
        const f4 posLightClipSpace = mul(L.iesMatrix, xyzw(a_rayPos, 1.0f));
        const f3 posLightSpaceNDC = xyz(posLightClipSpace) / posLightClipSpace.w;
        const f2 shadowTexCoord = mk2(posLightSpaceNDC.x * 0.5f + 0.5f, posLightSpaceNDC.y * 0.5f + 0.5f);
        lightColor = lightColor * sc.tex_sample(L.texId, shadowTexCoord);
      }
    }
    else if (L.texId != uint(-1)) {                                   // :163-170: the environment map seen along the shadow ray
      float sintheta = 0.0f;
      const f2 texCoord = sphereMapTo2DTexCoord(a_rayDir, &sintheta);
      const f2 texCoordT = mulRows2x4(L.samplerRow0, L.samplerRow1, texCoord);
      lightColor = lightColor * sc.tex_sample(L.texId, texCoordT);
    }
    return lightColor;
  }

  f4 EnvironmentColor(f3 a_dir, float& outPdf, f4 a_wavelengths = f4{0, 0, 0, 0}) const   // :175-210
  {
    f4 color = p.envColor;
    if (p.spectralMode != 0 && p.envSpecId != uint(-1)) {                    // :181-188: the environment's spectrum
      color = SampleUniformSpectrum(sc.specOffsetSz[2 * p.envSpecId], a_wavelengths);
      color = color * (p.envSpecMult / 106.856895f);
    }
    const uint envTexId = p.envTexId;
    if (envTexId != uint(-1)) {
      float sinTheta = 1.0f;
      const f2 tc = sphereMapTo2DTexCoord(a_dir, &sinTheta);
      const f2 texCoordT = mulRows2x4(p.envSamRow0, p.envSamRow1, tc);
      if (sinTheta != 0.f && p.envEnableSam != 0 && p.integratorType == INTEGRATOR_MIS_PT && p.envLightId != uint(-1)) {
        const LightSource& L = sc.lights[p.envLightId];
        const float mapPdf = evalMap2DPdf(texCoordT, sc.arrays1f.data(), (int)L.pdfTableOffset, (int)L.pdfTableSizeX, (int)L.pdfTableSizeY);
        outPdf = (mapPdf * 1.0f) / (2.f * kPI * kPI * std::max(std::abs(sinTheta), 1e-20f));
      }
      color = color * sc.tex_sample(envTexId, texCoordT);
    }
    return color;
  }

  // ---- materials (integrator_pt_mat.cpp) -----------------------------------------------------------------------------
  f4 fourScalarMatParams(const Material& m, f2 tc) const   // :151-167 / :360-376
  {
    f4 four = mk4(1, 1, 1, 1);
    if ((m.cflags & FLAG_FOUR_TEXTURES) != 0) {
      const f2 tc2 = mulRows2x4(m.row0[2], m.row1[2], tc), tc3 = mulRows2x4(m.row0[3], m.row1[3], tc);
      const f4 color2 = sc.tex_sample(m.texid[2], tc2), color3 = sc.tex_sample(m.texid[3], tc3);
      if ((m.cflags & FLAG_PACK_FOUR_PARAMS_IN_TEXTURE) != 0) four = color2; else four = mk4(color2.x, color3.x, 1, 1);
    }
    return four;
  }

  // NormalMapTransform (include/cmaterial.h) + Integrator::BumpMapping (integrator_pt_mat.cpp:94-107). LiteMath (absent) supplies
  // make_float3x3 / inverse3x3: taken as rows (tan, bitan, n) and the exact inverse by cofactors, inv(M) = [r1 x r2 | r2 x r0 | r0 x r1] / det.
  f3 BumpMapping(const Material& m, f3 n, f3 tan, f2 tc) const
  {
    const f4 normalTex = sc.tex_sample(m.texid[1], mulRows2x4(m.row0[1], m.row1[1], tc));
    f3 ts = mk3(2.0f * normalTex.x - 1.0f, 2.0f * normalTex.y - 1.0f, normalTex.z);
    if ((m.cflags & FLAG_NMAP_INVERT_X) != 0) ts.x *= -1.0f;
    if ((m.cflags & FLAG_NMAP_INVERT_Y) != 0) ts.y *= -1.0f;
    if ((m.cflags & FLAG_NMAP_SWAP_XY) != 0) { const float t = ts.x; ts.x = ts.y; ts.y = t; }
    const f3 bitan = cross(n, tan);
    const f3 c0 = cross(bitan, n), c1 = cross(n, tan), c2 = cross(tan, bitan);
    const float det = dot(tan, c0);
    const f3 w = (ts.x * c0 + ts.y * c1 + ts.z * c2) / det;
    return normalize(w);
  }

  BsdfSample MaterialSampleAndEval(uint a_materialId, RandomGen* a_gen, f3 v, f3 n, f3 tan, f2 tc, MisData* a_misPrev, uint a_currRayFlags, Record* rec, uint bounce, f4 wavelengths = f4{0, 0, 0, 0}) const   // :109-306
  {
    BsdfSample res;
    res.val = mk4(0, 0, 0, 0); res.pdf = 1.0f; res.dir = mk3(0, 1, 0); res.ior = 1.0f; res.flags = a_currRayFlags;
    // blend descent (BlendSampleAndEval, integrator_pt_mat.cpp:23-54, loop :123-130): one generator step per layer, BEFORE the float4
    uint currMatId = a_materialId;
    while (sc.materials[currMatId].mtype == MAT_TYPE_BLEND) {
      const Material& bm = sc.materials[currMatId];
      const f4 weightDat = sc.tex_sample(bm.texid[0], mulRows2x4(bm.row0[0], bm.row1[0], tc));
      const float weight = bm.data[BLEND_WEIGHT] * weightDat.x;
      const float select = rndFloat1(a_gen);           // GetRandomNumbersMatB (integrator_pt.cpp:37)
      if (select < weight) { res.pdf *= weight; res.val = res.val * weight; currMatId = bm.datai[1]; }
      else                 { res.pdf *= 1.0f - weight; res.val = res.val * (1.0f - weight); currMatId = bm.datai[0]; }
    }
    const Material& m = sc.materials[currMatId];
    const uint mtype = m.mtype;
    const uint normalMapId = m.texid[1];                // :131-139: the leaf's normal map bends the shading normal
    const f3 geomNormal = n;
    const f3 shadeNormal = (normalMapId != 0xFFFFFFFFu) ? BumpMapping(m, geomNormal, tan, tc) : n;
    const f2 texCoordT = mulRows2x4(m.row0[0], m.row1[0], tc);
    const f4 texColor = sc.tex_sample(m.texid[0], texCoordT);
    const f4 rands = rndFloat4(a_gen);                  // GetRandomNumbersMats, drawn for every material type (:147)
    if (rec && rec->enabled) rec->mat[bounce] = rands;
    const f4 four = fourScalarMatParams(m, tc);

    switch (mtype) {
      case MAT_TYPE_GLTF: {
        const f4 color = m.colors[GLTF_COLOR_BASE] * texColor;
        BsdfSampleT<f4> r; r.val = res.val; r.dir = res.dir; r.pdf = res.pdf; r.flags = res.flags; r.ior = res.ior;
        gltfSampleAndEval<f4>(m, rands, v, shadeNormal, tc, color, four, &r);
        res.val = r.val; res.dir = r.dir; res.pdf = r.pdf; res.flags = r.flags; res.ior = r.ior;
      } break;
      case MAT_TYPE_CONDUCTOR: {
        const f3 alphaTex = xyz(texColor);
        const f2 alpha = mk2(m.data[CONDUCTOR_ROUGH_V], m.data[CONDUCTOR_ROUGH_U]);
        const f4 etaSpec = SampleMatParamSpectrum(currMatId, wavelengths, CONDUCTOR_ETA, 0), kSpec = SampleMatParamSpectrum(currMatId, wavelengths, CONDUCTOR_K, 1);
        if (trEffectivelySmooth(alpha)) conductorSmoothSampleAndEval(m, etaSpec, kSpec, rands, v, shadeNormal, tc, &res);
        else                            conductorRoughSampleAndEval(m, etaSpec, kSpec, rands, v, shadeNormal, tc, alphaTex, &res);
      } break;
      case MAT_TYPE_DIFFUSE: {
        f4 reflSpec = SampleMatColorSpectrumTexture(currMatId, wavelengths, DIFFUSE_COLOR, 0, tc);      // integrator_pt_mat.cpp:256-260
        if (p.spectralMode == 0) reflSpec = reflSpec * texColor;
        diffuseSampleAndEval(m, reflSpec, rands, v, shadeNormal, tc, &res);
      } break;
      case MAT_TYPE_GLASS: glassSampleAndEval(m, rands, v, geomNormal, &res, &a_misPrev->ior); break;    // integrator_pt_mat.cpp:178-183: the geometric normal
      case MAT_TYPE_PLASTIC: {                                                                     // :270-282 (RGB mode)
        f4 reflSpec = SampleMatColorSpectrumTexture(currMatId, wavelengths, 0, 0, tc);                  // PLASTIC_COLOR (integrator_pt_mat.cpp:268-271)
        if (p.spectralMode == 0) reflSpec = reflSpec * texColor;
        plasticSampleAndEval(m, reflSpec, rands, v, shadeNormal, &res, sc.arrays1f.data(), m.datai[0]);
      } break;
      case MAT_TYPE_DIELECTRIC: {
        const f4 intIORSpec = SampleMatParamSpectrum(currMatId, wavelengths, DIELECTRIC_ETA_INT, 0);   // (integrator_pt_mat.cpp:280)
        const uint specId = m.spdid[0];
        dielectricSmoothSampleAndEval(m, intIORSpec, a_misPrev->ior, rands, v, shadeNormal, tc, &res);
        res.flags |= (specId < 0xFFFFFFFFu) ? RAY_FLAG_WAVES_DIVERGED : 0;
        a_misPrev->ior = res.ior;
      } break;
      case MAT_TYPE_THIN_FILM: {                                                                   // :197-249: the GEOMETRIC normal, always "diverged"
        const f3 alphaTex = xyz(texColor);
        const f2 alpha = mk2(m.data[FILM_ROUGH_V], m.data[FILM_ROUGH_U]);
        const FilmArgs a = filmArgs(currMatId, tc, wavelengths, false);
        if (trEffectivelySmooth(alpha))
          filmSmoothSampleAndEval(m, a.extIOR, a.filmIOR, a.intIOR, a.thickness, a.wavelengths_spec, a_misPrev->ior, rands, v, n, &res, sc.precompThinFilms.data(), a.precomp_offset, a.spectral_mode, a.precomp_flag);
        else
          filmRoughSampleAndEval(m, a.extIOR, a.filmIOR, a.intIOR, a.thickness, a.wavelengths_spec, a_misPrev->ior, rands, v, n, alphaTex, &res, sc.precompThinFilms.data(), a.precomp_offset, a.spectral_mode, a.precomp_flag);
        res.flags |= RAY_FLAG_WAVES_DIVERGED;
        a_misPrev->ior = res.ior;
      } break;
      default: break;
    }
    if (normalMapId != 0xFFFFFFFFu) {                   // :298-303: the caller multiplies by cos to the geometric normal
      const float cosThetaOut1 = std::abs(dot(res.dir, geomNormal));
      const float cosThetaOut2 = std::abs(dot(res.dir, shadeNormal));
      res.val = res.val * (cosThetaOut2 / std::max(cosThetaOut1, 1e-10f));
    }
    return res;
  }

  BsdfEval MaterialEval(uint a_materialId, f3 l, f3 v, f3 n, f3 tan, f2 tc, f4 wavelengths = f4{0, 0, 0, 0}) const   // :308-528
  {
    BsdfEval res; res.val = mk4(0, 0, 0, 0); res.pdf = 0.0f;
    // blend tree walk (:316-333, 511-527): a stack of (material id, weight) pairs, BLEND_STACK_SIZE deep
    struct IdW { uint id; float weight; };
    IdW currMat = { a_materialId, 1.0f };
    IdW stack[BLEND_STACK_SIZE]; stack[0] = currMat;
    int top = 0; bool needPop = false;
    do {
      if (needPop) { top--; currMat = stack[std::max(top, 0)]; } else needPop = true;
      const Material& m = sc.materials[currMat.id];
      f3 shadeNormal = n;
      const float weight = currMat.weight;
      float bumpCosMult = 1.0f;
      if (m.texid[1] != 0xFFFFFFFFu) {                                                // :341-355
        shadeNormal = BumpMapping(m, n, tan, tc);
        const float cosThetaOut1 = std::max(dot(l, n), 0.0f), cosThetaOut2 = std::max(dot(l, shadeNormal), 0.0f);
        bumpCosMult = cosThetaOut2 / std::max(cosThetaOut1, 1e-6f);
        if (cosThetaOut1 <= 0.0f) bumpCosMult = 0.0f;
      }
      const f2 texCoordT = mulRows2x4(m.row0[0], m.row1[0], tc);
      const f4 texColor = sc.tex_sample(m.texid[0], texCoordT);
      const f4 four = fourScalarMatParams(m, tc);
      BsdfEval currVal; currVal.val = mk4(0, 0, 0, 0); currVal.pdf = 0.0f;
      switch (m.mtype) {
        case MAT_TYPE_GLTF: {
          const f4 color = m.colors[GLTF_COLOR_BASE] * texColor;
          BsdfEvalT<f4> r; r.val = currVal.val; r.pdf = currVal.pdf;
          gltfEval<f4>(m, l, v, shadeNormal, tc, color, four, &r);
          res.val = res.val + r.val * weight * bumpCosMult;
          res.pdf += r.pdf * weight;
        } break;
        case MAT_TYPE_CONDUCTOR: {
          const f3 alphaTex = xyz(texColor);
          const f2 alpha = mk2(m.data[CONDUCTOR_ROUGH_V], m.data[CONDUCTOR_ROUGH_U]);
          if (!trEffectivelySmooth(alpha)) {
            const f4 etaSpec = SampleMatParamSpectrum(currMat.id, wavelengths, CONDUCTOR_ETA, 0), kSpec = SampleMatParamSpectrum(currMat.id, wavelengths, CONDUCTOR_K, 1);
            conductorRoughEval(m, etaSpec, kSpec, l, v, shadeNormal, tc, alphaTex, &currVal);
          }
          res.val = res.val + currVal.val * weight * bumpCosMult;
          res.pdf += currVal.pdf * weight;
        } break;
        case MAT_TYPE_DIFFUSE: {
          f4 reflSpec = SampleMatColorSpectrumTexture(currMat.id, wavelengths, DIFFUSE_COLOR, 0, tc);   // integrator_pt_mat.cpp:474-477
          if (p.spectralMode == 0) reflSpec = reflSpec * texColor;
          diffuseEval(m, reflSpec, l, v, shadeNormal, tc, &currVal);
          res.val = res.val + currVal.val * weight * bumpCosMult;
          res.pdf += currVal.pdf * weight;
        } break;
        case MAT_TYPE_PLASTIC: {                                                      // :484-499
          f4 reflSpec = SampleMatColorSpectrumTexture(currMat.id, wavelengths, 0, 0, tc);               // (integrator_pt_mat.cpp:490-493)
          if (p.spectralMode == 0) reflSpec = reflSpec * texColor;
          plasticEval(m, reflSpec, l, v, shadeNormal, &currVal, sc.arrays1f.data(), m.datai[0]);
          res.val = res.val + currVal.val * weight * bumpCosMult;
          res.pdf += currVal.pdf * weight;
        } break;
        case MAT_TYPE_THIN_FILM: {                                                    // :422-470
          const f3 alphaTex = xyz(texColor);
          const f2 alpha = mk2(m.data[FILM_ROUGH_V], m.data[FILM_ROUGH_U]);
          if (!trEffectivelySmooth(alpha)) {
            const FilmArgs a = filmArgs(currMat.id, tc, wavelengths, true);
            filmRoughEval(m, a.extIOR, a.filmIOR, a.intIOR, a.thickness, a.wavelengths_spec, l, v, n, alphaTex, &currVal, sc.precompThinFilms.data(), a.precomp_offset, a.spectral_mode, a.precomp_flag);
          }
          res.val = res.val + currVal.val * weight * bumpCosMult;
          res.pdf += currVal.pdf * weight;
        } break;
        case MAT_TYPE_GLASS:                                                          // cmat_glass.h:281-287: adds zero
        case MAT_TYPE_DIELECTRIC: break;                                              // cmat_dielectric.h:59-63: val and pdf are always zero
        case MAT_TYPE_BLEND: {                                                        // BlendEval (:56-77) + :511-522
          const float w = m.data[BLEND_WEIGHT] * texColor.x;
          const IdW p1 = { m.datai[0], currMat.weight * (1.0f - w) }, p2 = { m.datai[1], currMat.weight * w };
          currMat = p1;
          needPop = false;                                                            // the first child is evaluated on the next trip, without a pop
          if (top + 1 <= (int)BLEND_STACK_SIZE) { stack[top] = p2; top++; }           // the second one waits on the stack
        } break;
        default: break;
      }
    } while (top > 0);
    return res;
  }

  // ---- per-path kernels (integrator_pt.cpp) --------------------------------------------------------------------------
  struct Path
  {
    f4 accumColor, accumThroughput, rayPosAndNear, rayDirAndFar;
    RandomGen gen; MisData mis; uint rayFlags;
    f4 hit1, hit2, hit3; uint instId;
    float time;                                   // motion blur: the path's time in [0, 1] (integrator_pt.cpp:112-115)
    f4 wavelengths = f4{0, 0, 0, 0};              // spectral rendering: the path's four wavelengths (zero in RGB mode, :145-148)
  };

  static inline bool isDeadRay(uint f) { return (f & RAY_FLAG_IS_DEAD) != 0; }
  static inline bool hasNonSpecular(uint f) { return (f & RAY_FLAG_HAS_NON_SPEC) != 0; }
  static inline uint extractMatId(uint f) { return f & 0x00FFFFFFu; }
  static inline uint packMatId(uint f, uint m) { return (f & 0xFF000000u) | (m & 0x00FFFFFFu); }

  // ---- lens simulation (integrator_pt.cpp:806-938, after pbrt's RealisticCamera); m_lines as {curvatureRadius, thickness, eta, apertureRadius} ----
  std::vector<f4> lensLines; f2 physSize = mk2(0, 0);
  static bool Quadratic(float A, float B, float C, float* t0, float* t1)
  {
    const float discrim = B * B - 4.0f * A * C;
    if (discrim < 0.f) return false;
    const float rootDiscrim = std::sqrt(discrim);
    const float q = (B < 0.0f) ? -.5f * (B - rootDiscrim) : -.5f * (B + rootDiscrim);
    *t0 = q / A; *t1 = C / q;
    if (*t0 > *t1) { const float t = *t0; *t0 = *t1; *t1 = t; }
    return true;
  }
  static bool Refract(f3 wi, f3 n, float eta, f3* wt)
  {
    const float cosThetaI = dot(n, wi);
    const float sin2ThetaI = std::max(0.0f, 1.0f - cosThetaI * cosThetaI);
    const float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    const float cosThetaT = std::sqrt(1 - sin2ThetaT);
    *wt = eta * (-1.0f) * wi + (eta * cosThetaI - cosThetaT) * n;
    return true;
  }
  static bool IntersectSphericalElement(float radius, float zCenter, f3 rayPos, f3 rayDir, float* t, f3* n)
  {
    const f3 o = rayPos - mk3(0, 0, zCenter);
    const float A = rayDir.x * rayDir.x + rayDir.y * rayDir.y + rayDir.z * rayDir.z;
    const float B = 2 * (rayDir.x * o.x + rayDir.y * o.y + rayDir.z * o.z);
    const float C = o.x * o.x + o.y * o.y + o.z * o.z - radius * radius;
    float t0, t1;
    if (!Quadratic(A, B, C, &t0, &t1)) return false;
    const bool useCloserT = (rayDir.z > 0.0f) != (radius < 0.0f);
    *t = useCloserT ? std::min(t0, t1) : std::max(t0, t1);
    if (*t < 0.0f) return false;
    *n = normalize(o + (*t) * rayDir);
    *n = (dot(*n, -1.0f * rayDir) < 0.f) ? (-1.0f) * (*n) : *n;       // faceforward
    return true;
  }
  bool TraceLensesFromFilm(f3& inoutRayPos, f3& inoutRayDir) const
  {
    float elementZ = 0;
    f3 p = mk3(inoutRayPos.x, inoutRayPos.y, -inoutRayPos.z), d = mk3(inoutRayDir.x, inoutRayDir.y, -inoutRayDir.z);
    for (size_t i = 0; i < lensLines.size(); i++) {
      const f4 e = lensLines[i];
      elementZ -= e.y;
      float t; f3 n = mk3(0, 0, 0);
      const bool isStop = (e.x == 0.0f);
      if (isStop) {
        if (d.z >= 0.0f) return false;
        t = (elementZ - p.z) / d.z;
      } else if (!IntersectSphericalElement(e.x, elementZ + e.x, p, d, &t, &n)) return false;
      const f3 pHit = p + t * d;
      if (pHit.x * pHit.x + pHit.y * pHit.y > e.w * e.w) return false;
      p = pHit;
      if (!isStop) {
        const float etaI = e.z;
        float etaT = (i == lensLines.size() - 1) ? 1.0f : lensLines[i + 1].z;
        if (etaT == 0.0f) etaT = 1.0f;
        f3 wt;
        if (!Refract(normalize((-1.0f) * d), n, etaI / etaT, &wt)) return false;
        d = wt;
      }
    }
    inoutRayPos = mk3(p.x, p.y, -p.z); inoutRayDir = mk3(d.x, d.y, -d.z);
    return true;
  }

  // SampleCameraRay + kernel_InitEyeRay2 (:44-157); the camera part is shared with the DR replay
  void CameraRay(uint tid, f4 pixelOffsets, f4* rayPosAndNear, f4* rayDirAndFar) const
  {
    const uint XY = packedXY[tid];
    const uint x = (XY & 0x0000FFFFu), y = (XY & 0xFFFF0000u) >> 16;
    const float fx = float(x) + pixelOffsets.x, fy = float(y) + pixelOffsets.y;
    const float xCoordNormalized = (fx + float(p.winStartX)) / float(p.fbWidth);
    const float yCoordNormalized = (fy + float(p.winStartY)) / float(p.fbHeight);
    f3 rayDir = EyeRayDirNormalized(xCoordNormalized, yCoordNormalized, p.projInv);
    f3 rayPos = mk3(0, 0, 0);
    if (p.camLensRadius > 0.0f) {
      const float tFocus = p.camTargetDist / (-rayDir.z);
      const f3 focusPosition = rayPos + rayDir * tFocus;
      const f2 xy = p.camLensRadius * 2.0f * MapSamplesToDisc(mk2(pixelOffsets.z - 0.5f, pixelOffsets.w - 0.5f));
      rayPos.x += xy.x; rayPos.y += xy.y;
      rayDir = normalize(focusPosition - rayPos);
    }
    else if (!lensLines.empty()) {                      // m_enableOpticSim (:79-103)
      rayPos = mk3(0.25f * physSize.x * (2.0f * xCoordNormalized - 1.0f), 0.25f * physSize.y * (2.0f * yCoordNormalized - 1.0f), 0.0f);
      const f2 rareSam = (lensLines[0].w * 2.0f) * MapSamplesToDisc(mk2(pixelOffsets.z - 0.5f, pixelOffsets.w - 0.5f));
      rayDir = normalize(mk3(rareSam.x, rareSam.y, lensLines[0].y) - rayPos);
      if (!TraceLensesFromFilm(rayPos, rayDir)) { rayPos = mk3(0, -10000000.0f, 0.0f); rayDir = mk3(0, -1, 0); }
      else { rayDir = (-1.0f) * normalize(rayDir); rayPos = (-1.0f) * rayPos; }
    }
    transform_ray3f(p.worldViewInv, &rayPos, &rayDir);
    *rayPosAndNear = xyzw(rayPos, 0.0f);
    *rayDirAndFar = xyzw(rayDir, FLT_MAX);
  }

  void InitEyeRay2(uint tid, Path* s, Record* rec) const
  {
    s->accumColor = mk4(0, 0, 0, 0);
    s->accumThroughput = mk4(1, 1, 1, 1);
    RandomGen genLocal = randomGens[tid];
    s->rayFlags = 0;
    s->mis = makeInitialMisData();
    const f4 pixelOffsets = rndFloat4(&genLocal);      // GetRandomNumbersLens; no time / wavelength draws in RGB, static scenes
    if (rec && rec->enabled) rec->lens = pixelOffsets;
    CameraRay(tid, pixelOffsets, &s->rayPosAndNear, &s->rayDirAndFar);
    s->time = sc.motion ? rndFloat1(&genLocal) : 0.0f;  // GetRandomNumbersTime: only when m_normMatrices2Offs != 0 (:114-115)
    s->wavelengths = p.spectralMode != 0 ? SampleWavelengths(rndFloat1(&genLocal), LAMBDA_MIN, LAMBDA_MAX) : mk4(0, 0, 0, 0);   // GetRandomNumbersSpec (:116-118, 145-148)
    s->gen = genLocal;
  }

  // shading data of a hit, shared by kernel_RayTrace2 (:238-311) and PathTraceReplay (integrator_dr.cpp:843-892)
  void SurfaceFromHit(const orc_hit& hit, f4 rayPos, f4 rayDir, Path* s) const
  {
    uint currRayFlags = s->rayFlags;
    const uint geomId = hit.geomId;
    const uint triOffset = sc.matVertOffset[2 * geomId + 0], vertOffset = sc.matVertOffset[2 * geomId + 1];
    const f3 hitPos = xyz(rayPos) + hit.t * (1.f - 1e-6f) * xyz(rayDir);
    const f2 uv = mk2(hit.coords[0], hit.coords[1]);
    const uint A = sc.triIndices[(triOffset + hit.primId) * 3 + 0];
    const uint B = sc.triIndices[(triOffset + hit.primId) * 3 + 1];
    const uint C = sc.triIndices[(triOffset + hit.primId) * 3 + 2];
    const f4 data1 = (1.0f - uv.x - uv.y) * sc.vData8f[2 * (A + vertOffset)] + uv.y * sc.vData8f[2 * (B + vertOffset)] + uv.x * sc.vData8f[2 * (C + vertOffset)];
    const f4 data2 = (1.0f - uv.x - uv.y) * sc.vData8f[2 * (A + vertOffset) + 1] + uv.y * sc.vData8f[2 * (B + vertOffset) + 1] + uv.x * sc.vData8f[2 * (C + vertOffset) + 1];
    const f2 hitTexCoord = mk2(data1.w, data2.w);
    f3 hitNorm = mul3x3(sc.normMatrices[hit.instId], xyz(data1));
    f3 hitTang = mul3x3(sc.normMatrices[hit.instId], xyz(data2));
    if (sc.motion && !std::getenv("ORC_DBG_NO_NORMAL_LERP")) {   // :285-292: the end-of-motion matrix applied to the already transformed vectors, then lerp
      const f3 hitNorm2 = mul3x3(sc.normMatrices2[hit.instId], hitNorm), hitTang2 = mul3x3(sc.normMatrices2[hit.instId], hitTang);
      hitNorm = hitNorm + s->time * (hitNorm2 - hitNorm);
      hitTang = hitTang + s->time * (hitTang2 - hitTang);
    }
    hitNorm = normalize(hitNorm);
    hitTang = normalize(hitTang);
    const float flipNorm = dot(xyz(rayDir), hitNorm) > 0.001f ? -1.0f : 1.0f;
    hitNorm = flipNorm * hitNorm;
    hitTang = flipNorm * hitTang;
    if (flipNorm < 0.0f) currRayFlags |= RAY_FLAG_HAS_INV_NORMAL; else currRayFlags &= ~RAY_FLAG_HAS_INV_NORMAL;
    const uint midOriginal = sc.matIdByPrimId[triOffset + hit.primId];
    const uint midRemaped = RemapMaterialId(midOriginal, (int)hit.instId);
    s->rayFlags = packMatId(currRayFlags, midRemaped);
    s->hit1 = xyzw(hitPos, hitTexCoord.x);
    s->hit2 = xyzw(hitNorm, hitTexCoord.y);
    s->hit3 = xyzw(hitTang, hit.t);
    s->instId = hit.instId;
  }

  mutable long dbgCur = -1;                                  // debugging aid: ORC_DBG_TID=<tid> prints the rays of that thread (single-threaded runs)
  long dbgTid = std::getenv("ORC_DBG_TID") ? std::atol(std::getenv("ORC_DBG_TID")) : -1;
  void RayTrace2(uint bounce, Path* s, Record* rec) const   // :214-348
  {
    if (isDeadRay(s->rayFlags)) return;
    const orc_hit hit = sc.nearest_hit(s->rayPosAndNear, s->rayDirAndFar, false, s->time);
    if (dbgTid >= 0 && dbgCur == dbgTid)
      std::fprintf(stderr, "[orc dbg] bounce %u o %.9g %.9g %.9g d %.9g %.9g %.9g time %.9g -> inst %u prim %u t %.9g\n", bounce, s->rayPosAndNear.x, s->rayPosAndNear.y, s->rayPosAndNear.z,
                   s->rayDirAndFar.x, s->rayDirAndFar.y, s->rayDirAndFar.z, s->time, hit.instId, hit.primId, hit.t);
    if (rec && rec->enabled) rec->hit[bounce] = hit;
    if (hit.geomId != uint(-1)) SurfaceFromHit(hit, s->rayPosAndNear, s->rayDirAndFar, s);
    else {
      const uint flagsToAdd = (bounce == 0) ? (RAY_FLAG_PRIME_RAY_MISS | RAY_FLAG_IS_DEAD | RAY_FLAG_OUT_OF_SCENE) : (RAY_FLAG_IS_DEAD | RAY_FLAG_OUT_OF_SCENE);
      s->rayFlags = s->rayFlags | flagsToAdd;
    }
  }

  f4 SampleLightSource(uint bounce, Path* s, Record* rec) const   // :350-424
  {
    f4 shade = mk4(0, 0, 0, 0);
    const uint currRayFlags = s->rayFlags;
    if (isDeadRay(currRayFlags)) return shade;
    const uint matId = extractMatId(currRayFlags);
    const f3 ray_dir = xyz(s->rayDirAndFar);
    const f3 hpos = xyz(s->hit1), hnorm = xyz(s->hit2), htang = xyz(s->hit3);
    const f2 huv = mk2(s->hit1.w, s->hit2.w);

    const float rndId = rndFloat1(&s->gen);              // GetRandomNumbersLgts (:30-35): two generator steps, in this order
    const f4 r4 = rndFloat4(&s->gen);
    const f4 rands = mk4(r4.x, r4.y, r4.z, rndId);
    const int nLights = (int)sc.lights.size();
    const int lightId = std::min(int(std::floor(rands.w * float(nLights))), nLights - 1);
    if (rec && rec->enabled) rec->lgt[bounce] = rands;
    if (lightId < 0) return shade;

    const LightSample lSam = LightSampleRev(lightId, xyz(rands), hpos);
    const float hitDist = std::sqrt(dot(hpos - lSam.pos, hpos - lSam.pos));
    const f3 shadowRayDir = normalize(lSam.pos - hpos);
    const f3 shadowRayPos = hpos + hnorm * std::max(maxcomp(hpos), 1.0f) * 5e-6f;
    const bool inIllumArea = (dot(shadowRayDir, lSam.norm) < 0.0f) || lSam.isOmni || lSam.hasIES;
    const bool needShade = inIllumArea && !sc.any_hit(xyzw(shadowRayPos, 0.0f), xyzw(shadowRayDir, hitDist * 0.9995f), false, s->time);
    if (rec && rec->enabled) rec->inShadow[bounce] = needShade ? 0 : 1;
    if (needShade) {
      const BsdfEval bsdfV = MaterialEval(matId, shadowRayDir, (-1.0f) * ray_dir, hnorm, htang, huv, s->wavelengths);
      const float cosThetaOut = std::max(dot(shadowRayDir, hnorm), 0.0f);
      float lgtPdfW = LightPdfSelectRev(lightId) * LightEvalPDF(lightId, shadowRayPos, shadowRayDir, lSam.pos, lSam.norm, lSam.pdf);
      float misWeight = (p.integratorType == INTEGRATOR_MIS_PT) ? misWeightHeuristic(lgtPdfW, bsdfV.pdf) : 1.0f;
      const bool isDirect = (sc.lights[lightId].geomType == LIGHT_GEOM_DIRECT);
      const bool isPoint = (sc.lights[lightId].geomType == LIGHT_GEOM_POINT);
      if (isDirect) { misWeight = 1.0f; lgtPdfW = 1.0f; }
      else if (isPoint) misWeight = 1.0f;
      const bool isDirectLight = !hasNonSpecular(currRayFlags);
      if ((p.renderLayer == FB_DIRECT && !isDirectLight) || (p.renderLayer == FB_INDIRECT && isDirectLight)) misWeight = 0.0f;
      const f4 lightColor = LightIntensity(lightId, shadowRayPos, shadowRayDir, s->wavelengths);
      shade = (lightColor * bsdfV.val / lgtPdfW) * cosThetaOut * misWeight;
    }
    return shade;
  }

  void NextBounce(uint bounce, f4 shadeColor, Path* s, Record* rec) const   // :426-548
  {
    const uint currRayFlags = s->rayFlags;
    if (isDeadRay(currRayFlags)) return;
    const uint matId = extractMatId(currRayFlags);
    const f3 ray_dir = xyz(s->rayDirAndFar), ray_pos = xyz(s->rayPosAndNear);
    f3 hpos = xyz(s->hit1);
    const f3 hnorm = xyz(s->hit2), htang = xyz(s->hit3);
    const f2 huv = mk2(s->hit1.w, s->hit2.w);
    const float hitDist = s->hit3.w;
    const float prevPdfW = s->mis.matSamplePdf;
    const Material& m = sc.materials[matId];

    if (m.mtype == MAT_TYPE_LIGHT_SOURCE) {
      const f2 texCoordT = mulRows2x4(m.row0[0], m.row1[0], huv);
      const f4 texColor = sc.tex_sample(m.texid[0], texCoordT);
      const uint lightId = (uint)sc.remapInst[2 * s->instId + 1];
      const f4 emissColor = m.colors[EMISSION_COLOR];
      f4 lightIntensity = emissColor * texColor;
      if (lightId != 0xFFFFFFFFu) {
        const float lightCos = dot(ray_dir, xyz(sc.lights[lightId].norm));
        const float lightDirectionAtten = (lightCos < 0.0f || sc.lights[lightId].geomType == LIGHT_GEOM_SPHERE) ? 1.0f : 0.0f;
        lightIntensity = LightIntensity(lightId, ray_pos, ray_dir, s->wavelengths) * lightDirectionAtten;
      }
      float misWeight = 1.0f;
      if (p.integratorType == INTEGRATOR_MIS_PT) {
        if (bounce > 0 && lightId != 0xFFFFFFFFu) {
          const float lgtPdf = LightPdfSelectRev(lightId) * LightEvalPDF(lightId, ray_pos, ray_dir, hpos, hnorm, 1.0f);
          misWeight = misWeightHeuristic(prevPdfW, lgtPdf);
          if (prevPdfW <= 0.0f) misWeight = 1.0f;
        }
      } else if (p.integratorType == INTEGRATOR_SHADOW_PT && hasNonSpecular(currRayFlags)) misWeight = 0.0f;
      const bool isDirectLight = !hasNonSpecular(currRayFlags);
      const bool isFirstNonSpec = (currRayFlags & RAY_FLAG_FIRST_NON_SPEC) != 0;
      if (p.renderLayer == FB_INDIRECT && (isDirectLight || isFirstNonSpec)) misWeight = 0.0f;
      s->accumColor = s->accumColor + s->accumThroughput * lightIntensity * misWeight;
      s->rayFlags = currRayFlags | (RAY_FLAG_IS_DEAD | RAY_FLAG_HIT_LIGHT);
      return;
    }

    const BsdfSample matSam = MaterialSampleAndEval(matId, &s->gen, (-1.0f) * ray_dir, hnorm, htang, huv, &s->mis, currRayFlags, rec, bounce, s->wavelengths);
    const f4 bxdfVal = matSam.val * (1.0f / std::max(matSam.pdf, 1e-20f));
    const float cosTheta = std::abs(dot(matSam.dir, hnorm));
    s->mis.matSamplePdf = (matSam.flags & RAY_EVENT_S) != 0 ? -1.0f : matSam.pdf;
    s->mis.cosTheta = cosTheta;
    if (p.integratorType == INTEGRATOR_STUPID_PT) s->accumThroughput = s->accumThroughput * (cosTheta * bxdfVal);
    else {
      const f4 currThoroughput = s->accumThroughput;
      s->accumColor = s->accumColor + currThoroughput * shadeColor;
      s->accumThroughput = currThoroughput * cosTheta * bxdfVal;
    }
    if ((matSam.flags & RAY_EVENT_T) != 0) hpos = hpos + hitDist * ray_dir * 2.0f * 1e-6f;
    s->rayPosAndNear = xyzw(OffsRayPos(hpos, hnorm, matSam.dir), 0.0f);
    s->rayDirAndFar = xyzw(matSam.dir, FLT_MAX);
    uint nextFlags = ((currRayFlags & ~RAY_FLAG_FIRST_NON_SPEC) | matSam.flags);
    if (p.renderLayer == FB_DIRECT && hasNonSpecular(currRayFlags)) nextFlags |= RAY_FLAG_IS_DEAD;
    else if (!hasNonSpecular(currRayFlags) && hasNonSpecular(nextFlags)) nextFlags |= RAY_FLAG_FIRST_NON_SPEC;
    s->rayFlags = nextFlags;
  }

  void HitEnvironment(uint tid, Path* s) const   // :550-595
  {
    if ((s->rayFlags & RAY_FLAG_OUT_OF_SCENE) == 0) return;
    float envPdf = 1.0f;
    f4 envColor = EnvironmentColor(xyz(s->rayDirAndFar), envPdf, s->wavelengths);
    const bool isSpec = s->mis.matSamplePdf < 0.0f;                    // isSpecular (cglobals.h:300)
    const bool exitZero = (s->rayFlags & RAY_FLAG_PRIME_RAY_MISS) != 0;
    if (p.integratorType == INTEGRATOR_MIS_PT && p.envEnableSam != 0 && !isSpec && !exitZero) {
      const float lgtPdf = LightPdfSelectRev((int)p.envLightId) * envPdf;
      const float bsdfPdf = s->mis.matSamplePdf;
      envColor = envColor * misWeightHeuristic(bsdfPdf, lgtPdf);
    }
    else if (p.integratorType == INTEGRATOR_SHADOW_PT && p.envEnableSam != 0) envColor = mk4(0, 0, 0, 0);
    if (exitZero && p.envCamBackId != uint(-1)) {                      // the camera back plate for primary rays that miss
      const uint XY = tid < packedXY.size() ? packedXY[tid] : 0u;
      const uint x = (XY & 0x0000FFFF), y = (XY & 0xFFFF0000) >> 16;
      envColor = sc.tex_sample(p.envCamBackId, mk2((float(x) + 0.5f) / float(p.winWidth), (float(y) + 0.5f) / float(p.winHeight)));
    }
    if (p.integratorType == INTEGRATOR_STUPID_PT) s->accumColor = s->accumThroughput * envColor;
    else                                          s->accumColor = s->accumColor + s->accumThroughput * envColor;
  }

  void ContributeToImage(uint tid, uint channels, const Path* s, float* out_color, bool disableImageContrib)   // :598-657
  {
    randomGens[tid] = s->gen;
    if (disableImageContrib) return;
    const uint XY = packedXY[tid];
    const uint x = (XY & 0x0000FFFFu), y = (XY & 0xFFFF0000u) >> 16;
    const f4 tmpVal = s->accumColor * p.camRespoceRGB;
    f3 rgb = xyz(tmpVal);
    if (p.spectralMode != 0) rgb = SpectralCamRespoceToRGB(s->accumColor, s->wavelengths, s->rayFlags);   // :618-625
    const f4 colorRes = p.exposureMult * mk4(rgb.x, rgb.y, rgb.z, 1.0f);
    if (channels == 1) out_color[y * p.winWidth + x] += s->accumColor.x * p.exposureMult;
    else if (channels > 4) {                         // "always spectral rendering" (:642-654): one W x H layer per wavelength bin
      const f4 color = s->accumColor * p.exposureMult;
      const float* waves = &s->wavelengths.x; const float* cv = &color.x;
      for (int i = 0; i < 4; i++) {
        const float t = (waves[i] - LAMBDA_MIN) / (LAMBDA_MAX - LAMBDA_MIN);
        const int channelId = std::min(int(float(channels) * t), int(channels) - 1);
        out_color[(size_t)channelId * p.winWidth * p.winHeight + (size_t)y * p.winWidth + x] += cv[i];
      }
    }
    else {
      out_color[(y * p.winWidth + x) * channels + 0] += colorRes.x;
      out_color[(y * p.winWidth + x) * channels + 1] += colorRes.y;
      out_color[(y * p.winWidth + x) * channels + 2] += colorRes.z;
    }
  }

  f4 PathTrace(uint tid, uint channels, float* out_color, Record* rec, bool disableImageContrib)   // :719-759
  {
    Path s;
    InitEyeRay2(tid, &s, rec);
    dbgCur = (long)tid;
    for (uint depth = 0; depth < p.traceDepth; depth++) {
      RayTrace2(depth, &s, rec);
      if (isDeadRay(s.rayFlags)) break;
      const f4 shadeColor = SampleLightSource(depth, &s, rec);
      NextBounce(depth, shadeColor, &s, rec);
      if (isDeadRay(s.rayFlags)) break;
    }
    HitEnvironment(tid, &s);
    ContributeToImage(tid, channels, &s, out_color, disableImageContrib);
    return s.accumColor;
  }

  // PathTraceFromInputRays (integrator_pt.cpp:761-798): kernel_InitEyeRayFromInput (:159-199, camera-space ray through m_worldViewInv,
  // no lens randoms) + the PathTrace loop + kernel_CopyColorToOutput (:659-676: out_color[tid * channels ..] += accumColor, raw)
  void PathTraceFromInputRays(uint tid, uint channels, const float* rayPosAndW, const float* rayDirAndT, float* out_color)
  {
    Path s;
    s.accumColor = mk4(0, 0, 0, 0);
    s.accumThroughput = mk4(1, 1, 1, 1);
    s.rayFlags = 0;
    s.mis = makeInitialMisData();
    f3 rayPos = mk3(rayPosAndW[4 * tid + 0], rayPosAndW[4 * tid + 1], rayPosAndW[4 * tid + 2]);
    f3 rayDir = mk3(rayDirAndT[4 * tid + 0], rayDirAndT[4 * tid + 1], rayDirAndT[4 * tid + 2]);
    transform_ray3f(p.worldViewInv, &rayPos, &rayDir);
    s.rayPosAndNear = xyzw(rayPos, 0.0f);
    s.rayDirAndFar = xyzw(rayDir, FLT_MAX);
    s.time = sc.motion ? rayDirAndT[4 * tid + 3] : 0.0f;  // *time = rayDirData.time (:197)
    s.wavelengths = p.spectralMode != 0 ? splat4(rayPosAndW[4 * tid + 3]) : mk4(0, 0, 0, 0);   // *wavelengths = float4(rayPosData.wave) (:181-193)
    s.gen = randomGens[tid];
    for (uint depth = 0; depth < p.traceDepth; depth++) {
      RayTrace2(depth, &s, nullptr);
      if (isDeadRay(s.rayFlags)) break;
      const f4 shadeColor = SampleLightSource(depth, &s, nullptr);
      NextBounce(depth, shadeColor, &s, nullptr);
      if (isDeadRay(s.rayFlags)) break;
    }
    HitEnvironment(tid, &s);
    if (channels == 1) out_color[tid] += s.accumColor.x;
    else { out_color[tid * channels + 0] += s.accumColor.x; out_color[tid * channels + 1] += s.accumColor.y; out_color[tid * channels + 2] += s.accumColor.z; }
    if (channels == 4 && p.spectralMode != 0) out_color[tid * 4 + 3] += s.accumColor.w;   // kernel_CopyColorToOutput adds all four (:664-670): the fourth is a wavelength's sample here
    randomGens[tid] = s.gen;
  }

  void NaivePathTrace(uint tid, uint channels, float* out_color)   // :681-717
  {
    Path s;
    InitEyeRay2(tid, &s, nullptr);
    for (uint depth = 0; depth < p.traceDepth + 1; ++depth) {
      RayTrace2(depth, &s, nullptr);
      if (isDeadRay(s.rayFlags)) break;
      NextBounce(depth, mk4(0, 0, 0, 0), &s, nullptr);
      if (isDeadRay(s.rayFlags)) break;
    }
    HitEnvironment(tid, &s);
    ContributeToImage(tid, channels, &s, out_color, false);
  }

  // ---- differentiable rendering (diff_render/integrator_dr.cpp) ---------------------------------------------------------
  // Tex2DFetchAD (:95-161): fetch from the flat parameter vector when the texture is registered, else the ordinary sampler.
  // Returns a dual colour whose derivative slot `bounce` is the identity (d texColor / d texColor).
  AF4 Tex2DFetchAD(uint texId, f2 uv, const float* tex_data, int bounce, Scene::Taps* tapsOut, bool* isParam) const
  {
    const TexInfo& info = texAddressTable[texId];
    *isParam = false;
    if (info.offset != size_t(-1) && tex_data != nullptr) {
      const Texture& geo = sc.textures[texId];   // address modes come from the bound sampler (:110-113); size from the table
      Texture t; t.w = (uint)info.width; t.h = (uint)info.height; t.addrU = geo.addrU; t.addrV = geo.addrV;
      const Scene::Taps k = Scene::bilinear_taps(t, uv);
      *tapsOut = k; *isParam = true;
      f4 r;
      if (info.channels == 4) {
        const float* d = tex_data + info.offset;
        r.x = d[k.off[0] * 4 + 0] * k.w[0] + d[k.off[1] * 4 + 0] * k.w[1] + d[k.off[2] * 4 + 0] * k.w[2] + d[k.off[3] * 4 + 0] * k.w[3];
        r.y = d[k.off[0] * 4 + 1] * k.w[0] + d[k.off[1] * 4 + 1] * k.w[1] + d[k.off[2] * 4 + 1] * k.w[2] + d[k.off[3] * 4 + 1] * k.w[3];
        r.z = d[k.off[0] * 4 + 2] * k.w[0] + d[k.off[1] * 4 + 2] * k.w[1] + d[k.off[2] * 4 + 2] * k.w[2] + d[k.off[3] * 4 + 2] * k.w[3];
        r.w = d[k.off[0] * 4 + 3] * k.w[0] + d[k.off[1] * 4 + 3] * k.w[1] + d[k.off[2] * 4 + 3] * k.w[2] + d[k.off[3] * 4 + 3] * k.w[3];
      } else {
        const float* d = tex_data + info.offset;
        const float o = d[k.off[0]] * k.w[0] + d[k.off[1]] * k.w[1] + d[k.off[2]] * k.w[2] + d[k.off[3]] * k.w[3];
        r = mk4(o, o, o, o);
      }
      AF4 a(r);
      a.d[bounce] = splat4(1.0f);
      return a;
    }
    return AF4(sc.tex_sample(texId, uv));
  }

  struct ReplayOut { AF4 color; Scene::Taps taps[MAXB]; uint texId[MAXB]; bool isParam[MAXB]; };

  // PathTraceReplay (:790-1101): re-shade a recorded path; no BVH queries, no RNG. Only MAT_TYPE_GLTF is
  // differentiable / supported on this path (DR-specific MaterialSampleAndEval / MaterialEval, :461-612).
  void PathTraceReplay(uint tid, const Record& rec, const float* dparams, ReplayOut* out) const
  {
    AF4 accumColor(splat4(0.0f)), accumThroughput(splat4(1.0f));
    Path s; s.mis = makeInitialMisData(); s.rayFlags = 0;
    for (int b = 0; b < MAXB; b++) { out->isParam[b] = false; out->texId[b] = 0; }
    CameraRay(tid, rec.lens, &s.rayPosAndNear, &s.rayDirAndFar);

    for (uint bounce = 0; bounce < p.traceDepth; bounce++) {
      if (!isDeadRay(s.rayFlags)) {
        const orc_hit hit = rec.hit[bounce];
        if (hit.geomId != uint(-1)) SurfaceFromHit(hit, s.rayPosAndNear, s.rayDirAndFar, &s);
        else {
          const uint flagsToAdd = (bounce == 0) ? (RAY_FLAG_PRIME_RAY_MISS | RAY_FLAG_IS_DEAD | RAY_FLAG_OUT_OF_SCENE) : (RAY_FLAG_IS_DEAD | RAY_FLAG_OUT_OF_SCENE);
          s.rayFlags |= flagsToAdd;
        }
      }
      AF4 shadeColor(splat4(0.0f));
      if (!isDeadRay(s.rayFlags)) {        // kernel_SampleLightSource replayed (:903-956)
        const uint matId = extractMatId(s.rayFlags);
        const f3 ray_dir = xyz(s.rayDirAndFar);
        const f3 hpos = xyz(s.hit1), hnorm = xyz(s.hit2);
        const f2 huv = mk2(s.hit1.w, s.hit2.w);
        const f4 rands = rec.lgt[bounce];
        const int nLights = (int)sc.lights.size();
        const int lightId = std::min(int(std::floor(rands.w * float(nLights))), nLights - 1);
        if (lightId >= 0) {
          const LightSample lSam = LightSampleRev(lightId, xyz(rands), hpos);
          const f3 shadowRayDir = normalize(lSam.pos - hpos);
          const f3 shadowRayPos = hpos + hnorm * std::max(maxcomp(hpos), 1.0f) * 5e-6f;
          const bool inIllumArea = (dot(shadowRayDir, lSam.norm) < 0.0f) || lSam.isOmni || lSam.hasIES;
          const bool needShade = inIllumArea && (rec.inShadow[bounce] == 0);
          if (needShade) {
            const Material& m = sc.materials[matId];
            BsdfEvalT<AF4> bsdfV; bsdfV.val = AF4(splat4(0)); bsdfV.pdf = 0.0f;
            if (m.mtype == MAT_TYPE_GLTF) {
              const f2 texCoordT = mulRows2x4(m.row0[0], m.row1[0], huv);
              const AF4 texColor = Tex2DFetchAD(m.texid[0], texCoordT, dparams, (int)bounce, &out->taps[bounce], &out->isParam[bounce]);
              out->texId[bounce] = m.texid[0];
              const AF4 color = texColor * m.colors[GLTF_COLOR_BASE];
              BsdfEvalT<AF4> r; r.val = AF4(splat4(0)); r.pdf = 0.0f;
              gltfEval<AF4>(m, shadowRayDir, (-1.0f) * ray_dir, hnorm, huv, color, mk4(1, 1, 1, 1), &r);
              bsdfV.val = r.val; bsdfV.pdf = r.pdf;
            }
            const float cosThetaOut = std::max(dot(shadowRayDir, hnorm), 0.0f);
            float lgtPdfW = LightPdfSelectRev(lightId) * LightEvalPDF(lightId, shadowRayPos, shadowRayDir, lSam.pos, lSam.norm, lSam.pdf);
            float misWeight = (p.integratorType == INTEGRATOR_MIS_PT) ? misWeightHeuristic(lgtPdfW, bsdfV.pdf) : 1.0f;
            const bool isDirect = (sc.lights[lightId].geomType == LIGHT_GEOM_DIRECT);
            const bool isPoint = (sc.lights[lightId].geomType == LIGHT_GEOM_POINT);
            if (isDirect) { misWeight = 1.0f; lgtPdfW = 1.0f; }
            else if (isPoint) misWeight = 1.0f;
            const bool isDirectLight = !hasNonSpecular(s.rayFlags);
            if ((p.renderLayer == FB_DIRECT && !isDirectLight) || (p.renderLayer == FB_INDIRECT && isDirectLight)) misWeight = 0.0f;
            const f4 lightColor = LightIntensity(lightId, shadowRayPos, shadowRayDir);
            shadeColor = ((bsdfV.val * lightColor) / lgtPdfW) * cosThetaOut * misWeight;
          }
        }
      }
      if (!isDeadRay(s.rayFlags)) {        // kernel_NextBounce replayed (:964-1072)
        const uint matId = extractMatId(s.rayFlags);
        const f3 ray_dir = xyz(s.rayDirAndFar), ray_pos = xyz(s.rayPosAndNear);
        f3 hpos = xyz(s.hit1);
        const f3 hnorm = xyz(s.hit2);
        const f2 huv = mk2(s.hit1.w, s.hit2.w);
        const float hitDist = s.hit3.w;
        const float prevPdfW = s.mis.matSamplePdf;
        const Material& m = sc.materials[matId];
        if (m.mtype == MAT_TYPE_LIGHT_SOURCE) {
          const f2 texCoordT = mulRows2x4(m.row0[0], m.row1[0], huv);
          const f4 texColor = sc.tex_sample(m.texid[0], texCoordT);
          const uint lightId = (uint)sc.remapInst[2 * s.instId + 1];
          f4 lightIntensity = m.colors[EMISSION_COLOR] * texColor;
          if (lightId != 0xFFFFFFFFu) {
            const float lightCos = dot(ray_dir, xyz(sc.lights[lightId].norm));
            const float atten = (lightCos < 0.0f || sc.lights[lightId].geomType == LIGHT_GEOM_SPHERE) ? 1.0f : 0.0f;
            lightIntensity = LightIntensity(lightId, ray_pos, ray_dir) * atten;
          }
          float misWeight = 1.0f;
          if (p.integratorType == INTEGRATOR_MIS_PT) {
            if (bounce > 0 && lightId != 0xFFFFFFFFu) {
              const float lgtPdf = LightPdfSelectRev(lightId) * LightEvalPDF(lightId, ray_pos, ray_dir, hpos, hnorm, 1.0f);
              misWeight = misWeightHeuristic(prevPdfW, lgtPdf);
              if (prevPdfW <= 0.0f) misWeight = 1.0f;
            }
          } else if (p.integratorType == INTEGRATOR_SHADOW_PT && hasNonSpecular(s.rayFlags)) misWeight = 0.0f;
          const bool isDirectLight = !hasNonSpecular(s.rayFlags);
          const bool isFirstNonSpec = (s.rayFlags & RAY_FLAG_FIRST_NON_SPEC) != 0;
          if (p.renderLayer == FB_INDIRECT && (isDirectLight || isFirstNonSpec)) misWeight = 0.0f;
          accumColor = accumColor + (accumThroughput * lightIntensity) * misWeight;
          s.rayFlags |= (RAY_FLAG_IS_DEAD | RAY_FLAG_HIT_LIGHT);
        } else {
          BsdfSampleT<AF4> matSam; matSam.val = AF4(splat4(0)); matSam.pdf = 1.0f; matSam.dir = mk3(0, 1, 0); matSam.ior = 1.0f; matSam.flags = s.rayFlags;
          if (m.mtype == MAT_TYPE_GLTF) {
            const f2 texCoordT = mulRows2x4(m.row0[0], m.row1[0], huv);
            const AF4 texColor = Tex2DFetchAD(m.texid[0], texCoordT, dparams, (int)bounce, &out->taps[bounce], &out->isParam[bounce]);
            out->texId[bounce] = m.texid[0];
            const AF4 color = texColor * m.colors[GLTF_COLOR_BASE];
            gltfSampleAndEval<AF4>(m, rec.mat[bounce], (-1.0f) * ray_dir, hnorm, huv, color, mk4(1, 1, 1, 1), &matSam);
          }
          const AF4 bxdfVal = matSam.val * (1.0f / std::max(matSam.pdf, 1e-20f));
          const float cosTheta = std::abs(dot(matSam.dir, hnorm));
          s.mis.matSamplePdf = (matSam.flags & RAY_EVENT_S) != 0 ? -1.0f : matSam.pdf;
          s.mis.cosTheta = cosTheta;
          if (p.integratorType == INTEGRATOR_STUPID_PT) accumThroughput = mulAA(accumThroughput, bxdfVal * cosTheta);
          else {
            accumColor = accumColor + mulAA(accumThroughput, shadeColor);
            accumThroughput = mulAA(accumThroughput * cosTheta, bxdfVal);
          }
          if ((matSam.flags & RAY_EVENT_T) != 0) hpos = hpos + hitDist * ray_dir * 2.0f * 1e-6f;
          s.rayPosAndNear = xyzw(OffsRayPos(hpos, hnorm, matSam.dir), 0.0f);
          s.rayDirAndFar = xyzw(matSam.dir, FLT_MAX);
          uint nextFlags = ((s.rayFlags & ~RAY_FLAG_FIRST_NON_SPEC) | matSam.flags);
          if (p.renderLayer == FB_DIRECT && hasNonSpecular(s.rayFlags)) nextFlags |= RAY_FLAG_IS_DEAD;
          else if (!hasNonSpecular(s.rayFlags) && hasNonSpecular(nextFlags)) nextFlags |= RAY_FLAG_FIRST_NON_SPEC;
          s.rayFlags = nextFlags;
        }
      }
    }
    // environment term, added unconditionally in the replay (:1077-1098)
    accumColor = accumColor + accumThroughput * p.envColor;
    out->color = accumColor;
  }
};

static int g_threads = 0;

} // namespace orc

using namespace orc;

struct orc_ctx { Ctx c; };

static void copy_params(Params& p, const orc_params* s)
{
  std::memcpy(&p.projInv, s->projInv, 64);
  std::memcpy(&p.worldViewInv, s->worldViewInv, 64);
  p.winStartX = s->winStartX; p.winStartY = s->winStartY; p.winWidth = s->winWidth; p.winHeight = s->winHeight;
  p.fbWidth = s->fbWidth; p.fbHeight = s->fbHeight;
  p.traceDepth = s->traceDepth; p.integratorType = s->integratorType; p.renderLayer = s->renderLayer;
  p.tileSize = s->tileSize; p.spectralMode = s->spectralMode;
  p.exposureMult = s->exposureMult; p.camLensRadius = s->camLensRadius; p.camTargetDist = s->camTargetDist;
  std::memcpy(&p.camRespoceRGB, s->camRespoceRGB, 16);
  std::memcpy(&p.envColor, s->envColor, 16);
  p.envTexId = s->envTexId; p.envLightId = s->envLightId; p.envCamBackId = s->envCamBackId; p.envEnableSam = s->envEnableSam;
  std::memcpy(&p.envSamRow0, s->envSamRow0, 16); std::memcpy(&p.envSamRow1, s->envSamRow1, 16);
  p.envSpecId = s->envSpecIdPlus1 - 1u; p.envSpecMult = s->envSpecMult;
}

extern "C" {

orc_ctx* orc_create(const orc_scene_desc* scene, const orc_params* params)
{
  orc_ctx* h = new orc_ctx();
  h->c.sc.load(scene);
  copy_params(h->c.p, params);
  h->c.texAddressTable.assign(h->c.sc.textures.size(), TexInfo{size_t(-1), 0, 0, 0});   // LoadSceneEnd (integrator_dr.cpp:24-31)
  h->c.gradSize = 0;
  return h;
}
void orc_destroy(orc_ctx* h) { delete h; }
void orc_set_params(orc_ctx* h, const orc_params* p) { copy_params(h->c.p, p); }
void orc_set_threads(int n) { g_threads = n; }

void orc_pack_xy(orc_ctx* h, uint32_t* out)
{
  h->c.PackXY();
  if (out) std::memcpy(out, h->c.packedXY.data(), h->c.packedXY.size() * 4);
}
void orc_init_random_gens(orc_ctx* h, uint32_t count)
{
  h->c.randomGens.resize(count);
  for (uint32_t i = 0; i < count; i++) h->c.randomGens[i] = RandomGenInit((int)i);
}
void orc_get_random_gens(orc_ctx* h, uint32_t* out, uint32_t count) { std::memcpy(out, h->c.randomGens.data(), (size_t)count * 8); }
void orc_set_random_gens(orc_ctx* h, const uint32_t* in, uint32_t count) { h->c.randomGens.resize(count); std::memcpy(h->c.randomGens.data(), in, (size_t)count * 8); }

static int nthreads() { return g_threads > 0 ? g_threads : omp_get_max_threads(); }

void orc_path_trace_block(orc_ctx* h, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out_color, uint32_t passNum)
{
  Ctx& c = h->c;
  #pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads())
  for (long i = (long)tidBegin; i < (long)(tidBegin + tidCount); ++i)
    for (uint32_t j = 0; j < passNum; ++j)
      c.PathTrace((uint)i, channels, out_color, nullptr, false);
}
void orc_naive_path_trace_block(orc_ctx* h, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out_color, uint32_t passNum)
{
  Ctx& c = h->c;
  #pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads())
  for (long i = (long)tidBegin; i < (long)(tidBegin + tidCount); ++i)
    for (uint32_t j = 0; j < passNum; ++j)
      c.NaivePathTrace((uint)i, channels, out_color);
}

// PathTraceFromInputRaysBlock (integrator_pt_host.cpp:92-103); rays: RayPosAndW / RayDirAndT (cam_plugin/CamPluginAPI.h:27-37), 16 bytes each
void orc_path_trace_from_input_rays_block(orc_ctx* h, uint32_t tid, uint32_t channels, const float* rayPosAndW, const float* rayDirAndT, float* out_color, uint32_t passNum)
{
  Ctx& c = h->c;
  #pragma omp parallel for schedule(dynamic, 64) num_threads(nthreads())
  for (long i = 0; i < (long)tid; ++i)
    for (uint32_t j = 0; j < passNum; ++j)
      c.PathTraceFromInputRays((uint)i, channels, rayPosAndW, rayDirAndT, out_color);
}

void orc_ray_nearest(orc_ctx* h, const float* posNear4, const float* dirFar4, uint32_t n, orc_hit* out, int bruteForce)
{
  const Scene& sc = h->c.sc;
  #pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads())
  for (long i = 0; i < (long)n; i++)
    out[i] = sc.nearest_hit(((const f4*)posNear4)[i], ((const f4*)dirFar4)[i], bruteForce != 0);
}
void orc_ray_any(orc_ctx* h, const float* posNear4, const float* dirFar4, uint32_t n, uint32_t* out, int bruteForce)
{
  const Scene& sc = h->c.sc;
  #pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads())
  for (long i = 0; i < (long)n; i++)
    out[i] = sc.any_hit(((const f4*)posNear4)[i], ((const f4*)dirFar4)[i], bruteForce != 0) ? 1u : 0u;
}

void orc_set_optics(orc_ctx* h, const float* lines4, uint32_t n, float physSizeX, float physSizeY)
{
  h->c.lensLines.clear();
  for (uint32_t i = 0; i < n; i++) h->c.lensLines.push_back(mk4(lines4[4 * i], lines4[4 * i + 1], lines4[4 * i + 2], lines4[4 * i + 3]));
  h->c.physSize = mk2(physSizeX, physSizeY);
}

void orc_ray_nearest_motion(orc_ctx* h, const float* posNear4, const float* dirFar4, uint32_t n, float time, orc_hit* out, int bruteForce)
{
  const Scene& sc = h->c.sc;
  for (uint32_t i = 0; i < n; i++) out[i] = sc.nearest_hit(((const f4*)posNear4)[i], ((const f4*)dirFar4)[i], bruteForce != 0, time);
}
void orc_ray_any_motion(orc_ctx* h, const float* posNear4, const float* dirFar4, uint32_t n, float time, uint32_t* out, int bruteForce)
{
  const Scene& sc = h->c.sc;
  for (uint32_t i = 0; i < n; i++) out[i] = sc.any_hit(((const f4*)posNear4)[i], ((const f4*)dirFar4)[i], bruteForce != 0, time) ? 1u : 0u;
}

// IntegratorDR::PutDiffTex2D (integrator_dr.cpp:33-53)
int orc_put_diff_tex2d(orc_ctx* h, uint32_t texId, uint32_t width, uint32_t height, uint32_t channels, uint64_t* outOffset, uint64_t* outSize)
{
  Ctx& c = h->c;
  if (texId >= c.texAddressTable.size()) {
    std::printf("[orc_put_diff_tex2d]: bad tex id = %u\n", texId);
    *outOffset = (uint64_t)-1; *outSize = 0;
    return 1;
  }
  TexInfo& t = c.texAddressTable[texId];
  t.offset = c.gradSize; t.width = (int)width; t.height = (int)height; t.channels = (int)channels;
  const size_t currSize = size_t(width) * size_t(height) * size_t(channels);
  *outOffset = c.gradSize; *outSize = currSize;
  c.gradSize += currSize;
  return 0;
}

// PixelLossPT + the per-sample body of PathTraceDR (integrator_dr.cpp:1103-1132, 1156-1187).
// The reverse sweep Enzyme generates is replaced by forward-mode duals over the <= traceDepth texture fetches.
static float dr_sample(Ctx& c, uint tid, uint channels, float* out_color, const float* refImg, const float* data, float* grad, Record& rec)
{
  rec.enabled = true;
  c.PathTrace(tid, channels, out_color, &rec, true);      // (1) record; image contribution disabled (:1138)
  rec.enabled = false;
  Ctx::ReplayOut ro;
  c.PathTraceReplay(tid, rec, data, &ro);                 // (2) replay
  const uint XY = c.packedXY[tid];
  const uint x = (XY & 0x0000FFFFu), y = (XY & 0xFFFF0000u) >> 16;
  const uint pitch = (uint)c.p.winWidth;
  const uint yRef = (uint)c.p.winHeight - y - 1;
  const f4 colorRend = ro.color.v;
  const f4 colorRef = mk4(refImg[(yRef * pitch + x) * channels + 0], refImg[(yRef * pitch + x) * channels + 1], refImg[(yRef * pitch + x) * channels + 2], 0.0f);
  const f4 diff = colorRend - colorRef;
  // Mirror of the product's hpt_set_option("dr_skip_nonfinite", 1) - NOT reference behaviour (PixelLossPT adds every sample,
  // integrator_dr.cpp:1124-1131): a sample whose radiance is not finite gives neither colour, loss nor gradient.
  if (c.drSkipNonFinite && !std::isfinite(diff.x + diff.y + diff.z)) return 0.0f;
  out_color[(y * pitch + x) * channels + 0] += colorRend.x;
  out_color[(y * pitch + x) * channels + 1] += colorRend.y;
  out_color[(y * pitch + x) * channels + 2] += colorRend.z;
  const float loss = diff.x * diff.x + diff.y * diff.y + diff.z * diff.z;
  if (grad) {
    for (uint b = 0; b < c.p.traceDepth && b < (uint)MAXB; b++) {
      if (!ro.isParam[b]) continue;
      const TexInfo& info = c.texAddressTable[ro.texId[b]];
      const f4 dC = ro.color.d[b];                       // dC_ch / d texColor_ch at bounce b
      float g[3] = { 2.0f * diff.x * dC.x, 2.0f * diff.y * dC.y, 2.0f * diff.z * dC.z };
      if (c.drSkipNonFinite && !std::isfinite(g[0] + g[1] + g[2])) g[0] = g[1] = g[2] = 0.0f;
      for (int k = 0; k < 4; k++) {
        const float w = ro.taps[b].w[k];
        if (info.channels == 4) {
          grad[info.offset + (size_t)ro.taps[b].off[k] * 4 + 0] += g[0] * w;
          grad[info.offset + (size_t)ro.taps[b].off[k] * 4 + 1] += g[1] * w;
          grad[info.offset + (size_t)ro.taps[b].off[k] * 4 + 2] += g[2] * w;
        } else {
          grad[info.offset + (size_t)ro.taps[b].off[k]] += (g[0] + g[1] + g[2]) * w;
        }
      }
    }
  }
  return loss;
}

int orc_set_option(orc_ctx* h, const char* name, int value)
{
  if (std::strcmp(name, "dr_skip_nonfinite") == 0) { h->c.drSkipNonFinite = value != 0; return 0; }
  return 1;
}

float orc_path_trace_dr(orc_ctx* h, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out_color, uint32_t passNum,
                        const float* refImg, const float* data, float* dataGrad, uint64_t gradSize)
{
  Ctx& c = h->c;
  std::memset(dataGrad, 0, sizeof(float) * gradSize);
  const int T = nthreads();
  std::vector<std::vector<float>> grads(T, std::vector<float>(gradSize, 0.0f));   // private gradient per thread (:1143-1147)
  std::vector<double> lossT(T, 0.0);
  #pragma omp parallel for schedule(static) num_threads(T)
  for (long i = (long)tidBegin; i < (long)(tidBegin + tidCount); ++i) {
    const int th = omp_get_thread_num();
    Record rec; rec.enabled = false;
    for (uint32_t passId = 0; passId < passNum; passId++) {
      const float lossVal = dr_sample(c, (uint)i, channels, out_color, refImg, data, grads[th].data(), rec);
      lossT[th] += double(lossVal) / double(passNum);
    }
  }
  for (int t = 0; t < T; t++)
    for (size_t j = 0; j < gradSize; j++) dataGrad[j] += grads[t][j];                // (:1202-1204)
  double avgLoss = 0.0;
  for (int t = 0; t < T; t++) avgLoss += lossT[t];
  return float(avgLoss / double(c.p.winWidth * c.p.winHeight));                        // (:1206)
}

void orc_path_trace_dr_fd(orc_ctx* h, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, uint32_t passNum,
                          const float* refImg, const float* data, uint64_t gradSize,
                          const uint64_t* idx, uint32_t nIdx, float hstep, double* outDeriv)
{
  Ctx& c = h->c;
  for (uint32_t k = 0; k < nIdx; k++) outDeriv[k] = 0.0;
  std::vector<RandomGen> saved = c.randomGens;
  std::vector<float> scratch((size_t)c.p.winWidth * c.p.winHeight * channels, 0.0f);
  // record every sample once, then re-shade it with perturbed parameters
  std::vector<Record> recs((size_t)tidCount * passNum);
  for (uint32_t i = 0; i < tidCount; i++)
    for (uint32_t j = 0; j < passNum; j++) {
      Record& r = recs[(size_t)i * passNum + j];
      r.enabled = true;
      c.PathTrace(tidBegin + i, channels, scratch.data(), &r, true);
      r.enabled = false;
    }
  c.randomGens = saved;
  std::vector<float> pd(data, data + gradSize);
  const uint pitch = (uint)c.p.winWidth;
  for (uint32_t k = 0; k < nIdx; k++) {
    double lossPM[2] = {0.0, 0.0};
    for (int s = 0; s < 2; s++) {
      pd[idx[k]] = data[idx[k]] + (s == 0 ? hstep : -hstep);
      for (uint32_t i = 0; i < tidCount; i++)
        for (uint32_t j = 0; j < passNum; j++) {
          Ctx::ReplayOut ro;
          c.PathTraceReplay(tidBegin + i, recs[(size_t)i * passNum + j], pd.data(), &ro);
          const uint XY = c.packedXY[tidBegin + i];
          const uint x = (XY & 0x0000FFFFu), y = (XY & 0xFFFF0000u) >> 16;
          const uint yRef = (uint)c.p.winHeight - y - 1;
          const double dx = double(ro.color.v.x) - refImg[(yRef * pitch + x) * channels + 0];
          const double dy = double(ro.color.v.y) - refImg[(yRef * pitch + x) * channels + 1];
          const double dz = double(ro.color.v.z) - refImg[(yRef * pitch + x) * channels + 2];
          lossPM[s] += dx * dx + dy * dy + dz * dz;
        }
    }
    pd[idx[k]] = data[idx[k]];
    outDeriv[k] = (lossPM[0] - lossPM[1]) / (2.0 * double(hstep));
  }
}

void orc_rng_kat(int seed, uint32_t nDraws, uint32_t* outState2, float* outFloat4PerDraw)
{
  RandomGen g = RandomGenInit(seed);
  outState2[0] = g.sx; outState2[1] = g.sy;
  for (uint32_t i = 0; i < nDraws; i++) {
    const f4 r = rndFloat4(&g);
    outFloat4PerDraw[4 * i + 0] = r.x; outFloat4PerDraw[4 * i + 1] = r.y; outFloat4PerDraw[4 * i + 2] = r.z; outFloat4PerDraw[4 * i + 3] = r.w;
  }
  outState2[2] = g.sx; outState2[3] = g.sy;
}

void orc_tex_sample(orc_ctx* h, uint32_t texId, const float* uv2, uint32_t n, float* out4)
{
  for (uint32_t i = 0; i < n; i++) {
    const f4 r = h->c.sc.tex_sample(texId, mk2(uv2[2 * i], uv2[2 * i + 1]));
    out4[4 * i + 0] = r.x; out4[4 * i + 1] = r.y; out4[4 * i + 2] = r.z; out4[4 * i + 3] = r.w;
  }
}

// scalar probes of the restated BSDF / sampling helpers, for closed-form unit tests
int orc_probe(const char* name, const float* a, float* out)
{
  const std::string n(name);
  if (n == "FrDielectricPBRT") { out[0] = FrDielectricPBRT(a[0], a[1], a[2]); return 0; }
  if (n == "misWeightHeuristic") { out[0] = misWeightHeuristic(a[0], a[1]); return 0; }
  if (n == "FrComplexConductor") { out[0] = FrComplexConductor(a[0], cmk(a[1], a[2])); return 0; }
  if (n == "FrDielectricDetailedV2") { const f4 r = FrDielectricDetailedV2(a[0], a[1]); out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w; return 0; }
  if (n == "ggxEvalPDF") { out[0] = ggxEvalPDF(mk3(a[0], a[1], a[2]), mk3(a[3], a[4], a[5]), mk3(0, 0, 1), a[6]); return 0; }
  if (n == "ggxEvalBSDF") { out[0] = ggxEvalBSDF(mk3(a[0], a[1], a[2]), mk3(a[3], a[4], a[5]), mk3(0, 0, 1), a[6]); return 0; }
  if (n == "ggxSample") { const f3 r = ggxSample(mk2(a[0], a[1]), mk3(a[2], a[3], a[4]), mk3(0, 0, 1), a[5]); out[0] = r.x; out[1] = r.y; out[2] = r.z; return 0; }
  if (n == "trD") { out[0] = trD(mk3(a[0], a[1], a[2]), mk2(a[3], a[4])); return 0; }
  if (n == "lambertSample") { const f3 r = lambertSample(mk2(a[0], a[1]), mk3(0, 0, 1), mk3(a[2], a[3], a[4])); out[0] = r.x; out[1] = r.y; out[2] = r.z; return 0; }
  if (n == "MapSamplesToDisc") { const f2 r = MapSamplesToDisc(mk2(a[0], a[1])); out[0] = r.x; out[1] = r.y; return 0; }
  if (n == "plasticEval" || n == "plasticSample") {
    // a[0..3] roughness, ior ratio, spec weight, internal reflectance; a[4] nonlinear; a[5..7] reflectance; a[8..10] v; a[11..13] l or rands.xyz;
    // a[14..77] the transmittance table; n = +z
    Material m; std::memset(&m, 0, sizeof(m));
    for (int k = 0; k < 4; k++) m.data[k] = a[k];
    m.nonlinear = (uint)a[4];
    const f4 refl = mk4(a[5], a[6], a[7], 0.0f);
    const f3 v = mk3(a[8], a[9], a[10]);
    if (n == "plasticEval") {
      BsdfEval r; r.val = mk4(0, 0, 0, 0); r.pdf = 0.0f;
      plasticEval(m, refl, mk3(a[11], a[12], a[13]), v, mk3(0, 0, 1), &r, a + 14, 0);
      out[0] = r.val.x; out[1] = r.val.y; out[2] = r.val.z; out[3] = r.pdf;
    } else {
      BsdfSample r; r.val = mk4(0, 0, 0, 0); r.pdf = 1.0f; r.dir = mk3(0, 1, 0); r.flags = 0; r.ior = 1.0f;
      plasticSampleAndEval(m, refl, mk4(a[11], a[12], a[13], 0.0f), v, mk3(0, 0, 1), &r, a + 14, 0);
      out[0] = r.dir.x; out[1] = r.dir.y; out[2] = r.dir.z; out[3] = r.val.x; out[4] = r.val.y; out[5] = r.val.z; out[6] = r.pdf;
    }
    return 0;
  }
  if (n == "SampleWavelengths") { const f4 r = Ctx::SampleWavelengths(a[0], a[1], a[2]); out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w; return 0; }
  if (n == "XYZToRGB") { const f3 r = Ctx::XYZToRGB(mk3(a[0], a[1], a[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; return 0; }
  if (n == "FrFilm") {          // a: cosThetaI, etaI (re, im), etaF (re, im), etaT (re, im), thickness, lambda -> refl, refr, FrFilmRefl
    const FrReflRefr r = FrFilm(a[0], cmk(a[1], a[2]), cmk(a[3], a[4]), cmk(a[5], a[6]), a[7], a[8]);
    out[0] = r.refl; out[1] = r.refr; out[2] = FrFilmRefl(a[0], cmk(a[1], a[2]), cmk(a[3], a[4]), cmk(a[5], a[6]), a[7], a[8]); return 0;
  }
  if (n == "orennayarFunc") { out[0] = orennayarFunc(mk3(a[0], a[1], a[2]), mk3(a[3], a[4], a[5]), mk3(0, 0, 1), a[6]); return 0; }
  return 1;
}

// AdamOptimizer<float>::step (diff_render/adam.h:43-62)
void orc_adam_step(float* state, const float* grad, float* momentum, float* gsquare, uint64_t n, int iter)
{
  const int factorGamma = iter / 100 + 1;
  const float alpha = 0.5f, beta = 0.25f, gamma = 0.25f / float(factorGamma), epsilon = 1e-8f;
  for (uint64_t i = 0; i < n; i++) {
    const float g = grad[i];
    momentum[i] = momentum[i] * beta + g * (1.0f - beta);
    gsquare[i] = 2.0f * (gsquare[i] * alpha + (g * g) * (1.0f - alpha));
  }
  for (uint64_t i = 0; i < n; i++) state[i] -= (gamma * momentum[i] / (std::sqrt(gsquare[i] + epsilon)));
}


// RegLossImage2D4f (diff_render/integrator_dr.cpp:317-351): sum over interior pixels of sqrt(|p0-top|^2 + |p0-bottom|^2 + |p0-left|^2 +
// |p0-right|^2) on rgb (dot3); the odd line-order loop at :340-347 visits every line 1..h-2 exactly once.
double orc_reg_loss_image2d4f(int w, int h, const float* data)
{
  double summ = 0.0;
  for (int y = 1; y < h - 1; y++) {
    double line = 0.0;
    for (int x = 1; x < w - 1; x++) {
      float s = 0.0f;
      const int nb[4] = { (y + 1) * w + x, (y - 1) * w + x, y * w + x - 1, y * w + x + 1 };
      // the reference adds dot3(left) + dot3(right) + dot3(top) + dot3(bottom) in float, then takes the sqrt in double
      float acc[4];
      for (int k = 0; k < 4; k++) {
        const float dx = data[(y * w + x) * 4 + 0] - data[nb[k] * 4 + 0], dy = data[(y * w + x) * 4 + 1] - data[nb[k] * 4 + 1], dz = data[(y * w + x) * 4 + 2] - data[nb[k] * 4 + 2];
        acc[k] = dx * dx + dy * dy + dz * dz;
      }
      s = acc[2] + acc[3] + acc[0] + acc[1];
      line += std::sqrt(double(s));
    }
    summ += line;
  }
  return summ;
}

// Image2D4fRegularizer (integrator_dr.cpp:361-367): grad += d RegLossImage2D4f / d data (Enzyme accumulates into the shadow argument).
// Restated analytically: d sqrt(S)/d p0 = (4 p0 - sum of neighbours)/sqrt(S), d sqrt(S)/d p_k = -(p0 - p_k)/sqrt(S); 0 where S == 0.
void orc_image2d4f_regularizer(int w, int h, const float* data, float* grad)
{
  for (int y = 1; y < h - 1; y++)
    for (int x = 1; x < w - 1; x++) {
      const int c = y * w + x;
      const int nb[4] = { (y + 1) * w + x, (y - 1) * w + x, y * w + x - 1, y * w + x + 1 };
      double S = 0.0;
      for (int k = 0; k < 4; k++) for (int ch = 0; ch < 3; ch++) { const double d = double(data[c * 4 + ch]) - double(data[nb[k] * 4 + ch]); S += d * d; }
      if (!(S > 0.0)) continue;
      const double inv = 1.0 / std::sqrt(S);
      for (int k = 0; k < 4; k++) for (int ch = 0; ch < 3; ch++) {
        const double d = (double(data[c * 4 + ch]) - double(data[nb[k] * 4 + ch])) * inv;
        grad[c * 4 + ch] += float(d); grad[nb[k] * 4 + ch] -= float(d);
      }
    }
}

} // extern "C"
