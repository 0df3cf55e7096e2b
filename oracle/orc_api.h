// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md). C interface of the CPU restatement of
// HydraCore3's Integrator::PathTraceBlock / IntegratorDR::PathTraceDR.  PARITY UNPINNED: the reference
// ships no golden vectors for this path and cannot be built in this image.
//
// The plain-data structs below describe the flat buffers Integrator::LoadScene produces
// (integrator_pt.h:472-500); their binary layout deliberately equals include/hydra_hip.h's so that one
// ctypes structure in the tests can feed both the oracle and the HIP library with the same bytes.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_texture_desc {
  uint32_t width, height;
  uint32_t format;    // 0: RGBA8 packed in uint32 (r = low byte), 1: RGBA32F, 2: R32F
  uint32_t flags;     // bit 0: sRGB decode, rgb^2.2 after filtering (HydraSampler::inputGamma, integrator_pt.h:71)
  uint32_t addressU;  // 0 WRAP, 2 CLAMP (LiteImage::Sampler::AddressMode numbering)
  uint32_t addressV;
  uint32_t filter;    // 0 NEAREST, 1 LINEAR
  uint32_t reserved;
  const void* data;   // row-major texels, row 0 first
} orc_texture_desc;

typedef struct orc_scene_desc {
  uint32_t numGeoms, numInsts, numVerts, numTris;
  const float*    vPos4f;         // float4 x numVerts, object-space positions (AddGeom_Triangles3f input, stride 16)
  const float*    vData8f;        // 8 floats x numVerts: {norm.xyz, u | tang.xyz, v}   (m_vData8f, integrator_pt.h:479-484)
  const uint32_t* triIndices;     // 3 x numTris, mesh-local vertex indices             (m_triIndices)
  const uint32_t* matIdByPrimId;  // numTris                                            (m_matIdByPrimId)
  const uint32_t* matVertOffset;  // uint2 x numGeoms: (triOffset, vertOffset)          (m_matVertOffset)
  const uint32_t* geomTriCount;   // numGeoms
  const uint32_t* geomVertCount;  // numGeoms
  const uint32_t* instGeomId;     // numInsts: mesh instanced by instance i             (AddInstance order)
  const float*    instMatrices;   // column-major float4x4 x numInsts
  const float*    normMatrices;   // column-major float4x4 x numInsts                   (m_normMatrices)
  const int32_t*  remapInst;      // int2 x numInsts: (remap list id, light id)         (m_remapInst)
  const int32_t*  allRemapLists;  // lists followed by their offsets                    (m_allRemapLists)
  uint32_t        allRemapListsLen;
  uint32_t        allRemapListsSize; // m_allRemapListsSize: where the offsets start
  const void*     materials;      // 320-byte Material records (include/cmaterial.h:187-203)
  uint32_t        numMaterials;
  uint32_t        numLights;
  const void*     lights;         // 320-byte LightSource records (include/clight.h:19-56)
  const orc_texture_desc* textures;
  uint32_t        numTextures;
  uint32_t        numArrays1f;
  const float*    arrays1f;       // m_arrays1f: pdf table of the sampled environment map (may be NULL)
  // motion blur (integrator_pt_scene.cpp:848-897): normMatrices holds 2 x numInsts matrices when normMatrices2Offs != 0 (= numInsts)
  const float*    instMatricesMotion; // numInsts column-major float4x4: the instance matrix at time 1 (may be NULL)
  const uint32_t* instHasMotion;      // numInsts flags (may be NULL)
  uint32_t        normMatrices2Offs;  // m_normMatrices2Offs
  uint32_t        reserved;
  // spectral rendering: m_spec_values (1 nm from 360), m_spec_offset_sz (uint2 per spectrum), m_cie_xyz (float4 x 471), camera response ids
  const float*    specValues;
  const uint32_t* specOffsetSz;
  uint32_t        numSpecValues, numSpectra;
  const float*    cieXYZ;
  uint32_t        numCieXYZ;
  int32_t         camResponseSpectrumId[3];
  uint32_t        camResponseType;
  uint32_t        reserved2;
  // thin films (integrator_pt.h:587-590): m_films_thickness_vec, m_films_spec_id_vec, m_films_eta_k_vec, m_precomp_thin_films (all may be NULL / 0)
  const float*    filmsThickness;
  const uint32_t* filmsSpecId;
  const float*    filmsEtaK;
  const float*    precompThinFilms;
  uint32_t        numFilmsThickness, numFilmsSpecId, numFilmsEtaK, numPrecompThinFilms;
  /* spectra given by textures (KSPEC_SPD_TEX; LoadSceneSpectrumData, integrator_pt_scene.cpp:363-377): m_spec_tex_ids_wavelengths = uint2 {texture,
   * wavelength in nm} per band, m_spec_tex_offset_sz = uint2 {first band, bands} per spectrum id ({0xFFFFFFFF, 0}: a tabulated spectrum). NULL / 0
   * without such spectra. */
  const uint32_t* specTexIdsWavelengths;
  const uint32_t* specTexOffsetSz;
  uint32_t        numSpecTexBands, reserved3;
} orc_scene_desc;

typedef struct orc_params {
  float    projInv[16];       // m_projInv (column-major)
  float    worldViewInv[16];  // m_worldViewInv
  int32_t  winStartX, winStartY, winWidth, winHeight, fbWidth, fbHeight;
  uint32_t traceDepth;        // m_traceDepth
  uint32_t integratorType;    // m_intergatorType: 0 naive, 1 shadow, 2 MIS
  uint32_t renderLayer;       // m_renderLayer: 0 colour, 1 direct, 2 indirect
  uint32_t tileSize;          // m_tileSize
  uint32_t spectralMode;      // m_spectral_mode
  uint32_t envSpecIdPlus1;    // m_envSpecId + 1 (0 = none)
  float    exposureMult, camLensRadius, camTargetDist, envSpecMult;   // m_envSpecMult
  float    camRespoceRGB[4];
  float    envColor[4];
  uint32_t envTexId, envLightId, envCamBackId, envEnableSam;   // m_envTexId, m_envLightId, m_envCamBackId, m_envEnableSam (0xFFFFFFFF: none)
  float    envSamRow0[4], envSamRow1[4];                       // m_envSamRow0, m_envSamRow1
} orc_params;

// CRT_Hit (external/CrossRT/CrossRT.h:23-30)
typedef struct orc_hit {
  float t; uint32_t primId, instId, geomId; float coords[4];
} orc_hit;

typedef struct orc_ctx orc_ctx;

orc_ctx* orc_create(const orc_scene_desc* scene, const orc_params* params);
void     orc_destroy(orc_ctx*);
void     orc_set_params(orc_ctx*, const orc_params*);
void     orc_set_threads(int n);

// kernel_PackXY over the whole window (integrator_rt.cpp:13-31, integrator_pt_host.cpp:19-27)
void     orc_pack_xy(orc_ctx*, uint32_t* out_packedXY /* may be NULL */);
// InitRandomGens (integrator_pt.cpp:13-21)
void     orc_init_random_gens(orc_ctx*, uint32_t count);
void     orc_get_random_gens(orc_ctx*, uint32_t* out_uint2, uint32_t count);
void     orc_set_random_gens(orc_ctx*, const uint32_t* in_uint2, uint32_t count);

// PathTraceBlock (integrator_pt_host.cpp:57-73) restricted to tid in [tidBegin, tidBegin+tidCount)
void     orc_path_trace_block(orc_ctx*, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out_color, uint32_t passNum);
void     orc_naive_path_trace_block(orc_ctx*, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out_color, uint32_t passNum);
void     orc_path_trace_from_input_rays_block(orc_ctx*, uint32_t tid, uint32_t channels, const float* rayPosAndW, const float* rayDirAndT, float* out_color, uint32_t passNum);

// ISceneObject::RayQuery_NearestHit / RayQuery_AnyHit semantics (CrossRT.h:157-176), batched.
// bruteForce != 0 tests every triangle of every instance (no BVH).
void     orc_ray_nearest(orc_ctx*, const float* posNear4, const float* dirFar4, uint32_t n, orc_hit* out, int bruteForce);
void     orc_ray_any(orc_ctx*, const float* posNear4, const float* dirFar4, uint32_t n, uint32_t* out, int bruteForce);
// SetLines / SetPhysSize with m_enableOpticSim: n x {curvatureRadius, thickness, eta, apertureRadius}, film side first; n = 0 turns it off
void     orc_set_optics(orc_ctx*, const float* lines4, uint32_t n, float physSizeX, float physSizeY);
// RayQuery_NearestHitMotion / RayQuery_AnyHitMotion (CrossRT.h:157,174): the same at a time in [0, 1] of the moving instances
void     orc_ray_nearest_motion(orc_ctx*, const float* posNear4, const float* dirFar4, uint32_t n, float time, orc_hit* out, int bruteForce);
void     orc_ray_any_motion(orc_ctx*, const float* posNear4, const float* dirFar4, uint32_t n, float time, uint32_t* out, int bruteForce);

// IntegratorDR::PutDiffTex2D / PathTraceDR (diff_render/integrator_dr.cpp:33-53, 1135-1218)
int      orc_put_diff_tex2d(orc_ctx*, uint32_t texId, uint32_t width, uint32_t height, uint32_t channels, uint64_t* outOffset, uint64_t* outSize);
float    orc_path_trace_dr(orc_ctx*, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out_color, uint32_t passNum,
                           const float* refImg, const float* data, float* dataGrad, uint64_t gradSize);
// options that mirror the product's hpt_set_option where a parity test needs both sides switched: "dr_skip_nonfinite"
int      orc_set_option(orc_ctx*, const char* name, int value);
// Finite-difference companion: replays exactly the samples orc_path_trace_dr would draw (same RNG state
// on entry) and returns d(sum over samples of |c - ref|^2)/d data[idx[i]] by central differences with step h.
void     orc_path_trace_dr_fd(orc_ctx*, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, uint32_t passNum,
                              const float* refImg, const float* data, uint64_t gradSize,
                              const uint64_t* idx, uint32_t nIdx, float h, double* outDeriv);

// probes used by unit tests
void     orc_rng_kat(int seed, uint32_t nDraws, uint32_t* outState2, float* outFloat4PerDraw);
int      orc_probe(const char* name, const float* args, float* out);
void     orc_tex_sample(orc_ctx*, uint32_t texId, const float* uv2, uint32_t n, float* out4);
// AdamOptimizer<float>::step (diff_render/adam.h:43-62)
void     orc_adam_step(float* state, const float* grad, float* momentum, float* gsquare, uint64_t n, int iter);
// RegLossImage2D4f / Image2D4fRegularizer (diff_render/integrator_dr.cpp:317-367)
double   orc_reg_loss_image2d4f(int w, int h, const float* data);
void     orc_image2d4f_regularizer(int w, int h, const float* data, float* grad);

#ifdef __cplusplus
}
#endif
