/* hydra_hip.h -- C ABI of the MI355X-native (gfx950) path-tracing core.
 *
 * This is the whole drop-in boundary: plain pointers and sizes, no C++ types, no torch types. A host adapter
 * (hydracore3_amd/csrc/integrator_hip.h: class IntegratorHIP, class BVH2SceneHIP) exposes it through the
 * reference's own class surface; INTEGRATION.md shows the subclass a HydraCore3 maintainer would add.
 *
 * Every entry point names the reference interface it replaces (paths relative to the HydraCore3 tree).
 * All functions return 0 on success and a non-zero code on failure unless stated otherwise;
 * hpt_last_error() gives the message. Nothing throws across this boundary. One hpt_ctx per GPU; all calls
 * on a context come from one host thread (as in the reference: main.cpp, drmain.cpp).
 */
#ifndef HYDRA_HIP_H
#define HYDRA_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPT_OK            0
#define HPT_ERR_ARG       1   /* bad argument (the reference prints and returns, e.g. EmbreeRT.cpp:144-159) */
#define HPT_ERR_HIP       2   /* HIP runtime error */
#define HPT_ERR_STATE     3   /* call order violated (e.g. render before CommitDeviceData) */
#define HPT_ERR_UNSUPPORTED 4 /* scene uses a feature outside the hot path's scope */
#define HPT_ERR_NOMEM     5   /* a host allocation failed (std::bad_alloc / std::length_error caught at the boundary: sizes no host can hold) */

typedef struct hpt_ctx hpt_ctx;

/* texture + sampler: LiteImage::ICombinedImageSampler (integrator_pt.h:577, integrator_pt_scene_tex.cpp:19-103) */
typedef struct hpt_texture_desc {
  uint32_t width, height;
  uint32_t format;    /* 0: RGBA8 in a uint32 (r = low byte), 1: RGBA32F, 2: R32F */
  uint32_t flags;     /* bit 0: sRGB decode (rgb^2.2 after filtering; HydraSampler::inputGamma, integrator_pt.h:71) */
  uint32_t addressU;  /* LiteImage::Sampler::AddressMode: 0 WRAP, 2 CLAMP */
  uint32_t addressV;
  uint32_t filter;    /* LiteImage::Sampler::Filter: 0 NEAREST, 1 LINEAR */
  uint32_t reserved;
  const void* data;   /* host pointer, row-major, row 0 first */
} hpt_texture_desc;

/* The flat scene vectors Integrator::LoadScene fills (integrator_pt.h:472-500), as {pointer,count} pairs. */
typedef struct hpt_scene_desc {
  uint32_t numGeoms, numInsts, numVerts, numTris;
  const float*    vPos4f;         /* float4 x numVerts: positions handed to AddGeom_Triangles3f (integrator_pt_scene.cpp:799-800) */
  const float*    vData8f;        /* m_vData8f: {norm.xyz, u | tang.xyz, v} x numVerts (integrator_pt.h:479-484) */
  const uint32_t* triIndices;     /* m_triIndices: 3 x numTris, mesh-local */
  const uint32_t* matIdByPrimId;  /* m_matIdByPrimId: numTris */
  const uint32_t* matVertOffset;  /* m_matVertOffset: uint2 x numGeoms (triOffset, vertOffset) */
  const uint32_t* geomTriCount;   /* numGeoms */
  const uint32_t* geomVertCount;  /* numGeoms */
  const uint32_t* instGeomId;     /* numInsts: geomId passed to AddInstance, in instance order (integrator_pt_scene.cpp:852-885) */
  const float*    instMatrices;   /* numInsts column-major float4x4 (CrossRT.h:134, EmbreeRT.cpp:256) */
  const float*    normMatrices;   /* m_normMatrices: numInsts column-major float4x4 (integrator_pt_scene.cpp:877) */
  const int32_t*  remapInst;      /* m_remapInst: int2 x numInsts (remap list id, light id) */
  const int32_t*  allRemapLists;  /* m_allRemapLists: lists then offsets (integrator_pt_scene.cpp:909-924) */
  uint32_t        allRemapListsLen;
  uint32_t        allRemapListsSize; /* m_allRemapListsSize */
  const void*     materials;      /* m_materials: 320-byte Material records (include/cmaterial.h:187-203) */
  uint32_t        numMaterials;
  uint32_t        numLights;
  const void*     lights;         /* m_lights: 320-byte LightSource records (include/clight.h:19-56) */
  const hpt_texture_desc* textures; /* m_textures */
  uint32_t        numTextures;
  uint32_t        numArrays1f;
  const float*    arrays1f;       /* m_arrays1f: the environment light's pdf table (integrator_pt_scene.cpp:464-475); may be NULL */
  /* motion blur (integrator_pt_scene.cpp:848-897): with normMatrices2Offs != 0 (= numInsts) normMatrices holds 2 x numInsts matrices, the
   * second half for the end of the motion; instMatricesMotion / instHasMotion are read only when the geometry comes with the tables
   * (vPos4f != NULL) and stand for AddInstanceMotion(geomId, {matrix, matrix_motion}, 2) (:864-868). All may be NULL / 0. */
  const float*    instMatricesMotion; /* numInsts column-major float4x4: the instance matrix at time 1 */
  const uint32_t* instHasMotion;      /* numInsts flags */
  uint32_t        normMatrices2Offs;  /* m_normMatrices2Offs */
  uint32_t        reserved;
  /* spectral rendering (integrator_pt.h: m_spec_values, m_spec_offset_sz, m_cie_xyz, m_camResponseSpectrumId / m_camResponseType): all may be
   * NULL / 0 for RGB rendering. specValues: every spectrum resampled at 1 nm from LAMBDA_MIN = 360 (Spectrum::ResampleUniform, 471 floats each);
   * specOffsetSz: uint2 {offset, size} per spectrum id (0xFFFFFFFF offset: a spectrum given by textures, see specTexOffsetSz below);
   * cieXYZ: float4 {x, y, z, 0} x 471, the CIE 1931 observer at 360..830 nm (LoadScene fills it from Get_CIE_X/Y/Z, integrator_pt_scene.cpp:962-969). */
  const float*    specValues;
  const uint32_t* specOffsetSz;
  uint32_t        numSpecValues, numSpectra;
  const float*    cieXYZ;
  uint32_t        numCieXYZ;
  int32_t         camResponseSpectrumId[3];   /* -1: none (then SpectrumToXYZ + XYZToRGB) */
  uint32_t        camResponseType;            /* m_camResponseType: 0 = CAM_RESPONCE_XYZ, 1 = CAM_RESPONCE_RGB (integrator_pt.h:531-534) */
  uint32_t        reserved2;
  /* thin films (MAT_TYPE_THIN_FILM, integrator_pt.h:587-590): m_films_thickness_vec, m_films_spec_id_vec, m_films_eta_k_vec and the loader's
   * reflectance / transmittance tables m_precomp_thin_films (LoadThinFilmMaterial, integrator_pt_scene_mat.cpp:1020-1193; hpt_film_precompute
   * computes one material's). All may be NULL / 0 for scenes without films. */
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
} hpt_scene_desc;

/* The plain-data members UpdateMembersPlainData() refreshes before every *Block call (integrator_pt.h:268, main.cpp:398). */
typedef struct hpt_params {
  float    projInv[16];       /* m_projInv, column-major */
  float    worldViewInv[16];  /* m_worldViewInv */
  int32_t  winStartX, winStartY, winWidth, winHeight, fbWidth, fbHeight; /* SetViewport / SetFrameBufferSize (integrator_pt.h:365-392) */
  uint32_t traceDepth;        /* m_traceDepth */
  uint32_t integratorType;    /* m_intergatorType: 0 naive, 1 shadow, 2 MIS (integrator_pt.h:330-332) */
  uint32_t renderLayer;       /* m_renderLayer: FB_COLOR / FB_DIRECT / FB_INDIRECT (integrator_pt.h:406-408) */
  uint32_t tileSize;          /* m_tileSize */
  uint32_t spectralMode;      /* m_spectral_mode: 0 = RGB; 1 = four wavelengths per path (needs hpt_scene_desc's spectral tables; PathTraceBlock, NaivePathTraceBlock and
                               * PathTraceFromInputRaysBlock, every material / light / camera feature of the RGB path; channels: 1, 3 - 4, or more = wavelength layers;
                               * PathTraceDR is refused with HPT_ERR_UNSUPPORTED) */
  uint32_t envSpecIdPlus1;    /* m_envSpecId + 1 (integrator_pt.h:524; 0 = none, so that a zeroed struct means "no environment spectrum") */
  float    exposureMult, camLensRadius, camTargetDist, envSpecMult;   /* envSpecMult: m_envSpecMult (spectral mode: the environment spectrum's multiplier) */
  float    camRespoceRGB[4];  /* m_camRespoceRGB */
  float    envColor[4];       /* m_envColor */
  /* the environment of LoadSceneLights (integrator_pt_scene.cpp:441-478), read by EnvironmentColor / kernel_HitEnvironment
   * (integrator_pt_lgt.cpp:175-210, integrator_pt.cpp:550-595): 0xFFFFFFFF = none */
  uint32_t envTexId;          /* m_envTexId: index into the texture table */
  uint32_t envLightId;        /* m_envLightId: the LIGHT_GEOM_ENV entry of m_lights when the map is sampled explicitly */
  uint32_t envCamBackId;      /* m_envCamBackId: texture shown to primary rays that miss */
  uint32_t envEnableSam;      /* m_envEnableSam */
  float    envSamRow0[4], envSamRow1[4]; /* m_envSamRow0 / m_envSamRow1 */
} hpt_params;

/* CRT_Hit (external/CrossRT/CrossRT.h:23-30) */
typedef struct hpt_hit {
  float t; uint32_t primId, instId, geomId; float coords[4];
} hpt_hit;

/* ---- lifetime ------------------------------------------------------------------------------------------------- */
int  hpt_create(int device, hpt_ctx** out);          /* Integrator::Integrator + CreateSceneRT (integrator_pt.h:127-137, CrossRT.h:195) */
void hpt_destroy(hpt_ctx* ctx);                      /* Integrator::~Integrator + DeleteSceneRT (CrossRT.h:196) */
const char* hpt_last_error(hpt_ctx* ctx);
int  hpt_device_info(hpt_ctx* ctx, int* numCUs, int* wavefront, char* name, size_t nameLen);
/* Device memory for callers of the *_dev entry points that do not link the HIP runtime themselves (a plain C/C++ host such as
 * diff_render/drmain.cpp:174-261 keeping its texture, gradient and Adam moments resident). No reference counterpart: the reference's
 * generated GPU class owns its buffers (main.cpp:221-224). kind: 1 host->device, 2 device->host, 3 device->device; copies are
 * synchronous, memset is asynchronous on the null stream. */
/* Integrator::SetLines + SetPhysSize with m_enableOpticSim (integrator_pt.h:353-362; LoadOpticsFromNode integrator_pt_scene.cpp:1078-1141): the
 * lens stack traced from the film by SampleCameraRay (integrator_pt.cpp:79-103, 806-938). lines4 = n x {curvatureRadius, thickness, eta,
 * apertureRadius}, film side first (radius 0 = the aperture stop); n = 0 switches the simulation off. Takes effect at the next call. */
int  hpt_set_optics(hpt_ctx* ctx, const float* lines4, uint32_t n, float physSizeX, float physSizeY);
/* precomputeThinFilmSpectral / precomputeThinFilmRGB (integrator_pt_scene_mat.cpp:791-1018), what LoadThinFilmMaterial (:1020-1193) appends to
 * m_precomp_thin_films for one MAT_TYPE_THIN_FILM: reflectance / transmittance from outside and from inside over wavelength x angle (spectral
 * mode) or [thickness x] angle x rgb (RGB mode). layers = the films plus the substrate (FILM_LAYERS_COUNT); eta / k / their spectrum ids per
 * layer, thickness per film; cieXYZ (float4 x 471) is read in RGB mode only. outTable == NULL asks for the size; *outPrecomputed = 0 when the
 * reference's loader would not precompute (spectral mode, one film with a thickness map: FILM_PRECOMP_FLAG stays 0). Host code; no context. */
typedef struct hpt_film_params {
  int32_t  spectralMode; float extIOR; uint32_t layers; int32_t thicknessMap;
  float    thicknessMin, thicknessMax; uint32_t numSpectra, reserved;
  const float* eta; const float* k; const uint32_t* etaSpecId; const uint32_t* kSpecId; const float* thickness;
  const float* specValues; const uint32_t* specOffsetSz; const float* cieXYZ;
} hpt_film_params;
int  hpt_film_precompute(const hpt_film_params* params, float* outTable, uint64_t outCapacity, uint64_t* outCount, int* outPrecomputed);
/* The JPEG reader of the fixture loaders (LoadTextureAndMakeCombined reads ".jpg" / ".jpeg" through LiteImage, integrator_pt_scene_tex.cpp:24-33):
 * 8-bit baseline / progressive Huffman JPEG, grey or YCbCr, to RGBA8 rows in file order (csrc/jpeg_decode.h). outRGBA8 == NULL asks for the
 * size. HPT_ERR_UNSUPPORTED for files outside that subset. Host code; no context. */
int  hpt_decode_jpeg(const uint8_t* file, uint64_t fileSize, uint32_t* outWidth, uint32_t* outHeight, uint8_t* outRGBA8, uint64_t outCapacity);
/* mi::fresnel_coat_precompute (mi_materials.cpp:377-451), what LoadPlasticMaterial (integrator_pt_scene_mat.cpp:675-757) stores for a
 * MAT_TYPE_PLASTIC: the 64-entry rough-transmittance table (appended to m_arrays1f, its offset in Material::datai[0]) and the two scalars
 * Material::data[PLASTIC_PRECOMP_REFLECTANCE = 3], data[PLASTIC_SPEC_SAMPLE_WEIGHT = 2]. Host code; no context, no device. RGB mode. */
int  hpt_plastic_precompute(float alpha, float intIor, float extIor, const float* diffuseReflectance4, const float* specularReflectance4,
                            float* outTransmittance64, float* outInternalReflectance, float* outSpecularSamplingWeight);
int  hpt_device_malloc(hpt_ctx* ctx, size_t bytes, void** outDev);
int  hpt_device_free(hpt_ctx* ctx, void* dev);
int  hpt_device_copy(hpt_ctx* ctx, void* dst, const void* src, size_t bytes, int kind);
int  hpt_device_memset(hpt_ctx* ctx, void* dev, int value, size_t bytes);

/* ---- ISceneObject: BVH2 replacement of the Embree backend (external/CrossRT/CrossRT.h:45-176) ------------------ */
int      hpt_clear_geom(hpt_ctx* ctx);                                                   /* ClearGeom            :56 */
uint32_t hpt_add_geom_triangles3f(hpt_ctx* ctx, const float* vpos3f, size_t vertNumber, const uint32_t* triIndices,
                                  size_t indNumber, uint32_t flags, size_t vByteStride); /* AddGeom_Triangles3f  :73-74; returns geomId or 0xFFFFFFFF */
int      hpt_update_geom_triangles3f(hpt_ctx* ctx, uint32_t geomId, const float* vpos3f, size_t vertNumber,
                                     const uint32_t* triIndices, size_t indNumber, uint32_t flags, size_t vByteStride); /* UpdateGeom_Triangles3f :85-86 */
int      hpt_clear_scene(hpt_ctx* ctx);                                                  /* ClearScene           :105 */
uint32_t hpt_add_instance(hpt_ctx* ctx, uint32_t geomId, const float matrix16[16]);      /* AddInstance          :118; returns instId or 0xFFFFFFFF */
/* AddInstanceMotion :127 (EmbreeRT.cpp:264-292): matrixNumber key transforms, linearly interpolated at the ray's time; 2 are supported
 * (what LoadSceneInstances passes). The automatic layout choice keeps two-level for scenes with a moving instance (the single-level layout
 * inverts the interpolated matrix per triangle record); both layouts and both schedules render them, bit-identically. */
uint32_t hpt_add_instance_motion(hpt_ctx* ctx, uint32_t geomId, const float* matrices /* matrixNumber x 16, column-major */, uint32_t matrixNumber);
int      hpt_update_instance(hpt_ctx* ctx, uint32_t instId, const float matrix16[16]);   /* UpdateInstance       :134 */
/* CommitScene :109-110. options = BuildOptions (CrossRT.h:8-14): BUILD_HIGH (4, the reference's default) or 0 build the tree with the host's binned-SAH
 * builder; BUILD_LOW (1) / BUILD_MEDIUM (2) without BUILD_HIGH build the single-level tree ON THE DEVICE (a linear BVH and its 4-wide compressed form,
 * milliseconds for 10^6 triangles: scenes whose topology changes every frame) when the scene takes the single-level layout (hpt_set_accel_layout) and no
 * instance moves; hpt_set_option("device_build", 1 / 0) forces / forbids that whatever the options say. Hits never depend on which builder ran. */
int      hpt_commit_scene(hpt_ctx* ctx, uint32_t options);
/* RayQuery_NearestHit / RayQuery_AnyHit (:148,:165), batched: n rays from host memory, results to host memory. */
int      hpt_ray_query_nearest(hpt_ctx* ctx, const float* posAndNear4, const float* dirAndFar4, uint32_t n, hpt_hit* out);
int      hpt_ray_query_any(hpt_ctx* ctx, const float* posAndNear4, const float* dirAndFar4, uint32_t n, uint32_t* out);
/* RayQuery_NearestHitMotion / RayQuery_AnyHitMotion (:157,:174): the same at a time in [0, 1] of the moving instances (one time per batch). */
int      hpt_ray_query_nearest_motion(hpt_ctx* ctx, const float* posAndNear4, const float* dirAndFar4, uint32_t n, float time, hpt_hit* out);
int      hpt_ray_query_any_motion(hpt_ctx* ctx, const float* posAndNear4, const float* dirAndFar4, uint32_t n, float time, uint32_t* out);

/* ---- scene tables --------------------------------------------------------------------------------------------- */
/* CommitDeviceData() (integrator_pt.h:265): uploads every scene vector. If desc->vPos4f is non-NULL the geometry
 * is (re)registered through the ISceneObject calls above in mesh / instance order and committed; otherwise the
 * acceleration structure committed earlier through hpt_add_geom_* / hpt_add_instance / hpt_commit_scene is used. */
int  hpt_upload_scene(hpt_ctx* ctx, const hpt_scene_desc* desc);
int  hpt_update_params(hpt_ctx* ctx, const hpt_params* params);                          /* UpdateMembersPlainData :268 */
int  hpt_update_materials(hpt_ctx* ctx, size_t first, size_t count, const void* mats);   /* Update_m_materials     :468 */
int  hpt_update_lights(hpt_ctx* ctx, size_t first, size_t count, const void* lights);    /* Update_m_lights        :469 */
/* Update_m_matIdOffsets() (integrator_pt.h:470): re-upload m_matVertOffset (uint2 x numGeoms: first triangle / first vertex of every mesh in
 * m_matIdByPrimId / m_triIndices and m_vData8f) after the host changed it; the ranges are checked against the uploaded tables. */
int  hpt_update_mat_id_offsets(hpt_ctx* ctx, const uint32_t* matVertOffset, size_t numGeoms);
int  hpt_pack_xy(hpt_ctx* ctx, uint32_t tidX, uint32_t tidY);                            /* PackXYBlock (integrator_pt_host.cpp:19-27) */
int  hpt_get_packed_xy(hpt_ctx* ctx, uint32_t* out, uint32_t count);
int  hpt_init_random_gens(hpt_ctx* ctx, uint32_t maxThreads);                            /* InitRandomGens (integrator_pt.cpp:13-21) */
/* The same seeding (RandomGenInit, crandom.h:25-36) for thread ids firstSeed .. firstSeed + maxThreads - 1: rank r of a sample-sharded
 * multi-GPU render seeds its generators as threads r * maxThreads ... of one big InitRandomGens call, i.e. decorrelated sub-streams. */
int  hpt_init_random_gens_from(hpt_ctx* ctx, uint32_t maxThreads, uint32_t firstSeed);
int  hpt_get_random_gens(hpt_ctx* ctx, uint32_t* outUint2, uint32_t count);              /* m_randomGens is integrator-owned, device copy authoritative */
int  hpt_set_random_gens(hpt_ctx* ctx, const uint32_t* inUint2, uint32_t count);

/* ---- the hot path --------------------------------------------------------------------------------------------- */
/* Integrator::PathTraceBlock(tid, channels, out_color, a_passNum) (integrator_pt.h:260, integrator_pt_host.cpp:57-73).
 * Processes tid in [tidBegin, tidBegin+tidCount) (the reference always passes the whole window: tidBegin = 0,
 * tidCount = W*H; a sub-range is what one rank of a multi-GPU job renders). out_color is the caller's full
 * W*H*channels host framebuffer; radiance is ACCUMULATED into it (+=), un-normalised, as in the reference. */
int  hpt_path_trace_block(hpt_ctx* ctx, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out_color, uint32_t passNum);
/* NaivePathTraceBlock (integrator_pt.h:259, integrator_pt_host.cpp:39-55) */
int  hpt_naive_path_trace_block(hpt_ctx* ctx, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out_color, uint32_t passNum);
/* Multi-GPU split of one frame (no reference counterpart: the reference is single-device). With stride > 1 the k-th work item of a
 * *_block call renders tid = tidBegin + (k / chunk) * chunk * stride + k % chunk, i.e. a rank passes tidBegin = rank * chunk and
 * renders every stride-th chunk of the tile-swizzled tid space (items past the viewport are dropped). chunk must be a multiple of 64
 * (one 8x8 tile). stride <= 1 restores the contiguous window [tidBegin, tidBegin + tidCount). */
int  hpt_set_tid_interleave(hpt_ctx* ctx, uint32_t chunk, uint32_t stride);
/* Same with the framebuffer already resident in device memory (kernel_slicer's generated class keeps out_color on the
 * device between calls, kmake_mega.json:18). stream is a hipStream_t (NULL = default stream); asynchronous. */
int  hpt_path_trace_block_dev(hpt_ctx* ctx, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out_color_dev,
                              uint32_t passNum, int naive, void* stream);
/* PathTraceFromInputRaysBlock(tid, channels, in_rayPosAndNear, in_rayDirAndFar, out_color, a_passNum) (integrator_pt.h:261,
 * integrator_pt.cpp:159-199, 761-798): the path tracer fed with the caller's rays (RayPosAndW / RayDirAndT, 16 bytes each, camera space:
 * cam_plugin/CamPluginAPI.h:27-37) instead of camera rays; out_color[tid * channels ..] += raw accumColor, m_randomGens[tid] advanced.
 * Needs hpt_init_random_gens(>= tid); no PackXYBlock. Host-pointer and device-pointer forms. */
int  hpt_path_trace_from_input_rays_block(hpt_ctx* ctx, uint32_t tid, uint32_t channels, const float* rayPosAndW, const float* rayDirAndT, float* out_color, uint32_t passNum);
int  hpt_path_trace_from_input_rays_block_dev(hpt_ctx* ctx, uint32_t tid, uint32_t channels, const float* rayPosAndWDev, const float* rayDirAndTDev, float* outDev, uint32_t passNum, void* stream);

/* ---- differentiable rendering (diff_render/integrator_dr.h:42-47, 103) ------------------------------------------ */
int  hpt_put_diff_tex2d(hpt_ctx* ctx, uint32_t texId, uint32_t width, uint32_t height, uint32_t channels,
                        uint64_t* outOffset, uint64_t* outSize);                         /* PutDiffTex2D (integrator_dr.cpp:33-53) */
int  hpt_reset_diff_tex(hpt_ctx* ctx);                                                   /* LoadSceneEnd (integrator_dr.cpp:24-31) */
/* IntegratorDR::PathTraceDR (integrator_dr.cpp:1135-1218). a_dataGrad is overwritten (zeroed first), out_color accumulated,
 * *outLoss receives the value the reference returns. Host-pointer form. */
int  hpt_path_trace_dr(hpt_ctx* ctx, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out_color, uint32_t passNum,
                       const float* refImg, const float* data, float* dataGrad, size_t gradSize, float* outLoss);
/* Device-pointer form: all four arrays resident; *lossAccumDev (one float, device) receives sum over samples of loss / passNum
 * (divide by W*H for the reference's return value); dataGradDev is ACCUMULATED into (zero it yourself). Asynchronous. */
int  hpt_path_trace_dr_dev(hpt_ctx* ctx, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out_color_dev, uint32_t passNum,
                           const float* refImgDev, const float* dataDev, float* dataGradDev, size_t gradSize, float* lossAccumDev, void* stream);
/* AdamOptimizer<float>::step (diff_render/adam.h:43-62) on device arrays. */
int  hpt_adam_step_dev(hpt_ctx* ctx, float* stateDev, const float* gradDev, float* momentumDev, float* gSquareDev, size_t n, int iter, void* stream);
/* Image2D4fRegularizer(w, h, data, grad) (diff_render/integrator_dr.cpp:317-367; drmain.cpp:213-217): grad += d RegLossImage2D4f / d data,
 * the total-variation-like texture regulariser (rgb of a w x h x 4 texture), hand-derived instead of __enzyme_autodiff. Device pointers. */
int  hpt_image2d4f_regularizer_dev(hpt_ctx* ctx, int w, int h, const float* data, float* grad, void* stream);
int  hpt_image2d4f_regularizer(hpt_ctx* ctx, int w, int h, const float* data, float* grad);       /* host pointers, the reference's signature */

/* ---- timing / instrumentation ----------------------------------------------------------------------------------- */
/* GetExecutionTime(name, out[4]) (integrator_pt.h:266, main.cpp:416-419): out[0] exec ms, [1] host->device, [2] device->host, [3] overhead */
int  hpt_get_execution_time(hpt_ctx* ctx, const char* funcName, float out[4]);
/* Counters of the last hpt_*_block call made with instrumentation enabled: [0..7] = {rays, nodesVisited, trisTested,
 * surfaceHits, shadowRays, paths, instancesEntered, texFetches} (feeds the algorithmic-bytes roofline);
 * [8..12] = wave-cycles (s_memtime) spent in {queue+regeneration, closest-hit traversal, shading, shadow traversal, path end},
 * [13] = bounce-loop trips summed over waves, [14] / [15] = wave-level iterations of the inner-node / triangle loops
 * (lane utilisation of traversal = [1] / (64 * [14]), [2] / (64 * [15])). The instrumented kernel is a separate build: never timed. */
int  hpt_set_instrumentation(hpt_ctx* ctx, int enabled);
int  hpt_get_counters(hpt_ctx* ctx, uint64_t out[16]);
/* The instrumented PathTraceDR launch (hpt_set_instrumentation(1) before hpt_path_trace_dr*; megakernel schedule) also fills: out[0] = adjoint
 * records stored (one per bounce), out[1] = of those with a parameter texture (gradient taps), out[2] / out[3] = wave-cycles of the record
 * stores / of the reverse sweeps (same clock as hpt_get_counters' phase slots, whose "path end" slot then excludes them), out[4] = wave-trips
 * that ran a sweep, out[5] = lanes in those sweeps, out[6] = atomic wave-instructions issued, out[7] = bounces the sweeps walked,
 * out[8] = records written to HBM (a path's last record stays in registers when the path ends in the same trip); out[9..15] reserved (0).
 * No counterpart in the reference (diff_render/integrator_dr.cpp:1135-1218 has no profile hooks). */
int  hpt_get_dr_counters(hpt_ctx* ctx, uint64_t out[16]);
/* Launch geometry of the persistent kernel: blocks per CU (0 = automatic). */
int  hpt_set_launch_config(hpt_ctx* ctx, int blocksPerCU);
/* Acceleration-structure layout chosen at the next hpt_commit_scene: 0 = automatic (the triangle sweep for scenes of <= 32 instanced
 * triangles, one world-space BVH2 over all instanced triangles for heavy static scenes, scenes from 4 096 instanced triangles and scenes of
 * six or more instances, else two-level), 1 = force the two-level TLAS/BLAS layout, 2 = force the single-level one, 3 = force the sweep
 * (no tree: every wave tests every triangle, fetched with scalar loads; cost linear in the triangle count).
 * All layouts intersect triangles in object space with the same arithmetic and return bit-identical hits. */
int  hpt_set_accel_layout(hpt_ctx* ctx, int layout);
/* How hpt_path_trace_block(_dev) schedules the work: 1 = one persistent megakernel (a lane keeps its path from camera to end),
 * 2 = wavefront (a shade kernel and a persistent trace kernel with ballot/prefix-sum ray compaction and ray replacement, path state
 * in HBM), 3 = the megakernel with block-local ray repacking (the rays of a workgroup's 256 lanes pooled in LDS and drained by its four
 * waves with replacement; gltf / emissive scenes, PathTraceDR and spectral rendering), 4 = the wavefront schedule in one launch (every workgroup
 * owns a tile-interleaved set of pool slots for the whole call and alternates between shading them and draining its own ray queue; heavy static gltf /
 * emissive scenes on the 4-wide tree - elsewhere the automatic choice is taken; never chosen automatically), 0 = automatic: wavefront when the committed BVH is
 * expected to cost a ray >= 20 inner-node visits (hpt_get_accel_info's surface-area estimate) and the call has >= 2^19 pixels, the
 * block-local form for lean scenes from an estimate of 8, else the plain megakernel. All give bit-identical frames. refillBelow (1..64,
 * 0 = keep): a trace wave refills from the ray queue when fewer lanes than this still hold a ray; traceBlocksPerCU 0 = automatic.
 * groups (0 = automatic): the pixels of a call are cut into this many groups with their own path pool, ray queue and HIP stream, so
 * that the tail of one group's trace pass (a few long rays) overlaps the other groups' shade and trace passes.
 * The naive integrator and input-ray batches always use the megakernel (scenes with moving instances run under either); PathTraceDR follows the same
 * automatic choice as PathTraceBlock (its wavefront form keeps the adjoint records per pool slot), and so does spectral rendering (m_spectral_mode = 1:
 * its own shade kernel with four samples per radiance word in front of the same trace kernel). The wavefront call returns once the frame is nearly done
 * (it polls a device progress word); results are complete after the stream is synchronised, as for the megakernel. */
int  hpt_set_schedule(hpt_ctx* ctx, int schedule, int refillBelow, int traceBlocksPerCU, int groups);
/* Tuning knobs without a place in the reference's interface (results never depend on them):
 *   "wf_grace"  trips a wavefront trace wave keeps going after the ray queue ran dry before it parks its unfinished rays (traversal
 *               state + stack to HBM) for the next round's trace pass, which resumes them first; 0 = run every ray to the end.
 *               Only applied in rounds with at least 2 rays per lane of the trace grid.
 *   "refit"     1 (default): CommitScene after UpdateInstance / UpdateGeom_Triangles3f refits the single-level tree on the device; 0: rebuild.
 *   "node_min"  voted exit of the inner-node loop (0..63, applied at the next hpt_commit_scene; default chosen per scene).
 * Diagnostic switches (these DO change which kernels run or what they compute; never set in production):
 *   "dr_skip_nonfinite"     PathTraceDR: 1 = a sample whose radiance is not finite contributes neither colour, loss nor gradient (what an
 *                           optimisation loop wants: one NaN poisons Adam's moments for good); 0 (default) = PixelLossPT as the reference
 *                           has it, which adds every sample (diff_render/integrator_dr.cpp:1124-1131)
 *   "shade_records"         0: gltf / emissive scenes on the single-level layout gather vertex data through the index chain instead of the 64-byte
 *                           per-triangle shading records (kernel studies; environment: HPT_SHADE_RECORDS=0)
 *   "build_threads"         n: host threads CommitScene builds its trees with (meshes in parallel, big trees split into subtrees); 0 = the cores this
 *                           process may use, at most 16. The tree is the same whatever n is (only the numbering of its nodes differs)
 *   "wide_nodes"            0: the wavefront trace kernel walks the BVH2 instead of the 4-wide compressed tree of static single-level scenes
 *                           (kernel studies; environment: HPT_WIDE_NODES=0)
 *   "stats_wide"            1: the instrumented probe (hpt_set_instrumentation) counts the walk over the 4-wide tree
 *   "force_full_materials"  1: never pick the kernels specialised for gltf + emissive scenes (kernel studies)
 *   "dbg_no_normal_lerp"    1: moving instances without the reference's normal interpolation (integrator_pt.cpp:285-292); the checker has the
 *                           same switch (ORC_DBG_NO_NORMAL_LERP) - used to show where the rare path divergences under motion blur come from */
int  hpt_set_option(hpt_ctx* ctx, const char* name, int value);
/* ---- multi-GPU: one context = one GPU = one rank -----------------------------------------------------
 * The path shards without a data-path exchange; these three calls are the only traffic over xGMI. RCCL is loaded on first use.
 *   hpt_comm_get_unique_id : ncclGetUniqueId (128 bytes) on rank 0; the host distributes it (MPI, sockets, a file ...)
 *   hpt_comm_init          : ncclCommInitRank for this context's device
 *   hpt_reduce_framebuffer : in-place ncclReduce(sum, fp32) of the zero-initialised framebuffers to `root`, once per frame
 *   hpt_allreduce_grad     : in-place ncclAllReduce(sum, fp32) of a_dataGrad (or the one-float loss), once per optimisation iteration */
int  hpt_comm_get_unique_id(hpt_ctx* ctx, void* id128);
int  hpt_comm_init(hpt_ctx* ctx, int nranks, int rank, const void* id128);
int  hpt_comm_destroy(hpt_ctx* ctx);
int  hpt_reduce_framebuffer(hpt_ctx* ctx, float* frameDev, size_t count, int root, void* stream);
int  hpt_allreduce_grad(hpt_ctx* ctx, float* gradDev, size_t count, void* stream);
/* Schedule the last hpt_path_trace_block(_dev) call used (1 / 2) and, for the wavefront one, its number of shade+trace rounds. */
/* What CommitScene built: out[0] = expected inner-node visits per ray (surface-area estimate over the committed BVH; the quantity the automatic
 * schedule / layout choice is measured against), out[1] = instanced triangles, out[2] = instances, out[3] = layout (0 two-level, 1 single-level, 2 triangle sweep). */
int  hpt_get_accel_info(hpt_ctx* ctx, float out[4]);
/* The last hpt_commit_scene: out[0] = host milliseconds (BVH build, or for a refit the matrix inversions and record updates), out[1] = upload ms,
 * out[2] = device refit ms, out[3] = 1 when the committed single-level tree was REFITTED on the device (UpdateInstance / UpdateGeom_Triangles3f
 * with unchanged topology: boxes recomputed bottom-up by two small kernels, CrossRT.h:85-86, 134), 2 when it was BUILT on the device (out[2] = the
 * build's milliseconds) and 0 when it was built on the host. hpt_set_option("refit", 0) forces the build. */
int  hpt_get_commit_time(hpt_ctx* ctx, float out[4]);
int  hpt_get_schedule(hpt_ctx* ctx, int* lastSchedule, uint32_t* lastIterations);
/* What the last hpt_path_trace_* call walked (no counterpart in the reference: lets a test assert WHICH kernels produced a frame):
 * out[0] = schedule (1 / 2), out[1] = 1 when rays walked the 4-wide compressed tree, out[2] = 1 when surface data came from the 64-byte
 * shading records, out[3] = 1 when the traversal stacks had an HBM part. */
int  hpt_get_last_launch(hpt_ctx* ctx, uint32_t out[4]);
/* Duration of the last path-tracing kernel, measured with HIP events on the stream it ran on (ms). */
int  hpt_last_kernel_ms(hpt_ctx* ctx, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* HYDRA_HIP_H */
