// Host-side C++ adapter over the C ABI (include/hydra_hip.h), shaped like the reference's own classes so that it
// drops in where HydraCore3 creates its integrator (main.cpp:194-235, diff_render/drmain.cpp:126, hydra_api/hydra_cpu.cpp:8).
//
//   hydra_hip::BVH2SceneHIP   <->  ISceneObject                      (external/CrossRT/CrossRT.h:45-176)
//   hydra_hip::IntegratorHIP  <->  Integrator                        (integrator_pt.h:123-703): the virtual hooks the
//                                  kernel_slicer-generated Integrator_Generated overrides (main.cpp:221-224) --
//                                  PathTraceBlock, NaivePathTraceBlock, PackXYBlock, CommitDeviceData,
//                                  UpdateMembersPlainData, GetExecutionTime, Update_m_materials / Update_m_lights
//   hydra_hip::IntegratorDRHIP <-> IntegratorDR                      (diff_render/integrator_dr.h:27-136)
//
// The reference's headers cannot be included here (they need the absent LiteMath), so this header carries the same
// member names over plain arrays; INTEGRATION.md shows the dozen lines that derive the real `Integrator` from it.
// Header-only; link against hydracore3_amd/libhydra_hip.so. No exceptions cross the C boundary: failures print the
// library's message and follow the reference's convention (message + return, integrator_pt_scene.cpp:85-89).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/hydra_hip.h"

namespace hydra_hip {

struct float4x4 { float m[16]; };     // column-major, as LiteMath::float4x4::m_col

struct Material                        // include/cmaterial.h:187-203
{
  uint32_t mtype, cflags, lightId, nonlinear;
  uint32_t texid[4], spdid[4], datai[4];
  float colors[4][4], row0[4][4], row1[4][4], data[16];
};
struct LightSource                     // include/clight.h:19-56
{
  float matrix[16], iesMatrix[16], samplerRow0[4], samplerRow1[4], samplerRow0Inv[4], samplerRow1Inv[4], pos[4], intensity[4], norm[4];
  float size[2]; float pdfA; uint32_t geomType, distType, flags, pdfTableOffset, pdfTableSize, specId, texId, iesId; float mult;
  uint32_t pdfTableSizeX, pdfTableSizeY, camBackTexId; float lightCos1, lightCos2; uint32_t matId; float dummy2, dummy3;
};
static_assert(sizeof(Material) == 320 && sizeof(LightSource) == 320, "records must match the reference layout");

struct CRT_Hit { float t; uint32_t primId, instId, geomId; float coords[4]; };   // CrossRT.h:23-30

// ---- ISceneObject-shaped wrapper of the BVH2 builder + traversal ---------------------------------------------------------
class BVH2SceneHIP
{
public:
  explicit BVH2SceneHIP(hpt_ctx* ctx) : m_ctx(ctx) {}
  const char* Name() const { return "BVH2SceneHIP"; }
  void     ClearGeom() { hpt_clear_geom(m_ctx); }
  uint32_t AddGeom_Triangles3f(const float* a_vpos3f, size_t a_vertNumber, const uint32_t* a_triIndices, size_t a_indNumber,
                               uint32_t a_flags = 4 /*BUILD_HIGH*/, size_t vByteStride = sizeof(float) * 3)
  { return hpt_add_geom_triangles3f(m_ctx, a_vpos3f, a_vertNumber, a_triIndices, a_indNumber, a_flags, vByteStride); }
  void     UpdateGeom_Triangles3f(uint32_t a_geomId, const float* a_vpos3f, size_t a_vertNumber, const uint32_t* a_triIndices, size_t a_indNumber,
                                  uint32_t a_flags = 4, size_t vByteStride = sizeof(float) * 3)
  { hpt_update_geom_triangles3f(m_ctx, a_geomId, a_vpos3f, a_vertNumber, a_triIndices, a_indNumber, a_flags, vByteStride); }
  void     ClearScene() { hpt_clear_scene(m_ctx); }
  uint32_t AddInstance(uint32_t a_geomId, const float4x4& a_matrix) { return hpt_add_instance(m_ctx, a_geomId, a_matrix.m); }
  uint32_t AddInstanceMotion(uint32_t a_geomId, const float4x4* a_matrices, uint32_t a_matrixNumber) { return hpt_add_instance_motion(m_ctx, a_geomId, a_matrices[0].m, a_matrixNumber); }
  void     UpdateInstance(uint32_t a_instanceId, const float4x4& a_matrix) { hpt_update_instance(m_ctx, a_instanceId, a_matrix.m); }
  void     CommitScene(uint32_t options = 4) { hpt_commit_scene(m_ctx, options); }
  // single-ray forms of the reference interface; the batched C entry points are what a throughput caller should use
  CRT_Hit  RayQuery_NearestHit(const float posAndNear[4], const float dirAndFar[4])
  { CRT_Hit h; std::memset(&h, 0xFF, sizeof(h)); hpt_ray_query_nearest(m_ctx, posAndNear, dirAndFar, 1, reinterpret_cast<hpt_hit*>(&h)); return h; }
  bool     RayQuery_AnyHit(const float posAndNear[4], const float dirAndFar[4])
  { uint32_t r = 0; hpt_ray_query_any(m_ctx, posAndNear, dirAndFar, 1, &r); return r != 0; }
  // the *Motion forms (CrossRT.h:157,174): moving instances are evaluated at `time` in [0, 1]
  CRT_Hit  RayQuery_NearestHitMotion(const float p[4], const float d[4], float time)
  { CRT_Hit h; std::memset(&h, 0xFF, sizeof(h)); hpt_ray_query_nearest_motion(m_ctx, p, d, 1, time, reinterpret_cast<hpt_hit*>(&h)); return h; }
  bool     RayQuery_AnyHitMotion(const float p[4], const float d[4], float time = 0.0f)
  { uint32_t r = 0; hpt_ray_query_any_motion(m_ctx, p, d, 1, time, &r); return r != 0; }
private:
  hpt_ctx* m_ctx;
};

struct TextureData { uint32_t width = 1, height = 1, format = 0, flags = 0, addressU = 2, addressV = 2, filter = 0; std::vector<uint8_t> texels; };

// ---- Integrator-shaped front end -----------------------------------------------------------------------------------------
class IntegratorHIP
{
public:
  explicit IntegratorHIP(int a_maxThreads = 1, int a_device = 0) : m_maxThreadId(uint32_t(a_maxThreads))
  {
    if (hpt_create(a_device, &m_ctx) != HPT_OK) { std::printf("[IntegratorHIP]: no HIP device %d\n", a_device); m_ctx = nullptr; return; }
    m_pAccelStruct = new BVH2SceneHIP(m_ctx);
    TextureData white; white.texels.assign(4, 0xFF);                      // m_textures[0]: white dummy (integrator_pt_scene_tex.cpp:7-16)
    m_textures.push_back(white);
  }
  virtual ~IntegratorHIP() { delete m_pAccelStruct; if (m_ctx) hpt_destroy(m_ctx); }
  IntegratorHIP(const IntegratorHIP&) = delete;
  IntegratorHIP& operator=(const IntegratorHIP&) = delete;
  bool valid() const { return m_ctx != nullptr; }

  // ---- settings (integrator_pt.h:363-411) ----
  void SetIntegratorType(uint32_t a_type) { m_intergatorType = a_type; }
  void SetFrameBufferSize(int a_width, int a_height) { m_fbWidth = a_width; m_fbHeight = a_height; }
  void SetFrameBufferLayer(uint32_t a_layer) { m_renderLayer = a_layer; }
  void SetViewport(int a_xStart, int a_yStart, int a_width, int a_height)
  {
    m_winStartX = a_xStart; m_winStartY = a_yStart; m_winWidth = a_width; m_winHeight = a_height;
    if (m_fbWidth == 0 || m_fbHeight == 0) SetFrameBufferSize(a_width, a_height);
    const int sizeX = a_width - a_xStart, sizeY = a_height - a_yStart;
    if (sizeX % 8 == 0 && sizeY % 8 == 0) m_tileSize = 8; else if (sizeX % 4 == 0 && sizeY % 4 == 0) m_tileSize = 4;
    else if (sizeX % 2 == 0 && sizeY % 2 == 0) m_tileSize = 2; else m_tileSize = 1;
    m_maxThreadId = uint32_t(a_width * a_height);
  }
  void SetWorldViewInv(const float4x4& m) { m_worldViewInv = m; }
  void SetProjInv(const float4x4& m) { m_projInv = m; }
  void SetAccelStruct(BVH2SceneHIP*) {}                                    // the accelerator is part of the context

  // ---- hooks of the generated class ----
  virtual void CommitDeviceData()
  {
    if (!m_ctx) return;
    std::vector<hpt_texture_desc> td(m_textures.size());
    for (size_t i = 0; i < td.size(); i++) {
      const TextureData& t = m_textures[i];
      td[i] = hpt_texture_desc{ t.width, t.height, t.format, t.flags, t.addressU, t.addressV, t.filter, 0u, t.texels.data() };
    }
    std::vector<uint32_t> instGeom(m_instGeomId);
    hpt_scene_desc d; std::memset(&d, 0, sizeof(d));
    // with motion blur m_normMatrices holds a second half for the end of the motion, starting at m_normMatrices2Offs (= the instance count)
    d.numGeoms = uint32_t(m_matVertOffset.size() / 2); d.numInsts = m_normMatrices2Offs ? m_normMatrices2Offs : uint32_t(m_normMatrices.size());
    d.numVerts = uint32_t(m_vData8f.size() / 8); d.numTris = uint32_t(m_matIdByPrimId.size());
    d.vPos4f = nullptr;                                                    // geometry went in through m_pAccelStruct (LoadSceneGeometry)
    d.vData8f = m_vData8f.data(); d.triIndices = m_triIndices.data(); d.matIdByPrimId = m_matIdByPrimId.data();
    d.matVertOffset = m_matVertOffset.data(); d.geomTriCount = nullptr; d.geomVertCount = nullptr;
    d.instGeomId = instGeom.data(); d.instMatrices = nullptr; d.normMatrices = reinterpret_cast<const float*>(m_normMatrices.data());
    d.remapInst = m_remapInst.data(); d.allRemapLists = m_allRemapLists.data();
    d.allRemapListsLen = uint32_t(m_allRemapLists.size()); d.allRemapListsSize = m_allRemapListsSize;
    d.materials = m_materials.data(); d.numMaterials = uint32_t(m_materials.size());
    d.lights = m_lights.data(); d.numLights = uint32_t(m_lights.size());
    d.textures = td.data(); d.numTextures = uint32_t(td.size());
    d.arrays1f = m_arrays1f.empty() ? nullptr : m_arrays1f.data(); d.numArrays1f = uint32_t(m_arrays1f.size());   // pdf table of a sampled environment map, plastic tables
    d.normMatrices2Offs = m_normMatrices2Offs;                              // the moving instances themselves went in through AddInstanceMotion
    // spectral rendering: the tables LoadScene builds (integrator_pt_scene.cpp:358-419, 962-969), as the host holds them
    d.specValues = m_spec_values.empty() ? nullptr : m_spec_values.data(); d.numSpecValues = uint32_t(m_spec_values.size());
    d.specOffsetSz = m_spec_offset_sz.empty() ? nullptr : m_spec_offset_sz.data(); d.numSpectra = uint32_t(m_spec_offset_sz.size() / 2);
    d.cieXYZ = m_cie_xyz.empty() ? nullptr : m_cie_xyz.data(); d.numCieXYZ = uint32_t(m_cie_xyz.size() / 4);
    for (int k = 0; k < 3; k++) d.camResponseSpectrumId[k] = m_camResponseSpectrumId[k];
    d.camResponseType = uint32_t(m_camResponseType);
    // thin films: what LoadThinFilmMaterial appended (integrator_pt.h:587-590)
    d.filmsThickness = m_films_thickness_vec.empty() ? nullptr : m_films_thickness_vec.data(); d.numFilmsThickness = uint32_t(m_films_thickness_vec.size());
    d.filmsSpecId = m_films_spec_id_vec.empty() ? nullptr : m_films_spec_id_vec.data(); d.numFilmsSpecId = uint32_t(m_films_spec_id_vec.size());
    d.filmsEtaK = m_films_eta_k_vec.empty() ? nullptr : m_films_eta_k_vec.data(); d.numFilmsEtaK = uint32_t(m_films_eta_k_vec.size());
    d.precompThinFilms = m_precomp_thin_films.empty() ? nullptr : m_precomp_thin_films.data(); d.numPrecompThinFilms = uint32_t(m_precomp_thin_films.size());
    if (!m_spec_tex_ids_wavelengths.empty() && m_spec_tex_offset_sz.size() == m_spec_offset_sz.size()) {   // spectra given by textures (integrator_pt.h: uint2 vectors)
      d.specTexIdsWavelengths = m_spec_tex_ids_wavelengths.data(); d.numSpecTexBands = uint32_t(m_spec_tex_ids_wavelengths.size() / 2); d.specTexOffsetSz = m_spec_tex_offset_sz.data();
    }
    report(hpt_upload_scene(m_ctx, &d), "CommitDeviceData");
    if (m_randomGensInit != m_maxThreadId) { hpt_init_random_gens(m_ctx, m_maxThreadId); m_randomGensInit = m_maxThreadId; }   // InitRandomGens
  }
  virtual void UpdateMembersPlainData()
  {
    if (!m_ctx) return;
    hpt_params p; std::memset(&p, 0, sizeof(p));
    std::memcpy(p.projInv, m_projInv.m, 64); std::memcpy(p.worldViewInv, m_worldViewInv.m, 64);
    p.winStartX = m_winStartX; p.winStartY = m_winStartY; p.winWidth = m_winWidth; p.winHeight = m_winHeight; p.fbWidth = m_fbWidth; p.fbHeight = m_fbHeight;
    p.traceDepth = m_traceDepth; p.integratorType = m_intergatorType; p.renderLayer = m_renderLayer; p.tileSize = m_tileSize; p.spectralMode = uint32_t(m_spectral_mode);
    p.envSpecIdPlus1 = m_envSpecId + 1u; p.envSpecMult = m_envSpecMult;     // (uint(-1) + 1 = 0: none)
    p.exposureMult = m_exposureMult; p.camLensRadius = m_camLensRadius; p.camTargetDist = m_camTargetDist;
    std::memcpy(p.camRespoceRGB, m_camRespoceRGB, 16); std::memcpy(p.envColor, m_envColor, 16);
    p.envTexId = m_envTexId; p.envLightId = m_envLightId; p.envCamBackId = m_envCamBackId; p.envEnableSam = m_envEnableSam;
    std::memcpy(p.envSamRow0, m_envSamRow0, 16); std::memcpy(p.envSamRow1, m_envSamRow1, 16);
    report(hpt_update_params(m_ctx, &p), "UpdateMembersPlainData");
    // m_lines / m_physSize / m_enableOpticSim (integrator_pt.h:206-228, 353-362): the lens simulation of SampleCameraRay
    report(hpt_set_optics(m_ctx, m_enableOpticSim && !m_lines.empty() ? &m_lines[0].curvatureRadius : nullptr,
                          m_enableOpticSim ? uint32_t(m_lines.size()) : 0u, m_physSize[0], m_physSize[1]), "UpdateMembersPlainData (optics)");
  }
  struct LensElementInterface { float curvatureRadius, thickness, eta, apertureRadius; };     // integrator_pt.h:206-212
  void SetLines(const std::vector<LensElementInterface>& a_lines) { m_lines = a_lines; }      // :356-362
  void SetPhysSize(float x, float y) { m_physSize[0] = x; m_physSize[1] = y; }               // :354
  uint32_t m_enableOpticSim = 0;
  std::vector<LensElementInterface> m_lines;
  float m_physSize[2] = {0, 0};
  virtual void PackXYBlock(uint32_t tidX, uint32_t tidY, uint32_t /*a_passNum*/)
  { if (m_ctx) { UpdateMembersPlainData(); report(hpt_pack_xy(m_ctx, tidX, tidY), "PackXYBlock"); } }
  virtual void PathTraceBlock(uint32_t tid, uint32_t channels, float* out_color, uint32_t a_passNum)
  { if (m_ctx) report(hpt_path_trace_block(m_ctx, 0, tid, channels, out_color, a_passNum), "PathTraceBlock"); }
  virtual void NaivePathTraceBlock(uint32_t tid, uint32_t channels, float* out_color, uint32_t a_passNum)
  { if (m_ctx) report(hpt_naive_path_trace_block(m_ctx, 0, tid, channels, out_color, a_passNum), "NaivePathTraceBlock"); }
  // cam_plugin/CamPluginAPI.h:27-37: both structs are 16 bytes (origin + wavelength, direction + time), camera space
  struct RayPosAndW { float origin[3]; float wave; };
  struct RayDirAndT { float direction[3]; float time; };
  virtual void PathTraceFromInputRaysBlock(uint32_t tid, uint32_t channels, const RayPosAndW* in_rayPosAndNear, const RayDirAndT* in_rayDirAndFar,
                                           float* out_color, uint32_t a_passNum)
  { if (m_ctx) report(hpt_path_trace_from_input_rays_block(m_ctx, tid, channels, (const float*)in_rayPosAndNear, (const float*)in_rayDirAndFar, out_color, a_passNum), "PathTraceFromInputRaysBlock"); }
  virtual void GetExecutionTime(const char* a_funcName, float a_out[4]) { if (m_ctx) hpt_get_execution_time(m_ctx, a_funcName, a_out); }
  virtual void Update_m_materials(size_t a_first, size_t a_count) { if (m_ctx) report(hpt_update_materials(m_ctx, a_first, a_count, m_materials.data() + a_first), "Update_m_materials"); }
  virtual void Update_m_lights(size_t a_first, size_t a_count) { if (m_ctx) report(hpt_update_lights(m_ctx, a_first, a_count, m_lights.data() + a_first), "Update_m_lights"); }
  virtual void Update_m_matIdOffsets() { if (m_ctx) report(hpt_update_mat_id_offsets(m_ctx, m_matVertOffset.data(), m_matVertOffset.size() / 2), "Update_m_matIdOffsets"); }   // integrator_pt.h:470
  virtual void SceneRestrictions(uint32_t a_restrictions[4]) const
  { a_restrictions[0] = 1u << 20; a_restrictions[1] = 1u << 27; a_restrictions[2] = (1u << 28) - 1u; a_restrictions[3] = (1u << 28) - 1u; }   // 28-bit triangle references

  // ---- scene vectors, named as in integrator_pt.h:472-500 ----
  std::vector<Material>    m_materials;
  std::vector<LightSource> m_lights;
  std::vector<uint32_t>    m_matVertOffset;    // uint2 per geom
  std::vector<uint32_t>    m_matIdByPrimId, m_triIndices;
  std::vector<float>       m_vData8f;          // 8 floats per vertex
  std::vector<int>         m_allRemapLists, m_remapInst;   // m_remapInst: int2 per instance
  uint32_t                 m_allRemapListsSize = 0;
  std::vector<float4x4>    m_normMatrices;
  uint32_t                 m_normMatrices2Offs = 0;   // integrator_pt.h:498: where the end-of-motion normal matrices start (0: no motion blur)
  std::vector<float>       m_arrays1f;                // integrator_pt.h:485
  std::vector<float>       m_spec_values;             // integrator_pt.h:580: every spectrum at 1 nm from LAMBDA_MIN
  std::vector<uint32_t>    m_spec_offset_sz;          // integrator_pt.h:581: uint2 {offset, size} per spectrum
  std::vector<float>       m_cie_xyz;                 // integrator_pt.h:585: float4 {x, y, z, 0} per nm, 471 entries
  int                      m_camResponseSpectrumId[3] = {-1, -1, -1};   // integrator_pt.h:533
  int                      m_camResponseType = 0;     // integrator_pt.h:534: 0 = CAM_RESPONCE_XYZ, 1 = CAM_RESPONCE_RGB
  std::vector<uint32_t>    m_spec_tex_ids_wavelengths, m_spec_tex_offset_sz;   // spectra given by textures: uint2 each (LoadSceneSpectrumData, integrator_pt_scene.cpp:363-377)
  std::vector<float>       m_films_thickness_vec;     // integrator_pt.h:587-590: thin films
  std::vector<uint32_t>    m_films_spec_id_vec;
  std::vector<float>       m_films_eta_k_vec;
  std::vector<float>       m_precomp_thin_films;
  std::vector<uint32_t>    m_instGeomId;
  std::vector<TextureData> m_textures;
  BVH2SceneHIP*            m_pAccelStruct = nullptr;

  float4x4 m_projInv{}, m_worldViewInv{};
  int      m_winStartX = 0, m_winStartY = 0, m_winWidth = 0, m_winHeight = 0, m_fbWidth = 0, m_fbHeight = 0;
  uint32_t m_traceDepth = 10, m_renderLayer = 0, m_spp = 1024, m_tileSize = 8, m_maxThreadId = 0;
  uint32_t m_intergatorType = 0;
  int      m_spectral_mode = 0;
  float    m_exposureMult = 1.0f, m_camLensRadius = 0.0f, m_camTargetDist = 0.0f;
  float    m_camRespoceRGB[4] = {1, 1, 1, 1}, m_envColor[4] = {0, 0, 0, 0};
  uint32_t m_envTexId = 0xFFFFFFFFu, m_envLightId = 0xFFFFFFFFu, m_envCamBackId = 0xFFFFFFFFu, m_envEnableSam = 0;   // integrator_pt.h (environment map)
  float    m_envSamRow0[4] = {1, 0, 0, 0}, m_envSamRow1[4] = {0, 1, 0, 0};
  uint32_t m_envSpecId = 0xFFFFFFFFu; float m_envSpecMult = 1.0f;          // integrator_pt.h:524-525: the environment's spectrum (spectral mode)

  hpt_ctx* context() const { return m_ctx; }

protected:
  void report(int rc, const char* where) const { if (rc != HPT_OK) std::printf("[IntegratorHIP::%s]: %s\n", where, hpt_last_error(m_ctx)); }
  hpt_ctx* m_ctx = nullptr;
  uint32_t m_randomGensInit = 0;
};

// ---- IntegratorDR-shaped front end (diff_render/integrator_dr.h:27-136) -------------------------------------------------------
class IntegratorDRHIP : public IntegratorHIP
{
public:
  explicit IntegratorDRHIP(int a_maxThreads = 1, int a_device = 0) : IntegratorHIP(a_maxThreads, a_device) {}
  void LoadSceneEnd() { if (m_ctx) hpt_reset_diff_tex(m_ctx); }
  std::pair<size_t, size_t> PutDiffTex2D(uint32_t texId, uint32_t width, uint32_t height, uint32_t channels)
  {
    uint64_t off = 0, size = 0;
    if (!m_ctx || hpt_put_diff_tex2d(m_ctx, texId, width, height, channels, &off, &size) != HPT_OK) return std::make_pair(size_t(-1), size_t(0));
    return std::make_pair(size_t(off), size_t(size));
  }
  virtual void SetMaxThreadsAndBounces(int /*a_maxThreads*/, int a_maxBounce) { m_traceDepth = uint32_t(a_maxBounce); }   // no per-CPU-thread records on the GPU
  float PathTraceDR(uint32_t tid, uint32_t channels, float* out_color, uint32_t a_passNum,
                    const float* a_refImg, const float* a_data, float* a_dataGrad, size_t a_gradSize)
  {
    float loss = 0.0f;
    if (m_ctx) report(hpt_path_trace_dr(m_ctx, 0, tid, channels, out_color, a_passNum, a_refImg, a_data, a_dataGrad, a_gradSize, &loss), "PathTraceDR");
    return loss;
  }
  // Image2D4fRegularizer (diff_render/integrator_dr.cpp:361-367; drmain.cpp:213-217): grad += d RegLossImage2D4f / d data
  void Image2D4fRegularizer(int w, int h, const float* data, float* grad)
  { if (m_ctx) report(hpt_image2d4f_regularizer(m_ctx, w, h, data, grad), "Image2D4fRegularizer"); }
};

} // namespace hydra_hip
