// The HR2 render-driver leg on the HIP core: hydra_api/hydra_cpu.h:20-28 (struct IRenderDriver), hydra_cpu.cpp:4-107
// (HydraCore3RenderDriver): an API client that holds its meshes in memory hands them to the driver as POINTERS (RDScene_Input::pMeshPtrs,
// hydra_api's CommandBuffer::CommitToStorage :109-127), the driver converts them to Mesh4fInput and calls LoadScene; Render() is
// SetFrameBufferSize -> SetViewport -> UpdateMembersPlainData -> PackXYBlock -> the block call. Same class and member names; the scene
// description arrives as XML text + folder (the reference passes its parsed hydra_xml::HydraScene, which lives in LiteScene and stays there).
// The reference's Render currently calls CastSingleRayBlock with the PathTraceBlock call commented out (hydra_cpu.cpp:105-106); this driver makes the
// PathTraceBlock call - the hot path this repository replaces.
#pragma once
#include <memory>
#include <string>
#include <unordered_map>
#include "scene_loader.h"

namespace hydra_hip {
namespace HR2 {

// one mesh of the client, by pointers (hydra_api.h: the value type of CommandBuffer::meshPtrById; strides in floats as hydra_cpu.cpp:38-45 reads them)
struct MeshPointers
{
  const float* vPosPtr = nullptr; uint32_t vPosStride = 4;
  const float* vNormPtr = nullptr; uint32_t vNormStride = 4;
  const float* vTangPtr = nullptr; uint32_t vTangStride = 4;
  const float* vTexCoordPtr = nullptr; uint32_t vTexCoordStride = 2;
  uint32_t vertNum = 0;
  const uint32_t* indicesPtr = nullptr; uint32_t indicesNum = 0;
  const uint32_t* matIdPtr = nullptr; uint32_t matIdAll = 0, matIdNum = 1;
};
struct RDScene_Input { const std::unordered_map<int, MeshPointers>* pMeshPtrs = nullptr; uint32_t geomInstNum = 0, lghtInstNum = 0; };

// update flags as hydra_cpu.cpp:31-32 derives them from hydra_xml's object kinds: 1 << (XML_OBJ_GEOMETRY + 1), 1 << (XML_OBJ_SCENE + 1) (hydraxml.h:37-48: GEOMETRY = 3, SCENE = 7)
static constexpr uint32_t SCN_UPDATE_GEOMETRY = 1u << 4, SCN_UPDATE_INSTANCES = 1u << 8, SCN_UPDATE_ALL = 0xFFFFFFFFu;

struct IRenderDriver
{
  virtual ~IRenderDriver() {}
  virtual bool LoadScene(const std::string& a_xmlText, const std::string& a_folder, const RDScene_Input& a_input, uint32_t a_updateFlags) = 0;
  virtual void CommitDeviceData() = 0;
  virtual void Render(uint32_t startX, uint32_t startY, int32_t sizeX, uint32_t sizeY, uint32_t channels, float* data, uint32_t a_passNumber) = 0;
};

struct HydraHipRenderDriver : IRenderDriver
{
  explicit HydraHipRenderDriver(int device = 0) { if (hpt_create(device, &m_ctx) != HPT_OK) m_ctx = nullptr; }
  ~HydraHipRenderDriver() override { if (m_ctx) hpt_destroy(m_ctx); }
  bool valid() const { return m_ctx != nullptr; }
  const std::string& lastError() const { return m_err; }

  bool LoadScene(const std::string& a_xmlText, const std::string& a_folder, const RDScene_Input& a_input, uint32_t a_updateFlags) override
  {
    if (a_updateFlags & SCN_UPDATE_GEOMETRY) {                               // "put / convert mesh pointers / data, then pass data to LoadScene" (hydra_cpu.cpp:34-73)
      m_meshPtrs.clear();
      if (a_input.pMeshPtrs) for (const auto& p : *a_input.pMeshPtrs) {
        const MeshPointers& mp = p.second;
        if (mp.vNormStride != 4 || mp.vTangStride != 4 || mp.vTexCoordStride != 2) { m_err = "[HydraHipRenderDriver::LoadScene]: unsupported normal / tangent / texcoord stride"; return false; }
        Mesh4fInput in;
        in.vPosPtr = mp.vPosPtr; in.vPosByteStride = mp.vPosStride * (uint32_t)sizeof(float);
        in.vNormPtr4f = mp.vNormPtr; in.vTangPtr4f = mp.vTangPtr; in.vTexCoord2f = mp.vTexCoordPtr; in.vertNum = mp.vertNum;
        in.indicesPtr = mp.indicesPtr; in.indicesNum = mp.indicesNum;
        in.matIdPtr = mp.matIdPtr; in.matIdAll = mp.matIdAll; in.matIdNum = mp.matIdNum;
        m_meshPtrs[p.first] = in;
      }
    }
    // Integrator::LoadScene(a_scn, a_updateFlags): the tables of the whole scene from the description + the mesh pointers
    if (!LoadHydraXmlText(a_xmlText, a_folder, 0, 0, m_scene, m_err, false, &m_meshPtrs)) return false;
    if ((a_updateFlags & SCN_UPDATE_INSTANCES) != 0 && a_input.geomInstNum != 0 && a_input.geomInstNum < m_scene.instGeomId.size()) {   // a_scn.m_numInstances = a_input.geomInstNum
      const size_t n = a_input.geomInstNum;
      m_scene.instGeomId.resize(n); m_scene.instMatrices.resize(16 * n); m_scene.normMatrices.resize(16 * n); m_scene.remapInst.resize(2 * n);
      m_scene.instMatricesMotion.resize(16 * n); m_scene.instHasMotion.resize(n);
    }
    m_loaded = true;
    return true;
  }

  void CommitDeviceData() override                                            // Integrator::CommitDeviceData: geometry, instances, CommitScene, every table
  {
    if (!m_ctx || !m_loaded) return;
    const int rc = m_scene.upload(m_ctx);
    if (rc != HPT_OK) m_err = hpt_last_error(m_ctx);
    m_committed = rc == HPT_OK;
  }

  void Render(uint32_t startX, uint32_t startY, int32_t sizeX, uint32_t sizeY, uint32_t channels, float* data, uint32_t a_passNumber) override
  {
    if (!m_ctx || !m_committed) return;
    // SetFrameBufferSize(sizeX, sizeY); SetViewport(startX, startY, sizeX, sizeY); UpdateMembersPlainData(); PackXYBlock(sizeX, sizeY, 1) (hydra_cpu.cpp:96-104)
    m_scene.width = sizeX; m_scene.height = (int)sizeY;
    hpt_params p = m_scene.params();
    p.fbWidth = sizeX; p.fbHeight = (int)sizeY; p.winStartX = (int)startX; p.winStartY = (int)startY; p.winWidth = sizeX; p.winHeight = (int)sizeY;
    {                                                                         // SetViewport's tile size (integrator_pt.h:379-389)
      const int sx = sizeX - (int)startX, sy = (int)sizeY - (int)startY;
      p.tileSize = (sx % 8 == 0 && sy % 8 == 0) ? 8u : (sx % 4 == 0 && sy % 4 == 0) ? 4u : (sx % 2 == 0 && sy % 2 == 0) ? 2u : 1u;
    }
    int rc = hpt_update_params(m_ctx, &p);
    if (rc == HPT_OK) rc = hpt_pack_xy(m_ctx, (uint32_t)sizeX, sizeY);
    if (rc == HPT_OK && m_gens != (uint32_t)(sizeX * (int)sizeY)) { rc = hpt_init_random_gens(m_ctx, (uint32_t)(sizeX * (int)sizeY)); m_gens = (uint32_t)(sizeX * (int)sizeY); }   // (the reference's constructor seeds 1024 x 1024 generators once)
    if (rc == HPT_OK) rc = hpt_path_trace_block(m_ctx, 0, (uint32_t)(sizeX * (int)sizeY), channels, data, a_passNumber);      // m_pImpl->PathTraceBlock(sizeX*sizeY, channels, data, a_passNumber)
    if (rc != HPT_OK) m_err = hpt_last_error(m_ctx);
  }

  hpt_ctx* m_ctx = nullptr;
  LoadedScene m_scene;
  std::unordered_map<int, Mesh4fInput> m_meshPtrs;                             // Integrator::m_LSMeshPtrs
  std::string m_err;
  bool m_loaded = false, m_committed = false; uint32_t m_gens = 0;
};

inline std::shared_ptr<IRenderDriver> MakeHydraRenderHIP(int device = 0) { return std::make_shared<HydraHipRenderDriver>(device); }   // HR2::MakeHydraRenderCPU's counterpart (hydra_cpu.cpp:20-23)

} // namespace HR2
} // namespace hydra_hip
