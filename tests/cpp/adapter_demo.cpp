// Drives the HIP core purely through the C++ adapter (hydracore3_amd/csrc/integrator_hip.h), the way HydraCore3's
// main.cpp drives its Integrator (main.cpp:249-267, 395-419): geometry through the ISceneObject calls, scene vectors,
// CommitDeviceData, SetViewport, PackXYBlock, UpdateMembersPlainData, PathTraceBlock, GetExecutionTime.
// Scene: a Lambertian plane filling the view under a constant environment -> every sample is exactly albedo * env.
// Exit code 0 = pass.  Needs a GPU to run; compiling + linking it is part of build().
#include <cmath>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <string>
#include "../../hydracore3_amd/csrc/scene_loader.h"        // (brings integrator_hip.h): only to READ a Hydra XML scene into plain tables

using namespace hydra_hip;

// adapter_demo <scene.xml> <width> <height> <spp> <out.bin> [--spectral]: a whole Hydra scene through the ADAPTER CLASSES only - geometry and (moving)
// instances through BVH2SceneHIP (AddGeom_Triangles3f / AddInstance / AddInstanceMotion / CommitScene), the scene vectors through the
// Integrator-named members (m_materials ... m_arrays1f, m_normMatrices + m_normMatrices2Offs, the m_env* ids), then CommitDeviceData,
// PackXYBlock, UpdateMembersPlainData, PathTraceBlock - the calls main.cpp makes. The test compares the frame with the ctypes front end.
static int renderScene(const char* xml, int W, int H, int spp, const char* out, bool spectral)
{
  LoadedScene sc; std::string err;
  if (!LoadHydraXml(xml, W, H, sc, err, spectral)) { std::printf("adapter_demo: %s\n", err.c_str()); return 1; }
  IntegratorHIP integ(W * H, 0);
  if (!integ.valid()) { std::printf("adapter_demo: no GPU\n"); return 2; }
  BVH2SceneHIP* acc = integ.m_pAccelStruct;
  acc->ClearGeom();
  for (size_t g = 0; g < sc.geomTriCount.size(); g++) {                    // LoadSceneGeometry (integrator_pt_scene.cpp:790-846)
    const uint32_t triOff = sc.matVertOffset[2 * g], vertOff = sc.matVertOffset[2 * g + 1];
    acc->AddGeom_Triangles3f(sc.vPos4f.data() + 4 * size_t(vertOff), sc.geomVertCount[g], sc.triIndices.data() + 3 * size_t(triOff), 3 * size_t(sc.geomTriCount[g]), 4, 16);
  }
  acc->ClearScene();
  for (size_t i = 0; i < sc.instGeomId.size(); i++) {                       // LoadSceneInstances (:848-897)
    float4x4 two[2]; std::memcpy(two[0].m, sc.instMatrices.data() + 16 * i, 64);
    if (sc.normMatrices2Offs && sc.instHasMotion[i]) { std::memcpy(two[1].m, sc.instMatricesMotion.data() + 16 * i, 64); acc->AddInstanceMotion(sc.instGeomId[i], two, 2); }
    else acc->AddInstance(sc.instGeomId[i], two[0]);
  }
  acc->CommitScene();
  integ.m_matVertOffset = sc.matVertOffset; integ.m_matIdByPrimId = sc.matIdByPrimId; integ.m_triIndices = sc.triIndices; integ.m_vData8f = sc.vData8f;
  integ.m_normMatrices.resize(sc.normMatrices.size() / 16); std::memcpy(integ.m_normMatrices.data(), sc.normMatrices.data(), sc.normMatrices.size() * 4);
  integ.m_normMatrices2Offs = sc.normMatrices2Offs;
  integ.m_instGeomId = sc.instGeomId; integ.m_remapInst.assign(sc.remapInst.begin(), sc.remapInst.end());
  integ.m_allRemapLists.assign(sc.allRemapLists.begin(), sc.allRemapLists.end()); integ.m_allRemapListsSize = sc.allRemapListsSize;
  integ.m_materials = sc.materials; integ.m_lights = sc.lights; integ.m_arrays1f = sc.arrays1f;
  // spectral rendering: m_spec_values / m_spec_offset_sz / m_cie_xyz and the camera response, then m_spectral_mode (main.cpp: --spectral)
  integ.m_spec_values = sc.specValues; integ.m_spec_offset_sz = sc.specOffsetSz; integ.m_cie_xyz = sc.cieXYZ;
  for (int k = 0; k < 3; k++) integ.m_camResponseSpectrumId[k] = sc.camResponseSpectrumId[k];
  integ.m_camResponseType = int(sc.camResponseType); std::memcpy(integ.m_camRespoceRGB, sc.camRespoceRGB, 16);
  integ.m_spec_tex_ids_wavelengths = sc.specTexIdsWavelengths; integ.m_spec_tex_offset_sz = sc.specTexOffsetSz;
  integ.m_films_thickness_vec = sc.filmsThickness; integ.m_films_spec_id_vec = sc.filmsSpecId; integ.m_films_eta_k_vec = sc.filmsEtaK; integ.m_precomp_thin_films = sc.precompThinFilms;
  integ.m_spectral_mode = int(sc.spectralMode); integ.m_envSpecId = sc.envSpecId; integ.m_envSpecMult = sc.envSpecMult;
  integ.m_textures.clear();
  for (const LoadedTexture& t : sc.textures) { TextureData d; d.width = t.width; d.height = t.height; d.format = t.format; d.flags = t.flags; d.addressU = t.addressU; d.addressV = t.addressV; d.filter = t.filter; d.texels = t.bytes; integ.m_textures.push_back(d); }
  const hpt_params p = sc.params();
  float4x4 pi, wv; std::memcpy(pi.m, p.projInv, 64); std::memcpy(wv.m, p.worldViewInv, 64);
  integ.SetProjInv(pi); integ.SetWorldViewInv(wv);
  integ.m_traceDepth = p.traceDepth; integ.SetIntegratorType(p.integratorType); integ.m_camTargetDist = p.camTargetDist;
  std::memcpy(integ.m_envColor, p.envColor, 16);
  integ.m_envTexId = p.envTexId; integ.m_envLightId = p.envLightId; integ.m_envCamBackId = p.envCamBackId; integ.m_envEnableSam = p.envEnableSam;
  std::memcpy(integ.m_envSamRow0, p.envSamRow0, 16); std::memcpy(integ.m_envSamRow1, p.envSamRow1, 16);
  if (!sc.lensLines.empty()) {
    std::vector<IntegratorHIP::LensElementInterface> lines(sc.lensLines.size() / 4);
    std::memcpy(lines.data(), sc.lensLines.data(), sc.lensLines.size() * 4);
    integ.SetLines(lines); integ.SetPhysSize(sc.physSize[0], sc.physSize[1]); integ.m_enableOpticSim = 1;
  }
  integ.SetFrameBufferSize(W, H); integ.SetViewport(0, 0, W, H);
  integ.CommitDeviceData();
  integ.PackXYBlock(W, H, 1);
  integ.UpdateMembersPlainData();
  std::vector<float> frame(size_t(W) * H * 4, 0.0f);
  integ.PathTraceBlock(W * H, 4, frame.data(), spp);
  // the ISceneObject queries answer through the same object, *Motion forms at the ray's time
  const float o[4] = { float(sc.camPos[0]), float(sc.camPos[1]), float(sc.camPos[2]), 0.0f };
  const float d[4] = { float(sc.camLookAt[0] - sc.camPos[0]), float(sc.camLookAt[1] - sc.camPos[1]), float(sc.camLookAt[2] - sc.camPos[2]), 1e30f };
  const CRT_Hit h0 = acc->RayQuery_NearestHitMotion(o, d, 0.0f), h1 = acc->RayQuery_NearestHitMotion(o, d, 1.0f);
  std::printf("adapter_demo: %s %dx%d @ %d spp; central ray (unnormalised direction) hits instance %d at t = %.5f (time 0) / instance %d at t = %.5f (time 1)\n",
              xml, W, H, spp, int(h0.instId), h0.t, int(h1.instId), h1.t);
  if (FILE* f = std::fopen(out, "wb")) { std::fwrite(frame.data(), 4, frame.size(), f); std::fclose(f); } else return 1;
  return 0;
}

static float4x4 identity() { float4x4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }

int main(int argc, char** argv)
{
  if (argc >= 6) return renderScene(argv[1], std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]), argv[5], argc > 6 && std::string(argv[6]) == "--spectral");
  const int W = 64, H = 64, SPP = 4;
  IntegratorHIP integ(W * H, 0);
  if (!integ.valid()) { std::printf("adapter_demo: no GPU\n"); return 2; }

  // geometry: one quad (2 triangles) at y = 0, stride-16 positions as LoadSceneGeometry passes them (integrator_pt_scene.cpp:799-800)
  const float pos[16] = { -50, 0, 50, 1,   50, 0, 50, 1,   50, 0, -50, 1,   -50, 0, -50, 1 };
  const uint32_t idx[6] = { 0, 1, 2, 0, 2, 3 };
  const uint32_t geomId = integ.m_pAccelStruct->AddGeom_Triangles3f(pos, 4, idx, 6, 4, 16);
  const uint32_t instId = integ.m_pAccelStruct->AddInstance(geomId, identity());
  integ.m_pAccelStruct->CommitScene();
  if (geomId != 0 || instId != 0) { std::printf("adapter_demo: bad ids\n"); return 1; }

  integ.m_matVertOffset = { 0, 0 };
  integ.m_matIdByPrimId = { 0, 0 };
  integ.m_triIndices.assign(idx, idx + 6);
  for (int v = 0; v < 4; v++) { const float d[8] = { 0, 1, 0, 0.0f, 1, 0, 0, 0.0f }; integ.m_vData8f.insert(integ.m_vData8f.end(), d, d + 8); }
  integ.m_normMatrices = { identity() };
  integ.m_instGeomId = { 0 };
  integ.m_remapInst = { -1, -1 };
  integ.m_allRemapLists = { 0 };
  Material m{};                                           // diffuse-only ConvertOldHydraMaterial result (integrator_pt_scene_mat.cpp:410-419)
  m.mtype = 1; m.cflags = 1; m.lightId = 0xFFFFFFFFu; m.texid[1] = 0xFFFFFFFFu;
  for (int i = 0; i < 4; i++) { m.row0[i][0] = 1.0f; m.row1[i][1] = 1.0f; m.spdid[i] = 0xFFFFFFFFu; }
  m.colors[0][0] = 0.2f; m.colors[0][1] = 0.5f; m.colors[0][2] = 0.9f;
  m.data[4] = 1.0f;                                       // GLTF_FLOAT_GLOSINESS
  integ.m_materials = { m };
  integ.m_envColor[0] = 1.0f; integ.m_envColor[1] = 2.0f; integ.m_envColor[2] = 0.5f;

  // camera at (0,2,0) looking straight down, up = -z: inverse view = [right | up | back | eye], 40 degree fov
  float4x4 wvInv{};
  const float right[3] = { 1, 0, 0 }, up[3] = { 0, 0, -1 }, back[3] = { 0, 1, 0 }, eye[3] = { 0, 2, 0 };
  for (int r = 0; r < 3; r++) { wvInv.m[0 + r] = right[r]; wvInv.m[4 + r] = up[r]; wvInv.m[8 + r] = back[r]; wvInv.m[12 + r] = eye[r]; }
  wvInv.m[15] = 1.0f;
  const float zn = 0.01f, zf = 100.0f, t = zn * std::tan(40.0f * 3.14159265f / 360.0f);
  float4x4 projInv{};                                      // inverse of the OpenGL-style frustum(-t,t,-t,t,zn,zf)
  projInv.m[0] = t / zn; projInv.m[5] = t / zn; projInv.m[11] = (zn - zf) / (2.0f * zf * zn); projInv.m[14] = -1.0f; projInv.m[15] = (zf + zn) / (2.0f * zf * zn);
  integ.SetProjInv(projInv); integ.SetWorldViewInv(wvInv);
  integ.m_traceDepth = 2;
  integ.SetIntegratorType(2);                              // INTEGRATOR_MIS_PT
  integ.SetFrameBufferSize(W, H);
  integ.SetViewport(0, 0, W, H);

  integ.CommitDeviceData();
  integ.PackXYBlock(W, H, 1);
  integ.UpdateMembersPlainData();
  std::vector<float> realColor(size_t(W) * H * 4, 0.0f);
  integ.PathTraceBlock(W * H, 4, realColor.data(), SPP);
  float timings[4] = { 0, 0, 0, 0 };
  integ.GetExecutionTime("PathTraceBlock", timings);

  double maxErr = 0.0;
  const float expect[3] = { 0.2f * 1.0f, 0.5f * 2.0f, 0.9f * 0.5f };
  for (int p = 0; p < W * H; p++)
    for (int c = 0; c < 3; c++) maxErr = std::fmax(maxErr, std::fabs(realColor[4 * p + c] / SPP - expect[c]) / expect[c]);
  std::printf("adapter_demo: PathTraceBlock(exec) = %.3f ms, max relative error vs albedo*env = %.3e\n", timings[0], maxErr);
  return maxErr < 1e-4 ? 0 : 1;
}
