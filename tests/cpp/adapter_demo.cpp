// Drives the HIP core purely through the C++ adapter (hydracore3_amd/csrc/integrator_hip.h), the way HydraCore3's
// main.cpp drives its Integrator (main.cpp:249-267, 395-419): geometry through the ISceneObject calls, scene vectors,
// CommitDeviceData, SetViewport, PackXYBlock, UpdateMembersPlainData, PathTraceBlock, GetExecutionTime.
// Scene: a Lambertian plane filling the view under a constant environment -> every sample is exactly albedo * env.
// Exit code 0 = pass.  Needs a GPU to run; compiling + linking it is part of build().
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../hydracore3_amd/csrc/integrator_hip.h"

using namespace hydra_hip;

static float4x4 identity() { float4x4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }

int main()
{
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
