// The driver loop of cam_plugin/main_with_cam.cpp:96-166 on the C ABI: an ICamRaysAPI2-shaped camera (cam_plugin/CamPluginAPI.h:39-77) makes
// batches of camera-space rays, Integrator::PathTraceFromInputRaysBlock traces them, the camera adds the colours to the frame.
//
//   hydra_hip_camrays <scene.xml> <width> <height> <spp> <out.bin> [tile]
//
// The camera here is a plain pinhole of the scene's own field of view with a per-ray jitter (own code standing in for a plugin); the frame
// it produces must agree with PathTraceBlock's up to Monte-Carlo noise, which tests/test_gpu_parity.py checks.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../hydracore3_amd/csrc/scene_loader.h"

struct RayPosAndW { float origin[3]; float wave; };      // cam_plugin/CamPluginAPI.h:27-31
struct RayDirAndT { float direction[3]; float time; };   // :33-37

// the interface of ICamRaysAPI2 (CamPluginAPI.h:39-77), the methods the driver loop calls
class PinholeCamRays
{
public:
  void SetParameters(int w, int h, float fovDeg) { m_w = w; m_h = h; m_tan = std::tan(fovDeg * 3.14159265358979323846f / 360.0f); }
  void SetBatchSize(int tile) { m_tile = tile; }
  // rays of the pixels [subPassId * tile, ...) of the frame in row-major order, camera space (looking down -z), one jittered sample per pixel
  void MakeRaysBlock(RayPosAndW* pos, RayDirAndT* dir, uint32_t n, int subPassId)
  {
    const float aspect = float(m_w) / float(m_h);
    for (uint32_t i = 0; i < n; i++) {
      const uint64_t pix = ((uint64_t)subPassId * (uint64_t)m_tile + i) % ((uint64_t)m_w * m_h);
      const int x = int(pix % m_w), y = int(pix / m_w);
      m_state = m_state * 6364136223846793005ull + 1442695040888963407ull; const float jx = float((m_state >> 40) & 0xFFFFFF) / 16777216.0f;
      m_state = m_state * 6364136223846793005ull + 1442695040888963407ull; const float jy = float((m_state >> 40) & 0xFFFFFF) / 16777216.0f;
      const float sx = (2.0f * (x + jx) / float(m_w) - 1.0f) * m_tan * aspect, sy = (2.0f * (y + jy) / float(m_h) - 1.0f) * m_tan;
      const float inv = 1.0f / std::sqrt(sx * sx + sy * sy + 1.0f);
      pos[i] = RayPosAndW{ { 0.0f, 0.0f, 0.0f }, 0.0f };
      dir[i] = RayDirAndT{ { sx * inv, sy * inv, -inv }, 0.0f };
    }
  }
  void AddSamplesContributionBlock(float* out4f, const float* colors4f, uint32_t n, uint32_t w, uint32_t h, int subPassId)
  {
    for (uint32_t i = 0; i < n; i++) {
      const uint64_t pix = ((uint64_t)subPassId * (uint64_t)m_tile + i) % ((uint64_t)w * h);
      for (int k = 0; k < 4; k++) out4f[4 * pix + (uint64_t)k] += colors4f[4 * (size_t)i + (size_t)k];
    }
  }
private:
  int m_w = 0, m_h = 0, m_tile = 0; float m_tan = 1.0f; uint64_t m_state = 0x853c49e6748fea9bull;
};

int main(int argc, char** argv)
{
  if (argc < 6) { std::fprintf(stderr, "usage: %s <scene.xml> <width> <height> <spp> <out.bin> [tile]\n", argv[0]); return 2; }
  const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), spp = std::atoi(argv[4]);
  const int MEGA_TILE_SIZE = argc > 6 ? std::atoi(argv[6]) : 512 * 512;                      // main_with_cam.cpp:96
  if (W <= 0 || H <= 0 || spp <= 0 || MEGA_TILE_SIZE <= 0) { std::fprintf(stderr, "[hydra_hip_camrays]: bad sizes\n"); return 2; }
  hydra_hip::LoadedScene sc; std::string err;
  if (!hydra_hip::LoadHydraXml(argv[1], W, H, sc, err)) { std::fprintf(stderr, "[hydra_hip_camrays]: %s\n", err.c_str()); return 1; }
  hpt_ctx* ctx = nullptr;
  if (hpt_create(0, &ctx) != HPT_OK) { std::fprintf(stderr, "[hydra_hip_camrays]: no HIP device\n"); return 1; }
  if (sc.upload(ctx) != HPT_OK) { std::fprintf(stderr, "[hydra_hip_camrays]: %s\n", hpt_last_error(ctx)); return 1; }
  if (hpt_init_random_gens(ctx, (uint32_t)MEGA_TILE_SIZE) != HPT_OK) { std::fprintf(stderr, "[hydra_hip_camrays]: %s\n", hpt_last_error(ctx)); return 1; }

  PinholeCamRays cam;
  cam.SetParameters(W, H, (float)sc.fov);
  cam.SetBatchSize(MEGA_TILE_SIZE);
  std::vector<RayPosAndW> rayPos((size_t)MEGA_TILE_SIZE);
  std::vector<RayDirAndT> rayDir((size_t)MEGA_TILE_SIZE);
  std::vector<float> rayCol((size_t)MEGA_TILE_SIZE * 4), realColor((size_t)W * H * 4, 0.0f);
  const int passNum = (W * H + MEGA_TILE_SIZE - 1) / MEGA_TILE_SIZE;                          // (the reference assumes the tile divides the frame)
  double execMs = 0.0;
  for (int passId = 0; passId < spp; passId++)                                                // CAM_PASSES_NUM passes of SAMPLES_PER_RAY = 1
    for (int subPassId = 0; subPassId < passNum; subPassId++) {
      const uint32_t n = (uint32_t)std::min<long long>(MEGA_TILE_SIZE, (long long)W * H - (long long)subPassId * MEGA_TILE_SIZE);
      std::fill(rayCol.begin(), rayCol.end(), 0.0f);
      cam.MakeRaysBlock(rayPos.data(), rayDir.data(), n, subPassId);
      if (hpt_path_trace_from_input_rays_block(ctx, n, 4, &rayPos[0].origin[0], &rayDir[0].direction[0], rayCol.data(), 1) != HPT_OK) {
        std::fprintf(stderr, "[hydra_hip_camrays]: %s\n", hpt_last_error(ctx)); return 1;
      }
      cam.AddSamplesContributionBlock(realColor.data(), rayCol.data(), n, (uint32_t)W, (uint32_t)H, subPassId);
      float t[4]; hpt_get_execution_time(ctx, "PathTraceFromInputRaysBlock", t); execMs += t[0];
    }
  FILE* f = std::fopen(argv[5], "wb");
  if (!f) { std::fprintf(stderr, "cannot write %s\n", argv[5]); return 1; }
  std::fwrite(realColor.data(), sizeof(float), realColor.size(), f); std::fclose(f);
  double s = 0.0; for (size_t i = 0; i < realColor.size(); i += 4) s += realColor[i] + realColor[i + 1] + realColor[i + 2];
  std::printf("[hydra_hip_camrays]: %dx%d @ %d spp in tiles of %d, mean radiance %.5f, PathTraceFromInputRays(exec, total) = %.3f ms\n", W, H, spp, MEGA_TILE_SIZE,
              s / (3.0 * W * H * spp), execMs);
  hpt_destroy(ctx);
  return 0;
}
