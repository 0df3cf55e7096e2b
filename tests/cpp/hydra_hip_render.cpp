// Renders a Hydra XML scene through the C ABI without Python:  hydra_hip_render <scene.xml> <width> <height> <spp> <out.bin> [--tables | --ppm preview.ppm] [--spectral]
//   default   : scene_loader.h -> LoadedScene::upload -> hpt_path_trace_block; writes the raw float4 frame (un-normalised, as the callee
//               accumulates it) to <out.bin> and prints the mean radiance per sample
//   --tables  : no GPU needed - dumps the loaded tables as [name '\0'][u64 byte count][bytes] records for the loader test
//   --spectral: m_spectral_mode = 1 (four wavelengths per path; the scene's spectra, the CIE observer fit of scene_loader.h)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../hydracore3_amd/csrc/scene_loader.h"

template <class T> static void blob(FILE* f, const char* name, const std::vector<T>& v) { std::fputs(name, f); std::fputc(0, f); const uint64_t n = v.size() * sizeof(T); std::fwrite(&n, 8, 1, f); if (n) std::fwrite(v.data(), 1, n, f); }

int main(int argc, char** argv)
{
  if (argc == 4 && std::string(argv[1]) == "--exr") {                                  // decoder check: hydra_hip_render --exr <file.exr> <out.bin> = {w, h, w * h * 4 floats in file order}
    std::vector<uint8_t> raw; std::vector<float> rgba; uint32_t w = 0, h = 0; std::string err;
    if (!hydra_hip::detail::readFile(argv[2], raw) || !hydra_hip::detail::decodeExr(raw, w, h, rgba, err)) { std::fprintf(stderr, "[hydra_hip_render]: %s\n", err.c_str()); return 1; }
    FILE* f = std::fopen(argv[3], "wb"); if (!f) return 1;
    std::fwrite(&w, 4, 1, f); std::fwrite(&h, 4, 1, f); std::fwrite(rgba.data(), 4, rgba.size(), f); std::fclose(f);
    return 0;
  }
  if (argc < 6) { std::fprintf(stderr, "usage: %s <scene.xml> <width> <height> <spp> <out.bin> [--tables | --ppm preview.ppm]\n", argv[0]); return 2; }
  const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), spp = std::atoi(argv[4]);
  const bool tables = argc > 6 && std::string(argv[6]) == "--tables";
  bool spectral = false;
  for (int k = 6; k < argc; k++) spectral = spectral || std::string(argv[k]) == "--spectral";
  hydra_hip::LoadedScene sc; std::string err;
  if (!hydra_hip::LoadHydraXml(argv[1], W, H, sc, err, spectral)) { std::fprintf(stderr, "[hydra_hip_render]: %s\n", err.c_str()); return 1; }
  FILE* f = std::fopen(argv[5], "wb");
  if (!f) { std::fprintf(stderr, "cannot write %s\n", argv[5]); return 1; }
  if (tables) {
    const hpt_params p = sc.params();
    blob(f, "vPos4f", sc.vPos4f); blob(f, "vData8f", sc.vData8f); blob(f, "triIndices", sc.triIndices); blob(f, "matIdByPrimId", sc.matIdByPrimId);
    blob(f, "matVertOffset", sc.matVertOffset); blob(f, "geomTriCount", sc.geomTriCount); blob(f, "geomVertCount", sc.geomVertCount);
    blob(f, "instGeomId", sc.instGeomId); blob(f, "instMatrices", sc.instMatrices); blob(f, "normMatrices", sc.normMatrices);
    blob(f, "remapInst", sc.remapInst); blob(f, "allRemapLists", sc.allRemapLists);
    blob(f, "materials", sc.materials); blob(f, "lights", sc.lights);
    blob(f, "params", std::vector<hpt_params>(1, p));
    blob(f, "arrays1f", sc.arrays1f);
    blob(f, "lensLines", sc.lensLines); blob(f, "physSize", std::vector<float>(sc.physSize, sc.physSize + 2));
    blob(f, "instMatricesMotion", sc.instMatricesMotion); blob(f, "instHasMotion", sc.instHasMotion);
    blob(f, "normMatrices2Offs", std::vector<uint32_t>(1, sc.normMatrices2Offs));
    blob(f, "specValues", sc.specValues); blob(f, "specOffsetSz", sc.specOffsetSz); blob(f, "cieXYZ", sc.cieXYZ);
    blob(f, "specTexIdsWavelengths", sc.specTexIdsWavelengths); blob(f, "specTexOffsetSz", sc.specTexOffsetSz);
    blob(f, "filmsThickness", sc.filmsThickness); blob(f, "filmsSpecId", sc.filmsSpecId); blob(f, "filmsEtaK", sc.filmsEtaK); blob(f, "precompThinFilms", sc.precompThinFilms);
    blob(f, "camResponse", std::vector<int32_t>{ sc.camResponseSpectrumId[0], sc.camResponseSpectrumId[1], sc.camResponseSpectrumId[2], (int32_t)sc.camResponseType });
    for (size_t i = 0; i < sc.textures.size(); i++) {
      const hydra_hip::LoadedTexture& t = sc.textures[i];
      blob(f, "texHeader", std::vector<uint32_t>{ t.width, t.height, t.format, t.flags, t.addressU, t.addressV, t.filter });
      blob(f, "texBytes", t.bytes);
    }
    std::fclose(f);
    return 0;
  }
  hpt_ctx* ctx = nullptr;
  if (hpt_create(0, &ctx) != HPT_OK) { std::fprintf(stderr, "[hydra_hip_render]: no HIP device\n"); return 1; }
  int rc = sc.upload(ctx);
  if (rc != HPT_OK) { std::fprintf(stderr, "[hydra_hip_render]: %s\n", hpt_last_error(ctx)); return 1; }
  std::vector<float> frame((size_t)W * H * 4, 0.0f);
  rc = hpt_path_trace_block(ctx, 0, (uint32_t)(W * H), 4, frame.data(), (uint32_t)spp);
  if (rc != HPT_OK) { std::fprintf(stderr, "[hydra_hip_render]: %s\n", hpt_last_error(ctx)); return 1; }
  float t[4]; hpt_get_execution_time(ctx, "PathTraceBlock", t);
  std::fwrite(frame.data(), sizeof(float), frame.size(), f); std::fclose(f);
  if (argc > 6 && std::string(argv[6]) == "--ppm" && argc > 7) {                       // 8-bit preview: radiance / spp, gamma 2.2, top row first
    if (FILE* pf = std::fopen(argv[7], "wb")) {
      std::fprintf(pf, "P6\n%d %d\n255\n", W, H);
      for (int y = H - 1; y >= 0; y--) for (int x = 0; x < W; x++) for (int k = 0; k < 3; k++) {
        const float v = std::pow(std::min(std::max(frame[((size_t)y * W + x) * 4 + k] / float(spp), 0.0f), 1.0f), 1.0f / 2.2f);
        std::fputc((int)(v * 255.0f + 0.5f), pf);
      }
      std::fclose(pf);
    }
  }
  double s = 0.0; for (size_t i = 0; i < frame.size(); i += 4) s += frame[i] + frame[i + 1] + frame[i + 2];
  std::printf("[hydra_hip_render]: %dx%d @ %d spp, mean radiance %.5f, kernel %.3f ms\n", W, H, spp, s / (3.0 * W * H * spp), t[0]);
  hpt_destroy(ctx);
  return 0;
}
