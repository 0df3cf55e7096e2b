// The optimisation loop of diff_render/drmain.cpp:174-261 on the C ABI, everything resident on the device, no Python:
//
//   hydra_hip_dr <scene.xml> <width> <height> <spp> <iterations> <texId> <texW> <texH> <out_prefix> [--ref ref.bin] [--ref-spp N]
//                [--reg lambda] [--dump-every K]
//
//   reference image  --ref ref.bin : width*height*4 floats, radiance per sample, rows stored bottom-up as the reference's EXR reader
//                                    hands them to PixelLossPT (integrator_dr.cpp:1103-1132)
//                    default       : what `drmain -grad 0` produces first - a forward render of the scene with its own texture at
//                                    --ref-spp (default 4*spp), normalised by 1/spp
//   loop             texture := 1.0 (drmain.cpp:180), PutDiffTex2D(texId, texW, texH, 4) (:185), then per iteration
//                    PathTraceDR -> loss, gradient (:207), optional Image2D4fRegularizer * lambda (:213-217, commented out there),
//                    AdamOptimizer::step (:244); the loss of every iteration is printed as drmain prints it
//   output           <out_prefix>_tex.bin (texW*texH*4 floats), <out_prefix>_frame.bin (last frame, per-sample radiance),
//                    <out_prefix>_loss.txt (one loss per line)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../hydracore3_amd/csrc/scene_loader.h"

#define CHK(call) do { const int rc_ = (call); if (rc_ != HPT_OK) { std::fprintf(stderr, "[hydra_hip_dr]: %s failed: %s\n", #call, hpt_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv)
{
  if (argc < 10) { std::fprintf(stderr, "usage: %s <scene.xml> <width> <height> <spp> <iterations> <texId> <texW> <texH> <out_prefix> [--ref ref.bin] [--ref-spp N] [--reg lambda] [--dump-every K]\n", argv[0]); return 2; }
  const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), spp = std::atoi(argv[4]), iters = std::atoi(argv[5]);
  const uint32_t texId = (uint32_t)std::atoi(argv[6]), texW = (uint32_t)std::atoi(argv[7]), texH = (uint32_t)std::atoi(argv[8]);
  const std::string prefix = argv[9];
  std::string refPath; int refSpp = 4 * spp, dumpEvery = 0; float lambda = 0.0f;
  for (int i = 10; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--ref" && i + 1 < argc) refPath = argv[++i];
    else if (a == "--ref-spp" && i + 1 < argc) refSpp = std::atoi(argv[++i]);
    else if (a == "--reg" && i + 1 < argc) lambda = (float)std::atof(argv[++i]);
    else if (a == "--dump-every" && i + 1 < argc) dumpEvery = std::atoi(argv[++i]);
    else { std::fprintf(stderr, "[hydra_hip_dr]: unknown option %s\n", a.c_str()); return 2; }
  }
  if (W <= 0 || H <= 0 || spp <= 0 || iters <= 0 || texW == 0 || texH == 0) { std::fprintf(stderr, "[hydra_hip_dr]: bad sizes\n"); return 2; }
  hydra_hip::LoadedScene sc; std::string err;
  if (!hydra_hip::LoadHydraXml(argv[1], W, H, sc, err)) { std::fprintf(stderr, "[hydra_hip_dr]: %s\n", err.c_str()); return 1; }
  hpt_ctx* ctx = nullptr;
  if (hpt_create(0, &ctx) != HPT_OK) { std::fprintf(stderr, "[hydra_hip_dr]: no HIP device\n"); return 1; }
  CHK(sc.upload(ctx));
  const size_t nPix = (size_t)W * H, nFrame = nPix * 4;

  // ---- the reference image ------------------------------------------------------------------------------------------------------------
  std::vector<float> ref(nFrame, 0.0f);
  if (!refPath.empty()) {
    FILE* f = std::fopen(refPath.c_str(), "rb");
    if (!f || std::fread(ref.data(), sizeof(float), nFrame, f) != nFrame) { std::fprintf(stderr, "[hydra_hip_dr]: cannot read %zu floats from %s\n", nFrame, refPath.c_str()); return 1; }
    std::fclose(f);
  } else {
    std::vector<float> frame(nFrame, 0.0f);
    CHK(hpt_path_trace_block(ctx, 0, (uint32_t)nPix, 4, frame.data(), (uint32_t)refSpp));
    const float norm = 1.0f / float(refSpp);
    for (int y = 0; y < H; y++)                                              // bottom-up rows, as the loss reads them
      for (int x = 0; x < W * 4; x++) ref[(size_t)(H - 1 - y) * W * 4 + x] = frame[(size_t)y * W * 4 + x] * norm;
    CHK(hpt_init_random_gens(ctx, (uint32_t)nPix));                          // the optimisation starts from the generators a fresh process has
  }

  // ---- resident state: texture parameters, gradient, Adam moments, frame, reference, loss --------------------------------------------
  uint64_t texOffset = 0, texSize = 0;
  CHK(hpt_put_diff_tex2d(ctx, texId, texW, texH, 4, &texOffset, &texSize));
  if (texSize != (uint64_t)texW * texH * 4) { std::fprintf(stderr, "[hydra_hip_dr]: PutDiffTex2D returned size %llu\n", (unsigned long long)texSize); return 1; }
  const size_t n = (size_t)texSize;
  void *dData = nullptr, *dGrad = nullptr, *dMom = nullptr, *dSq = nullptr, *dFrame = nullptr, *dRef = nullptr, *dLoss = nullptr;
  CHK(hpt_device_malloc(ctx, n * 4, &dData)); CHK(hpt_device_malloc(ctx, n * 4, &dGrad)); CHK(hpt_device_malloc(ctx, n * 4, &dMom));
  CHK(hpt_device_malloc(ctx, n * 4, &dSq)); CHK(hpt_device_malloc(ctx, nFrame * 4, &dFrame)); CHK(hpt_device_malloc(ctx, nFrame * 4, &dRef));
  CHK(hpt_device_malloc(ctx, 4, &dLoss));
  { std::vector<float> ones(n, 1.0f); CHK(hpt_device_copy(ctx, dData, ones.data(), n * 4, 1)); }      // std::fill(imgData, 1.0f) (drmain.cpp:180)
  CHK(hpt_device_memset(ctx, dMom, 0, n * 4)); CHK(hpt_device_memset(ctx, dSq, 0, n * 4));
  CHK(hpt_device_copy(ctx, dRef, ref.data(), nFrame * 4, 1));

  std::vector<float> losses, frame(nFrame);
  double kernelMs = 0.0;
  for (int iter = 0; iter < iters; iter++) {
    CHK(hpt_device_memset(ctx, dFrame, 0, nFrame * 4));                      // std::fill(realColor, 0) (:201)
    CHK(hpt_device_memset(ctx, dGrad, 0, n * 4));                            // memset(a_dataGrad, 0) (integrator_dr.cpp:1139)
    CHK(hpt_device_memset(ctx, dLoss, 0, 4));
    CHK(hpt_path_trace_dr_dev(ctx, 0, (uint32_t)nPix, 4, (float*)dFrame, (uint32_t)spp, (const float*)dRef, (const float*)dData, (float*)dGrad, n, (float*)dLoss, nullptr));
    if (lambda != 0.0f) {                                                    // imgGrad += lambda * d RegLoss (:213-217): lambda folded in by scaling a scratch gradient
      void* dReg = nullptr; CHK(hpt_device_malloc(ctx, n * 4, &dReg)); CHK(hpt_device_memset(ctx, dReg, 0, n * 4));
      CHK(hpt_image2d4f_regularizer_dev(ctx, (int)texW, (int)texH, (const float*)dData, (float*)dReg, nullptr));
      std::vector<float> g(n), r(n);
      CHK(hpt_device_copy(ctx, g.data(), dGrad, n * 4, 2)); CHK(hpt_device_copy(ctx, r.data(), dReg, n * 4, 2));
      for (size_t i = 0; i < n; i++) g[i] += lambda * r[i];
      CHK(hpt_device_copy(ctx, dGrad, g.data(), n * 4, 1)); CHK(hpt_device_free(ctx, dReg));
    }
    float lossSum = 0.0f;
    CHK(hpt_device_copy(ctx, &lossSum, dLoss, 4, 2));                        // synchronises with the launch above
    const float loss = lossSum / float(nPix);                               // avgLoss /= W*H (integrator_dr.cpp:1206)
    float ms = 0.0f; CHK(hpt_last_kernel_ms(ctx, &ms));
    kernelMs += ms;
    losses.push_back(loss);
    std::printf("[hydra_hip_dr]: Render(%02d, spp = %d) .., loss = %g, time = %.3f ms\n", iter, spp, loss, ms);
    CHK(hpt_adam_step_dev(ctx, (float*)dData, (const float*)dGrad, (float*)dMom, (float*)dSq, n, iter, nullptr));   // pOpt->step(imgData, imgGrad, iter) (:244)
    if (dumpEvery > 0 && iter % dumpEvery == 0) {
      CHK(hpt_device_copy(ctx, frame.data(), dFrame, nFrame * 4, 2));
      char name[512]; std::snprintf(name, sizeof(name), "%s_%02d.bin", prefix.c_str(), iter);
      FILE* f = std::fopen(name, "wb"); if (f) { const float norm = 1.0f / float(spp); for (float& v : frame) v *= norm; std::fwrite(frame.data(), 4, nFrame, f); std::fclose(f); }
    }
  }
  std::vector<float> tex(n);
  CHK(hpt_device_copy(ctx, tex.data(), dData, n * 4, 2));
  CHK(hpt_device_copy(ctx, frame.data(), dFrame, nFrame * 4, 2));
  { const float norm = 1.0f / float(spp); for (float& v : frame) v *= norm; }
  auto dump = [&](const std::string& name, const void* p, size_t bytes) { FILE* f = std::fopen(name.c_str(), "wb"); if (!f) return false; std::fwrite(p, 1, bytes, f); std::fclose(f); return true; };
  if (!dump(prefix + "_tex.bin", tex.data(), n * 4) || !dump(prefix + "_frame.bin", frame.data(), nFrame * 4)) { std::fprintf(stderr, "[hydra_hip_dr]: cannot write %s_*.bin\n", prefix.c_str()); return 1; }
  { FILE* f = std::fopen((prefix + "_loss.txt").c_str(), "w"); if (f) { for (float l : losses) std::fprintf(f, "%.9g\n", l); std::fclose(f); } }
  std::printf("[hydra_hip_dr]: %d iterations, loss %g -> %g, %.3f ms of PathTraceDR per iteration\n", iters, losses.front(), losses.back(), kernelMs / iters);
  for (void* p : { dData, dGrad, dMom, dSq, dFrame, dRef, dLoss }) hpt_device_free(ctx, p);
  hpt_destroy(ctx);
  return 0;
}
