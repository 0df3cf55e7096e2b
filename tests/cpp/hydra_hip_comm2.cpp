// Two processes on the C ABI's RCCL entry points (hpt_comm_get_unique_id / hpt_comm_init / hpt_reduce_framebuffer / hpt_allreduce_grad), the way a C++ host
// without PyTorch shards PathTraceBlock over GPUs:  hydra_hip_comm2 <scene.xml> <width> <height> <spp> <idfile>
//   * the parent forks BEFORE anything touches the GPU (a GPU-initialised process must not be replaced or duplicated); rank 0 creates the
//     ncclUniqueId and hands it over through <idfile>; rank r opens device r % deviceCount;
//   * each rank renders its interleaved 1024-tid chunks (hpt_set_tid_interleave, the bit-identical pixel split) into a zeroed full frame;
//   * hpt_reduce_framebuffer(SUM, root 0) assembles the frame; rank 0 renders the whole frame alone and compares: must be bit-identical;
//   * hpt_allreduce_grad over a small vector checks the all-reduce (every rank contributes rank + 1).
// Exit code 0: passed. 77: RCCL refused the communicator - on a box with ONE GPU both ranks land on the same device, which RCCL rejects
// ("Duplicate GPU detected"); the message is printed and the test that runs this tool treats it as "not runnable here", not as a pass.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unistd.h>
#include <sys/wait.h>
#include "../../hydracore3_amd/csrc/scene_loader.h"

using namespace hydra_hip;

static int rankMain(int rank, int world, const char* xml, int W, int H, int spp, const std::string& idFile)
{
  LoadedScene sc; std::string err;
  if (!LoadHydraXml(xml, W, H, sc, err)) { std::fprintf(stderr, "[comm2 rank %d]: %s\n", rank, err.c_str()); return 1; }
  char info[256] = {0};
  hpt_ctx* probe = nullptr;
  int ndev = 1;
  for (int d = 7; d >= 0; d--) { if (hpt_create(d, &probe) == HPT_OK) { ndev = d + 1; hpt_destroy(probe); break; } }
  hpt_ctx* ctx = nullptr;
  if (hpt_create(rank % ndev, &ctx) != HPT_OK) { std::fprintf(stderr, "[comm2 rank %d]: no HIP device\n", rank); return 1; }
  (void)info;
  unsigned char id[128];
  if (rank == 0) {
    if (hpt_comm_get_unique_id(ctx, id) != HPT_OK) { std::fprintf(stderr, "[comm2]: %s\n", hpt_last_error(ctx)); return 77; }
    FILE* f = std::fopen((idFile + ".tmp").c_str(), "wb"); if (!f) return 1;
    std::fwrite(id, 1, sizeof(id), f); std::fclose(f);
    std::rename((idFile + ".tmp").c_str(), idFile.c_str());
  } else {
    FILE* f = nullptr;
    for (int tries = 0; tries < 600 && !f; tries++) { f = std::fopen(idFile.c_str(), "rb"); if (!f) usleep(100000); }
    if (!f || std::fread(id, 1, sizeof(id), f) != sizeof(id)) { std::fprintf(stderr, "[comm2 rank %d]: no unique id\n", rank); return 1; }
    std::fclose(f);
  }
  if (hpt_comm_init(ctx, world, rank, id) != HPT_OK) {
    std::fprintf(stderr, "[comm2 rank %d of %d on device %d of %d]: RCCL refused the communicator: %s\n", rank, world, rank % ndev, ndev, hpt_last_error(ctx));
    return 77;
  }
  if (sc.upload(ctx) != HPT_OK) { std::fprintf(stderr, "[comm2 rank %d]: %s\n", rank, hpt_last_error(ctx)); return 1; }
  const uint32_t N = (uint32_t)(W * H), chunk = 1024u;
  // rank r renders chunks r, r + world, ... (hydracore3_amd/sharding.py: tid_interleave)
  const uint32_t nChunks = (N + chunk - 1) / chunk, mine = (nChunks - rank + world - 1) / world;
  hpt_set_tid_interleave(ctx, chunk, (uint32_t)world);
  void* dFrame = nullptr; const size_t bytes = (size_t)N * 4 * sizeof(float);
  if (hpt_device_malloc(ctx, bytes, &dFrame) != HPT_OK || hpt_device_memset(ctx, dFrame, 0, bytes) != HPT_OK) return 1;
  int rc = hpt_path_trace_block_dev(ctx, (uint32_t)rank * chunk, mine * chunk, 4, (float*)dFrame, (uint32_t)spp, 0, nullptr);
  if (rc == HPT_OK) rc = hpt_reduce_framebuffer(ctx, (float*)dFrame, (size_t)N * 4, 0, nullptr);
  // all-reduce of a small "gradient": every rank contributes rank + 1
  std::vector<float> g(4096, float(rank + 1)); void* dG = nullptr;
  if (rc == HPT_OK) rc = hpt_device_malloc(ctx, g.size() * 4, &dG);
  if (rc == HPT_OK) rc = hpt_device_copy(ctx, dG, g.data(), g.size() * 4, 1);
  if (rc == HPT_OK) rc = hpt_allreduce_grad(ctx, (float*)dG, g.size(), nullptr);
  if (rc == HPT_OK) rc = hpt_device_copy(ctx, g.data(), dG, g.size() * 4, 2);
  if (rc != HPT_OK) { std::fprintf(stderr, "[comm2 rank %d]: %s\n", rank, hpt_last_error(ctx)); return 1; }
  const float want = float(world * (world + 1) / 2);
  for (float v : g) if (v != want) { std::fprintf(stderr, "[comm2 rank %d]: all-reduce gave %g, expected %g\n", rank, v, want); return 1; }
  int result = 0;
  if (rank == 0) {
    std::vector<float> sharded((size_t)N * 4), solo((size_t)N * 4, 0.0f);
    hpt_device_copy(ctx, sharded.data(), dFrame, bytes, 2);
    hpt_ctx* one = nullptr;
    if (hpt_create(0, &one) != HPT_OK || sc.upload(one) != HPT_OK || hpt_path_trace_block(one, 0, N, 4, solo.data(), (uint32_t)spp) != HPT_OK) { std::fprintf(stderr, "[comm2]: single-GPU render failed\n"); return 1; }
    size_t differ = 0; for (size_t i = 0; i < solo.size(); i++) differ += std::memcmp(&solo[i], &sharded[i], 4) != 0;
    std::printf("[hydra_hip_comm2]: %d ranks on %d device(s), %dx%d @ %d spp: reduced frame differs from the single-GPU frame in %zu floats; all-reduce ok\n", world, ndev, W, H, spp, differ);
    result = differ == 0 ? 0 : 1;
    hpt_destroy(one);
  }
  hpt_device_free(ctx, dFrame); hpt_device_free(ctx, dG);
  hpt_comm_destroy(ctx);
  hpt_destroy(ctx);
  return result;
}

int main(int argc, char** argv)
{
  if (argc < 6) { std::fprintf(stderr, "usage: %s <scene.xml> <width> <height> <spp> <idfile>\n", argv[0]); return 2; }
  const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), spp = std::atoi(argv[4]), world = 2;
  std::remove(argv[5]);
  const pid_t child = fork();                                        // before any HIP call: neither process has initialised the GPU yet
  if (child < 0) return 1;
  alarm(150);                                                        // both processes: a collective that never completes must not outlive the test
  if (child == 0) { const int rc = rankMain(1, world, argv[1], W, H, spp, argv[5]); std::fflush(nullptr); _exit(rc); }
  const int rc0 = rankMain(0, world, argv[1], W, H, spp, argv[5]);
  int status = 0; waitpid(child, &status, 0);
  const int rc1 = WIFEXITED(status) ? WEXITSTATUS(status) : 1;
  std::remove(argv[5]);
  if (rc0 == 77 || rc1 == 77) return 77;
  return rc0 ? rc0 : rc1;
}
