// Does a wave64 VALU instruction cost less when only one 32-lane half of EXEC is populated? (gfx950: SIMD-32, 2 passes per wave64 op)
// Build + run on a GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/halfwave profiles/probes/halfwave_probe.hip && /tmp/halfwave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) probe(float* out, int mode, int iters)
{
  const unsigned lane = threadIdx.x & 63u;
  bool on = true;
  if (mode == 1) on = lane < 32u;            // low half only
  if (mode == 2) on = lane >= 32u;           // high half only
  if (mode == 3) on = (lane & 1u) == 0u;     // 32 lanes spread over both halves
  if (mode == 4) on = lane < 16u;
  if (mode == 5) on = lane == 0u;
  if (mode == 6) on = lane < 33u;            // one lane of the upper half
  float a = (float)threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f, f = 0.3f, g = 0.7f, h = 0.9f;
  if (on) {
    for (int i = 0; i < iters; i++) {
      // 8 independent chains: issue-bound, not latency-bound
      asm volatile("v_fma_f32 %0, %0, %8, %0\n\tv_fma_f32 %1, %1, %8, %1\n\tv_fma_f32 %2, %2, %8, %2\n\tv_fma_f32 %3, %3, %8, %3\n\t"
                   "v_fma_f32 %4, %4, %8, %4\n\tv_fma_f32 %5, %5, %8, %5\n\tv_fma_f32 %6, %6, %8, %6\n\tv_fma_f32 %7, %7, %8, %7"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(1.0000001f));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h;
}

int main()
{
  const int blocks = 256 * 8, iters = 20000;      // 8 blocks per CU x 4 waves = 8 waves per SIMD
  float* d; hipMalloc(&d, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[7] = { "all 64 lanes", "lanes 0-31", "lanes 32-63", "even lanes (32)", "lanes 0-15", "lane 0", "lanes 0-32" };
  for (int wavesPerSimd : { 8, 4, 2, 1 }) {
    const int nb = 256 * wavesPerSimd;
    for (int mode = 0; mode < 7; mode++) {
      probe<<<nb, 256>>>(d, mode, 100);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      probe<<<nb, 256>>>(d, mode, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double inst = (double)nb * 4 * iters * 8;                 // wave-instructions
      printf("%d waves/SIMD, %-16s: %8.3f ms, %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", wavesPerSimd, names[mode], ms, ms * 1e-3 * 2.4e9 / (inst / 1024.0));
    }
  }
  return 0;
}
