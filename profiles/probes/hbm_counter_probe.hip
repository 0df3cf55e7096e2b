// Calibration of rocprofv3's FETCH_SIZE for the access pattern of a BVH walk on gfx950: every lane fetches one random
// 64-byte record per step with four global_load_dwordx4 (as traceRayFlat / wfTraceKernel fetch a node), the next index depends on the data.
// Known byte count = lanes x steps x 64 B. Tables: 2 MiB (L2-resident: the counter should read ~0), 128 MiB (Infinity-Cache resident: the guide
// says such hits ARE counted), 4 GiB (beyond every cache: HBM). A coalesced 16-B-per-lane streaming read of the same bytes is the control
// (MI355X_MICROARCH.md: FETCH_SIZE reports exactly half of those bytes).
//   hipcc -O3 --offload-arch=gfx950 profiles/probes/hbm_counter_probe.hip -o /tmp/hbm_probe
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/hbmprobe -- /tmp/hbm_probe          (FETCH_SIZE takes 3 of the 4 TCC slots:
//   rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/hbmprobe2 -- /tmp/hbm_probe    more in one pass is refused, and the refused run hangs)
// profiles/probes/hbm_counter_probe.py turns the counter CSV into the ratios.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256, 5) walk64(const float4* __restrict__ recs, unsigned mask, int steps, unsigned* out)
{
  unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  float acc = 0.0f;
  for (int s = 0; s < steps; s++) {
    idx &= mask;
    const float4* p = recs + (size_t)idx * 4;
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc += a.x + b.y + c.z + d.w;
    idx = idx * 1664525u + 1013904223u + __float_as_uint(a.w);
  }
  if (acc == 123.456f) out[0] = idx;
  out[1 + (threadIdx.x & 63)] = idx;
}
// the same, one 16-byte piece of the record only (a 64-byte line touched through one dwordx4: what a triangle or shading-record fetch's first load sees)
__global__ void __launch_bounds__(256, 5) walk16(const float4* __restrict__ recs, unsigned mask, int steps, unsigned* out)
{
  unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  float acc = 0.0f;
  for (int s = 0; s < steps; s++) {
    idx &= mask;
    const float4 a = recs[(size_t)idx * 4];
    acc += a.x;
    idx = idx * 1664525u + 1013904223u + __float_as_uint(a.w);
  }
  if (acc == 123.456f) out[0] = idx;
  out[1 + (threadIdx.x & 63)] = idx;
}
__global__ void __launch_bounds__(256, 5) stream16(const float4* __restrict__ recs, size_t n4, unsigned* out)
{
  float acc = 0.0f;
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256u) { const float4 a = recs[i]; acc += a.x + a.w; }
  if (acc == 123.456f) out[0] = 1u;
}

int main()
{
  const size_t big = size_t(4) << 30;
  float4* d = nullptr; unsigned* out = nullptr;
  if (hipMalloc(&d, big) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
  (void)hipMemset(d, 0, big);                         // a.w = 0: the index chain is the LCG alone (every lane its own pseudo-random sequence)
  const int blocks = 256 * 5, steps = 256;
  const double lanes = double(blocks) * 256.0;
  struct { const char* name; size_t bytes; } tabs[3] = { {"2MiB", size_t(2) << 20}, {"128MiB", size_t(128) << 20}, {"4GiB", big} };
  for (int rep = 0; rep < 2; rep++)             // (first pass warms the caches for the small tables)
    for (auto& t : tabs) {
      const unsigned mask = (unsigned)(t.bytes / 64 - 1);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0); walk64<<<blocks, 256>>>(d, mask, steps, out); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      std::printf("KNOWN walk64 %s rep %d bytes %.0f ms %.3f -> %.1f GB/s\n", t.name, rep, lanes * steps * 64.0, ms, lanes * steps * 64.0 / ms / 1e6);
      hipEventRecord(e0); walk16<<<blocks, 256>>>(d, mask, steps, out); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      std::printf("KNOWN walk16 %s rep %d bytes %.0f ms %.3f (16 B used per 64-B record; lines touched: %.0f B)\n", t.name, rep, lanes * steps * 16.0, ms, lanes * steps * 64.0);
    }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); stream16<<<256 * 8, 256>>>(d, big / 16, out); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::printf("KNOWN stream16 4GiB rep 0 bytes %.0f ms %.3f -> %.1f GB/s\n", double(big), ms, double(big) / ms / 1e6);
  hipFree(d); hipFree(out);
  return 0;
}
