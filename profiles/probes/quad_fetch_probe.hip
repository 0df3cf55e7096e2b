// How should a lane fetch its own random 64-byte record (a BVH node) on gfx950?
//   walk64   four global_load_dwordx4 per lane, every lane its own line (what the trace kernels do today): 4 x 64 distinct lines per wave and step
//   walk16   one dwordx4 per lane (a quarter of the record): the request-count control
//   walk64q  the four lanes of a quad fetch the quad's four records TOGETHER: in load j lane i reads piece i of the record of quad lane j
//            (64 contiguous bytes per quad and instruction), then the 4 x 4 pieces are transposed inside the quad with DPP moves
// Same records, same dependent index chain, same bytes used. Timing only (HIP events).
//   hipcc -O3 --offload-arch=gfx950 profiles/probes/quad_fetch_probe.hip -o /tmp/quad_probe && /tmp/quad_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ float sum4(const float4 v) { return v.x + v.y + v.z + v.w; }

__global__ void __launch_bounds__(256, 5) walk64(const float4* __restrict__ recs, unsigned mask, int steps, unsigned* out)
{
  unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  float acc = 0.0f;
  for (int s = 0; s < steps; s++) {
    idx &= mask;
    const float4* p = recs + (size_t)idx * 4;
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc += sum4(a) + 2.0f * sum4(b) + 3.0f * sum4(c) + 4.0f * sum4(d);
    idx = idx * 1664525u + 1013904223u + __float_as_uint(a.w);
  }
  if (acc == 123.456f) out[0] = idx;
  if (blockIdx.x == 0u && threadIdx.x < 64u) out[1 + threadIdx.x] = idx;
}
__global__ void __launch_bounds__(256, 5) walk16(const float4* __restrict__ recs, unsigned mask, int steps, unsigned* out)
{
  unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  float acc = 0.0f;
  for (int s = 0; s < steps; s++) {
    idx &= mask;
    const float4 a = recs[(size_t)idx * 4];
    acc += sum4(a);
    idx = idx * 1664525u + 1013904223u + __float_as_uint(a.w);
  }
  if (acc == 123.456f) out[0] = idx;
  if (blockIdx.x == 0u && threadIdx.x < 64u) out[1 + threadIdx.x] = idx;
}

template <int J> __device__ __forceinline__ unsigned quadBcast(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, J * 0x55, 0xF, 0xF, true); }   // quad_perm(J, J, J, J)
template <int J> __device__ __forceinline__ float4 quadBcast4(const float4 v)
{
  return make_float4(__uint_as_float(quadBcast<J>(__float_as_uint(v.x))), __uint_as_float(quadBcast<J>(__float_as_uint(v.y))),
                     __uint_as_float(quadBcast<J>(__float_as_uint(v.z))), __uint_as_float(quadBcast<J>(__float_as_uint(v.w))));
}
__device__ __forceinline__ float4 sel4(bool c, const float4 a, const float4 b) { return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }

// piece k of MY record: lane k of the quad holds it in X[me]
#define PIECE(K, OUT) do { \
    const float4 t0 = quadBcast4<K>(X0), t1 = quadBcast4<K>(X1), t2 = quadBcast4<K>(X2), t3 = quadBcast4<K>(X3); \
    OUT = sel4(me == 0u, t0, sel4(me == 1u, t1, sel4(me == 2u, t2, t3))); } while (0)

__global__ void __launch_bounds__(256, 5) walk64q(const float4* __restrict__ recs, unsigned mask, int steps, unsigned* out)
{
  unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  const unsigned me = threadIdx.x & 3u;
  float acc = 0.0f;
  for (int s = 0; s < steps; s++) {
    idx &= mask;
    const float4 X0 = recs[(size_t)quadBcast<0>(idx) * 4 + me];
    const float4 X1 = recs[(size_t)quadBcast<1>(idx) * 4 + me];
    const float4 X2 = recs[(size_t)quadBcast<2>(idx) * 4 + me];
    const float4 X3 = recs[(size_t)quadBcast<3>(idx) * 4 + me];
    float4 a, b, c, d;
    PIECE(0, a); PIECE(1, b); PIECE(2, c); PIECE(3, d);
    acc += sum4(a) + 2.0f * sum4(b) + 3.0f * sum4(c) + 4.0f * sum4(d);
    idx = idx * 1664525u + 1013904223u + __float_as_uint(a.w);
  }
  if (acc == 123.456f) out[0] = idx;
  if (blockIdx.x == 0u && threadIdx.x < 64u) out[1 + threadIdx.x] = idx;
}

int main(int argc, char** argv)
{
  const bool zero = argc > 1 && argv[1][0] == 'z';      // all-zero table: the index chain is the LCG alone (as profiles/probes/hbm_counter_probe.hip has it)
  const size_t big = size_t(1) << 30;
  float4* d = nullptr; unsigned* out = nullptr; unsigned* outq = nullptr;
  if (hipMalloc(&d, big) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess || hipMalloc(&outq, 4096) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
  // records with content (the transposition is checked: both kernels must end on the same index chain)
  if (zero) (void)hipMemset(d, 0, big);
  else {
    const size_t n = big / 4;
    unsigned* h = (unsigned*)std::malloc(big);
    unsigned x = 12345u;
    for (size_t i = 0; i < n; i++) { x = x * 1664525u + 1013904223u; h[i] = (x >> 9) | 0x3F800000u; }   // floats in [1, 2)
    (void)hipMemcpy(d, h, big, hipMemcpyHostToDevice);
    std::free(h);
  }
  const int blocks = 256 * 5, steps = 256;
  const double lanes = double(blocks) * 256.0;
  struct { const char* name; size_t bytes; } tabs[4] = { {"2MiB", size_t(2) << 20}, {"16MiB", size_t(16) << 20}, {"128MiB", size_t(128) << 20}, {"1GiB", big} };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++)
    for (auto& t : tabs) {
      const unsigned mask = (unsigned)(t.bytes / 64 - 1);
      float ms = 0;
      hipEventRecord(e0); walk64<<<blocks, 256>>>(d, mask, steps, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      std::printf("walk64  %-6s rep %d  %.3f ms  %.1f G records/s  %.1f GB/s\n", t.name, rep, ms, lanes * steps / ms / 1e6, lanes * steps * 64.0 / ms / 1e6);
      hipEventRecord(e0); walk16<<<blocks, 256>>>(d, mask, steps, outq); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      std::printf("walk16  %-6s rep %d  %.3f ms  %.1f G records/s\n", t.name, rep, ms, lanes * steps / ms / 1e6);
      hipEventRecord(e0); walk64q<<<blocks, 256>>>(d, mask, steps, outq); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      unsigned ha[65], hb[65];
      (void)hipMemcpy(ha, out, 65 * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hb, outq, 65 * 4, hipMemcpyDeviceToHost);
      bool same = true; for (int i = 1; i < 65; i++) same = same && ha[i] == hb[i];
      std::printf("walk64q %-6s rep %d  %.3f ms  %.1f G records/s  %.1f GB/s  %s\n", t.name, rep, ms, lanes * steps / ms / 1e6, lanes * steps * 64.0 / ms / 1e6, same ? "(same chain as walk64)" : "(CHAIN DIFFERS)");
    }
  hipFree(d); hipFree(out); hipFree(outq);
  return 0;
}
