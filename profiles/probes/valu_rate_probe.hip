// Issue cost of f32 VALU instruction kinds on gfx950, all SIMDs busy: plain v_mul / v_add / v_fma, the packed forms (two floats per lane
// per instruction) and the IEEE-division helpers. Cycles are derived from the wall time at the clock the run itself measures
// (s_memtime ticks over wall time), so DVFS does not bias the figure.
// Build + run on a GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate profiles/probes/valu_rate_probe.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) probe(float* out, int iters, unsigned long long* ticks)
{
  float a = (float)threadIdx.x * 1e-3f + 1.0f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f, f = 0.3f, g = 0.7f, h = 0.9f;
  const float k = 1.0000001f;
  f2 pa = {a, b}, pb = {c, d}, pc = {e, f}, pd = {g, h}; const f2 pk = {k, k};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (KIND == 0)       // 8 independent v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %0\n\tv_fma_f32 %1, %1, %8, %1\n\tv_fma_f32 %2, %2, %8, %2\n\tv_fma_f32 %3, %3, %8, %3\n\t"
                   "v_fma_f32 %4, %4, %8, %4\n\tv_fma_f32 %5, %5, %8, %5\n\tv_fma_f32 %6, %6, %8, %6\n\tv_fma_f32 %7, %7, %8, %7"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(k));
    else if (KIND == 1)  // 8 independent v_mul_f32
      asm volatile("v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\t"
                   "v_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(k));
    else if (KIND == 2)  // 8 independent v_add_f32
      asm volatile("v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\t"
                   "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(k));
    else if (KIND == 3)  // 8 v_pk_mul_f32 on 4 independent pairs (2 rounds)
      asm volatile("v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                   "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4"
                   : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pk));
    else if (KIND == 4)  // 8 v_pk_fma_f32
      asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n\tv_pk_fma_f32 %1, %1, %4, %1\n\tv_pk_fma_f32 %2, %2, %4, %2\n\tv_pk_fma_f32 %3, %3, %4, %3\n\t"
                   "v_pk_fma_f32 %0, %0, %4, %0\n\tv_pk_fma_f32 %1, %1, %4, %1\n\tv_pk_fma_f32 %2, %2, %4, %2\n\tv_pk_fma_f32 %3, %3, %4, %3"
                   : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pk));
    else if (KIND == 5)  // 8 v_pk_add_f32
      asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4\n\t"
                   "v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4"
                   : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pk));
    else if (KIND == 6)  // 8 v_rcp_f32
      asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t"
                   "v_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b));
    else if (KIND == 7)  // 8 v_cndmask_b32 (vcc select)
      asm volatile("v_cmp_gt_f32 vcc, %0, %8\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t"
                   "v_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(k) : "vcc");
    else if (KIND == 8)  // one full IEEE division a = a / k per chain x 2 chains (what the triangle test's 1 / det costs)
    { a = a / k; c = c / k; }
    else if (KIND == 9)  // 8 v_fma_f32 with an SGPR operand
      asm volatile("v_fma_f32 %0, %0, %8, %0\n\tv_fma_f32 %1, %1, %8, %1\n\tv_fma_f32 %2, %2, %8, %2\n\tv_fma_f32 %3, %3, %8, %3\n\t"
                   "v_fma_f32 %4, %4, %8, %4\n\tv_fma_f32 %5, %5, %8, %5\n\tv_fma_f32 %6, %6, %8, %6\n\tv_fma_f32 %7, %7, %8, %7"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "s"(k));
    else if (KIND == 10) // 8 v_cmp_gt_f32 writing VCC
      asm volatile("v_cmp_gt_f32 vcc, %0, %8\n\tv_cmp_gt_f32 vcc, %1, %8\n\tv_cmp_gt_f32 vcc, %2, %8\n\tv_cmp_gt_f32 vcc, %3, %8\n\t"
                   "v_cmp_gt_f32 vcc, %4, %8\n\tv_cmp_gt_f32 vcc, %5, %8\n\tv_cmp_gt_f32 vcc, %6, %8\n\tv_cmp_gt_f32 vcc, %7, %8"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(k) : "vcc");
    else if (KIND == 11) // 8 v_cndmask_b32, VCC written once before the loop
      asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t"
                   "v_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(k) : "vcc");
    else if (KIND == 12) // 8 v_cmp_gt_f32 writing an SGPR pair (VOP3)
      asm volatile("v_cmp_gt_f32 s[20:21], %0, %8\n\tv_cmp_gt_f32 s[22:23], %1, %8\n\tv_cmp_gt_f32 s[24:25], %2, %8\n\tv_cmp_gt_f32 s[26:27], %3, %8\n\t"
                   "v_cmp_gt_f32 s[20:21], %4, %8\n\tv_cmp_gt_f32 s[22:23], %5, %8\n\tv_cmp_gt_f32 s[24:25], %6, %8\n\tv_cmp_gt_f32 s[26:27], %7, %8"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(k) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    else if (KIND == 13) // 8 v_cndmask_b32 with an SGPR-pair mask (VOP3), mask constant
      asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\n\tv_cndmask_b32 %1, %1, %8, s[20:21]\n\tv_cndmask_b32 %2, %2, %8, s[20:21]\n\tv_cndmask_b32 %3, %3, %8, s[20:21]\n\t"
                   "v_cndmask_b32 %4, %4, %8, s[20:21]\n\tv_cndmask_b32 %5, %5, %8, s[20:21]\n\tv_cndmask_b32 %6, %6, %8, s[20:21]\n\tv_cndmask_b32 %7, %7, %8, s[20:21]"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(k) : "s20", "s21");
    else if (KIND == 14) // 8 v_min_f32
      asm volatile("v_min_f32 %0, %0, %8\n\tv_min_f32 %1, %1, %8\n\tv_min_f32 %2, %2, %8\n\tv_min_f32 %3, %3, %8\n\t"
                   "v_min_f32 %4, %4, %8\n\tv_min_f32 %5, %5, %8\n\tv_min_f32 %6, %6, %8\n\tv_min_f32 %7, %7, %8"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(k));
    else if (KIND == 15) // 8 v_med3_f32
      asm volatile("v_med3_f32 %0, %0, %8, %1\n\tv_med3_f32 %1, %1, %8, %2\n\tv_med3_f32 %2, %2, %8, %3\n\tv_med3_f32 %3, %3, %8, %4\n\t"
                   "v_med3_f32 %4, %4, %8, %5\n\tv_med3_f32 %5, %5, %8, %6\n\tv_med3_f32 %6, %6, %8, %7\n\tv_med3_f32 %7, %7, %8, %0"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(k));
    else if (KIND == 16) // v_cmp -> VCC followed by ONE dependent v_cndmask, then 6 independent v_mul
      asm volatile("v_cmp_gt_f32 vcc, %0, %8\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\t"
                   "v_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(k) : "vcc");
    else if (KIND == 17) // 8 v_and_b32 (integer)
      asm volatile("v_and_b32 %0, %0, %8\n\tv_and_b32 %1, %1, %8\n\tv_and_b32 %2, %2, %8\n\tv_and_b32 %3, %3, %8\n\t"
                   "v_and_b32 %4, %4, %8\n\tv_and_b32 %5, %5, %8\n\tv_and_b32 %6, %6, %8\n\tv_and_b32 %7, %7, %8"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b) : "v"(k));
    else if (KIND == 18) // 8 v_sqrt_f32
      asm volatile("v_sqrt_f32 %0, %0\n\tv_sqrt_f32 %1, %1\n\tv_sqrt_f32 %2, %2\n\tv_sqrt_f32 %3, %3\n\t"
                   "v_sqrt_f32 %4, %4\n\tv_sqrt_f32 %5, %5\n\tv_sqrt_f32 %6, %6\n\tv_sqrt_f32 %7, %7"
                   : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h), "+v"(b));
    else if (KIND == 19) // one dependent chain of 8 v_mul_f32 (latency)
      asm volatile("v_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\t"
                   "v_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1"
                   : "+v"(a) : "v"(k));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + pa.x + pa.y + pb.x + pb.y + pc.x + pc.y + pd.x + pd.y;
}

template <int KIND> static void run(const char* name, float* d, unsigned long long* dt, int wavesPerSimd, double instPerIter)
{
  const int nb = 256 * wavesPerSimd, iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<KIND><<<nb, 256>>>(d, 100, dt); hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<KIND><<<nb, 256>>>(d, iters, dt);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long ticks = 0; hipMemcpy(&ticks, dt, 8, hipMemcpyDeviceToHost);
  const double instPerSimd = (double)wavesPerSimd * iters * instPerIter;      // wave-instructions each SIMD issues
  // __builtin_readcyclecounter = s_memtime: a constant 100 MHz counter on this chip, so the shader clock is not derived from it; cycles are quoted at 2.4 GHz
  printf("%d waves/SIMD  %-28s %8.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz (block 0: %llu ticks)\n", wavesPerSimd, name, ms, ms * 1e-3 * 2.4e9 / instPerSimd, ticks);
}

int main()
{
  float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  unsigned long long* dt; hipMalloc(&dt, 8);
  for (int w : { 4, 1 }) {
    run<0>("v_fma_f32", d, dt, w, 8); run<1>("v_mul_f32", d, dt, w, 8); run<2>("v_add_f32", d, dt, w, 8);
    run<3>("v_pk_mul_f32 (2 flt/lane)", d, dt, w, 8); run<4>("v_pk_fma_f32", d, dt, w, 8); run<5>("v_pk_add_f32", d, dt, w, 8);
    run<6>("v_rcp_f32", d, dt, w, 8); run<7>("v_cmp + 7 v_cndmask", d, dt, w, 8); run<8>("2 IEEE divisions (per div)", d, dt, w, 2);
    run<9>("v_fma_f32, SGPR operand", d, dt, w, 8);
    run<10>("v_cmp -> vcc", d, dt, w, 8); run<11>("v_cndmask, vcc constant", d, dt, w, 8); run<12>("v_cmp -> sgpr pair", d, dt, w, 8);
    run<13>("v_cndmask, sgpr mask", d, dt, w, 8); run<14>("v_min_f32", d, dt, w, 8); run<15>("v_med3_f32", d, dt, w, 8);
    run<16>("cmp + cndmask + 6 mul", d, dt, w, 8); run<17>("v_and_b32", d, dt, w, 8); run<18>("v_sqrt_f32", d, dt, w, 8);
    run<19>("dependent v_mul chain", d, dt, w, 8);
  }
  return 0;
}
