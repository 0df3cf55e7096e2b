// Persistent path-tracing kernel for MI355X (gfx950, wave64) and its small helper kernels.
//
// Replaces the reference's  #pragma omp parallel for over pixels x passes  (integrator_pt_host.cpp:57-73) and the
// per-path kernel chain of Integrator::PathTrace (integrator_pt.cpp:719-759):
//
//   * one lane owns one pixel (tid in the reference's tile-swizzled order, integrator_rt.cpp:13-31) for all of its passes,
//     because the pixel's RNG stream continues from pass to pass (integrator_pt.cpp:139,605); radiance is summed in VGPRs
//     in the reference's order and the framebuffer is read-modified-written once per pixel instead of once per sample;
//   * lanes never wait for each other between samples: a lane whose path ended regenerates its next sample at the top of
//     the bounce loop, so every trip through the loop traces one closest-hit ray and one shadow ray for (almost) all 64
//     lanes of the wave;
//   * pixels are handed out by a persistent-threads work queue: lanes that ran out of passes are compacted with a wave
//     ballot + mbcnt prefix sum, and ONE atomicAdd per wave fetches a contiguous run of tids for them;
//   * the grid is sized to the chip (blocksPerCU x 256 CUs), not to the image.
//
// Exit condition every wave reaches: the queue counter only grows, a lane that draws an index past the end never asks
// again, and a wave leaves the loop when none of its lanes holds a live path or a pixel.
#include <hip/hip_runtime.h>
#include "hpt_decl.h"

namespace hpt {

template <bool STATS, bool DR, int MODE, bool DEEP, bool FLAT, bool MOTION, bool SWEEP>
__global__ void HPT_PT_BOUNDS(DR, MODE) pathTraceKernel(const DevScene S, const Job job)
{
  constexpr bool NAIVE = (MODE == 1 || MODE == 5), INRAYS = (MODE == 2 || MODE == 6), LEAN = (MODE == 3), FILM = (MODE >= 4 && MODE <= 6);   // (MODE 7 = MODE 0 built for one more wave per SIMD)
  __shared__ uint stackMem[LDS_STACK * 256];
  const uint glane = blockIdx.x * 256u + threadIdx.x;                    // slot in the per-lane HBM buffers
  TravStack stk; stk.lds = &stackMem[threadIdx.x]; stk.ovf = job.stackOverflow + glane; stk.ovfStride = job.gridLanes;

  // ---- per-lane state -----------------------------------------------------------------------------------------------
  // Cold per-PIXEL state (touched once per path, not per bounce) is parked in LDS so that it does not occupy VGPRs
  // across the two traversals and the shading code: running framebuffer value, packed (x,y), tid, passes left.
  __shared__ float coldPix[3 * 256];
  __shared__ uint  coldU[3 * 256];
  __shared__ uint  drStage[DR ? 4 * DR_STAGE_DWORDS : 1];               // PathTraceDR: the waves' staging areas of the cooperative gradient scatter
#define PIX(k)      coldPix[(k) * 256 + threadIdx.x]
#define PIX_XY      coldU[0 * 256 + threadIdx.x]
#define PIX_TID     coldU[1 * 256 + threadIdx.x]
#define PIX_PASSES  coldU[2 * 256 + threadIdx.x]
  bool havePixel = false, alive = false, drained = false;
  uint bounce = 0, flags = 0;
  Rng  gen; gen.sx = gen.sy = 0;
  V3   rpos = v3(0, 0, 0), rdir = v3(0, 0, 1);
  V3   accum = v3(0, 0, 0), thr = v3(1, 1, 1);
  float misPdf = 1.0f, misIor = 1.0f;
  float pathTime = 0.0f;                 // motion blur: the path's time in [0, 1] (only the MOTION variants ever change or read it)
  TravStats st; st.nodes = st.tris = st.insts = st.waveNodeIters = st.waveTriIters = 0;
  uint nRays = 0, nShadow = 0, nHits = 0, nPaths = 0;
  float lossLocal = 0.0f;
  const uint maxBounce = NAIVE ? S.traceDepth + 1u : S.traceDepth;
  // diagnostic stamps (STATS build only; never in a timed kernel): where a wave's cycles go, phase by phase
  unsigned long long tPh[7] = {0, 0, 0, 0, 0, 0, 0}, tTrips = 0, tPrev = 0;
  unsigned long long nRec = 0, nRecTex = 0, nSweepTrips = 0, nSweepLanes = 0, nAtomInst = 0, nSweepBounces = 0, nRecStored = 0;   // PathTraceDR probe (STATS && DR)
#define STAMP(i) do { if (STATS) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tPh[i] += tn - tPrev; tPrev = tn; } } while (0)
  if (STATS) tPrev = __builtin_amdgcn_s_memtime();

  while (true) {
    // ---- (1) a finished pixel goes back to HBM: one read-modify-write per pixel and call ------------------------------
    if (!alive && havePixel && PIX_PASSES == 0u) {
      const uint XY = PIX_XY;
      const uint pixel = INRAYS ? PIX_TID : ((XY & 0xFFFF0000u) >> 16) * (uint)S.winWidth + (XY & 0x0000FFFFu);
      if (job.channels == 1) job.outColor[pixel] = PIX(0);
      else { float* o = job.outColor + (size_t)pixel * job.channels; o[0] = PIX(0); o[1] = PIX(1); o[2] = PIX(2); }
      job.gens[PIX_TID] = gen;                                           // kernel_ContributeToImage: m_randomGens[tid] = *gen (:605)
      havePixel = false;
    }
    // ---- (2) work queue: ballot the idle lanes, one atomic per wave, prefix-sum the ranks --------------------------------
    {
      const bool need = !alive && !havePixel && !drained;
      const unsigned long long mask = __ballot(need);
      if (mask != 0ull) {
        uint base = 0;
        if (need && mbcnt64(mask) == 0u) base = atomicAdd(job.queue, (uint)__popcll(mask));
        base = __shfl(base, (int)(__ffsll((long long)mask) - 1));
        if (need) {
          const uint k = base + mbcnt64(mask);
          const uint tid = job.tidBegin + (k / job.tidChunk) * job.tidChunk * job.tidStride + (k % job.tidChunk);
          if (k < job.tidCount && tid < job.tidEnd) {
            const uint XY = INRAYS ? 0u : job.packedXY[tid];
            gen = job.gens[tid];
            const uint pixel = INRAYS ? tid : ((XY & 0xFFFF0000u) >> 16) * (uint)S.winWidth + (XY & 0x0000FFFFu);
            if (job.channels == 1) { PIX(0) = job.outColor[pixel]; PIX(1) = 0.0f; PIX(2) = 0.0f; }
            else { const float* o = job.outColor + (size_t)pixel * job.channels; PIX(0) = o[0]; PIX(1) = o[1]; PIX(2) = o[2]; }
            PIX_XY = XY; PIX_TID = tid; PIX_PASSES = job.passNum;
            havePixel = true;
          } else drained = true;
        }
      }
    }
    // ---- (3) regenerate: next pass of the pixel (kernel_InitEyeRay2) -----------------------------------------------------
    if (!alive && havePixel) {                                              // (a pixel with no pass left was flushed in (1))
      PIX_PASSES = PIX_PASSES - 1u;
      accum = v3(0, 0, 0); thr = v3(1, 1, 1); flags = 0; bounce = 0;
      misPdf = 1.0f; misIor = 1.0f;
      if (INRAYS) {                                                        // kernel_InitEyeRayFromInput: no lens randoms
        const float4 ip = job.inRayPos[PIX_TID], id4 = job.inRayDir[PIX_TID];
        const V3 org = v3(ip.x, ip.y, ip.z), dir = v3(id4.x, id4.y, id4.z);
        const V3 p1 = mul4x3(S.worldViewInv, org);                         // transform_ray3f (cglobals.h:254-263)
        const V3 p2 = mul4x3(S.worldViewInv, org + 100.0f * dir);
        rpos = p1; rdir = normalize(p2 - p1);
        if (MOTION) pathTime = id4.w;                                      // *time = rayDirData.time (integrator_pt.cpp:197)
      } else {
        const V4 lens = rng_float4(gen);                                   // GetRandomNumbersLens
        const uint XY = PIX_XY;
        cameraRay<!(DR || LEAN)>(S, XY & 0x0000FFFFu, (XY & 0xFFFF0000u) >> 16, lens, rpos, rdir);
        if (MOTION) pathTime = rng_float1(gen);                            // GetRandomNumbersTime (integrator_pt.cpp:114-115): one step per path, after the lens
      }
      alive = true;
      if (STATS) nPaths++;
    }
    if (!__any(alive)) break;
    STAMP(0); if (STATS) tTrips++;

    // ---- (4) closest hit: kernel_RayTrace2 -> RayQuery_NearestHit ----------------------------------------------------------
    HitRec hit; hit.inst = 0xFFFFFFFFu; hit.prim = 0; hit.t = 0; hit.u = hit.v = 0;
    if (alive) {
      traceAny<false, STATS, DEEP, FLAT, MOTION, SWEEP>(S, rpos, rdir, 0.0f, HPT_FLT_MAX, hit, stk, st, pathTime);
      if (STATS) nRays++;
    }

    STAMP(1);
    // ---- (5) surface, next-event estimation set-up, emission, BSDF sampling -------------------------------------------------
    bool wantShadow = false;
    V3 shPos = v3(0, 0, 0), shDir = v3(0, 0, 1); float shFar = 0.0f;
    V3 contrib = v3(0, 0, 0);                        // thr * shadeColor, added after the shadow ray returns
    // DR record of this bounce
    V3 recA = v3(0, 0, 0), recS = v3(0, 0, 0), recdA = v3(0, 0, 0), recdS = v3(0, 0, 0); Taps recTaps; uint recTex = 0xFFFFFFFFu;
    V3 tailR = v3(0, 0, 0);                          // emission picked up at the terminating vertex, per unit throughput
    const V3 thrBefore = thr;
    bool didBounce = false;
    for (int k = 0; k < 4; k++) { recTaps.off[k] = 0; recTaps.w[k] = 0.0f; }
    recTaps.fx = recTaps.fy = 0.0f; recTaps.base = recTaps.ch = 0u;

    if (alive) {
      if (STATS && hit.inst != 0xFFFFFFFFu) nHits++;
      didBounce = shadeVertex<DR, NAIVE, LEAN, MOTION, FILM>(S, job.data, hit, rpos, rdir, accum, thr, misPdf, misIor, flags, bounce, gen,
                                                 wantShadow, shPos, shDir, shFar, contrib, recA, recS, recdA, recdS, recTaps, recTex, tailR, pathTime);
    }

    STAMP(2);
    // ---- (6) shadow rays: RayQuery_AnyHit ---------------------------------------------------------------------------------------
    if (wantShadow) {
      HitRec sh;
      const bool occluded = traceAny<true, STATS, DEEP, FLAT, MOTION, SWEEP>(S, shPos, shDir, 0.0f, shFar, sh, stk, st, pathTime);
      if (STATS) { nRays++; nShadow++; }
      if (!occluded) accum = accum + contrib; else if (DR) { recS = v3(0, 0, 0); recdS = v3(0, 0, 0); }
    } else if (DR) { recS = v3(0, 0, 0); recdS = v3(0, 0, 0); }

    STAMP(3);
    // ---- (7) bookkeeping: adjoint record, end of path ------------------------------------------------------------------------------
    bool closing = false;                              // DR: this lane's path ended in this trip and its gradient is to be scattered
    V3 sweepDiff = v3(0, 0, 0), sweepTail = v3(0, 0, 0);
    DrRec lastRec = drEmptyRecord(); bool lastInRegs = false;
    if (alive) {
      if (didBounce) bounce++;
      const bool ended = (flags & RAY_FLAG_IS_DEAD) != 0 || bounce >= maxBounce;
      if (DR && didBounce) {
        // the record of a bounce goes to HBM only when the path goes on: a path that ends here (bounce limit, FB_DIRECT) sweeps it from registers
        lastRec = drMakeRecord(recA, recS, recdA, recdS, thrBefore, recTex, recTaps);
        lastInRegs = ended;
        STAMP(4);
#ifndef HPT_DBG_DR_NOSTORE   // diagnostic builds only (profiles/dr_ab.sh): what do the record stores cost?
        if (!ended) drStoreRecord(job.record, job.recordLanes, glane, bounce - 1u, lastRec);
#endif
        if (STATS) { nRec++; if (recTex != 0xFFFFFFFFu) nRecTex++; if (!ended) nRecStored++; }
        STAMP(5);
      }
      if (ended) {
        // kernel_HitEnvironment (integrator_pt.cpp:550-595), constant environment colour.
        // The DR replay adds the environment term unconditionally (diff_render/integrator_dr.cpp:1077-1098).
        const V3 env = DR ? ld3(S.envColor) : environmentRadiance(S, rdir, misPdf, flags, INRAYS ? (PIX_TID < job.packedCount ? job.packedXY[PIX_TID] : 0u) : PIX_XY);
        if (DR) accum = accum + thr * env;
        else if ((flags & RAY_FLAG_OUT_OF_SCENE) != 0) {
          if (S.integratorType == INTEGRATOR_STUPID_PT) accum = thr * env; else accum = accum + thr * env;
        }
        if (DR) {
          // PixelLossPT (integrator_dr.cpp:1103-1132) + hand-derived reverse sweep replacing __enzyme_autodiff (:1172-1183), see drReverseSweep
          const uint pitch = (uint)S.winWidth;
          const uint XY = PIX_XY;
          const uint yRef = (uint)S.winHeight - ((XY & 0xFFFF0000u) >> 16) - 1u;
          const float* rp = job.refImg + ((size_t)yRef * pitch + (XY & 0x0000FFFFu)) * job.channels;
#ifdef HPT_DBG_DR_NOREF      // diagnostic builds only: the reference pixel's load at path end
          const V3 diff = v3(accum.x - 0.25f, accum.y - 0.25f, accum.z - 0.25f); (void)rp;
#else
          const V3 diff = v3(accum.x - rp[0], accum.y - rp[1], accum.z - rp[2]);
#endif
          // PixelLossPT adds every sample, finite or not (integrator_dr.cpp:1124-1131): that is the default here too. One sample in ~1e8 on
          // the 1M-triangle scene comes out non-finite (a 0/0 in a grazing GGX term of the reference's formulas, unguarded there too) and in
          // an optimisation loop a single NaN gradient poisons Adam's moments for good, so hpt_set_option("dr_skip_nonfinite", 1) lets such
          // a sample contribute neither loss, colour nor gradient.
          const bool sane = job.drSkipNonFinite == 0u || __builtin_isfinite(diff.x + diff.y + diff.z);
          if (sane) {
            lossLocal += (diff.x * diff.x + diff.y * diff.y + diff.z * diff.z) / float(job.passNum);
            PIX(0) += accum.x; PIX(1) += accum.y; PIX(2) += accum.z;           // out_color += colorRend (:1124-1126)
            closing = true; sweepDiff = diff; sweepTail = tailR + env;
          }
        } else {
          // kernel_ContributeToImage (integrator_pt.cpp:598-657)
          const V3 c = accum * ld3(S.camRespoceRGB);
          if (INRAYS) { PIX(0) += accum.x; PIX(1) += accum.y; PIX(2) += accum.z; }        // kernel_CopyColorToOutput: raw accumColor
          else if (job.channels == 1) PIX(0) += accum.x * S.exposureMult;
          else { PIX(0) += S.exposureMult * c.x; PIX(1) += S.exposureMult * c.y; PIX(2) += S.exposureMult * c.z; }
        }
        alive = false;
      }
    }
    if (DR) {
      // the reverse sweep, by the whole wave: the lanes whose paths go on help scatter the gradients of those that closed one (drReverseSweep)
      STAMP(4);
      if (__any(closing)) {
        if (STATS) { if (closing) { nSweepLanes++; nSweepBounces += bounce; } if (lane_id() == 0u) nSweepTrips++; }
#ifndef HPT_DBG_DR_NOSWEEP   // diagnostic builds only
        drReverseSweep(S, job.record, job.recordLanes, glane, closing, bounce, sweepTail, sweepDiff, job.grad, job.drSkipNonFinite != 0u,
                       drStage + (threadIdx.x >> 6) * DR_STAGE_DWORDS, lastRec, lastInRegs, STATS ? &nAtomInst : nullptr);
#endif
      }
      STAMP(6);
    }
    STAMP(4);
  }
#undef STAMP

#undef PIX
#undef PIX_XY
#undef PIX_TID
#undef PIX_PASSES
  if (STATS) {
    // wave-reduce, one atomic per counter per wave
    unsigned long long v[8] = { nRays, st.nodes, st.tris, nHits, nShadow, nPaths, st.insts, 0ull };
    { unsigned long long a = st.waveNodeIters, b = st.waveTriIters;
      for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o); b += __shfl_down(b, o); }
      if ((threadIdx.x & 63) == 0) { atomicAdd(&job.counters->v[14], a); atomicAdd(&job.counters->v[15], b); } }
    if ((threadIdx.x & 63) == 0) { for (int i = 0; i < 5; i++) atomicAdd(&job.counters->v[8 + i], tPh[i]); atomicAdd(&job.counters->v[13], tTrips); }
    if (DR) {
      if ((threadIdx.x & 63) == 0) { atomicAdd(&job.counters->v[18], tPh[5]); atomicAdd(&job.counters->v[19], tPh[6]); }
      unsigned long long w[7] = { nRec, nRecTex, nSweepTrips, nSweepLanes, nAtomInst, nSweepBounces, nRecStored };
      const int slot[7] = { 16, 17, 20, 21, 22, 23, 24 };
      for (int i = 0; i < 7; i++) {
        unsigned long long x = w[i];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&job.counters->v[slot[i]], x);
      }
    }
    for (int i = 0; i < 8; i++) {
      unsigned long long x = v[i];
      for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
      if ((threadIdx.x & 63) == 0 && x) atomicAdd(&job.counters->v[i], x);
    }
  }
  if (DR) {
    float x = lossLocal;
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(job.lossAccum, x);
  }
}

// ---- explicit instantiations: one group per translation unit (-DHPT_INST_GROUP=n, see __graft_entry__.build) ----------------------------
// DEEP: the scene's BVH can need more than LDS_STACK stack entries; FLAT: single-level world-space BVH vs two-level TLAS/BLAS
#define HPT_INST4(STATS, DR, MODE) \
  template __global__ void pathTraceKernel<STATS, DR, MODE, false, false, false>(const DevScene, const Job); \
  template __global__ void pathTraceKernel<STATS, DR, MODE, true,  false, false>(const DevScene, const Job); \
  template __global__ void pathTraceKernel<STATS, DR, MODE, false, true,  false>(const DevScene, const Job); \
  template __global__ void pathTraceKernel<STATS, DR, MODE, true,  true,  false>(const DevScene, const Job); \
  template __global__ void pathTraceKernel<STATS, DR, MODE, false, false, false, true>(const DevScene, const Job);   /* the triangle sweep of tiny scenes */
#ifndef HPT_INST_GROUP
#error "compile hpt_kernels.hip with -DHPT_INST_GROUP=1..15"
#endif
#if HPT_INST_GROUP == 1      // gltf + emissive scenes (every benchmark workload)
HPT_INST4(false, false, 3)
#elif HPT_INST_GROUP == 2    // every BSDF branch
HPT_INST4(false, false, 0)
#elif HPT_INST_GROUP == 3    // NaivePathTrace
HPT_INST4(false, false, 1)
#elif HPT_INST_GROUP == 4    // PathTraceFromInputRays
HPT_INST4(false, false, 2)
#elif HPT_INST_GROUP == 5    // instrumented
HPT_INST4(true, false, 0)
#elif HPT_INST_GROUP == 6    // PathTraceDR
HPT_INST4(false, true, 0)
#elif HPT_INST_GROUP == 15   // PathTraceDR, instrumented (phase stamps, record / sweep / atomic counts: profiles/dr_phases.py)
HPT_INST4(true, true, 0)
#elif HPT_INST_GROUP == 9    // scenes with thin films (MODE 4 / 5 / 6 = 0 / 1 / 2 + MAT_TYPE_THIN_FILM)
HPT_INST4(false, false, 4)
#elif HPT_INST_GROUP == 10
HPT_INST4(false, false, 5)
#elif HPT_INST_GROUP == 11
HPT_INST4(false, false, 6)
#elif HPT_INST_GROUP == 14   // every BSDF branch, built for 4 waves per SIMD (scenes with few material types)
HPT_INST4(false, false, 7)
#elif HPT_INST_GROUP == 12   // thin films and moving instances
template __global__ void pathTraceKernel<false, false, 4, false, false, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 4, true,  false, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 5, false, false, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 5, true,  false, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 6, false, false, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 6, true,  false, true>(const DevScene, const Job);
#elif HPT_INST_GROUP == 13   // thin films and moving instances, single-level layout
template __global__ void pathTraceKernel<false, false, 4, false, true, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 4, true,  true, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 5, false, true, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 5, true,  true, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 6, false, true, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 6, true,  true, true>(const DevScene, const Job);
#elif HPT_INST_GROUP == 7    // moving instances
template __global__ void pathTraceKernel<false, false, 0, false, false, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 0, true,  false, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 1, false, false, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 1, true,  false, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 2, false, false, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 2, true,  false, true>(const DevScene, const Job);
#elif HPT_INST_GROUP == 8    // moving instances, single-level layout
template __global__ void pathTraceKernel<false, false, 0, false, true, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 0, true,  true, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 1, false, true, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 1, true,  true, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 2, false, true, true>(const DevScene, const Job);
template __global__ void pathTraceKernel<false, false, 2, true,  true, true>(const DevScene, const Job);
#endif

} // namespace hpt
