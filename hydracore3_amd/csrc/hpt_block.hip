// Third schedule of the same path tracer: the persistent megakernel with BLOCK-LOCAL ray repacking, for scenes between the two existing
// schedules' sweet spots - medium trees (the reference's 8 202-triangle test_228: 9 node visits per ray on average, 48 for the slowest
// lane of a wave) and calls with too few pixels to keep the wavefront schedule's chip-wide ray queue supplied (a multi-GPU share of a frame).
//
// Why. In hpt_kernels.hip a lane keeps its ray until the slowest lane of its wave has finished: measured on the test_228 class the node loop
// runs with 0.19 of the lanes busy and the two traversals are 60 % of the wave cycles (profiles/dr_phases.py). hpt_wavefront.hip repairs
// that with a chip-wide ray queue in HBM, which costs a pool of path state through HBM per bounce and two kernel launches per bounce - more
// than it gains on such scenes (531 vs 667 Mpaths/s). Here the exchange happens inside a workgroup, through LDS:
//
//   S  every lane owns one pixel's current path, state in VGPRs exactly as in the megakernel: it folds the result of its last shadow ray into
//      the path, shades the vertex its closest-hit ray found (the SAME shadeVertex), closes / regenerates paths, and appends its next rays -
//      the closest-hit ray of the next bounce AND the shadow ray of this vertex, both known after shading - to the block's ray pool in LDS
//      (a wave ballot + mbcnt prefix sum per wave, one LDS atomic per wave);
//   T  the block's four waves then drain the pool together with ray replacement: a wave whose lanes have thinned out below `refillBelow`
//      leaves the node loop and refills the idle lanes from the pool (one LDS atomic per refill), so lanes are busy whatever the spread of
//      per-ray work, and the slowest ray of 512 bounds the phase instead of the slowest of each 64 twice per bounce.
//
// Two workgroup barriers per bounce, no HBM traffic beyond what the megakernel has. The arithmetic per path is the megakernel's (same
// functions, same order, the pending light sample added when its shadow ray returns - before the next vertex is shaded - as the wavefront
// shade pass does), so frames and generator streams are bit-identical to both other schedules (tests/test_gpu_parity.py).
// Exit: the work queue only grows; a lane that draws an index past the end never asks again; a block leaves when a shade phase appended no ray.
#include <hip/hip_runtime.h>
#include "hpt_decl.h"
#include "hpt_block_trace.h"

namespace hpt {

template <bool DR, bool LEAN, bool DEEP, bool FLAT, bool WIDE>
__global__ void __launch_bounds__(256, HPT_BW_WAVES(DR, LEAN)) pathTraceBlockKernel(const DevScene S, const Job job, uint refillBelow, uint nodeMin)
{
  __shared__ uint stackMem[LDS_STACK * 256];
  __shared__ uint pool[8 * BW_POOL];                   // ray entries, field-major: o.xyz, tfar | d.xyz, owner lane | kind << 31; reused for the ray's result
  __shared__ float coldPix[3 * 256];
  __shared__ uint  coldU[3 * 256];
  __shared__ uint  poolTail, poolHead;
  // (PathTraceDR: the cooperative gradient scatter stages its values in the wave's own columns of the traversal stack, which is idle while paths are shaded)
  const uint glane = blockIdx.x * 256u + threadIdx.x;
  const uint lane = threadIdx.x & 63u;
  TravStack stk; stk.lds = &stackMem[threadIdx.x]; stk.ovf = job.stackOverflow + glane; stk.ovfStride = job.gridLanes;
#define PIX(k)      coldPix[(k) * 256 + threadIdx.x]
#define PIX_XY      coldU[0 * 256 + threadIdx.x]
#define PIX_TID     coldU[1 * 256 + threadIdx.x]
#define PIX_PASSES  coldU[2 * 256 + threadIdx.x]
  bool havePixel = false, alive = false, drained = false, pend = false, ending = false;
  uint bounce = 0, flags = 0, myNear = 0, myShad = 0;
  Rng  gen; gen.sx = gen.sy = 0;
  V3   rpos = v3(0, 0, 0), rdir = v3(0, 0, 1), accum = v3(0, 0, 0), thr = v3(1, 1, 1), contrib = v3(0, 0, 0);
  float misPdf = 1.0f, misIor = 1.0f, lossLocal = 0.0f;
  const uint maxBounce = S.traceDepth;
  HitRec hit; hit.inst = 0xFFFFFFFFu; hit.prim = 0; hit.t = 0; hit.u = hit.v = 0;      // the result of this lane's closest-hit ray of the last round
  bool occluded = false;                                                                 // ... of its shadow ray
  if (threadIdx.x == 0u) { poolTail = 0u; poolHead = 0u; }
  __syncthreads();

  while (true) {
    // ================= S: results of the last round, shading, path ends, regeneration, new rays =================================
    bool wantShadow = false;
    V3 shPos = v3(0, 0, 0), shDir = v3(0, 0, 1); float shFar = 0.0f;
    bool closing = false; uint sweepBounce = 0; V3 sweepDiff = v3(0, 0, 0), sweepTail = v3(0, 0, 0);
    DrRec lastRec = drEmptyRecord(); bool lastInRegs = false;
    V3 tailR = v3(0, 0, 0);
    // (6') the shadow ray traced in the last round: add the candidate contribution in the megakernel's order
    if (pend) {
      if (!occluded) accum = accum + contrib;
      else if (DR && bounce > 0u) drClearShadowTerm(job.record, job.recordLanes, glane, bounce - 1u);
      pend = false;
    }
    bool finalize = ending;                              // the path ended in the last round, only its shadow ray was outstanding
    ending = false;
    if (alive) {
      V3 recA = v3(0, 0, 0), recS = v3(0, 0, 0), recdA = v3(0, 0, 0), recdS = v3(0, 0, 0); Taps recTaps; uint recTex = 0xFFFFFFFFu;
      for (int k = 0; k < 4; k++) { recTaps.off[k] = 0; recTaps.w[k] = 0.0f; }
      recTaps.fx = recTaps.fy = 0.0f; recTaps.base = recTaps.ch = 0u;
      const V3 thrBefore = thr;
      const bool didBounce = shadeVertex<DR, false, LEAN, false, false>(S, job.data, hit, rpos, rdir, accum, thr, misPdf, misIor, flags, bounce, gen,
                                                                        wantShadow, shPos, shDir, shFar, contrib, recA, recS, recdA, recdS, recTaps, recTex, tailR, 0.0f);
      if (didBounce) bounce++;
      const bool ended = (flags & RAY_FLAG_IS_DEAD) != 0 || bounce >= maxBounce;
      if (DR && didBounce) {
        if (!wantShadow) { recS = v3(0, 0, 0); recdS = v3(0, 0, 0); }     // (an occluded sample is cleared when its shadow ray comes back)
        lastRec = drMakeRecord(recA, recS, recdA, recdS, thrBefore, recTex, recTaps);
        lastInRegs = ended && !wantShadow;                                 // swept in this round: never leaves the registers
        if (!lastInRegs) drStoreRecord(job.record, job.recordLanes, glane, bounce - 1u, lastRec);
      }
      if (ended) {
        if (!DR && (flags & RAY_FLAG_OUT_OF_SCENE) != 0) {                 // kernel_HitEnvironment (integrator_pt.cpp:550-595)
          const V3 env = environmentRadiance(S, rdir, misPdf, flags, PIX_XY);
          if (S.integratorType == INTEGRATOR_STUPID_PT) accum = thr * env; else accum = accum + thr * env;
        }
        alive = false;
        if (wantShadow) ending = true; else finalize = true;
      }
    }
    if (finalize && DR) {
      // PixelLossPT (integrator_dr.cpp:1103-1132): environment term unconditionally (:1077-1098), per-sample loss, out_color += colorRend
      const V3 env = ld3(S.envColor);
      accum = accum + thr * env;
      const uint XY = PIX_XY;
      const uint yRef = (uint)S.winHeight - ((XY & 0xFFFF0000u) >> 16) - 1u;
      const float* rp = job.refImg + ((size_t)yRef * (uint)S.winWidth + (XY & 0x0000FFFFu)) * job.channels;
      const V3 diff = v3(accum.x - rp[0], accum.y - rp[1], accum.z - rp[2]);
      if (job.drSkipNonFinite == 0u || __builtin_isfinite(diff.x + diff.y + diff.z)) {
        lossLocal += (diff.x * diff.x + diff.y * diff.y + diff.z * diff.z) / float(job.passNum);
        PIX(0) += accum.x; PIX(1) += accum.y; PIX(2) += accum.z;
        closing = true; sweepBounce = bounce; sweepDiff = diff; sweepTail = tailR + env;
      }
    } else if (finalize) {                                                  // kernel_ContributeToImage (integrator_pt.cpp:598-657)
      const V3 c = accum * ld3(S.camRespoceRGB);
      if (job.channels == 1) PIX(0) += accum.x * S.exposureMult;
      else { PIX(0) += S.exposureMult * c.x; PIX(1) += S.exposureMult * c.y; PIX(2) += S.exposureMult * c.z; }
    }
    if (DR && __any(closing))
      drReverseSweep(S, job.record, job.recordLanes, glane, closing, sweepBounce, sweepTail, sweepDiff, job.grad, job.drSkipNonFinite != 0u,
                     stackMem + (threadIdx.x & ~63u), lastRec, lastInRegs, nullptr, 256u);
    const bool idle = !alive && !ending;                                    // no path in flight: next pass of the pixel, or next pixel
    // (1) a finished pixel goes back to HBM
    if (idle && havePixel && PIX_PASSES == 0u) {
      const uint XY = PIX_XY;
      const uint pixel = ((XY & 0xFFFF0000u) >> 16) * (uint)S.winWidth + (XY & 0x0000FFFFu);
      if (job.channels == 1) job.outColor[pixel] = PIX(0);
      else { float* o = job.outColor + (size_t)pixel * job.channels; o[0] = PIX(0); o[1] = PIX(1); o[2] = PIX(2); }
      job.gens[PIX_TID] = gen;
      havePixel = false;
    }
    // (2) work queue: ballot the lanes without a pixel, one atomic per wave
    {
      const bool need = idle && !havePixel && !drained;
      const unsigned long long mask = __ballot(need);
      if (mask != 0ull) {
        uint base = 0;
        if (need && mbcnt64(mask) == 0u) base = atomicAdd(job.queue, (uint)__popcll(mask));
        base = __shfl(base, (int)(__ffsll((long long)mask) - 1));
        if (need) {
          const uint k = base + mbcnt64(mask);
          const uint tid = job.tidBegin + (k / job.tidChunk) * job.tidChunk * job.tidStride + (k % job.tidChunk);
          if (k < job.tidCount && tid < job.tidEnd) {
            const uint XY = job.packedXY[tid];
            gen = job.gens[tid];
            const uint pixel = ((XY & 0xFFFF0000u) >> 16) * (uint)S.winWidth + (XY & 0x0000FFFFu);
            if (job.channels == 1) { PIX(0) = job.outColor[pixel]; PIX(1) = 0.0f; PIX(2) = 0.0f; }
            else { const float* o = job.outColor + (size_t)pixel * job.channels; PIX(0) = o[0]; PIX(1) = o[1]; PIX(2) = o[2]; }
            PIX_XY = XY; PIX_TID = tid; PIX_PASSES = job.passNum;
            havePixel = true;
          } else drained = true;
        }
      }
    }
    // (3) regenerate: next pass of the pixel (kernel_InitEyeRay2)
    if (idle && havePixel) {
      PIX_PASSES = PIX_PASSES - 1u;
      accum = v3(0, 0, 0); thr = v3(1, 1, 1); flags = 0; bounce = 0; misPdf = 1.0f; misIor = 1.0f;
      const V4 lens = rng_float4(gen);
      const uint XY = PIX_XY;
      cameraRay<!(DR || LEAN)>(S, XY & 0x0000FFFFu, (XY & 0xFFFF0000u) >> 16, lens, rpos, rdir);
      alive = true;
    }
    // (4) append this lane's rays to the block's pool: ballot + prefix sum, one LDS atomic per wave
    {
      const unsigned long long mn = __ballot(alive), ms = __ballot(wantShadow);
      const uint cn = (uint)__popcll(mn), cs = (uint)__popcll(ms);
      uint base = 0;
      if (lane == 0u && cn + cs != 0u) base = atomicAdd(&poolTail, cn + cs);
      base = __shfl(base, 0);
      if (alive) {
        const uint e = base + mbcnt64(mn);
        pool[0 * BW_POOL + e] = __float_as_uint(rpos.x); pool[1 * BW_POOL + e] = __float_as_uint(rpos.y); pool[2 * BW_POOL + e] = __float_as_uint(rpos.z);
        pool[3 * BW_POOL + e] = __float_as_uint(HPT_FLT_MAX);
        pool[4 * BW_POOL + e] = __float_as_uint(rdir.x); pool[5 * BW_POOL + e] = __float_as_uint(rdir.y); pool[6 * BW_POOL + e] = __float_as_uint(rdir.z);
        pool[7 * BW_POOL + e] = 0u;
        myNear = e;
      }
      if (wantShadow) {
        const uint e = base + cn + mbcnt64(ms);
        pool[0 * BW_POOL + e] = __float_as_uint(shPos.x); pool[1 * BW_POOL + e] = __float_as_uint(shPos.y); pool[2 * BW_POOL + e] = __float_as_uint(shPos.z);
        pool[3 * BW_POOL + e] = __float_as_uint(shFar);
        pool[4 * BW_POOL + e] = __float_as_uint(shDir.x); pool[5 * BW_POOL + e] = __float_as_uint(shDir.y); pool[6 * BW_POOL + e] = __float_as_uint(shDir.z);
        pool[7 * BW_POOL + e] = 1u;
        myShad = e; pend = true;
      }
    }
    __syncthreads();
    const uint total = poolTail;
    if (total == 0u) break;                                                  // no lane of the block holds a path or a pixel any more

    // ================= T: the block drains its pool, with ray replacement (hpt_block_trace.h) =========================================
    blockTracePhase<DEEP, FLAT, WIDE>(S, stk, pool, &poolHead, total, refillBelow, nodeMin, lane);
    __syncthreads();
    // every lane takes its rays' results out of the pool before any wave appends the next round's rays over them
    if (alive) {
      hit.t = __uint_as_float(pool[0 * BW_POOL + myNear]); hit.u = __uint_as_float(pool[1 * BW_POOL + myNear]); hit.v = __uint_as_float(pool[2 * BW_POOL + myNear]);
      hit.prim = pool[3 * BW_POOL + myNear]; hit.inst = pool[4 * BW_POOL + myNear]; hit.slot = pool[5 * BW_POOL + myNear];
    }
    if (pend) occluded = pool[0 * BW_POOL + myShad] != 0u;
    if (threadIdx.x == 0u) { poolTail = 0u; poolHead = 0u; }
    __syncthreads();
  }
#undef PIX
#undef PIX_XY
#undef PIX_TID
#undef PIX_PASSES
  if (DR) {
    float x = lossLocal;
    for (int o2 = 32; o2 > 0; o2 >>= 1) x += __shfl_down(x, o2);
    if ((threadIdx.x & 63) == 0) atomicAdd(job.lossAccum, x);
  }
}

// explicit instantiations (HPT_BW_INST: 1 forward, 2 PathTraceDR)
#ifndef HPT_BW_INST
#define HPT_BW_INST 0
#endif
#define HPT_BWI(DR, LEAN) \
  template __global__ void pathTraceBlockKernel<DR, LEAN, false, false, false>(const DevScene, const Job, uint, uint); \
  template __global__ void pathTraceBlockKernel<DR, LEAN, true,  false, false>(const DevScene, const Job, uint, uint); \
  template __global__ void pathTraceBlockKernel<DR, LEAN, false, true,  false>(const DevScene, const Job, uint, uint); \
  template __global__ void pathTraceBlockKernel<DR, LEAN, true,  true,  false>(const DevScene, const Job, uint, uint); \
  template __global__ void pathTraceBlockKernel<DR, LEAN, false, true,  true>(const DevScene, const Job, uint, uint);   /* 4-wide compressed tree */ \
  template __global__ void pathTraceBlockKernel<DR, LEAN, true,  true,  true>(const DevScene, const Job, uint, uint);
#if HPT_BW_INST == 0 || HPT_BW_INST == 1
HPT_BWI(false, true)
#endif
#if HPT_BW_INST == 0 || HPT_BW_INST == 2
HPT_BWI(true, true)
#endif
#if HPT_BW_INST == 0 || HPT_BW_INST == 3
  template __global__ void pathTraceBlockKernel<false, false, false, false, false>(const DevScene, const Job, uint, uint);
  template __global__ void pathTraceBlockKernel<false, false, true,  false, false>(const DevScene, const Job, uint, uint);
  template __global__ void pathTraceBlockKernel<false, false, false, true,  false>(const DevScene, const Job, uint, uint);
  template __global__ void pathTraceBlockKernel<false, false, true,  true,  false>(const DevScene, const Job, uint, uint);
#endif

} // namespace hpt
