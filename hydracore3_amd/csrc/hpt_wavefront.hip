// Wavefront schedule of the same path tracer, for scenes whose rays are expensive (deep BVHs, 10^5..10^7 triangles).
//
// Why a second schedule.  In the persistent megakernel (hpt_kernels.hip) a lane keeps its ray until the slowest lane of the wave
// has finished its own: measured on the 1M-triangle scene the node loop runs 237 wave iterations per ray while the mean lane needs
// 43 (profiles/phases.py: 18 % lane utilisation, 92 % of the wave cycles inside the two traversals).  Here the two halves of a
// bounce are separate kernels that meet in HBM:
//
//   wfShadeKernel   one lane per pool slot (= one pixel of the launch, which owns its RNG stream for all passes exactly as in the
//                   megakernel): folds the previous shadow-ray result into the path, shades the closest hit (hpt_shade.h: the
//                   SAME function the megakernel inlines), ends / regenerates paths, and appends the slot to the ray queues
//                   with a wave ballot + mbcnt prefix sum per wave and ONE atomicAdd per block (ray compaction);
//   wfTraceKernel   persistent waves pull rays from the compacted queue (a per-wave stash in LDS is topped up 64 rays at a time with
//                   one atomic; the queue is cut into 64 ranges with their own heads, the waves of an XCD start in neighbouring
//                   ranges). Traversal state is resumable: whenever fewer than `refillBelow` lanes of a wave still hold a ray, the
//                   wave leaves the loop, refills the idle lanes and continues, so the node and triangle loops run with (almost)
//                   full waves whatever the spread of per-ray work; when the queue has run dry, unfinished rays are suspended to
//                   HBM after a few more trips and resumed first by the next round's pass. 96 VGPRs, 5 waves per SIMD.
//
// Price: path state (148 bytes per slot) and rays travel through HBM/MALL once per bounce - negligible against a 40-node
// traversal, dominant against a 7-node one, which is why the Cornell-box class of scenes - and calls with too few pixels to keep
// the trace lanes supplied - stay on the megakernel (hpt_host.hip: useWavefront). The arithmetic per path is identical in both
// schedules (same functions, same order, IEEE flags), so the two produce bit-identical frames; tests/test_gpu_parity.py holds
// them to that. wfShadeKernel<DR = true> carries the differentiable integrator (adjoint records per slot, reverse sweep at path end).
#include <hip/hip_runtime.h>
#include "hpt_decl.h"

namespace hpt {

#if !defined(HPT_WF_INST) || HPT_WF_INST != 2      // (small kernels: emitted once, with the shade kernels)
__global__ void wfInitKernel(WfPool P, uint n, uint passNum)
{
  const uint i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { P.status[i] = passNum << 8; P.inflight[i] = 0u; if (P.lossSlot) P.lossSlot[i] = 0.0f; }       // (the counters are zeroed by the host: the grid may be smaller than the counter block)
}
#endif

template <bool DR, bool LEAN, bool MOTION, bool FILM>
__global__ void HPT_WFS_BOUNDS(DR, LEAN) wfShadeKernel(const DevScene S, const WfPool P, const WfJob job)
{
  __shared__ uint drStage[DR ? 4 * DR_STAGE_DWORDS : 1];                   // the waves' staging areas of the cooperative gradient scatter (drReverseSweep)
  const uint s = blockIdx.x * 256u + threadIdx.x;
  uint* ctr = P.ctr + WF_CTR_WORDS * (job.iter & 1u);
  if (s <= WF_RANGES) P.ctr[WF_CTR_WORDS * ((job.iter + 1u) & 1u) + 32u * s] = 0u;   // counters of the NEXT round (its trace pass is long done)

  bool valid = s < job.itemCount;
  uint tid = 0;
  if (valid) {
    const uint k = job.itemBase + s;
    tid = job.tidBegin + (k / job.tidChunk) * job.tidChunk * job.tidStride + (k % job.tidChunk);
    valid = tid < job.tidEnd;
  }
  if (valid && ldP(&P.inflight[s]) != 0u) valid = false;       // a suspended ray of this pixel is still being traced: the pixel sits this round out
  uint st = valid ? ldP(&P.status[s]) : 0u;
  uint passes = st >> 8;
  bool alive = (st & WF_ALIVE) != 0u, pend = (st & WF_PEND) != 0u, ending = (st & WF_ENDING) != 0u;
  const bool active = valid && (alive || pend || ending || passes != 0u);
  bool wantShadow = false;
  // DR: the reverse sweep runs after the divergent part, by the whole wave (drReverseSweep); what a closing lane hands over to it
  bool closing = false; uint sweepBounce = 0; V3 sweepDiff = v3(0, 0, 0), sweepTail = v3(0, 0, 0);
  DrRec lastRec = drEmptyRecord(); bool lastInRegs = false;

  if (active) {
    Rng gen = ldP(&job.gens[tid]);
    const uint XY = job.packedXY[tid];
    V3 accum = v3(0, 0, 0), thr = v3(1, 1, 1), rpos = v3(0, 0, 0), rdir = v3(0, 0, 1);
    float misPdf = 1.0f, misIor = 1.0f; uint flags = 0, bounce = 0;
    float pathTime = 0.0f;
    if (MOTION && alive) pathTime = ldP(&P.time[s]);
    if (alive || ending) { const float4 a = ldP(&P.acc[s]); accum = v3(a.x, a.y, a.z); bounce = __float_as_uint(a.w); }
    if (DR && ending) { const float4 t4 = ldP(&P.thr[s]); thr = v3(t4.x, t4.y, t4.z); }     // the environment term of the finished path needs its throughput
    // (6') the shadow ray traced since the last visit: add the candidate contribution in the megakernel's order
    if (pend) {
      if (ldP(&P.occl[s]) == 0u) { const float4 c = ldP(&P.contrib[s]); accum = accum + v3(c.x, c.y, c.z); }
      else if (DR && bounce > 0u) drClearShadowTerm(job.record, job.itemCount, s, bounce - 1u);   // the light sample of the last vertex was occluded
      pend = false;
    }
    bool finalize = ending;                                                // path ended last time, only its shadow ray was outstanding
    ending = false;
    V3 shPos = v3(0, 0, 0), shDir = v3(0, 0, 1), contrib = v3(0, 0, 0); float shFar = 0.0f;
    V3 tailR = v3(0, 0, 0);                                                 // DR: emission picked up at the terminating vertex, per unit throughput

    if (alive) {
      const float4 ro = ldP(&P.rayO[s]), rd = ldP(&P.rayD[s]), t4 = ldP(&P.thr[s]), h4 = ldP(&P.hit[s]);
      rpos = v3(ro.x, ro.y, ro.z); misPdf = ro.w; rdir = v3(rd.x, rd.y, rd.z); misIor = rd.w;
      thr = v3(t4.x, t4.y, t4.z); flags = __float_as_uint(t4.w);
      HitRec hit; hit.t = h4.x; hit.u = h4.y; hit.v = h4.z; hit.prim = __float_as_uint(h4.w); hit.inst = ldP(&P.hitInst[s]);
      if (S.shadeTris != nullptr) hit.slot = __float_as_uint(h4.w);      // (with the shading records in use the trace pass reports the hit's record, not its primitive id)
      V3 rA = v3(0, 0, 0), rS = v3(0, 0, 0), rdA = v3(0, 0, 0), rdS = v3(0, 0, 0); Taps taps; uint recTex = 0xFFFFFFFFu;   // adjoint record of this vertex (DR)
      for (int k = 0; k < 4; k++) { taps.off[k] = 0; taps.w[k] = 0.0f; }
      taps.fx = taps.fy = 0.0f; taps.base = taps.ch = 0u;
      const V3 thrBefore = thr;
      const bool didBounce = shadeVertex<DR, false, LEAN, MOTION, FILM>(S, DR ? job.data : nullptr, hit, rpos, rdir, accum, thr, misPdf, misIor, flags, bounce, gen,
                                                    wantShadow, shPos, shDir, shFar, contrib, rA, rS, rdA, rdS, taps, recTex, tailR, pathTime);
      if (didBounce) bounce++;
      const bool ended = (flags & RAY_FLAG_IS_DEAD) != 0 || bounce >= S.traceDepth;
      if (DR && didBounce) {
        if (!wantShadow) { rS = v3(0, 0, 0); rdS = v3(0, 0, 0); }           // (an occluded sample is cleared when its shadow ray comes back)
        lastRec = drMakeRecord(rA, rS, rdA, rdS, thrBefore, recTex, taps);
        // a path that ends here with no shadow ray outstanding is swept in this pass: its last record never leaves the registers
        lastInRegs = ended && !wantShadow;
        if (!lastInRegs) drStoreRecord(job.record, job.itemCount, s, bounce - 1u, lastRec);
      }
      if (ended) {
        if (!DR && (flags & RAY_FLAG_OUT_OF_SCENE) != 0) {                  // kernel_HitEnvironment (integrator_pt.cpp:550-595)
          const V3 env = environmentRadiance(S, rdir, misPdf, flags, XY);
          if (S.integratorType == INTEGRATOR_STUPID_PT) accum = thr * env; else accum = accum + thr * env;
        }
        alive = false;
        if (wantShadow) ending = true; else finalize = true;
      }
    }
    if (finalize && DR) {
      // PixelLossPT (integrator_dr.cpp:1103-1132): the replay adds the environment term unconditionally (:1077-1098), after the last
      // shadow contribution as in the megakernel; per-sample loss against the y-flipped reference; out_color += colorRend; reverse sweep
      const V3 env = ld3(S.envColor);
      accum = accum + thr * env;
      const uint x = XY & 0x0000FFFFu, y = (XY & 0xFFFF0000u) >> 16;
      const float* rp = job.refImg + ((size_t)((uint)S.winHeight - y - 1u) * (uint)S.winWidth + x) * job.channels;
      const V3 diff = v3(accum.x - rp[0], accum.y - rp[1], accum.z - rp[2]);
      if (job.drSkipNonFinite == 0u || __builtin_isfinite(diff.x + diff.y + diff.z)) {   // (non-finite samples: see the megakernel)
        P.lossSlot[s] += (diff.x * diff.x + diff.y * diff.y + diff.z * diff.z) / float(job.passNum);
        float* o = job.outColor + ((size_t)y * (uint)S.winWidth + x) * job.channels;
        o[0] += accum.x; o[1] += accum.y; o[2] += accum.z;
        closing = true; sweepBounce = bounce; sweepDiff = diff; sweepTail = tailR + env;
      }
    } else if (finalize) {                                                   // kernel_ContributeToImage (integrator_pt.cpp:598-657)
      const uint pixel = ((XY & 0xFFFF0000u) >> 16) * (uint)S.winWidth + (XY & 0x0000FFFFu);
      const V3 c = accum * ld3(S.camRespoceRGB);
      if (job.channels == 1) job.outColor[pixel] += accum.x * S.exposureMult;
      else { float* o = job.outColor + (size_t)pixel * job.channels; o[0] += S.exposureMult * c.x; o[1] += S.exposureMult * c.y; o[2] += S.exposureMult * c.z; }
    }
    if (!alive && !ending && passes != 0u) {                                 // kernel_InitEyeRay2: next pass of this pixel
      passes--;
      accum = v3(0, 0, 0); thr = v3(1, 1, 1); flags = 0; bounce = 0; misPdf = 1.0f; misIor = 1.0f;
      const V4 lens = rng_float4(gen);
      cameraRay<!(DR || LEAN)>(S, XY & 0x0000FFFFu, (XY & 0xFFFF0000u) >> 16, lens, rpos, rdir);
      if (MOTION) { pathTime = rng_float1(gen); stP(&P.time[s], pathTime); }    // GetRandomNumbersTime (integrator_pt.cpp:114-115): one step per path, after the lens
      alive = true;
    }
    stP(&job.gens[tid], gen);
    if (alive) {
      stP(&P.rayO[s], make_float4(rpos.x, rpos.y, rpos.z, misPdf));
      stP(&P.rayD[s], make_float4(rdir.x, rdir.y, rdir.z, misIor));
    }
    if (alive || (DR && ending)) stP(&P.thr[s], make_float4(thr.x, thr.y, thr.z, __uint_as_float(flags)));
    if (alive || ending) stP(&P.acc[s], make_float4(accum.x, accum.y, accum.z, __uint_as_float(bounce)));
    if (wantShadow) {
      stP(&P.shO[s], make_float4(shPos.x, shPos.y, shPos.z, shFar));
      stP(&P.shD[s], make_float4(shDir.x, shDir.y, shDir.z, 0.0f));
      stP(&P.contrib[s], make_float4(contrib.x, contrib.y, contrib.z, 0.0f));
    }
    stP(&P.status[s], (passes << 8) | (alive ? WF_ALIVE : 0u) | (wantShadow ? WF_PEND : 0u) | (ending ? WF_ENDING : 0u));
  }
#ifndef HPT_DBG_DR_NOSWEEP   // diagnostic builds only (profiles/dr_ab.sh)
  if (DR && __any(closing))
    drReverseSweep(S, job.record, job.itemCount, s < job.itemCount ? s : 0u, closing, sweepBounce, sweepTail, sweepDiff, job.grad, job.drSkipNonFinite != 0u,
                   drStage + (threadIdx.x >> 6) * DR_STAGE_DWORDS, lastRec, lastInRegs, nullptr, 64u, false);
#endif
  // ray compaction: ballot + prefix sum, one atomic per wave and queue
  const bool qNear = active && alive, qShad = active && wantShadow;
  uint kn, ks;
  blockAppend(&ctr[0], qNear, qShad, kn, ks);
  uint* rayQ = P.rayQ[job.iter & 1u];
  if (qNear) stP(&rayQ[kn], s);
  if (qShad) stP(&rayQ[ks], s | 0x80000000u);
}

#if !defined(HPT_WF_INST) || HPT_WF_INST != 2
// DR: sum of the per-slot losses in double, one atomic per block; wfLossFinishKernel adds the total to the caller's float
__global__ void __launch_bounds__(256) wfLossReduceKernel(const float* lossSlot, uint n, double* acc)
{
  __shared__ double part[4];
  double x = 0.0;
  for (uint i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) x += (double)lossSlot[i];
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
  if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = x;
  __syncthreads();
  if (threadIdx.x == 0u) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}
__global__ void wfLossFinishKernel(const double* acc, float* loss) { *loss += (float)*acc; }
#endif

// ---- persistent traversal with ray replacement -------------------------------------------------------------------------------
#ifndef HPT_WF_XCD_RANGES
#define HPT_WF_XCD_RANGES 1
#endif
template <bool DEEP, bool FLAT, bool STATS, bool MOTION, bool WIDE>
__global__ void __launch_bounds__(256, HPT_WF_WAVES) wfTraceKernel(const DevScene S, const WfPool P, uint iter, uint refillBelow, uint grace,
                                                                   uint* stackOverflow, uint gridLanes, Counters* counters)
{
  __shared__ uint stackMem[LDS_STACK * 256];
  const uint glane = blockIdx.x * 256u + threadIdx.x;
  TravStack stk; stk.lds = &stackMem[threadIdx.x]; stk.ovf = stackOverflow + glane; stk.ovfStride = gridLanes;
  uint* ctr = P.ctr + WF_CTR_WORDS * (iter & 1u);
  uint* ctrNext = P.ctr + WF_CTR_WORDS * ((iter + 1u) & 1u);
  const uint* rayQ = P.rayQ[iter & 1u];
  uint* rayQNext = P.rayQ[(iter + 1u) & 1u];
  const uint* suspIn = P.susp[iter & 1u];
  uint* suspOut = P.susp[(iter + 1u) & 1u];
  const size_t SM = P.maxSusp;
  const uint total = ctr[0];
  if (total < 2u * gridLanes) grace = 0u;     // a round without surplus rays is all tail: suspending would only multiply the rounds
  if (S.rootRef == REF_NONE) {                                               // empty scene: every ray misses
    for (uint k = glane; k < total; k += gridLanes) {
      const uint q = rayQ[k];
      if ((q >> 31) == 0u) P.hitInst[q] = 0xFFFFFFFFu; else P.occl[q & 0x7FFFFFFFu] = 0u;
    }
    return;
  }

  constexpr uint STASH = MOTION ? 9u : 8u;                                 // dwords per stashed ray (motion blur: + the ray's time)
  __shared__ uint stashMem[4 * STASH * 64];
  uint* stash = stashMem + (threadIdx.x >> 6) * (STASH * 64);
  float rayTime = 0.0f;
  const uint lane = threadIdx.x & 63u;
  bool has = false, isAny = false, found = false, resumed = false;
  uint dryTrips = 0;
  // XCD-aware start range: blocks are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each with its own L2. The waves of
  // one XCD start in ITS eight neighbouring ranges - contiguous, pixel-coherent stretches of the queue - and only steal elsewhere later.
#if HPT_WF_XCD_RANGES
  uint range = (blockIdx.x % 8u) * (WF_RANGES / 8u) + ((blockIdx.x / 8u) * 4u + (threadIdx.x >> 6)) % (WF_RANGES / 8u), tried = 0, stashCount = 0;   // wave-uniform
#else
  uint range = (glane >> 6) % WF_RANGES, tried = 0, stashCount = 0;          // wave-uniform
#endif
  uint slot = 0, cur = REF_NONE, curInst = 0xFFFFFFFFu;
  int  sp = 0;
  V3 wo = v3(0, 0, 0), wd = v3(0, 0, 1), o = wo, d = wd, id = v3(0, 0, 0), oid = v3(0, 0, 0);
  float hitT = 0.0f, hitU = 0.0f, hitV = 0.0f; uint hitPrim = 0, hitInst = 0xFFFFFFFFu, hitSlot = 0xFFFFFFFFu;
  unsigned long long nodeLane = 0, nodeWave = 0, triLane = 0, triWave = 0, refills = 0, suspended = 0;
  unsigned long long tPh[4] = {0, 0, 0, 0}, tPrev = 0, trips = 0;            // STATS: wave cycles in refill / node loop / leaves / ray end
#define WSTAMP(i) do { if (STATS) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tPh[i] += tn - tPrev; tPrev = tn; } } while (0)
  unsigned long long tBegin = 0, tEmpty = 0;
  if (STATS) { tPrev = __builtin_amdgcn_s_memtime(); tBegin = tPrev; }

#define HPT_PUSH(v) do { if (DEEP) stkPush(stk, sp, (v)); else stk.lds[sp * 256] = (v); sp++; } while (0)
#define HPT_POP()   do { sp--; cur = DEEP ? stkPop(stk, sp) : stk.lds[sp * 256]; } while (0)

  while (true) {
    // ---- refill ---------------------------------------------------------------------------------------------------------
    // Idle lanes take rays from the wave's stash in LDS (8 dwords per ray, field-major: conflict-free). When the stash cannot
    // serve them all it is topped up to 64 rays first: ONE atomicAdd per top-up on the head of the current queue range, then
    // all 64 lanes fetch one ray each - the queue's memory latency is paid once per ~64 rays, not once per refill.
    // The queue is cut into WF_RANGES contiguous ranges with a head counter each (different L2 channels). A wave starts at
    // "its" range and moves on when that one is exhausted (an L2 load tells, before any atomic is spent on it).
    {
      const unsigned long long mask = __ballot(!has);
      const uint n = (uint)__popcll(mask);
      if (n != 0u) {
        if (stashCount < n && tried < WF_RANGES) {
          if (STATS && firstActiveLane()) refills++;
          while (stashCount < 64u && tried < WF_RANGES) {
            const uint rBegin = (uint)(((unsigned long long)total * range) / WF_RANGES);
            const uint rSize = (uint)(((unsigned long long)total * (range + 1u)) / WF_RANGES) - rBegin;
            uint* head = ctr + 32u * (1u + range);
            uint granted = 0;
            const uint want = 64u - stashCount;
            if (__hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < rSize) {   // read at L2: never a stale L1 copy
              uint base = 0;
              if (lane == 0u) base = atomicAdd(head, want);
              base = __shfl(base, 0);
              granted = base < rSize ? min(want, rSize - base) : 0u;
              if (lane < granted) {
                const uint k = rBegin + base + lane;
                const uint q = ldP(&rayQ[k]);
                const uint sl = q & 0x3FFFFFFFu;
                float4 a, b;
                if ((q >> 31) == 0u) { a = ldP(&P.rayO[sl]); b = ldP(&P.rayD[sl]); a.w = HPT_FLT_MAX; }
                else                 { a = ldP(&P.shO[sl]); b = ldP(&P.shD[sl]); }
                if ((q & 0x40000000u) != 0u) a.w = __uint_as_float(k);        // resumed ray: its saved state is record k
                uint* e = stash + (stashCount + lane);
                e[0 * 64] = __float_as_uint(a.x); e[1 * 64] = __float_as_uint(a.y); e[2 * 64] = __float_as_uint(a.z); e[3 * 64] = __float_as_uint(a.w);
                e[4 * 64] = __float_as_uint(b.x); e[5 * 64] = __float_as_uint(b.y); e[6 * 64] = __float_as_uint(b.z); e[7 * 64] = q;
                if (MOTION) e[8 * 64] = __float_as_uint(ldP(&P.time[sl]));       // both rays of a path carry the path's time
              }
              stashCount += granted;
            }
            if (granted < want) { range = (range + 1u) % WF_RANGES; tried++; }   // this range has nothing left
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
        const uint give = min(n, stashCount);
        if (!has && mbcnt64(mask) < give) {
          const uint* e = stash + (stashCount - 1u - mbcnt64(mask));
          wo = v3(__uint_as_float(e[0 * 64]), __uint_as_float(e[1 * 64]), __uint_as_float(e[2 * 64])); hitT = __uint_as_float(e[3 * 64]);
          wd = v3(__uint_as_float(e[4 * 64]), __uint_as_float(e[5 * 64]), __uint_as_float(e[6 * 64]));
          const uint q = e[7 * 64];
          if (MOTION) rayTime = __uint_as_float(e[8 * 64]);
          slot = q & 0x3FFFFFFFu; isAny = (q >> 31) != 0u; resumed = (q & 0x40000000u) != 0u;
          o = wo; d = wd; slabRay(wo, wd, id, oid);
          cur = WIDE ? S.root4 : S.rootRef; curInst = 0xFFFFFFFFu; sp = 0; found = false;
          hitPrim = 0xFFFFFFFFu; hitInst = 0xFFFFFFFFu; hitSlot = 0xFFFFFFFFu; hitU = 0.0f; hitV = 0.0f;
          if (resumed) {                                                    // pick the traversal up where the last pass left it
            const uint* r = suspIn + __float_as_uint(hitT);
            cur = r[0 * SM]; sp = (int)r[1 * SM]; curInst = r[2 * SM];
            hitT = __uint_as_float(r[3 * SM]); hitU = __uint_as_float(r[4 * SM]); hitV = __uint_as_float(r[5 * SM]);
            hitPrim = r[6 * SM]; hitInst = r[7 * SM]; found = r[8 * SM] != 0u; hitSlot = r[9 * SM];
            for (int k = 0; k < sp; k++) { const uint v = r[(WF_SUSP_WORDS + k) * SM]; if (DEEP) stkPush(stk, k, v); else stk.lds[k * 256] = v; }
            if (FLAT) curInst = 0xFFFFFFFFu;                                 // the object-space ray is rebuilt at the next triangle
            else if (curInst != 0xFFFFFFFFu) {
              if (MOTION && S.insts[curInst].pad0 != 0u) toObjectSpaceMotion(S.instMotion + 24u * curInst, rayTime, wo, wd, o, d);
              else toObjectSpace(S.insts, curInst, wo, wd, o, d);
              slabRay(o, d, id, oid);
            }
          }
          has = true;
        }
        stashCount -= give;
      }
    }
    if (!__any(has)) break;
    WSTAMP(0);
    const bool queueEmpty = (stashCount == 0u) && (tried >= WF_RANGES);      // nothing left to refill with: run the rays to the end
    if (STATS && queueEmpty && tEmpty == 0) tEmpty = __builtin_amdgcn_s_memtime();

    // ---- traverse until this lane's ray is done, or the wave has thinned out and the queue can refill it ------------------
    if (has) {
      while (true) {
        if (STATS) trips++;
        while ((cur & REF_LEAF) == 0u) {
          if (WIDE) {
            if (STATS) { nodeLane++; if (firstActiveLane()) nodeWave++; }
            wideNodeStep<DEEP>(S, stk, oid, id, hitT, cur, sp);
            if (S.nodeMin4 != 0u && (uint)__popcll(__ballot((cur & REF_LEAF) == 0u)) < S.nodeMin4) break;
            continue;
          }
          const float4* np = (const float4*)(S.nodes + cur);
          const float4 q0 = np[0], q1 = np[1], q2 = np[2];
          const uint4  q3 = ((const uint4*)np)[3];
          if (STATS) { nodeLane++; if (firstActiveLane()) nodeWave++; }
          bool h0, h1; float t0n, t1n;
          nodeSlabs(q0, q1, q2, oid, id, 0.0f, hitT, h0, h1, t0n, t1n);     // (flat layout: boxes in world space, id / oid stay the world ray's)
          if (h0 && h1) {
            const bool firstIs0 = t0n <= t1n;
            HPT_PUSH(firstIs0 ? q3.y : q3.x);
            cur = firstIs0 ? q3.x : q3.y;
          } else if (h0) cur = q3.x;
          else if (h1) cur = q3.y;
          else if (sp > 0) HPT_POP();
          else cur = REF_NONE;
          // voted exit: when only a few lanes are still walking inner nodes, the lanes that already hold a leaf are served first
          if (S.nodeMin != 0u && (uint)__popcll(__ballot((cur & REF_LEAF) == 0u)) < S.nodeMin) break;
        }
        WSTAMP(1);
        const uint leaf = cur;
        bool done = (leaf == REF_NONE);
        if (!done && (leaf & REF_LEAF) != 0u) {
          const uint cnt = (leaf >> 28) & 7u;
          if (FLAT || (cnt >= 1u && cnt <= 4u)) {
            const uint first = leaf & 0x0FFFFFFFu;
            for (uint k = 0; k < cnt; k++) {
              const float4* tp = (const float4*)(S.tris + first + k);
              const float4 a = tp[0], b = tp[1], c = tp[2];
              if (STATS) { triLane++; if (firstActiveLane()) triWave++; }
              uint inst = curInst;
              if (FLAT) {
                inst = __float_as_uint(b.w);
                if (inst != curInst) {                                          // world -> object space of this triangle's instance (at the ray's time, if it moves)
                  if (MOTION && S.insts[inst].pad0 != 0u) toObjectSpaceMotion(S.instMotion + 24u * inst, rayTime, wo, wd, o, d);
                  else toObjectSpace(S.insts, inst, wo, wd, o, d);
                  curInst = inst;
                }
              }
              if (triangleTest(a, b, c, o, d, 0.0f, inst, hitT, hitPrim, hitInst, hitU, hitV, found)) hitSlot = first + k;
            }
            if (isAny && found) done = true;
            else if (sp > 0) HPT_POP(); else done = true;
          } else if (cnt == 0u) {
            const uint inst = cur & 0x0FFFFFFFu;
            if (MOTION && S.insts[inst].pad0 != 0u) toObjectSpaceMotion(S.instMotion + 24u * inst, rayTime, wo, wd, o, d);
            else toObjectSpace(S.insts, inst, wo, wd, o, d);
            slabRay(o, d, id, oid);
            curInst = inst;
            HPT_PUSH(REF_RESTORE);
            cur = S.insts[inst].root;
          } else {
            o = wo; d = wd; slabRay(o, d, id, oid); curInst = 0xFFFFFFFFu;
            if (sp > 0) HPT_POP(); else done = true;
          }
        }
        WSTAMP(2);
        if (done) {
          if (isAny) stP(&P.occl[slot], found ? 1u : 0u);
          else { stP(&P.hit[slot], make_float4(hitT, hitU, hitV, __uint_as_float((FLAT && S.shadeTris != nullptr) ? hitSlot : hitPrim))); stP(&P.hitInst[slot], found ? hitInst : 0xFFFFFFFFu); }
          if (resumed) atomicAnd(&P.inflight[slot], isAny ? ~2u : ~1u);
          has = false;
          break;
        }
        if (!queueEmpty && (uint)__popcll(__ballot(true)) < refillBelow) break;
        if (queueEmpty && grace != 0u && ++dryTrips >= grace) break;          // nothing to refill with: a few more trips, then suspend
      }
    }
    WSTAMP(3);
    // ---- bounded tail: the queue is dry and the grace trips are used up - park the unfinished rays for the next round's pass ----
    if (queueEmpty && grace != 0u && __any(has && dryTrips >= grace)) {
      const unsigned long long m = __ballot(has);
      uint base = 0;
      const int leader = (int)(__ffsll((long long)m) - 1);
      if ((int)lane == leader) base = atomicAdd(&ctrNext[0], (uint)__popcll(m));
      base = __shfl(base, leader);
      if (has) {
        const uint rec = base + mbcnt64(m);                                   // < maxSusp: a lane suspends at most one ray per pass
        rayQNext[rec] = slot | (isAny ? 0x80000000u : 0u) | 0x40000000u;
        uint* r = suspOut + rec;
        r[0 * SM] = cur; r[1 * SM] = (uint)sp; r[2 * SM] = curInst;
        r[3 * SM] = __float_as_uint(hitT); r[4 * SM] = __float_as_uint(hitU); r[5 * SM] = __float_as_uint(hitV);
        r[6 * SM] = hitPrim; r[7 * SM] = hitInst; r[8 * SM] = found ? 1u : 0u; r[9 * SM] = hitSlot;
        for (int k = 0; k < sp; k++) r[(WF_SUSP_WORDS + k) * SM] = DEEP ? stkPop(stk, k) : stk.lds[k * 256];
        atomicOr(&P.inflight[slot], isAny ? 2u : 1u);
        has = false;
        if (STATS) suspended++;
      }
      break;
    }
  }
#undef WSTAMP
#undef HPT_PUSH
#undef HPT_POP
  if (STATS) {
    unsigned long long v[5] = { nodeLane, nodeWave, triLane, triWave, refills };
    { unsigned long long x = suspended; for (int o2 = 32; o2 > 0; o2 >>= 1) x += __shfl_down(x, o2); if ((threadIdx.x & 63) == 0 && x) atomicAdd(&counters->v[12], x); }
    for (int i = 0; i < 5; i++) {
      unsigned long long x = v[i];
      for (int o2 = 32; o2 > 0; o2 >>= 1) x += __shfl_down(x, o2);
      if ((threadIdx.x & 63) == 0 && x) atomicAdd(&counters->v[i], x);
    }
    if ((threadIdx.x & 63) == 0) {
      for (int i = 0; i < 4; i++) atomicAdd(&counters->v[5 + i], tPh[i]);
      unsigned long long tr = trips;                                         // trips of lane 0 of the wave ~ wave-level trips of the leaf loop
      atomicAdd(&counters->v[9], tr);
      const unsigned long long tEnd = __builtin_amdgcn_s_memtime();           // wave time with nothing left to refill from (tail) / total
      atomicAdd(&counters->v[10], tEmpty ? tEnd - tEmpty : 0ull); atomicAdd(&counters->v[11], tEnd - tBegin);
    }
  }
}

// ---- explicit instantiations (the host launches them through the declarations in hpt_decl.h) -----------------------------------------------
#ifndef HPT_WF_INST
#define HPT_WF_INST 0          // 0: everything in one translation unit; 1: shade kernels only; 2: trace kernels only
#endif
#if HPT_WF_INST == 0 || HPT_WF_INST == 1
template __global__ void wfShadeKernel<true, true>(const DevScene, const WfPool, const WfJob);
template __global__ void wfShadeKernel<false, true>(const DevScene, const WfPool, const WfJob);
template __global__ void wfShadeKernel<false, false>(const DevScene, const WfPool, const WfJob);
template __global__ void wfShadeKernel<false, false, true>(const DevScene, const WfPool, const WfJob);     // moving instances (every BSDF branch)
template __global__ void wfShadeKernel<false, false, false, true>(const DevScene, const WfPool, const WfJob);   // thin films
template __global__ void wfShadeKernel<false, false, true, true>(const DevScene, const WfPool, const WfJob);    // thin films and moving instances
#endif
#if HPT_WF_INST == 0 || HPT_WF_INST == 2
#define HPT_WFT(DEEP, FLAT, STATS) template __global__ void wfTraceKernel<DEEP, FLAT, STATS>(const DevScene, const WfPool, uint, uint, uint, uint*, uint, Counters*);
HPT_WFT(false, false, false) HPT_WFT(true, false, false) HPT_WFT(false, true, false) HPT_WFT(true, true, false)
HPT_WFT(false, false, true)  HPT_WFT(true, false, true)  HPT_WFT(false, true, true)  HPT_WFT(true, true, true)
#define HPT_WFTM(DEEP, FLAT) template __global__ void wfTraceKernel<DEEP, FLAT, false, true>(const DevScene, const WfPool, uint, uint, uint, uint*, uint, Counters*);
HPT_WFTM(false, false) HPT_WFTM(true, false) HPT_WFTM(false, true) HPT_WFTM(true, true)
template __global__ void wfTraceKernel<false, true, false, false, true>(const DevScene, const WfPool, uint, uint, uint, uint*, uint, Counters*);   // 4-wide compressed tree
template __global__ void wfTraceKernel<true,  true, false, false, true>(const DevScene, const WfPool, uint, uint, uint, uint*, uint, Counters*);
template __global__ void wfTraceKernel<false, true, true, false, true>(const DevScene, const WfPool, uint, uint, uint, uint*, uint, Counters*);   // instrumented (hpt_set_option "wf_stats")
template __global__ void wfTraceKernel<true,  true, true, false, true>(const DevScene, const WfPool, uint, uint, uint, uint*, uint, Counters*);
#endif

} // namespace hpt
