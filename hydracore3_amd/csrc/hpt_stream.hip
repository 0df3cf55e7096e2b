// Block-owned streaming form of the wavefront schedule (hpt_set_schedule(ctx, 4, ...); heavy static scenes with gltf / emissive materials).
//
// The wavefront schedule (hpt_wavefront.hip) separates the two halves of a bounce by KERNEL boundaries: every round all slots are shaded, then all
// rays are traced, and each trace pass ends in a tail (0.31 of that kernel's wave time comes after its queue ran dry, profiles/wf_stats.py) while the
// shade pass streams the pool through HBM with the chip's latency-bound lanes idle. The block-local schedule (hpt_block.hip) has neither boundary
// nor pool traffic, but its pool is the 512 rays of one workgroup. Here a slot's whole life stays inside ONE workgroup: every block owns a fixed,
// tile-interleaved set of slots (a few thousand pixels) for the whole call, with their state in HBM as in the wavefront pool (touched by this CU
// only: the vector L1 is shared by the workgroup, so a workgroup barrier orders the two phases - no agent-scope release / acquire, no sc1 traffic),
// and alternates between
//   shade   the lanes walk the block's slots, 256 at a time: the body of wfShadeKernel<false, LEAN> for one slot, rays appended to the block's queue;
//   trace   the block's four waves drain that queue with ray replacement: the loop of wfTraceKernel on the 4-wide tree, fed through an LDS head word;
// until a round queues no ray. Nothing separates the rounds of different blocks: while one block of a CU shades, the others trace, and a block's
// tail overlaps its neighbours' bulk. One launch per call, no host loop. Same functions, same order per path as the other schedules: bit-identical frames
// (tests/test_gpu_parity.py: test_streaming_schedule_equals_megakernel).
// Measured (1 M-triangle interior, 1920 x 1080): 305 ... 318 Mpaths/s against the wavefront schedule's 325 - the tail of a round is set by the rays per lane and
// round, which one slot per pixel fixes for either form, and the fused kernel spills 40 VGPRs at five waves per SIMD (four waves, no spills: 302). Opt-in
// (never the automatic choice): for callers that want one launch per call, and the base of a form without rounds.
#include <hip/hip_runtime.h>
#include "hpt_decl.h"

namespace hpt {

// One slot's shade step: the body of wfShadeKernel<false, true>. (As a real call - noinline, its own register allocation, 11 instead of 40 spilled VGPRs in the
// kernel - the interior drops from 305 to 212 Mpaths/s: the call's save / restore traffic costs more than the spills.)
HPT_DEV void streamShadeSlot(const DevScene& S, const WfPool& P, const WfJob& job, const uint s, bool& qNear, bool& qShad)
{
  bool valid = s < job.itemCount;
  uint tid = 0;
  if (valid) {
    const uint k = job.itemBase + s;
    tid = job.tidBegin + (k / job.tidChunk) * job.tidChunk * job.tidStride + (k % job.tidChunk);
    valid = tid < job.tidEnd;
  }
  uint st = valid ? ldP(&P.status[s]) : 0u;
  uint passes = st >> 8;
  bool alive = (st & WF_ALIVE) != 0u, pend = (st & WF_PEND) != 0u, ending = (st & WF_ENDING) != 0u;
  const bool active = valid && (alive || pend || ending || passes != 0u);
  bool wantShadow = false;
  if (active) {
    Rng gen = ldP(&job.gens[tid]);
    const uint XY = job.packedXY[tid];
    V3 accum = v3(0, 0, 0), thr = v3(1, 1, 1), rpos = v3(0, 0, 0), rdir = v3(0, 0, 1);
    float misPdf = 1.0f, misIor = 1.0f; uint flags = 0, bounce = 0;
    if (alive || ending) { const float4 a = ldP(&P.acc[s]); accum = v3(a.x, a.y, a.z); bounce = __float_as_uint(a.w); }
    if (pend) {                                                            // the shadow ray traced since the last visit, in the megakernel's order
      if (ldP(&P.occl[s]) == 0u) { const float4 c = ldP(&P.contrib[s]); accum = accum + v3(c.x, c.y, c.z); }
      pend = false;
    }
    bool finalize = ending;
    ending = false;
    V3 shPos = v3(0, 0, 0), shDir = v3(0, 0, 1), contrib = v3(0, 0, 0); float shFar = 0.0f;
    V3 tailR = v3(0, 0, 0);
    if (alive) {
      const float4 ro = ldP(&P.rayO[s]), rd = ldP(&P.rayD[s]), t4 = ldP(&P.thr[s]), h4 = ldP(&P.hit[s]);
      rpos = v3(ro.x, ro.y, ro.z); misPdf = ro.w; rdir = v3(rd.x, rd.y, rd.z); misIor = rd.w;
      thr = v3(t4.x, t4.y, t4.z); flags = __float_as_uint(t4.w);
      HitRec hit; hit.t = h4.x; hit.u = h4.y; hit.v = h4.z; hit.prim = __float_as_uint(h4.w); hit.inst = ldP(&P.hitInst[s]);
      if (S.shadeTris != nullptr) hit.slot = __float_as_uint(h4.w);
      V3 rA = v3(0, 0, 0), rS = v3(0, 0, 0), rdA = v3(0, 0, 0), rdS = v3(0, 0, 0); Taps taps; uint recTex = 0xFFFFFFFFu;
      for (int k = 0; k < 4; k++) { taps.off[k] = 0; taps.w[k] = 0.0f; }
      taps.fx = taps.fy = 0.0f; taps.base = taps.ch = 0u;
      const bool didBounce = shadeVertex<false, false, true, false, false>(S, nullptr, hit, rpos, rdir, accum, thr, misPdf, misIor, flags, bounce, gen,
                                                                         wantShadow, shPos, shDir, shFar, contrib, rA, rS, rdA, rdS, taps, recTex, tailR, 0.0f);
      if (didBounce) bounce++;
      const bool ended = (flags & RAY_FLAG_IS_DEAD) != 0 || bounce >= S.traceDepth;
      if (ended) {
        if ((flags & RAY_FLAG_OUT_OF_SCENE) != 0) {                         // kernel_HitEnvironment (integrator_pt.cpp:550-595)
          const V3 env = environmentRadiance(S, rdir, misPdf, flags, XY);
          if (S.integratorType == INTEGRATOR_STUPID_PT) accum = thr * env; else accum = accum + thr * env;
        }
        alive = false;
        if (wantShadow) ending = true; else finalize = true;
      }
    }
    if (finalize) {                                                          // kernel_ContributeToImage (integrator_pt.cpp:598-657)
      const uint pixel = ((XY & 0xFFFF0000u) >> 16) * (uint)S.winWidth + (XY & 0x0000FFFFu);
      const V3 c = accum * ld3(S.camRespoceRGB);
      if (job.channels == 1) job.outColor[pixel] += accum.x * S.exposureMult;
      else { float* o = job.outColor + (size_t)pixel * job.channels; o[0] += S.exposureMult * c.x; o[1] += S.exposureMult * c.y; o[2] += S.exposureMult * c.z; }
    }
    if (!alive && !ending && passes != 0u) {                                 // kernel_InitEyeRay2: next pass of this pixel
      passes--;
      accum = v3(0, 0, 0); thr = v3(1, 1, 1); flags = 0; bounce = 0; misPdf = 1.0f; misIor = 1.0f;
      const V4 lens = rng_float4(gen);
      cameraRay<false>(S, XY & 0x0000FFFFu, (XY & 0xFFFF0000u) >> 16, lens, rpos, rdir);
      alive = true;
    }
    stP(&job.gens[tid], gen);
    if (alive) {
      stP(&P.rayO[s], make_float4(rpos.x, rpos.y, rpos.z, misPdf));
      stP(&P.rayD[s], make_float4(rdir.x, rdir.y, rdir.z, misIor));
      stP(&P.thr[s], make_float4(thr.x, thr.y, thr.z, __uint_as_float(flags)));
    }
    if (alive || ending) stP(&P.acc[s], make_float4(accum.x, accum.y, accum.z, __uint_as_float(bounce)));
    if (wantShadow) {
      stP(&P.shO[s], make_float4(shPos.x, shPos.y, shPos.z, shFar));
      stP(&P.shD[s], make_float4(shDir.x, shDir.y, shDir.z, 0.0f));
      stP(&P.contrib[s], make_float4(contrib.x, contrib.y, contrib.z, 0.0f));
    }
    stP(&P.status[s], (passes << 8) | (alive ? WF_ALIVE : 0u) | (wantShadow ? WF_PEND : 0u) | (ending ? WF_ENDING : 0u));
  }
  qNear = active && alive; qShad = active && wantShadow;
}

template <bool DEEP>
__global__ void __launch_bounds__(256, HPT_STREAM_WAVES) streamKernel(const DevScene S, const WfPool P, const WfJob job, uint slotsPerBlock, uint refillBelow, uint* blockQueues,
                                                                      uint* stackOverflow, uint gridLanes)
{
  __shared__ uint stackMem[LDS_STACK * 256];
  __shared__ uint stashMem[4 * 8 * 64];
  __shared__ uint qTail, qHead;
  const uint glane = blockIdx.x * 256u + threadIdx.x;
  const uint lane = threadIdx.x & 63u;
  TravStack stk; stk.lds = &stackMem[threadIdx.x]; stk.ovf = stackOverflow + glane; stk.ovfStride = gridLanes;
  uint* stash = stashMem + (threadIdx.x >> 6) * (8 * 64);
  uint* myQ = blockQueues + (size_t)blockIdx.x * (2u * slotsPerBlock);
  if (threadIdx.x == 0u) { qTail = 0u; qHead = 0u; }
  __syncthreads();

  while (true) {
    // ---- shade: every slot of the block once (wfShadeKernel<false, true>'s body per slot) ----------------------------------------------------
    for (uint j = threadIdx.x; j < slotsPerBlock; j += 256u) {
      const uint s = (j >> 6) * (gridDim.x << 6) + (blockIdx.x << 6) + (j & 63u);        // tiles of 64 consecutive slots dealt round-robin to the blocks
      bool qNear, qShad;
      streamShadeSlot(S, P, job, s, qNear, qShad);
      // the slot's rays go to the block's queue: ballot + prefix sum per wave, one LDS atomic per wave
      const unsigned long long mn = __ballot(qNear), ms = __ballot(qShad);
      const uint cn = (uint)__popcll(mn), cs = (uint)__popcll(ms);
      uint base = 0;
      if (lane == 0u && cn + cs != 0u) base = atomicAdd(&qTail, cn + cs);
      base = __shfl(base, 0);
      if (qNear) stP(&myQ[base + mbcnt64(mn)], s);
      if (qShad) stP(&myQ[base + cn + mbcnt64(ms)], s | 0x80000000u);
    }
    __syncthreads();                                                             // (workgroup scope: the block's stores are in its CU's L1 / L2 for its own loads)
    const uint total = qTail;
    if (total == 0u) break;                                                      // block-uniform: every slot of the block has finished

    // ---- trace: the four waves drain the block's queue with ray replacement (wfTraceKernel's loop, 4-wide tree, no suspension) ----------------
    {
      bool has = false, isAny = false, found = false, dry = false;
      uint stashCount = 0, slot = 0, cur = REF_NONE, curInst = 0xFFFFFFFFu;
      int sp = 0;
      V3 wo = v3(0, 0, 0), wd = v3(0, 0, 1), o = wo, d = wd, id = v3(0, 0, 0), oid = v3(0, 0, 0);
      float hitT = 0.0f, hitU = 0.0f, hitV = 0.0f; uint hitPrim = 0, hitInst = 0xFFFFFFFFu, hitSlot = 0xFFFFFFFFu;
      while (true) {
        {
          const unsigned long long mask = __ballot(!has);
          const uint n = (uint)__popcll(mask);
          if (n != 0u) {
            if (stashCount < n && !dry) {
              const uint want = 64u - stashCount;
              uint b0 = 0;
              if (lane == 0u) b0 = atomicAdd(&qHead, want);
              b0 = __shfl(b0, 0);
              const uint granted = b0 < total ? min(want, total - b0) : 0u;
              if (granted < want) dry = true;
              if (lane < granted) {
                const uint q = ldP(&myQ[b0 + lane]);
                const uint sl = q & 0x3FFFFFFFu;
                float4 a, b;
                if ((q >> 31) == 0u) { a = ldP(&P.rayO[sl]); b = ldP(&P.rayD[sl]); a.w = HPT_FLT_MAX; }
                else                 { a = ldP(&P.shO[sl]); b = ldP(&P.shD[sl]); }
                uint* e = stash + (stashCount + lane);
                e[0 * 64] = __float_as_uint(a.x); e[1 * 64] = __float_as_uint(a.y); e[2 * 64] = __float_as_uint(a.z); e[3 * 64] = __float_as_uint(a.w);
                e[4 * 64] = __float_as_uint(b.x); e[5 * 64] = __float_as_uint(b.y); e[6 * 64] = __float_as_uint(b.z); e[7 * 64] = q;
              }
              stashCount += granted;
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              __builtin_amdgcn_wave_barrier();
            }
            const uint give = min(n, stashCount);
            if (!has && mbcnt64(mask) < give) {
              const uint* e = stash + (stashCount - 1u - mbcnt64(mask));
              wo = v3(__uint_as_float(e[0 * 64]), __uint_as_float(e[1 * 64]), __uint_as_float(e[2 * 64])); hitT = __uint_as_float(e[3 * 64]);
              wd = v3(__uint_as_float(e[4 * 64]), __uint_as_float(e[5 * 64]), __uint_as_float(e[6 * 64]));
              const uint q = e[7 * 64];
              slot = q & 0x3FFFFFFFu; isAny = (q >> 31) != 0u;
              o = wo; d = wd; slabRay(wo, wd, id, oid);
              cur = S.root4; curInst = 0xFFFFFFFFu; sp = 0; found = false;
              hitPrim = 0xFFFFFFFFu; hitInst = 0xFFFFFFFFu; hitSlot = 0xFFFFFFFFu; hitU = 0.0f; hitV = 0.0f;
              has = true;
            }
            stashCount -= give;
            __builtin_amdgcn_wave_barrier();
          }
        }
        if (!__any(has)) break;
        const bool queueEmpty = (stashCount == 0u) && dry;
        if (has) {
          while (true) {
            while ((cur & REF_LEAF) == 0u) {
              wideNodeStep<DEEP>(S, stk, oid, id, hitT, cur, sp);
              if (S.nodeMin4 != 0u && (uint)__popcll(__ballot((cur & REF_LEAF) == 0u)) < S.nodeMin4) break;
            }
            const uint leaf = cur;
            bool done = (leaf == REF_NONE);
            if (!done && (leaf & REF_LEAF) != 0u) {
              const uint cnt = (leaf >> 28) & 7u;
              const uint first = leaf & 0x0FFFFFFFu;
              for (uint k = 0; k < cnt; k++) {
                const float4* tp = (const float4*)(S.tris + first + k);
                const float4 a = tp[0], b = tp[1], c = tp[2];
                const uint inst = __float_as_uint(b.w);
                if (inst != curInst) { toObjectSpace(S.insts, inst, wo, wd, o, d); curInst = inst; }     // world -> object space of this triangle's instance
                if (triangleTest(a, b, c, o, d, 0.0f, inst, hitT, hitPrim, hitInst, hitU, hitV, found)) hitSlot = first + k;
              }
              if (isAny && found) done = true;
              else if (sp > 0) { sp--; cur = DEEP ? stkPop(stk, sp) : stk.lds[sp * 256]; } else done = true;
            }
            if (done) {
              if (isAny) stP(&P.occl[slot], found ? 1u : 0u);
              else { stP(&P.hit[slot], make_float4(hitT, hitU, hitV, __uint_as_float(S.shadeTris != nullptr ? hitSlot : hitPrim))); stP(&P.hitInst[slot], found ? hitInst : 0xFFFFFFFFu); }
              has = false;
              break;
            }
            if (!queueEmpty && (uint)__popcll(__ballot(true)) < refillBelow) break;
          }
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0u) { qTail = 0u; qHead = 0u; }
    __syncthreads();
  }
}

template __global__ void streamKernel<false>(const DevScene, const WfPool, const WfJob, uint, uint, uint*, uint*, uint);
template __global__ void streamKernel<true>(const DevScene, const WfPool, const WfJob, uint, uint, uint*, uint*, uint);

} // namespace hpt
