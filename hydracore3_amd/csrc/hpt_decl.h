// Launch interface between the host side (hpt_host.hip) and the path-tracing kernels, which are compiled as separate translation
// units (hpt_kernels.hip once per instantiation group, hpt_wavefront.hip) and linked into one libhydra_hip.so: argument structs,
// occupancy constants and the kernel templates' declarations. The host never sees a kernel body, so a change to one kernel family
// rebuilds only that family's objects and the objects build in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include "hpt_shade.h"

namespace hpt {

// ---- persistent megakernel (hpt_kernels.hip) -----------------------------------------------------------------------------------------
struct Job
{
  uint   tidBegin, tidCount;      // work items of this launch: item k renders tid = tidBegin + (k / chunk) * chunk * stride + k % chunk
  uint   tidChunk, tidStride;     // (stride 1 = one contiguous window; stride W = every W-th chunk: the interleaved multi-GPU split)
  uint   tidEnd;                  // number of threads of the whole frame (items mapping past it are dropped)
  uint   passNum, channels;
  float* outColor;                // full W*H*channels framebuffer (device)
  Rng*   gens;                    // m_randomGens (device, persistent)
  const uint* packedXY;           // m_packedXY
  uint  packedCount;              // entries in packedXY (the input-ray mode reads it only for the camera back plate)
  uint*  queue;                   // work-queue head (zeroed before launch)
  Counters* counters;             // instrumentation (STATS builds)
  // differentiable rendering
  const float* refImg;            // a_refImg
  const float* data;              // a_data
  float* grad;                    // a_dataGrad (atomically accumulated)
  float* lossAccum;               // sum over samples of loss / passNum
  float* record;                  // per-lane, per-bounce adjoint records: [bounce][field][lane]
  uint   recordLanes;             // total lanes of the grid (stride of the record buffer)
  uint   naive;                   // the spectral kernel only: NaivePathTrace (the RGB kernels have their own instantiations)
  uint   drSkipNonFinite;         // hpt_set_option("dr_skip_nonfinite"): samples whose radiance is not finite give neither loss, colour nor gradient
                                  // (0 = the reference's PixelLossPT, which adds them like any other)
  uint*  stackOverflow;           // HBM part of the traversal stacks: [depth - LDS_STACK][global lane]
  uint   gridLanes;
  const float4* inRayPos;         // PathTraceFromInputRays: RayPosAndW[tid] / RayDirAndT[tid] in camera space (MODE 2)
  const float4* inRayDir;
};

#ifndef HPT_MIN_WAVES
#define HPT_MIN_WAVES 4   // waves per SIMD the register allocator must fit (measured: 2 -> 725, 3 -> 913..1262, 4 -> 1005..1365 Mpaths/s on the Cornell box)
#endif
// MODE: 0 = PathTrace (MIS / shadow / stupid by m_intergatorType), 1 = NaivePathTrace, 2 = PathTraceFromInputRays (the caller's rays
// instead of camera rays, linear tid -> output index, raw accumColor: integrator_pt.cpp:159-199, 659-676, 761-798),
// 3 = PathTrace for scenes with gltf + emissive materials only (shadeVertex<LEAN>), 4 / 5 / 6 = 0 / 1 / 2 for scenes with thin films (shadeVertex<FILM>), 7 = 0 built for one more wave per SIMD (scenes with few material types:
// env_map 1584 -> 1670 Mpaths/s, while typed_materials loses 3 % there and stays on MODE 0)
// waves per SIMD the kernels with every BSDF branch (MODE 0 / 1 / 2) are compiled for; the lean and DR kernels keep HPT_MIN_WAVES.
// Measured (profiles/ab_full.sh, 1024^2 x 64 spp, Mpaths/s at 4 / 3 / 2 waves): Cornell forced onto this kernel 1410 / 1531 / 1264,
// legacy_materials 1585 / 1714 / 1528, env_map 1425 / 1507 / 1369, typed_materials 1123 / 1125 / 1105.
#ifndef HPT_FULL_WAVES
#define HPT_FULL_WAVES 3
#endif
#ifndef HPT_FILM_WAVES
#define HPT_FILM_WAVES 3   // the film kernels (MODE 4 / 5 / 6)
#endif
#ifndef HPT_DR_WAVES
#define HPT_DR_WAVES HPT_MIN_WAVES   // the PathTraceDR megakernel
#endif
#define HPT_PT_BOUNDS(DR, MODE) __launch_bounds__(256, (DR) ? HPT_DR_WAVES : ((MODE) == 3 ? HPT_MIN_WAVES : ((MODE) == 7 ? HPT_FULL_WAVES + 1 : ((MODE) >= 4 ? HPT_FILM_WAVES : HPT_FULL_WAVES))))
template <bool STATS, bool DR, int MODE, bool DEEP, bool FLAT, bool MOTION = false, bool SWEEP = false>
__global__ void HPT_PT_BOUNDS(DR, MODE) pathTraceKernel(const DevScene S, const Job job);

// spectral rendering (hpt_spectral.hip): one thread per pixel, four wavelengths per path
#ifndef HPT_SPEC_WAVES
#define HPT_SPEC_WAVES 4        // waves per SIMD the spectral kernel is compiled for: scopes 0 and 1
#endif
#ifndef HPT_SPEC_WIDE_WAVES
#define HPT_SPEC_WIDE_WAVES 3   // ... scope 2
#endif
template <bool DEEP, bool FLAT, bool SWEEP, bool MOTION, int SCOPE>   // SCOPE 0: the reference's spectral fixtures' needs; 1: + gltf; 2: + glass / blends / normal maps / environment maps / lens; 3: the same at 4 waves per SIMD
__global__ void __launch_bounds__(256, SCOPE == 2 ? HPT_SPEC_WIDE_WAVES : HPT_SPEC_WAVES) pathTraceSpectralKernel(const DevScene S, const Job job);

template <bool DEEP, bool FLAT, bool WIDE, int SCOPE>   // the same integrator under the block-local schedule (work queue, LDS ray pool, ray replacement; hpt_spectral.hip)
__global__ void __launch_bounds__(256, SCOPE == 2 ? HPT_SPEC_WIDE_WAVES : HPT_SPEC_WAVES) pathTraceBlockSpectralKernel(const DevScene S, const Job job, uint refillBelow, uint nodeMin);

// ---- megakernel with block-local ray repacking (hpt_block.hip) --------------------------------------------------------------------------
#ifndef HPT_BW_FWD_WAVES
#define HPT_BW_FWD_WAVES 4
#endif
#ifndef HPT_BW_DR_WAVES
#define HPT_BW_DR_WAVES 4
#endif
#ifndef HPT_BW_FULL_WAVES
#define HPT_BW_FULL_WAVES 3      // every BSDF branch: 74 .. 98 VGPRs spilled at 4 waves per SIMD
#endif
#define HPT_BW_WAVES(DR, LEAN) ((DR) ? HPT_BW_DR_WAVES : ((LEAN) ? HPT_BW_FWD_WAVES : HPT_BW_FULL_WAVES))
template <bool DR, bool LEAN, bool DEEP, bool FLAT, bool WIDE = false>   // WIDE: walk DevScene::nodes4 (heavy single-level scenes)
__global__ void __launch_bounds__(256, HPT_BW_WAVES(DR, LEAN)) pathTraceBlockKernel(const DevScene S, const Job job, uint refillBelow, uint nodeMin);

// ---- wavefront schedule (hpt_wavefront.hip) --------------------------------------------------------------------------------------------
static const uint WF_RANGES = 64u;                   // the trace kernel pulls rays from this many ranges of the queue (work stealing)
static const uint WF_CTR_WORDS = 32u * (1u + WF_RANGES);
static const uint WF_SUSP_WORDS = 10u;               // cur, sp, curInst, hitT, hitU, hitV, hitPrim, hitInst, found, (spare)
static const uint WF_ALIVE = 1u, WF_PEND = 2u, WF_ENDING = 4u;   // status bits; passes left in bits 8..31

struct WfPool
{
  float4* rayO;      // rpos.xyz, misPdf
  float4* rayD;      // rdir.xyz, misIor
  float4* thr;       // throughput.xyz, flags
  float4* acc;       // accumulated radiance.xyz, bounce
  float4* shO;       // shadow ray origin.xyz, far
  float4* shD;       // shadow ray direction.xyz
  float4* contrib;   // thr * shade of the pending light sample
  float4* hit;       // t, u, v, primId
  uint*   hitInst;   // instId or 0xFFFFFFFF
  uint*   occl;      // shadow ray result
  uint*   status;
  float*  lossSlot;  // DR: per-slot sum of the pixel's sample losses (reduced in double at the end: one float accumulator for 10^7..10^8
                     // samples loses the small increments - measured 1.5 % low on the 1M-triangle scene)
  float*  time;      // motion blur: the path's time in [0, 1] (drawn at regeneration, read by the trace pass and by the shading of every vertex)
  float4* waves;     // spectral rendering (wfShadeSpecKernel): the path's four wavelengths; thr / acc / contrib then hold four samples each and
  uint2*  fb;        // ... the flags and the bounce counter live here
  uint*   inflight;  // bit 0 / 1: the slot's closest-hit / shadow ray was suspended by a trace pass and has not finished yet
  uint*   rayQ[2];   // compacted ray queue of round (iteration & 1): slot id | (shadow ray ? 1 << 31 : 0) | (resumed ray ? 1 << 30 : 0)
  uint*   susp[2];   // traversal state of the rays a trace pass suspended, written for the NEXT round: [WF_SUSP_WORDS + stack][maxSusp],
                     // record k belongs to queue entry k of that round (suspended rays are queued first)
  uint    maxSusp;   // records per buffer (= lanes of the trace grid: a lane suspends at most one ray per pass)
  uint    suspStack; // stack entries per record
  uint*   ctr;       // two sets of WF_CTR_WORDS (set iteration & 1 is live): [0] rays queued by the shade pass;
                     // [32 * (1 + r)] head of queue range r for the trace pass (one 128-byte line each: the atomics of different
                     // ranges go to different L2 channels instead of serialising on one address)
};

struct WfJob
{
  uint   itemBase, itemCount;     // pool slot s renders work item itemBase + s (item -> tid as in Job)
  uint   tidBegin, tidChunk, tidStride, tidEnd;
  uint   passNum, channels, iter;
  uint   drSkipNonFinite;         // as in Job
  float* outColor;
  Rng*   gens;
  const uint* packedXY;
  // differentiable rendering (wfShadeKernel<DR = true>): a_refImg, a_data, a_dataGrad, loss accumulator, adjoint records [bounce][field][slot]
  const float* refImg; const float* data; float* grad; float* lossAccum; float* record;
};

#ifndef HPT_WF_SHADE_FULL_WAVES
#define HPT_WF_SHADE_FULL_WAVES 3      // the shade kernel with every BSDF branch: 1 M-triangle interior forced onto it 209 (4 waves) -> 214 Mpaths/s (profiles/ab_wfs.sh)
#endif
#ifndef HPT_WF_SHADE_WAVES
#define HPT_WF_SHADE_WAVES 5           // the lean forward shade kernel (98 VGPRs, no spills): 1 M triangles 4 / 5 / 6 waves -> 286.8 / 291.8 / 270.4 Mpaths/s (profiles/ab.sh)
#endif
#ifndef HPT_WF_SHADE_DR_WAVES
#define HPT_WF_SHADE_DR_WAVES 4        // the DR shade kernel holds the adjoint record's terms as well (112 VGPRs)
#endif
#ifndef HPT_WF_WAVES
#define HPT_WF_WAVES 5   // measured on the 1M-triangle scene: 4 -> 213, 5 -> 224, 6 -> 217 Mpaths/s (96 VGPRs: no spills; 24 KB of LDS per block)
#endif
#define HPT_WFS_BOUNDS(DR, LEAN) __launch_bounds__(256, (DR) ? HPT_WF_SHADE_DR_WAVES : ((LEAN) ? HPT_WF_SHADE_WAVES : HPT_WF_SHADE_FULL_WAVES))

// Ray compaction of the shade kernels. Every wave ballots its two kinds of rays and prefix-sums the lanes (mbcnt); the four waves of the block add
// their counts in LDS and ONE atomicAdd per block reserves the block's run of the queue. (One atomic per wave was measured at
// 0.97 ms per shade pass over 2M slots: ~100 K atomics on one address serialise in a single L2 channel.)
HPT_DEV void blockAppend(uint* counter, bool qNear, bool qShad, uint& posNear, uint& posShad)
{
  __shared__ uint waveCnt[4];
  __shared__ uint blockBase;
  const unsigned long long mn = __ballot(qNear), ms = __ballot(qShad);
  const uint cn = (uint)__popcll(mn), cs = (uint)__popcll(ms);
  const uint wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63u) == 0u) waveCnt[wave] = cn + cs;
  __syncthreads();
  if (threadIdx.x == 0u) {
    const uint tot = waveCnt[0] + waveCnt[1] + waveCnt[2] + waveCnt[3];
    blockBase = tot ? atomicAdd(counter, tot) : 0u;
  }
  __syncthreads();
  uint base = blockBase;
  for (uint w = 0; w < wave; w++) base += waveCnt[w];
  posNear = base + mbcnt64(mn);
  posShad = base + cn + mbcnt64(ms);
}

__global__ void wfInitKernel(WfPool P, uint n, uint passNum);
#ifndef HPT_STREAM_WAVES
#define HPT_STREAM_WAVES 5
#endif
template <bool DEEP>                                                    // block-owned streaming form of the wavefront schedule (hpt_stream.hip; schedule 4)
__global__ void __launch_bounds__(256, HPT_STREAM_WAVES) streamKernel(const DevScene S, const WfPool P, const WfJob job, uint slotsPerBlock, uint refillBelow, uint* blockQueues,
                                                                      uint* stackOverflow, uint gridLanes);
template <int SCOPE>                                                    // spectral rendering under the wavefront schedule (hpt_spectral.hip)
__global__ void __launch_bounds__(256, SCOPE == 2 ? HPT_SPEC_WIDE_WAVES : HPT_SPEC_WAVES) wfShadeSpecKernel(const DevScene S, const WfPool P, const WfJob job);
template <bool DR, bool LEAN, bool MOTION = false, bool FILM = false>   // FILM: shadeVertex<FILM> (thin films, hpt_film.h)
__global__ void HPT_WFS_BOUNDS(DR, LEAN) wfShadeKernel(const DevScene S, const WfPool P, const WfJob job);
template <bool DEEP, bool FLAT, bool STATS, bool MOTION = false, bool WIDE = false>   // WIDE: walk DevScene::nodes4 (4-wide compressed nodes) instead of the BVH2
__global__ void __launch_bounds__(256, HPT_WF_WAVES) wfTraceKernel(const DevScene S, const WfPool P, uint iter, uint refillBelow, uint grace,
                                                                   uint* stackOverflow, uint gridLanes, Counters* counters);
__global__ void __launch_bounds__(256) wfLossReduceKernel(const float* lossSlot, uint n, double* acc);
__global__ void wfLossFinishKernel(const double* acc, float* loss);

} // namespace hpt
