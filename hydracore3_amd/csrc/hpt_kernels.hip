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
#include "hpt_shade.h"

namespace hpt {

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
// 3 = PathTrace for scenes with gltf + emissive materials only (shadeVertex<LEAN>)
// waves per SIMD the kernels with every BSDF branch (MODE 0 / 1 / 2) are compiled for; the lean and DR kernels keep HPT_MIN_WAVES.
// Measured (profiles/ab_full.sh, 1024^2 x 64 spp, Mpaths/s at 4 / 3 / 2 waves): Cornell forced onto this kernel 1410 / 1531 / 1264,
// legacy_materials 1585 / 1714 / 1528, env_map 1425 / 1507 / 1369, typed_materials 1123 / 1125 / 1105 - 168 VGPRs instead of 128 take
// the spills from 160 to 44 registers and that outweighs the lost wave.
#ifndef HPT_FULL_WAVES
#define HPT_FULL_WAVES 3
#endif
template <bool STATS, bool DR, int MODE, bool DEEP, bool FLAT, bool MOTION = false>
__global__ void __launch_bounds__(256, (DR || MODE == 3) ? HPT_MIN_WAVES : HPT_FULL_WAVES) pathTraceKernel(const DevScene S, const Job job)
{
  constexpr bool NAIVE = (MODE == 1), INRAYS = (MODE == 2), LEAN = (MODE == 3);
  __shared__ uint stackMem[LDS_STACK * 256];
  const uint glane = blockIdx.x * 256u + threadIdx.x;                    // slot in the per-lane HBM buffers
  TravStack stk; stk.lds = &stackMem[threadIdx.x]; stk.ovf = job.stackOverflow + glane; stk.ovfStride = job.gridLanes;

  // ---- per-lane state -----------------------------------------------------------------------------------------------
  // Cold per-PIXEL state (touched once per path, not per bounce) is parked in LDS so that it does not occupy VGPRs
  // across the two traversals and the shading code: running framebuffer value, packed (x,y), tid, passes left.
  __shared__ float coldPix[3 * 256];
  __shared__ uint  coldU[3 * 256];
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
  unsigned long long tPh[5] = {0, 0, 0, 0, 0}, tTrips = 0, tPrev = 0;
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
      traceAny<false, STATS, DEEP, FLAT, MOTION>(S, rpos, rdir, 0.0f, HPT_FLT_MAX, hit, stk, st, pathTime);
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

    if (alive) {
      if (STATS && hit.inst != 0xFFFFFFFFu) nHits++;
      didBounce = shadeVertex<DR, NAIVE, LEAN, MOTION>(S, job.data, hit, rpos, rdir, accum, thr, misPdf, misIor, flags, bounce, gen,
                                                 wantShadow, shPos, shDir, shFar, contrib, recA, recS, recdA, recdS, recTaps, recTex, tailR, pathTime);
    }

    STAMP(2);
    // ---- (6) shadow rays: RayQuery_AnyHit ---------------------------------------------------------------------------------------
    if (wantShadow) {
      HitRec sh;
      const bool occluded = traceAny<true, STATS, DEEP, FLAT, MOTION>(S, shPos, shDir, 0.0f, shFar, sh, stk, st, pathTime);
      if (STATS) { nRays++; nShadow++; }
      if (!occluded) accum = accum + contrib; else if (DR) { recS = v3(0, 0, 0); recdS = v3(0, 0, 0); }
    } else if (DR) { recS = v3(0, 0, 0); recdS = v3(0, 0, 0); }

    STAMP(3);
    // ---- (7) bookkeeping: adjoint record, end of path ------------------------------------------------------------------------------
    if (alive) {
      if (DR && didBounce) {
        drStoreRecord(job.record, job.recordLanes, glane, bounce, recA, recS, recdA, recdS, thrBefore, recTex, recTaps);
      }
      if (didBounce) bounce++;
      if ((flags & RAY_FLAG_IS_DEAD) != 0 || bounce >= maxBounce) {
        // kernel_HitEnvironment (integrator_pt.cpp:550-595), constant environment colour.
        // The DR replay adds the environment term unconditionally (diff_render/integrator_dr.cpp:1077-1098).
        const V3 env = DR ? ld3(S.envColor) : environmentRadiance(S, rdir, misPdf, flags, INRAYS ? (PIX_TID < job.packedCount ? job.packedXY[PIX_TID] : 0u) : PIX_XY);
        if (DR) accum = accum + thr * env;
        else if ((flags & RAY_FLAG_OUT_OF_SCENE) != 0) {
          if (S.integratorType == INTEGRATOR_STUPID_PT) accum = thr * env; else accum = accum + thr * env;
        }
        if (DR) {
          // PixelLossPT (integrator_dr.cpp:1103-1132) + hand-derived reverse sweep replacing __enzyme_autodiff (:1172-1183).
          // With T_0 = 1, T_{b+1} = T_b A_b and C = sum_b T_b S_b + T_n tail:
          //   dC/dtex_b = T_b dS_b + T_b dA_b R_{b+1},   R_b = S_b + A_b R_{b+1},   R_n = tail
          const uint pitch = (uint)S.winWidth;
          const uint XY = PIX_XY;
          const uint yRef = (uint)S.winHeight - ((XY & 0xFFFF0000u) >> 16) - 1u;
          const float* rp = job.refImg + ((size_t)yRef * pitch + (XY & 0x0000FFFFu)) * job.channels;
          const V3 diff = v3(accum.x - rp[0], accum.y - rp[1], accum.z - rp[2]);
          // One sample in ~1e8 on the 1M-triangle scene comes out non-finite (a 0/0 in a grazing GGX term; the reference's formulas,
          // unguarded there too). In the optimisation loop a single NaN gradient poisons Adam's moments for good, so such a sample
          // contributes neither loss, colour nor gradient here - the one deliberate deviation from PixelLossPT.
          const bool sane = __builtin_isfinite(diff.x + diff.y + diff.z);
          if (sane) {
            lossLocal += (diff.x * diff.x + diff.y * diff.y + diff.z * diff.z) / float(job.passNum);
            PIX(0) += accum.x; PIX(1) += accum.y; PIX(2) += accum.z;           // out_color += colorRend (:1124-1126)
          }
          if (sane) drReverseSweep(S, job.record, job.recordLanes, glane, bounce, tailR + env, diff, job.grad);
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

// ---- helper kernels --------------------------------------------------------------------------------------------------------------
// kernel_PackXY over the window (integrator_rt.cpp:13-31)
__global__ void packXYKernel(uint* out, int W, int H, uint ts)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) return;
  uint offset = (uint)y * (uint)W + (uint)x;
  if (ts != 1u) {
    const uint inX = (uint)x % ts, inY = (uint)y % ts;
    const uint wBlocks = (uint)W / ts;
    offset = (((uint)x / ts) + ((uint)y / ts) * wBlocks) * ts * ts + inY * ts + inX;
  }
  out[offset] = (((uint)y << 16) & 0xFFFF0000u) | ((uint)x & 0x0000FFFFu);
}

// InitRandomGens (integrator_pt.cpp:13-21)
__global__ void initRandomGensKernel(Rng* gens, uint n, uint firstSeed)
{
  const uint i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) gens[i] = rng_init(firstSeed + i);
}

// batched RayQuery_NearestHit / RayQuery_AnyHit for the ISceneObject entry points
template <bool FLAT, bool MOTION = false>
__global__ void __launch_bounds__(256) rayQueryKernel(const DevScene S, const float4* posNear, const float4* dirFar, uint n, void* out, int anyHit, uint* stackOverflow, float time = 0.0f)
{
  __shared__ uint stackMem[LDS_STACK * 256];
  const uint i = blockIdx.x * 256u + threadIdx.x;
  TravStack stk; stk.lds = &stackMem[threadIdx.x]; stk.ovf = stackOverflow + i; stk.ovfStride = gridDim.x * 256u;
  if (i >= n) return;
  const float4 p = posNear[i], d = dirFar[i];
  HitRec h; TravStats st; st.nodes = st.tris = st.insts = st.waveNodeIters = st.waveTriIters = 0;
  if (anyHit) {
    const bool occ = traceAny<true, false, true, FLAT, MOTION>(S, v3(p.x, p.y, p.z), v3(d.x, d.y, d.z), p.w, d.w, h, stk, st, time);
    ((uint*)out)[i] = occ ? 1u : 0u;
  } else {
    const bool found = traceAny<false, false, true, FLAT, MOTION>(S, v3(p.x, p.y, p.z), v3(d.x, d.y, d.z), p.w, d.w, h, stk, st, time);
    // CRT_Hit (CrossRT.h:23-30) as the Embree backend fills it (EmbreeRT.cpp:343-360)
    float4* o = (float4*)out + 2 * (size_t)i;
    if (found) {
      o[0] = make_float4(h.t, __uint_as_float(h.prim), __uint_as_float(h.inst), __uint_as_float(S.insts[h.inst].geomId));
      o[1] = make_float4(h.v, h.u, 1.0f - h.v - h.u, 0.0f);
    } else {
      o[0] = make_float4(d.w, __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu), __uint_as_float(0xFFFFFFFFu));
      o[1] = make_float4(0, 0, 0, 0);
    }
  }
}

// Image2D4fRegularizer (diff_render/integrator_dr.cpp:317-367): grad += d/d data of  sum_{interior pixels} sqrt(sum_{4 neighbours} |p0 - p_k|^2_rgb).
// Hand-derived instead of Enzyme, gather form (no atomics): texel q receives its own term (4 q - sum nb)/sqrt(S_q) when it is interior,
// and -(n - q)/sqrt(S_n) from each interior neighbour n; terms with S == 0 contribute nothing.
HPT_DEV float regS(const float4* d, int w, int x, int y)
{
  const float4 p0 = d[y * w + x], a = d[(y + 1) * w + x], b = d[(y - 1) * w + x], c = d[y * w + x - 1], e = d[y * w + x + 1];
  float S = 0.0f;
  const float4 nb[4] = { a, b, c, e };
  for (int k = 0; k < 4; k++) { const float dx = p0.x - nb[k].x, dy = p0.y - nb[k].y, dz = p0.z - nb[k].z; S += dx * dx + dy * dy + dz * dz; }
  return S;
}
__global__ void image2D4fRegularizerKernel(int w, int h, const float4* data, float4* grad)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const float4 q = data[y * w + x];
  float gx = 0.0f, gy = 0.0f, gz = 0.0f;
  const bool interior = x >= 1 && x < w - 1 && y >= 1 && y < h - 1;
  if (interior) {
    const float S = regS(data, w, x, y);
    if (S > 0.0f) {
      const float inv = 1.0f / __builtin_sqrtf(S);
      const float4 a = data[(y + 1) * w + x], b = data[(y - 1) * w + x], c = data[y * w + x - 1], e = data[y * w + x + 1];
      gx += (4.0f * q.x - (a.x + b.x + c.x + e.x)) * inv; gy += (4.0f * q.y - (a.y + b.y + c.y + e.y)) * inv; gz += (4.0f * q.z - (a.z + b.z + c.z + e.z)) * inv;
    }
  }
  const int nx[4] = { x, x, x - 1, x + 1 }, ny[4] = { y + 1, y - 1, y, y };
  for (int k = 0; k < 4; k++) {
    const int X = nx[k], Y = ny[k];
    if (X >= 1 && X < w - 1 && Y >= 1 && Y < h - 1) {              // neighbour n is an interior pixel: q is one of ITS four neighbours
      const float S = regS(data, w, X, Y);
      if (S > 0.0f) {
        const float inv = 1.0f / __builtin_sqrtf(S);
        const float4 n = data[Y * w + X];
        gx -= (n.x - q.x) * inv; gy -= (n.y - q.y) * inv; gz -= (n.z - q.z) * inv;
      }
    }
  }
  float4 g = grad[y * w + x];
  g.x += gx; g.y += gy; g.z += gz;
  grad[y * w + x] = g;
}

// AdamOptimizer<float>::step (diff_render/adam.h:43-62): HBM-bound, 16 bytes per lane per array
__global__ void adamStepKernel(float* state, const float* grad, float* momentum, float* gsq, size_t n, float gamma)
{
  const float alpha = 0.5f, beta = 0.25f, epsilon = 1e-8f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float g = grad[i];
    const float mo = momentum[i] * beta + g * (1.0f - beta);
    const float gs = 2.0f * (gsq[i] * alpha + (g * g) * (1.0f - alpha));
    momentum[i] = mo; gsq[i] = gs;
    state[i] -= (gamma * mo / (__builtin_sqrtf(gs + epsilon)));
  }
}

} // namespace hpt
