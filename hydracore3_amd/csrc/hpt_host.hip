// Host side of the C ABI (include/hydra_hip.h): context, BVH2 build + upload, scene tables, launches, timing.
// Compiled by hipcc together with the kernels; exports only the extern "C" hpt_* symbols.
#include "plastic_precompute.h"
#include "film_precompute.h"
#include "jpeg_decode.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <new>
#include <stdexcept>
#include <vector>
#include <set>
#include <chrono>
#include <dlfcn.h>
#include <sched.h>
#include <thread>
#include <atomic>

#include "../../include/hydra_hip.h"
#include "hpt_decl.h"
#include "hpt_helpers.hip"
#include "bvh_build.h"
#include "hpt_lbvh.h"

static const float BLOCK_SAH_VISITS = 8.0f;                     // automatic schedule: from this many expected inner-node visits per ray the megakernel repacks rays block-locally (hpt_block.hip)
static const uint MAX_STACK = 64;                           // traversal stack entries per lane: LDS_STACK in LDS + the rest in an HBM overflow buffer
// Static scenes with at most this many INSTANCED triangles get the single-level world-space BVH (48 B + ~32 B of nodes per triangle:
// 32 M triangles = 2.6 GB of the 288 GB); beyond it (heavy instancing) the two-level TLAS/BLAS layout is kept.
static const size_t FLAT_TRI_BUDGET = size_t(32) << 20;

using namespace hpt;

// Scenes with at least this many instanced triangles count as "heavy": wavefront schedule, single-level BVH, voted exit of the node loop.
// A "heavy" scene is one whose committed BVH is expected to cost a ray at least this many inner-node visits (sah_node_visits, the
// surface-area estimate): it gets the wavefront schedule (when the call has enough pixels) and the voted node-loop exit. Measured
// (profiles/sah_info.py, crossover.sh, crossover_small.sh, full_kernel.py): Cornell box 8.5, test_228 (8 202 triangles) 10.3, the own fixtures
// 3.7 ... 4.8 - all fastest on the megakernel without the vote (test_228: 667 vs 531 Mpaths/s); the interior generator from 4 850 to 1 M
// triangles 30 ... 60 - all fastest on the wavefront schedule with the vote (4 850 triangles: 340 vs 254). The triangle count does not
// separate the two families (test_228 has more triangles than the smallest interior).
static const float  HEAVY_SAH_VISITS = 20.0f;
static const uint   WF_AUTO_PIXELS = 1u << 19;            // fewer pixels per call cannot keep the wavefront trace kernel's lanes supplied (see useWavefront)
static const size_t FLAT_AUTO_TRIS = size_t(1) << 12;      // instanced triangles from which the single-level layout is chosen whatever the instance count
// Tiny scenes (the Cornell-box class): no tree walk at all, the wave sweeps the instances' triangles with scalar loads (hpt_device.h: traceSweep).
// Cost is linear in the triangle count (one exact triangle test per lane and triangle, ~72 VALU instructions at 100 % lane utilisation) against
// ~2 450 issue slots per ray of the BVH walk on the Cornell box at 31 % utilisation: the crossover sits around 32 instanced triangles.
static const size_t SWEEP_MAX_TRIS = 32, SWEEP_MAX_INSTS = 8;
static const size_t MANY_INSTANCES = 6;                      // instances from which the single-level layout is chosen for light scenes too (see hpt_commit_scene)

namespace {

struct Geom
{
  std::vector<float> pos;        // xyz per vertex (tightly packed copy)
  std::vector<uint>  idx;
  Bvh2               bvh;        // local references
  std::vector<BvhTri> tris;      // BVH order
  bool               dirty = true;
};

struct Inst { uint geomId; float m[16]; bool motion = false; float m1[16] = {0}; };   // m1: the matrix at time 1 of a moving instance (AddInstanceMotion)

template <class T>
struct DevBuf                      // owning device array; freed on scope exit (locals on error paths) or by hpt_destroy (context members)
{
  T* p = nullptr; size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
  hipError_t alloc(size_t count) { if (count <= n && p) return hipSuccess; release(); n = count; return count ? hipMalloc((void**)&p, count * sizeof(T)) : hipSuccess; }
  hipError_t upload(const T* src, size_t count)
  {
    hipError_t e = alloc(count); if (e != hipSuccess) return e;
    return count ? hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice) : hipSuccess;
  }
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

} // namespace

struct hpt_ctx
{
  int device = 0, numCUs = 256;
  std::string err, devName;

  // ISceneObject state
  std::vector<Geom> geoms;
  std::vector<Inst> insts;
  bool accelCommitted = false;
  uint stackNeeded = 0;

  // device buffers
  DevBuf<BvhNode> dNodes; DevBuf<BvhTri> dTris; DevBuf<BvhInst> dInsts, dSweepInsts; DevBuf<BvhTri> dSweepTris; DevBuf<float> dSweepBoxes;
  DevBuf<uint> dTriIndices, dMatIdByPrim, dMatVertOffset, dPackedXY;
  DevBuf<float> dVData, dNormMat;
  DevBuf<int> dRemapInst, dRemapLists;
  DevBuf<MaterialRec> dMaterials; DevBuf<LightRec> dLights; DevBuf<TexRec> dTextures;
  std::vector<void*> texData; std::vector<TexRec> hTextures;
  DevBuf<float> dSpecValues; DevBuf<uint> dSpecOffsetSz; DevBuf<float4> dCieXYZ;   // spectral tables (m_spec_values, m_spec_offset_sz, m_cie_xyz)
  bool spectralOk = false; std::string spectralWhyNot;   // whether the uploaded scene is within the spectral kernel's scope
  bool fewMaterialTypes = false;                         // at most three reachable material types, no blends / plastic / normal maps: the full-material kernel built for 4 waves (MODE 7)
  bool spectralGltfMats = true, spectralHeavyMats = true;   // a reachable material needs scope 1 (gltf) / scope 2 (glass, blend, normal map) of the spectral kernel (hpt_spectral.hip)
  // thin films (integrator_pt.h:587-590): the tables a film material indexes, and what the uploaded materials say about them
  DevBuf<float> dFilmsEtaK, dPrecompFilms; DevBuf<uint> dFilmsSpecId, dSpecTexIdsWavelengths, dSpecTexOffsetSz;
  std::vector<uint> hFilmsSpecId; size_t numFilmsEtaK = 0, numPrecompFilms = 0; uint numSpectraHost = 0;
  bool hasFilm = false, filmTablesRGB = true, filmTablesSpectral = true;   // a MAT_TYPE_THIN_FILM is in the table; its precomputed tables have the size RGB / spectral rendering reads
  DevBuf<float> dArrays1f; size_t numArrays1f = 0;       // m_arrays1f (pdf table of a sampled environment map)
  DevBuf<float4> dLensLines;                             // m_lines of the lens simulation (hpt_set_optics)
  DevBuf<float> dInstMotion, dNormMat2;                  // motion blur: key matrices of the moving instances, end-of-motion normal matrices
  // device refit of the single-level layout (refit_flat): the committed tree's nodes grouped by level, scratch boxes, what the tree was built for
  DevBuf<uint> dLevelNodes; DevBuf<float> dTriBox, dNodeBounds, dInstO2W;
  DevBuf<BvhNode4> dNodes4; DevBuf<uint> dNodes4Src;     // the single-level tree collapsed to 4-wide compressed nodes; per child the BVH2 (node << 1 | side) its box comes from
  uint nodes4Count = 0, stackNeeded4 = 0;                // (0: no wide tree)
  DevBuf<float4> dShadeTris;                             // DevScene::shadeTris (64 B per triangle record of the single-level layout), built lazily before a launch
  bool shadeTrisDirty = true, shadeTrisEnabled = true;   // (hpt_set_option("shade_records", 0): the kernels gather vertex data through the index chain)
  int  buildThreads = 0;                                 // hpt_set_option("build_threads", n): host threads of CommitScene (0: automatic)
  bool statsWide = false;                                // hpt_set_option("stats_wide", 1): the instrumented probe walks the 4-wide tree (what a wavefront call on this scene does)
  bool wideEnabled = true;                               // hpt_set_option("wide_nodes", 0): the trace kernel walks the BVH2
  std::vector<uint> levelOffsets;                        // nodes of level l = dLevelNodes[levelOffsets[l] .. levelOffsets[l + 1])
  std::vector<BvhTri> flatTris;                         // host copy of the single-level layout's triangle records (BVH order)
  bool flatRefittable = false;                           // a single-level tree is committed and nothing but instance matrices / vertex positions changed since
  std::vector<uint> dirtyGeoms;                          // meshes whose vertices changed since the last commit (UpdateGeom_Triangles3f)
  bool refitEnabled = true;                              // hpt_set_option("refit", 0): always rebuild
  float tCommit[4] = {0, 0, 0, 0};                       // last CommitScene: host build ms, upload ms, device refit ms, 1 = refit / 0 = build
  float sahVisits = 0.0f;                                // expected inner-node visits per ray of the committed structure (sah_node_visits)
  bool anyMotion = false;                                // some instance moves: the MOTION kernel variants (either BVH layout, either schedule; no sweep, no refit)
  std::vector<MaterialRec> hMaterials;                   // host mirror of m_materials: the blend graph is validated as a whole
  std::vector<uint> hLightGeom;                          // geomType of every light, to validate m_envLightId
  std::vector<uint> hTriIndices;                         // host mirror of m_triIndices: Update_m_matIdOffsets re-validates the vertex indices a mesh will read
  bool envLightOk(uint id) const { return id < hLightGeom.size() && hLightGeom[id] == LIGHT_GEOM_ENV; }
  DevBuf<Rng> dGens;
  DevBuf<uint> dQueue, dStackOvf; DevBuf<Counters> dCounters;
  DevBuf<float> dFrame, dRecord, dRef, dData, dGrad, dLoss; DevBuf<double> dLossAcc;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // wavefront schedule (hpt_wavefront.hip): the pixels of a call are cut into groups, each with its own path pool, ray queue and
  // stream, so that the tail of one group's trace pass overlaps the other groups' work
  struct WfGroup
  {
    DevBuf<float4> f4[8], waves; DevBuf<uint2> fb; DevBuf<uint> u[9]; DevBuf<float> rec, lossSlot, time;
    hipStream_t stream = nullptr; hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; hipEvent_t done = nullptr;
    uint* progress = nullptr;            // pinned: rays queued after every WF_CHECK-th shade pass (0 = group finished)
    uint checkpoints = 0, itemBase = 0, itemCount = 0; unsigned long long it = 0; bool finished = false;
  };
  std::vector<WfGroup*> wfGroups;
  hipEvent_t wfFork = nullptr;
  int  wfGroupCount = 0;                 // 0 = automatic
  uint wfIterCap = 0;                    // diagnostic (hpt_set_option "dbg_wf_iter_cap"): rounds after which the wavefront loop gives up
  uint wfGrace = 16;                     // trips a trace wave keeps going after the queue ran dry before it suspends its rays (0 = never)
  // multi-GPU collectives (RCCL, loaded on first use: single-GPU users never touch it)
  void* rcclLib = nullptr; void* comm = nullptr; int commRanks = 0, commRank = 0;
  bool forceFull = false;                // diagnostic (hpt_set_option "force_full_materials")
  bool drSkipNonFinite = false;          // hpt_set_option "dr_skip_nonfinite": off = PixelLossPT as the reference has it
  bool leanMaterials = false;            // every material is gltf or emissive: the kernels without the other BSDF branches are used
  int  schedule = 0;                     // 0 automatic, 1 megakernel, 2 wavefront, 3 megakernel with block-local ray repacking (hpt_set_schedule)
  int  bwWide = -1;                        // hpt_set_option("bw_wide"): -1 the 4-wide tree on heavy scenes only, 0 never, 1 whenever the scene has one
  uint bwRefillBelow = 48, bwNodeMin = 4;   // (profiles/bw_sweep.py: test_228 class, refill 32 .. 64 x vote 0 / 4 / 8 / 16)
  uint bwUnused = 0;   // hpt_block.hip: a wave refills when fewer lanes hold a ray; its node loop's vote
  int  nodeMinOverride = -1;             // env HPT_NODE_MIN (tuning): overrides the per-scene choice of DevScene::nodeMin
  uint wfRefillBelow = 56;               // a trace wave refills from the queue when fewer lanes than this still hold a ray
  int  wfBlocksPerCU = 0;
  size_t instTris = 0;                   // instanced triangles of the committed scene
  // CommitScene on the device (hpt_lbvh.hip): -1 = by CommitScene's options (BUILD_LOW / BUILD_MEDIUM without BUILD_HIGH), 0 = never, 1 = whenever the single-level layout is built
  int deviceBuild = -1; void* lbvh = nullptr; unsigned long long geomVersion = 1, lbvhGeomVersion = 0; std::vector<size_t> lbvhPosOff, lbvhIdxOff;
  uint lastSchedule = 1, lastWfIters = 0;
  uint lastWide = 0, lastShadeRecords = 0, lastDeep = 0;          // what the last launch walked: 4-wide compressed tree, 64-byte shading records, HBM part of the stacks

  DevScene S;
  bool sceneUploaded = false, paramsSet = false;
  uint packedCount = 0;
  int  blocksPerCU = 0;
  int  accelLayout = 0;                  // 0 automatic (flat when it fits the budget), 1 force two-level, 2 force flat
  uint tidChunk = 0, tidStride = 1;      // hpt_set_tid_interleave
  bool instrument = false;
  uint64_t gradSize = 0;

  // GetExecutionTime slots
  float tPathTrace[4] = {0, 0, 0, 0}, tNaive[4] = {0, 0, 0, 0}, tDR[4] = {0, 0, 0, 0}, tFromRays[4] = {0, 0, 0, 0};
  float lastKernelMs = 0.0f;

  int fail(int code, const std::string& m) { err = m; std::fprintf(stderr, "[hydra_hip] %s\n", m.c_str()); return code; }
  int hipFail(hipError_t e, const char* what) { return fail(HPT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
};

#define HIPCHK(ctx, call) do { hipError_t _e = (call); if (_e != hipSuccess) return (ctx)->hipFail(_e, #call); } while (0)

// Nothing throws across the C boundary (include/hydra_hip.h): every extern "C" body is a function-try-block whose handler lands here. The host
// side builds std::vector / std::string / std::thread objects (scene tables, BVH build), so std::bad_alloc and std::system_error are real
// possibilities; they become an error code and a message for hpt_last_error like every other failure.
static int hptGuard(hpt_ctx* c, const char* where)
{
  int code = HPT_ERR_STATE; std::string what = "unknown exception";
  try { throw; }
  catch (const std::bad_alloc&) { code = HPT_ERR_NOMEM; what = "out of host memory (std::bad_alloc): sizes too large for this host"; }
  catch (const std::length_error& e) { code = HPT_ERR_NOMEM; what = std::string("container size limit exceeded (std::length_error): ") + e.what(); }
  catch (const std::exception& e) { what = e.what(); }
  catch (...) {}
  if (c) { try { return c->fail(code, std::string(where) + ": " + what); } catch (...) {} }
  return code;
}


// ---- lifetime -----------------------------------------------------------------------------------------------------------------
extern "C" int hpt_comm_destroy(hpt_ctx* c);

extern "C" int hpt_create(int device, hpt_ctx** out)
try {
  if (!out) return HPT_ERR_ARG;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) { std::fprintf(stderr, "[hydra_hip] no HIP device: %s\n", hipGetErrorString(e)); return HPT_ERR_HIP; }
  if (device < 0 || device >= n) return HPT_ERR_ARG;
  e = hipSetDevice(device);
  if (e != hipSuccess) return HPT_ERR_HIP;
  hpt_ctx* c = new hpt_ctx();
  c->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) { c->numCUs = prop.multiProcessorCount; c->devName = prop.gcnArchName; }
  (void)hipEventCreate(&c->ev0); (void)hipEventCreate(&c->ev1);
  if (const char* e = std::getenv("HPT_NODE_MIN")) c->nodeMinOverride = std::atoi(e) & 63;
  if (const char* e = std::getenv("HPT_WF_GRACE")) c->wfGrace = (uint)std::atoi(e);
  if (const char* e = std::getenv("HPT_SHADE_RECORDS")) c->shadeTrisEnabled = std::atoi(e) != 0;
  if (const char* e = std::getenv("HPT_WIDE_NODES")) c->wideEnabled = std::atoi(e) != 0;   // (read before any scene is committed: CommitScene derives DevScene::megaWide from it)
  std::memset(&c->S, 0, sizeof(DevScene));
  c->S.rootRef = REF_NONE;
  *out = c;
  return HPT_OK;
}
catch (...) { return hptGuard(nullptr, "hpt_create"); }

extern "C" void hpt_destroy(hpt_ctx* c)
try {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  (void)hpt_comm_destroy(c);
  lbvhDestroy(c->lbvh); c->lbvh = nullptr;
  c->dNodes.release(); c->dTris.release(); c->dInsts.release(); c->dSweepInsts.release(); c->dSweepTris.release(); c->dSweepBoxes.release(); c->dLevelNodes.release(); c->dShadeTris.release(); c->dNodes4.release(); c->dNodes4Src.release(); c->dTriBox.release(); c->dNodeBounds.release(); c->dInstO2W.release(); c->dTriIndices.release(); c->dMatIdByPrim.release();
  c->dMatVertOffset.release(); c->dPackedXY.release(); c->dVData.release(); c->dNormMat.release(); c->dRemapInst.release();
  c->dRemapLists.release(); c->dMaterials.release(); c->dLights.release(); c->dTextures.release(); c->dArrays1f.release(); c->dSpecValues.release(); c->dSpecOffsetSz.release(); c->dCieXYZ.release(); c->dFilmsEtaK.release(); c->dPrecompFilms.release(); c->dFilmsSpecId.release(); c->dSpecTexIdsWavelengths.release(); c->dSpecTexOffsetSz.release(); c->dGens.release();
  c->dQueue.release(); c->dStackOvf.release(); c->dCounters.release(); c->dFrame.release(); c->dRecord.release(); c->dRef.release(); c->dData.release();
  c->dGrad.release(); c->dLoss.release(); c->dLossAcc.release();
  for (hpt_ctx::WfGroup* g : c->wfGroups) {
    for (auto& b : g->f4) b.release();
    for (auto& b : g->u) b.release();
    g->rec.release(); g->lossSlot.release(); g->time.release();
    if (g->progress) (void)hipHostFree(g->progress);
    for (auto& e : g->ev) if (e) (void)hipEventDestroy(e);
    if (g->done) (void)hipEventDestroy(g->done);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
  }
  if (c->wfFork) (void)hipEventDestroy(c->wfFork);
  for (void* p : c->texData) if (p) (void)hipFree(p);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  delete c;
}
catch (...) { (void)hptGuard(c, "hpt_destroy"); }

extern "C" const char* hpt_last_error(hpt_ctx* c) try { return c ? c->err.c_str() : "null context"; }
catch (...) { return ""; }

extern "C" int hpt_device_info(hpt_ctx* c, int* numCUs, int* wavefront, char* name, size_t nameLen)
try {
  if (!c) return HPT_ERR_ARG;
  if (numCUs) *numCUs = c->numCUs;
  if (wavefront) *wavefront = 64;
  if (name && nameLen) { std::strncpy(name, c->devName.c_str(), nameLen - 1); name[nameLen - 1] = 0; }
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_device_info"); }

// ---- the JPEG reader of the scene loaders (host only) ----------------------------------------------------------------------------------------
extern "C" int hpt_decode_jpeg(const uint8_t* file, uint64_t fileSize, uint32_t* outWidth, uint32_t* outHeight, uint8_t* outRGBA8, uint64_t outCapacity)
try {
  if (!file || !outWidth || !outHeight) return HPT_ERR_ARG;
  std::vector<uint8_t> rgba; std::string err; uint32_t w = 0, h = 0;
  try {                                                        // (nothing throws across the C boundary)
    std::vector<uint8_t> f(file, file + fileSize);
    if (!hydra_hip::jpeg::decode(f, w, h, rgba, err)) return HPT_ERR_UNSUPPORTED;
  } catch (const std::exception&) { return HPT_ERR_UNSUPPORTED; }
  *outWidth = w; *outHeight = h;
  if (!outRGBA8) return HPT_OK;                                // (a size query)
  if (outCapacity < rgba.size()) return HPT_ERR_ARG;
  std::memcpy(outRGBA8, rgba.data(), rgba.size());
  return HPT_OK;
}
catch (...) { return hptGuard(nullptr, "hpt_decode_jpeg"); }

// ---- precomputeThinFilmSpectral / precomputeThinFilmRGB for the scene loaders (host only, no device needed) ------------------------------------
extern "C" int hpt_film_precompute(const hpt_film_params* fp, float* outTable, uint64_t outCapacity, uint64_t* outCount, int* outPrecomputed)
try {
  if (!fp || !outCount) return HPT_ERR_ARG;
  hydra_hip::film::Params p;
  p.spectralMode = fp->spectralMode; p.extIOR = fp->extIOR; p.layers = fp->layers; p.eta = fp->eta; p.k = fp->k; p.etaSpecId = fp->etaSpecId; p.kSpecId = fp->kSpecId;
  p.thickness = fp->thickness; p.thicknessMap = fp->thicknessMap; p.thicknessMin = fp->thicknessMin; p.thicknessMax = fp->thicknessMax;
  p.specValues = fp->specValues; p.specOffsetSz = fp->specOffsetSz; p.numSpectra = fp->numSpectra; p.cieXYZ = fp->cieXYZ;
  if (p.layers < 1 || p.layers > hydra_hip::film::MAX_LAYERS) return HPT_ERR_ARG;
  if (p.specValues && !p.specOffsetSz) return HPT_ERR_ARG;
  const bool pre = hydra_hip::film::precomputed(p);
  if (outPrecomputed) *outPrecomputed = pre ? 1 : 0;
  *outCount = pre ? (uint64_t)hydra_hip::film::tableSize(p) : 0u;
  if (!pre || !outTable) return HPT_OK;                        // (a size query)
  if (outCapacity < *outCount) return HPT_ERR_ARG;
  try { return hydra_hip::film::precompute(p, outTable) ? HPT_OK : HPT_ERR_ARG; } catch (const std::exception&) { return HPT_ERR_ARG; }
}
catch (...) { return hptGuard(nullptr, "hpt_film_precompute"); }

// ---- mi::fresnel_coat_precompute for the scene loaders (host only, no device needed) --------------------------------------------------------
extern "C" int hpt_plastic_precompute(float alpha, float intIor, float extIor, const float* diffuse4, const float* specular4,
                                      float* outTransmittance64, float* outInternalReflectance, float* outSpecularSamplingWeight)
try {
  if (!diffuse4 || !specular4 || !outTransmittance64 || !outInternalReflectance || !outSpecularSamplingWeight) return HPT_ERR_ARG;
  if (!(alpha > 0.0f) || !(intIor > 0.0f) || !(extIor > 0.0f)) return HPT_ERR_ARG;
  const hydra_hip::plastic::CoatPrecomputed p = hydra_hip::plastic::fresnelCoatPrecompute(alpha, intIor, extIor, diffuse4, specular4);
  std::memcpy(outTransmittance64, p.transmittance, sizeof(p.transmittance));
  *outInternalReflectance = p.internalReflectance; *outSpecularSamplingWeight = p.specularSamplingWeight;
  return HPT_OK;
}
catch (...) { return hptGuard(nullptr, "hpt_plastic_precompute"); }

// ---- device memory for callers of the *_dev entry points that do not link the HIP runtime themselves ---------------------------------
extern "C" int hpt_device_malloc(hpt_ctx* c, size_t bytes, void** outDev)
try {
  if (!c || !outDev) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  *outDev = nullptr;
  HIPCHK(c, hipMalloc(outDev, bytes ? bytes : 4));
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_device_malloc"); }
extern "C" int hpt_device_free(hpt_ctx* c, void* dev)
try {
  if (!c) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (dev) HIPCHK(c, hipFree(dev));
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_device_free"); }
extern "C" int hpt_device_copy(hpt_ctx* c, void* dst, const void* src, size_t bytes, int kind)
try {
  if (!c || (bytes && (!dst || !src)) || kind < 1 || kind > 3) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if (bytes) HIPCHK(c, hipMemcpy(dst, src, bytes, k));                      // synchronous: orders after the asynchronous *_dev launches on the null stream
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_device_copy"); }
extern "C" int hpt_device_memset(hpt_ctx* c, void* dev, int value, size_t bytes)
try {
  if (!c || (bytes && !dev)) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (bytes) HIPCHK(c, hipMemsetAsync(dev, value, bytes, nullptr));
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_device_memset"); }

// ---- ISceneObject ---------------------------------------------------------------------------------------------------------------
extern "C" int hpt_clear_geom(hpt_ctx* c)
try {
  if (!c) return HPT_ERR_ARG;
  c->geoms.clear(); c->insts.clear(); c->accelCommitted = false; c->flatRefittable = false; c->geomVersion++;
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_clear_geom"); }

static int fill_geom(hpt_ctx* c, Geom& g, const float* vpos, size_t nVert, const uint32_t* idx, size_t nIdx, size_t stride)
{
  if (stride == 0) stride = sizeof(float) * 3;
  if (stride % sizeof(float) != 0) return c->fail(HPT_ERR_ARG, "AddGeom_Triangles3f: vByteStride must be a multiple of sizeof(float)");
  if (!vpos || !idx) return c->fail(HPT_ERR_ARG, "AddGeom_Triangles3f: nullptr input");
  const size_t fs = stride / sizeof(float);
  g.pos.resize(nVert * 3);
  for (size_t i = 0; i < nVert; i++) { g.pos[3 * i + 0] = vpos[fs * i + 0]; g.pos[3 * i + 1] = vpos[fs * i + 1]; g.pos[3 * i + 2] = vpos[fs * i + 2]; }
  g.idx.assign(idx, idx + (nIdx / 3) * 3);
  for (uint v : g.idx) if (v >= nVert) return c->fail(HPT_ERR_ARG, "AddGeom_Triangles3f: index out of range");
  g.dirty = true; c->geomVersion++;
  return HPT_OK;
}

extern "C" uint32_t hpt_add_geom_triangles3f(hpt_ctx* c, const float* vpos, size_t nVert, const uint32_t* idx, size_t nIdx, uint32_t, size_t stride)
try {
  if (!c) return 0xFFFFFFFFu;
  Geom g;
  if (fill_geom(c, g, vpos, nVert, idx, nIdx, stride) != HPT_OK) return 0xFFFFFFFFu;
  c->geoms.push_back(std::move(g));
  c->accelCommitted = false; c->flatRefittable = false;
  return (uint32_t)(c->geoms.size() - 1);
}
catch (...) { (void)hptGuard(c, "hpt_add_geom_triangles3f"); return 0xFFFFFFFFu; }

extern "C" int hpt_update_geom_triangles3f(hpt_ctx* c, uint32_t geomId, const float* vpos, size_t nVert, const uint32_t* idx, size_t nIdx, uint32_t, size_t stride)
try {
  if (!c) return HPT_ERR_ARG;
  if (geomId >= c->geoms.size()) return c->fail(HPT_ERR_ARG, "UpdateGeom_Triangles3f: bad geomId");
  Geom& g = c->geoms[geomId];
  if (nIdx > g.idx.size() || nVert * 3 > g.pos.size()) return c->fail(HPT_ERR_ARG, "UpdateGeom_Triangles3f: growing a geometry is not supported");
  c->accelCommitted = false;
  // the committed single-level tree survives a change of vertex POSITIONS (same triangles): its boxes are refitted on the device
  if ((nIdx / 3) * 3 != g.idx.size() || std::memcmp(idx, g.idx.data(), g.idx.size() * sizeof(uint32_t)) != 0) c->flatRefittable = false;
  else c->dirtyGeoms.push_back(geomId);
  return fill_geom(c, g, vpos, nVert, idx, nIdx, stride);
}
catch (...) { return hptGuard(c, "hpt_update_geom_triangles3f"); }

extern "C" int hpt_clear_scene(hpt_ctx* c) try { if (!c) return HPT_ERR_ARG; c->insts.clear(); c->accelCommitted = false; c->flatRefittable = false; return HPT_OK; }
catch (...) { return hptGuard(c, "hpt_clear_scene"); }

extern "C" uint32_t hpt_add_instance(hpt_ctx* c, uint32_t geomId, const float m[16])
try {
  if (!c || !m || geomId >= c->geoms.size()) return 0xFFFFFFFFu;
  Inst in; in.geomId = geomId; std::memcpy(in.m, m, 64);
  c->insts.push_back(in);
  c->accelCommitted = false; c->flatRefittable = false;
  return (uint32_t)(c->insts.size() - 1);
}
catch (...) { (void)hptGuard(c, "hpt_add_instance"); return 0xFFFFFFFFu; }

extern "C" uint32_t hpt_add_instance_motion(hpt_ctx* c, uint32_t geomId, const float* matrices, uint32_t matrixNumber)
try {
  if (!c || !matrices || geomId >= c->geoms.size() || matrixNumber == 0) return 0xFFFFFFFFu;
  if (matrixNumber == 1) return hpt_add_instance(c, geomId, matrices);
  if (matrixNumber != 2) { c->fail(HPT_ERR_UNSUPPORTED, "AddInstanceMotion: two key matrices are supported (what LoadSceneInstances passes)"); return 0xFFFFFFFFu; }
  Inst in; in.geomId = geomId; std::memcpy(in.m, matrices, 64); std::memcpy(in.m1, matrices + 16, 64); in.motion = true;
  c->insts.push_back(in);
  c->accelCommitted = false; c->flatRefittable = false;
  return (uint32_t)(c->insts.size() - 1);
}
catch (...) { (void)hptGuard(c, "hpt_add_instance_motion"); return 0xFFFFFFFFu; }

extern "C" int hpt_update_instance(hpt_ctx* c, uint32_t instId, const float m[16])
try {
  if (!c || !m) return HPT_ERR_ARG;
  if (instId >= c->insts.size()) return HPT_OK;          // the reference silently ignores it (EmbreeRT.cpp:302-303)
  std::memcpy(c->insts[instId].m, m, 64);
  c->accelCommitted = false;
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_update_instance"); }

// Expected inner-node visits of a random ray through the root box (the surface-area heuristic the builder minimises): 1 for the root plus
// area(child) / area(root) for every child that is an inner node. The number that tells a light scene from a heavy one better than the
// triangle count does (see useWavefront).
static float sah_node_visits(const Bvh2& t)
{
  const float a0 = t.bounds.halfArea();
  if (t.nodes.empty() || !(a0 > 0.0f)) return 0.0f;
  double v = 1.0;
  for (const BvhNode& n : t.nodes) for (int k = 0; k < 2; k++) {
    const uint ref = k ? n.ref1 : n.ref0;
    if (ref == REF_NONE || (ref & REF_LEAF)) continue;
    const float* q = n.q + 6 * k;
    const float dx = q[1] - q[0], dy = q[3] - q[2], dz = q[5] - q[4];
    if (dx >= 0.0f) v += double(dx * dy + dy * dz + dz * dx) / a0;
  }
  return (float)v;
}

// world -> object rows of an instance matrix (column-major 4x4, affine): cofactor inverse in double, rounded once
static void inverse_rows(const float* m, float row0[4], float row1[4], float row2[4])
{
  const double a00 = m[0], a01 = m[4], a02 = m[8],  tx = m[12];
  const double a10 = m[1], a11 = m[5], a12 = m[9],  ty = m[13];
  const double a20 = m[2], a21 = m[6], a22 = m[10], tz = m[14];
  const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  const double det = a00 * c00 + a01 * c01 + a02 * c02;
  const double id = 1.0 / det;
  const double i00 = c00 * id, i01 = (a02 * a21 - a01 * a22) * id, i02 = (a01 * a12 - a02 * a11) * id;
  const double i10 = c01 * id, i11 = (a00 * a22 - a02 * a20) * id, i12 = (a02 * a10 - a00 * a12) * id;
  const double i20 = c02 * id, i21 = (a01 * a20 - a00 * a21) * id, i22 = (a00 * a11 - a01 * a10) * id;
  row0[0] = (float)i00; row0[1] = (float)i01; row0[2] = (float)i02; row0[3] = (float)(-(i00 * tx + i01 * ty + i02 * tz));
  row1[0] = (float)i10; row1[1] = (float)i11; row1[2] = (float)i12; row1[3] = (float)(-(i10 * tx + i11 * ty + i12 * tz));
  row2[0] = (float)i20; row2[1] = (float)i21; row2[2] = (float)i22; row2[3] = (float)(-(i20 * tx + i21 * ty + i22 * tz));
}

static int ceil_log2(size_t v) { int l = 0; while ((size_t(1) << l) < v) l++; return l; }

// object -> world rows (3 x 4) of every instance, as refitTriBoxesKernel reads them
static void inst_o2w_rows(const std::vector<Inst>& insts, std::vector<float>& out)
{
  out.assign(12 * std::max<size_t>(insts.size(), 1), 0.0f);
  for (size_t i = 0; i < insts.size(); i++) {
    const float* m = insts[i].m; float* o = &out[12 * i];
    for (int r = 0; r < 3; r++) { o[4 * r + 0] = m[r]; o[4 * r + 1] = m[4 + r]; o[4 * r + 2] = m[8 + r]; o[4 * r + 3] = m[12 + r]; }
  }
}

// UpdateInstance / UpdateGeom_Triangles3f + CommitScene on a committed single-level scene (CrossRT.h:85-86, 134, 110): the tree's topology stays,
// its boxes are recomputed on the device (refitTriBoxesKernel + one refitLevelKernel per level, deepest first). The host only inverts the
// instance matrices again and, for meshes whose vertices moved, rewrites their triangle records.
// host threads for CommitScene (hpt_set_option("build_threads", n); default: the cores this process may use, at most 16)
static int buildThreads(const hpt_ctx* c)
{
  if (c->buildThreads > 0) return c->buildThreads;
  unsigned n = std::thread::hardware_concurrency();
  cpu_set_t set; CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int k = CPU_COUNT(&set); if (k > 0) n = std::min<unsigned>(n ? n : (unsigned)k, (unsigned)k); }
  return (int)std::min<unsigned>(std::max<unsigned>(n, 1u), 16u);
}

// contiguous chunks of [0, n) on `threads` host threads (fn(begin, end)); small ranges stay on the caller's thread
template <class F> static void parallelRanges(size_t n, int threads, F fn)
{
  if (threads <= 1 || n < 32768) { fn((size_t)0, n); return; }
  const size_t per = std::max<size_t>(n / (8 * (size_t)threads), 4096);       // chunks are handed out dynamically: a thread that could not be started costs nothing
  std::atomic<size_t> next(0);
  auto work = [&]() { for (size_t b = next.fetch_add(per); b < n; b = next.fetch_add(per)) fn(b, std::min(n, b + per)); };
  std::vector<std::thread> pool;
  try { for (int t = 1; t < threads; t++) pool.emplace_back(work); } catch (...) {}   // (no thread to be had: the caller's thread does the rest)
  work();
  for (std::thread& t : pool) t.join();
}

// ---- CommitScene on the device: hpt_lbvh.hip builds the single-level BVH2, its triangle records and the 4-wide compressed tree -------------
static int commit_flat_on_device(hpt_ctx* c, size_t instTris, double tBuild0)
{
  if (!c->lbvh) c->lbvh = lbvhCreate();
  std::string err;
  if (c->lbvhGeomVersion != c->geomVersion) {                 // the meshes' positions and indices: uploaded once, kept until a mesh changes
    std::vector<const float*> pos; std::vector<size_t> posN; std::vector<const uint*> idx; std::vector<size_t> idxN;
    for (const Geom& g : c->geoms) { pos.push_back(g.pos.data()); posN.push_back(g.pos.size()); idx.push_back(g.idx.data()); idxN.push_back(g.idx.size()); }
    if (!lbvhUploadGeometry(c->lbvh, pos, posN, idx, idxN, c->lbvhPosOff, c->lbvhIdxOff, err)) return c->fail(HPT_ERR_HIP, "CommitScene (device build): " + err);
    c->lbvhGeomVersion = c->geomVersion;
  }
  const size_t ni = c->insts.size();
  std::vector<LbvhInstance> li(ni);
  std::vector<BvhInst> dinst(std::max<size_t>(ni, 1));
  for (size_t i = 0; i < ni; i++) {
    const float* m = c->insts[i].m;
    for (int r = 0; r < 3; r++) { li[i].objectToWorld[4 * r + 0] = m[r]; li[i].objectToWorld[4 * r + 1] = m[4 + r]; li[i].objectToWorld[4 * r + 2] = m[8 + r]; li[i].objectToWorld[4 * r + 3] = m[12 + r]; }
    const uint g = c->insts[i].geomId;
    li[i].triCount = (uint)(c->geoms[g].idx.size() / 3); li[i].posOffset = c->lbvhPosOff[g]; li[i].idxOffset = c->lbvhIdxOff[g];
    inverse_rows(m, dinst[i].row0, dinst[i].row1, dinst[i].row2);
    dinst[i].root = REF_NONE; dinst[i].geomId = g; dinst[i].pad0 = 0u; dinst[i].pad1 = 0;
  }
  const uint n = (uint)instTris;
  HIPCHK(c, c->dNodes.alloc(std::max<size_t>(n, 1)));
  HIPCHK(c, c->dTris.alloc(std::max<size_t>(n, 1)));
  const bool wantWide = true;
  HIPCHK(c, c->dNodes4.alloc(std::max<size_t>(n, 1)));
  HIPCHK(c, c->dInsts.upload(dinst.data(), dinst.size()));
  {                                                          // (both layouts' kernels read the motion rows' pointer; none of these instances moves)
    std::vector<float> mo(24 * std::max<size_t>(ni, 1), 0.0f);
    for (size_t i = 0; i < ni; i++) for (int key = 0; key < 2; key++) for (int k = 0; k < 12; k++) mo[24 * i + 12 * (size_t)key + k] = li[i].objectToWorld[k];
    HIPCHK(c, c->dInstMotion.upload(mo.data(), mo.size()));
    c->S.instMotion = c->dInstMotion.p;
  }
  const double tDev0 = now_ms();
  LbvhResult r;
  if (!lbvhBuild(c->lbvh, li.data(), (uint)ni, n, c->dNodes.p, c->dTris.p, c->dNodes4.p, (uint)c->dNodes4.n, wantWide, r, err)) return c->fail(HPT_ERR_HIP, "CommitScene (device build): " + err);
  const double tDev1 = now_ms();
  if (r.depth + 1u > MAX_STACK) return HPT_ERR_STATE;         // a degenerate input (the Morton curve cannot separate it): the host's depth-capped build takes over
  c->instTris = instTris; c->anyMotion = false;
  c->nodes4Count = 0; c->stackNeeded4 = 0; c->S.nodes4 = nullptr; c->S.root4 = REF_NONE; c->S.statsWide = 0; c->S.megaWide = 0;
  c->sahVisits = r.sahVisits;
  if (r.nodes4Count != 0u && (r.rootRef & REF_LEAF) == 0u) {
    c->nodes4Count = r.nodes4Count; c->stackNeeded4 = 3u * r.depth4 + 1u;
    c->S.nodes4 = c->dNodes4.p; c->S.root4 = 0u;
    c->S.megaWide = (c->wideEnabled && c->sahVisits >= HEAVY_SAH_VISITS) ? 1u : 0u;
  }
  c->flatTris.clear(); c->flatRefittable = false;             // (an update is answered by another device build, not by a refit)
  c->S.nodes = c->dNodes.p; c->S.tris = c->dTris.p; c->S.insts = c->dInsts.p;
  c->S.rootRef = r.rootRef; c->S.numInsts = (uint)ni; c->S.flatMode = 1; c->S.sweep = 0; c->S.sweepInsts = nullptr; c->S.sweepTris = nullptr;
  c->S.nodeMin = c->nodeMinOverride >= 0 ? (uint)c->nodeMinOverride : (c->sahVisits >= HEAVY_SAH_VISITS ? 16u : 0u);
  c->S.nodeMin4 = c->nodeMinOverride >= 0 ? (uint)c->nodeMinOverride : (c->sahVisits >= HEAVY_SAH_VISITS ? 32u : 0u);
  c->stackNeeded = r.depth + 1u;
  if (std::getenv("HPT_DEBUG_ACCEL")) std::fprintf(stderr, "[hydra_hip] device-built single-level BVH: %zu triangles, depth %u, sah visits %.2f; 4-wide: %u nodes, depth %u; %.2f ms of kernels\n",
                                                  instTris, r.depth, c->sahVisits, r.nodes4Count, r.depth4, tDev1 - tDev0);
  c->tCommit[0] = float(tDev0 - tBuild0); c->tCommit[1] = 0.0f; c->tCommit[2] = float(tDev1 - tDev0); c->tCommit[3] = 2.0f;   // host preparation + uploads, -, device build, "built on the device"
  c->accelCommitted = true; c->shadeTrisDirty = true;
  return HPT_OK;
}

static int refit_flat(hpt_ctx* c)
{
  const double t0 = now_ms();
  const size_t ni = c->insts.size();
  std::vector<BvhInst> dinst(std::max<size_t>(ni, 1));
  for (size_t i = 0; i < ni; i++) {
    inverse_rows(c->insts[i].m, dinst[i].row0, dinst[i].row1, dinst[i].row2);
    dinst[i].root = REF_NONE; dinst[i].geomId = c->insts[i].geomId; dinst[i].pad0 = dinst[i].pad1 = 0;
  }
  std::vector<float> o2w; inst_o2w_rows(c->insts, o2w);
  if (!c->dirtyGeoms.empty()) {                                 // moved vertices: the object-space records (v0, e1, e2) of those meshes' triangles
    std::vector<char> dirty(c->geoms.size(), 0);
    for (uint g : c->dirtyGeoms) if (g < dirty.size()) dirty[g] = 1;
    for (BvhTri& t : c->flatTris) {
      const uint gi = c->insts[t.instId].geomId;
      if (!dirty[gi]) continue;
      const Geom& g = c->geoms[gi];
      const float* A = &g.pos[3 * g.idx[3 * t.primId + 0]]; const float* B = &g.pos[3 * g.idx[3 * t.primId + 1]]; const float* C = &g.pos[3 * g.idx[3 * t.primId + 2]];
      for (int a = 0; a < 3; a++) { t.v0[a] = A[a]; t.e1[a] = B[a] - A[a]; t.e2[a] = C[a] - A[a]; }
    }
    for (uint g : c->dirtyGeoms) if (g < c->geoms.size()) c->geoms[g].dirty = true;   // a later two-level build must redo these BLAS
  }
  const double t1 = now_ms();
  HIPCHK(c, c->dInsts.upload(dinst.data(), dinst.size()));
  HIPCHK(c, c->dInstO2W.upload(o2w.data(), o2w.size()));
  if (!c->dirtyGeoms.empty()) HIPCHK(c, c->dTris.upload(c->flatTris.data(), c->flatTris.size()));
  c->dirtyGeoms.clear();
  const double t2 = now_ms();
  const uint nt = (uint)c->instTris;
  if (nt) refitTriBoxesKernel<<<dim3((nt + 255u) / 256u), dim3(256), 0, 0>>>(c->dTris.p, c->dInstO2W.p, nt, c->dTriBox.p);
  for (size_t l = c->levelOffsets.size() - 1; l-- > 0;) {
    const uint cnt = c->levelOffsets[l + 1] - c->levelOffsets[l];
    if (cnt) refitLevelKernel<<<dim3((cnt + 255u) / 256u), dim3(256), 0, 0>>>(c->dNodes.p, c->dLevelNodes.p + c->levelOffsets[l], cnt, c->dTriBox.p, c->dNodeBounds.p);
  }
  if (c->nodes4Count) refitNodes4Kernel<<<dim3((c->nodes4Count + 255u) / 256u), dim3(256), 0, 0>>>(c->dNodes4.p, c->dNodes4Src.p, c->dNodes.p, c->nodes4Count);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipDeviceSynchronize());
  c->S.insts = c->dInsts.p;
  c->tCommit[0] = float(t1 - t0); c->tCommit[1] = float(t2 - t1); c->tCommit[2] = float(now_ms() - t2); c->tCommit[3] = 1.0f;
  c->accelCommitted = true; c->shadeTrisDirty = true;
  return HPT_OK;
}

static int commit_flat_on_device(hpt_ctx* c, size_t instTris, double tBuild0);

extern "C" int hpt_commit_scene(hpt_ctx* c, uint32_t options)
try {
  if (!c) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (c->flatRefittable && c->refitEnabled && c->S.flatMode == 1u) return refit_flat(c);
  const double tBuild0 = now_ms();
  c->dirtyGeoms.clear();
  {
    // CommitScene(BUILD_LOW | BUILD_MEDIUM) (CrossRT.h:8-14) or hpt_set_option("device_build", 1): the single-level tree is built by kernels
    // (hpt_lbvh.hip: a few ms for 1 M triangles against a quarter of a second on the host; a linear BVH, coarser than the SAH tree)
    size_t nTris = 0; bool moving = false;
    for (const Inst& in : c->insts) { nTris += c->geoms[in.geomId].idx.size() / 3; moving = moving || in.motion; }
    const bool wantFlat = nTris <= FLAT_TRI_BUDGET && nTris < (size_t(1) << 28) && !moving &&
                          (c->accelLayout == 2 || (c->accelLayout == 0 && (nTris >= FLAT_AUTO_TRIS || c->insts.size() >= MANY_INSTANCES)));
    const bool fast = (options & 3u) != 0u && (options & 4u) == 0u;
    if (wantFlat && nTris > 0 && (c->deviceBuild == 1 || (c->deviceBuild < 0 && fast))) {
      const int rc = commit_flat_on_device(c, nTris, tBuild0);
      if (rc != HPT_ERR_STATE) return rc;                    // (HPT_ERR_STATE: the tree came out deeper than the traversal stack - the host's depth-capped build takes over)
    }
  }
  // ---- bottom level: one BVH2 per mesh over object-space triangles ----
  uint maxBlasDepth = 0;
  const int nThreads = buildThreads(c);
  // meshes are independent: the dirty ones are built by a pool of host threads (a mesh's own build stays sequential; a single big mesh - and
  // the single-level tree below - is split into subtrees instead, Bvh2Builder::build)
  std::vector<size_t> todo;
  size_t biggest = 0;
  for (size_t gi = 0; gi < c->geoms.size(); gi++) if (c->geoms[gi].dirty) { todo.push_back(gi); biggest = std::max(biggest, c->geoms[gi].idx.size() / 3); }
  const bool meshParallel = nThreads > 1 && todo.size() >= 4;
  auto buildMesh = [&](Geom& g, int threadsInside) {
    const size_t nt = g.idx.size() / 3;
    std::vector<Aabb> boxes(nt);
    for (size_t t = 0; t < nt; t++) {
      boxes[t].reset();
      for (int k = 0; k < 3; k++) boxes[t].grow(&g.pos[3 * g.idx[3 * t + k]]);
    }
    const int depthCap = std::max(24, ceil_log2((nt + BVH_LEAF_MAX - 1) / BVH_LEAF_MAX) + 2);
    g.bvh = Bvh2Builder::build(boxes, BVH_LEAF_MAX, depthCap, false, threadsInside);
    g.tris.resize(nt);
    for (size_t i = 0; i < nt; i++) {
      const uint p = g.bvh.order[i];
      const float* A = &g.pos[3 * g.idx[3 * p + 0]]; const float* B = &g.pos[3 * g.idx[3 * p + 1]]; const float* C = &g.pos[3 * g.idx[3 * p + 2]];
      BvhTri& t = g.tris[i];
      for (int a = 0; a < 3; a++) { t.v0[a] = A[a]; t.e1[a] = B[a] - A[a]; t.e2[a] = C[a] - A[a]; }
      t.primId = p; t.instId = 0; t.pad1 = 0;
    }
    g.dirty = false;
  };
  if (meshParallel) {
    std::atomic<size_t> next(0);
    auto work = [&]() { for (size_t k = next.fetch_add(1); k < todo.size(); k = next.fetch_add(1)) buildMesh(c->geoms[todo[k]], 1); };
    std::vector<std::thread> pool;
    try { for (int t = 1; t < nThreads; t++) pool.emplace_back(work); } catch (...) {}
    work();
    for (std::thread& t : pool) t.join();
  } else for (size_t gi : todo) buildMesh(c->geoms[gi], nThreads);
  for (const Geom& g : c->geoms) maxBlasDepth = std::max(maxBlasDepth, g.bvh.depth);
  // ---- single-level layout: one BVH2 over all instanced triangles, world-space boxes, object-space triangle records ----
  size_t instTris = 0;
  for (const Inst& in : c->insts) instTris += c->geoms[in.geomId].idx.size() / 3;
  c->instTris = instTris;
  // Automatic choice, measured: heavy static scenes (wavefront schedule) gain 8 % from the single-level layout (1M triangles: 191 -> 207
  // Mpaths/s; no instance enter / leave trips, triangle-loop lane utilisation 0.21 -> 0.38); the Cornell-box class on the megakernel loses
  // 8 % to it (looser world-space boxes around rotated instances, per-triangle ray transform) and keeps the two-level TLAS/BLAS layout.
  // Scenes of many small instances (the own fixtures: 6 ... 16 transformed spheres on a floor) also gain from it on the megakernel:
  // typed_materials 1031 -> 1127, legacy_materials 1574 -> 1752, env_map 1526 -> 1670 Mpaths/s (profiles/ab_layout.sh), the instance
  // enter / leave trips outweigh the looser boxes once a ray meets several overlapping BLAS boxes.
  // ... and so does the 8 202-triangle test_228 with its 3 instances (megakernel: two-level 617, single-level 667 Mpaths/s).
  // Scenes with moving instances keep the two-level layout unless told otherwise: there the ray is taken to the instance's space at the
  // path's time once per instance visit, under the single-level layout once per triangle record (legacy_materials, one moving sphere of
  // ten instances: two-level 1797, single-level 1065 Mpaths/s - profiles/ab_layout_full.sh).
  c->anyMotion = false;
  for (const Inst& in : c->insts) c->anyMotion = c->anyMotion || in.motion;
  const bool autoFlat = !c->anyMotion && (instTris >= FLAT_AUTO_TRIS || c->insts.size() >= MANY_INSTANCES);
  const bool flat = instTris <= FLAT_TRI_BUDGET && (c->accelLayout == 2 || (c->accelLayout == 0 && autoFlat));
  {                                                          // object->world rows (3x4) at both keys for the moving instances (both layouts read them)
    const size_t nim = c->insts.size();
    std::vector<float> mo(24 * std::max<size_t>(nim, 1), 0.0f);
    for (size_t i = 0; i < nim; i++) for (int key = 0; key < 2; key++) {
      const float* m = (key && c->insts[i].motion) ? c->insts[i].m1 : c->insts[i].m;
      float* o = &mo[24 * i + 12 * (size_t)key];
      for (int r = 0; r < 3; r++) { o[4 * r + 0] = m[r]; o[4 * r + 1] = m[4 + r]; o[4 * r + 2] = m[8 + r]; o[4 * r + 3] = m[12 + r]; }
    }
    HIPCHK(c, c->dInstMotion.upload(mo.data(), mo.size()));
    c->S.instMotion = c->dInstMotion.p;
  }
  if (flat) {
    const size_t ni = c->insts.size();
    std::vector<Aabb> boxes(instTris);
    std::vector<uint> triInst(instTris), triPrim(instTris);
    {
      size_t k = 0;
      for (size_t i = 0; i < ni; i++) { const size_t nt = c->geoms[c->insts[i].geomId].idx.size() / 3; for (size_t t = 0; t < nt; t++, k++) { triInst[k] = (uint)i; triPrim[k] = (uint)t; } }
    }
    parallelRanges(instTris, nThreads, [&](size_t kb, size_t ke) {
      for (size_t k = kb; k < ke; k++) {
        const size_t i = triInst[k], t = triPrim[k];
        const Geom& g = c->geoms[c->insts[i].geomId];
        Aabb b; b.reset();
        // a moving instance: every point travels on the segment between its two key positions, so the box over both keys bounds the triangle at any time
        for (int key = 0; key < (c->insts[i].motion ? 2 : 1); key++) {
          const float* m = key ? c->insts[i].m1 : c->insts[i].m;
          for (int kk = 0; kk < 3; kk++) {
            const float* p = &g.pos[3 * g.idx[3 * t + kk]];
            const float q[3] = { m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12], m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13], m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14] };
            b.grow(q);
          }
        }
        b.pad();                                            // covers the rounding of the world-space vertex positions too
        boxes[k] = b;
      }
    });
    const int depthCap = std::max(24, ceil_log2((instTris + BVH_LEAF_MAX - 1) / BVH_LEAF_MAX) + 2);
    Bvh2 tree = Bvh2Builder::build(boxes, BVH_LEAF_MAX, depthCap, false, nThreads);
    std::vector<BvhTri> tris(std::max<size_t>(instTris, 1));
    parallelRanges(instTris, nThreads, [&](size_t kb, size_t ke) {
      for (size_t k = kb; k < ke; k++) {
        const uint src = tree.order[k];
        const uint i = triInst[src], p = triPrim[src];
        const Geom& g = c->geoms[c->insts[i].geomId];
        const float* A = &g.pos[3 * g.idx[3 * p + 0]]; const float* B = &g.pos[3 * g.idx[3 * p + 1]]; const float* C = &g.pos[3 * g.idx[3 * p + 2]];
        BvhTri& t = tris[k];
        for (int a = 0; a < 3; a++) { t.v0[a] = A[a]; t.e1[a] = B[a] - A[a]; t.e2[a] = C[a] - A[a]; }   // same object-space record as the two-level path
        t.primId = p; t.instId = i; t.pad1 = 0;
      }
    });
    std::vector<BvhNode> nodes(tree.nodes);
    if (nodes.empty()) nodes.push_back(BvhNode());
    if (tris.size() >= (size_t(1) << 28)) return c->fail(HPT_ERR_UNSUPPORTED, "CommitScene: scene too large for 28-bit triangle references");
    std::vector<BvhInst> dinst(std::max<size_t>(ni, 1));
    for (size_t i = 0; i < ni; i++) {
      inverse_rows(c->insts[i].m, dinst[i].row0, dinst[i].row1, dinst[i].row2);
      dinst[i].root = REF_NONE; dinst[i].geomId = c->insts[i].geomId; dinst[i].pad0 = c->insts[i].motion ? 1u : 0u; dinst[i].pad1 = 0;
    }
    const double tUp0 = now_ms();
    HIPCHK(c, c->dNodes.upload(nodes.data(), nodes.size()));
    HIPCHK(c, c->dTris.upload(tris.data(), tris.size()));
    HIPCHK(c, c->dInsts.upload(dinst.data(), dinst.size()));
    // ---- the same tree as 4-wide compressed nodes (BvhNode4, hpt_types.h) for the heavy-scene trace kernel ----
    // Collapse: a node adopts its grandchildren, largest surface first, until it has four children or only leaves are left. Child boxes are
    // the (padded) BVH2 child boxes, quantised outwards in the node's frame; `src` remembers where each box lives so that a refit can requantise.
    c->nodes4Count = 0; c->stackNeeded4 = 0; c->S.nodes4 = nullptr; c->S.root4 = REF_NONE; c->S.statsWide = 0; c->S.megaWide = 0;
    double collapseMs = 0.0;
    if (tree.rootRef != REF_NONE && !(tree.rootRef & REF_LEAF) && !c->anyMotion) {
      const double tC0 = now_ms();
      std::vector<BvhNode4> n4; std::vector<uint> src4;
      uint depth4 = 1;
      collapseToWide(tree, n4, src4, depth4);
      collapseMs = now_ms() - tC0;
      if (n4.size() < (size_t(1) << 31)) {
        HIPCHK(c, c->dNodes4.upload(n4.data(), n4.size()));
        HIPCHK(c, c->dNodes4Src.upload(src4.data(), src4.size()));
        c->nodes4Count = (uint)n4.size(); c->stackNeeded4 = 3u * depth4 + 1u;       // a visit leaves at most three children waiting
        c->S.nodes4 = c->dNodes4.p; c->S.root4 = 0u;
        c->S.megaWide = (c->wideEnabled && sah_node_visits(tree) >= HEAVY_SAH_VISITS) ? 1u : 0u;   // heavy scenes rendered by the megakernel (calls below the wavefront's pixel threshold) walk it too
        if (std::getenv("HPT_DEBUG_ACCEL")) std::fprintf(stderr, "[hydra_hip] 4-wide tree: %zu nodes (BVH2: %zu), depth %u, stack bound %u\n", n4.size(), tree.nodes.size(), depth4, c->stackNeeded4);
      }
    }
    {                                                          // what refit_flat needs: the nodes grouped by level, scratch for the boxes
      std::vector<uint> level(tree.nodes.size(), 0u), order; order.reserve(tree.nodes.size());
      uint maxLevel = 0;
      if (tree.rootRef != REF_NONE && !(tree.rootRef & REF_LEAF)) {
        std::vector<uint> stack(1, tree.rootRef);
        while (!stack.empty()) {
          const uint id = stack.back(); stack.pop_back();
          for (uint ref : { tree.nodes[id].ref0, tree.nodes[id].ref1 })
            if (ref != REF_NONE && !(ref & REF_LEAF)) { level[ref] = level[id] + 1u; maxLevel = std::max(maxLevel, level[ref]); stack.push_back(ref); }
        }
      }
      c->levelOffsets.assign(tree.nodes.empty() ? 1 : maxLevel + 2, 0u);
      for (size_t i = 0; i < tree.nodes.size(); i++) c->levelOffsets[level[i] + 1]++;
      for (size_t l = 1; l < c->levelOffsets.size(); l++) c->levelOffsets[l] += c->levelOffsets[l - 1];
      std::vector<uint> fill(c->levelOffsets.begin(), c->levelOffsets.end() - 1), ids(std::max<size_t>(tree.nodes.size(), 1), 0u);
      for (size_t i = 0; i < tree.nodes.size(); i++) ids[fill[level[i]]++] = (uint)i;
      HIPCHK(c, c->dLevelNodes.upload(ids.data(), ids.size()));
      HIPCHK(c, c->dTriBox.alloc(6 * std::max<size_t>(instTris, 1)));
      HIPCHK(c, c->dNodeBounds.alloc(6 * std::max<size_t>(tree.nodes.size(), 1)));
      c->flatTris.swap(tris);
      c->flatRefittable = tree.rootRef != REF_NONE && !(tree.rootRef & REF_LEAF) && instTris > 0 && !c->anyMotion;   // (the refit kernels know one key only)
    }
    c->tCommit[0] = float(tUp0 - tBuild0 + collapseMs); c->tCommit[1] = float(now_ms() - tUp0 - collapseMs); c->tCommit[2] = 0.0f; c->tCommit[3] = 0.0f;   // the collapse is host build time
    c->S.nodes = c->dNodes.p; c->S.tris = c->dTris.p; c->S.insts = c->dInsts.p;
    c->S.rootRef = tree.rootRef; c->S.numInsts = (uint)ni; c->S.flatMode = 1; c->S.sweep = 0; c->S.sweepInsts = nullptr; c->S.sweepTris = nullptr;
    c->sahVisits = sah_node_visits(tree);
    c->S.nodeMin = c->nodeMinOverride >= 0 ? (uint)c->nodeMinOverride : (c->sahVisits >= HEAVY_SAH_VISITS ? 16u : 0u);
    // the 4-wide walk votes earlier: 1M triangles, node_min 0 / 8 / 16 / 24 / 32 / 48 -> 176 / 254 / 273 / 286 / 288 / 269 Mpaths/s (profiles/sweep_wide.sh)
    c->S.nodeMin4 = c->nodeMinOverride >= 0 ? (uint)c->nodeMinOverride : (c->sahVisits >= HEAVY_SAH_VISITS ? 32u : 0u);
    c->stackNeeded = tree.depth + 1u;
    if (std::getenv("HPT_DEBUG_ACCEL")) std::fprintf(stderr, "[hydra_hip] single-level BVH: %zu triangles, %zu nodes, depth %u, stack %u (LDS part %d), sah visits %.2f, nodeMin %u\n", instTris, tree.nodes.size(), tree.depth, c->stackNeeded, LDS_STACK, c->sahVisits, c->S.nodeMin);
    if (c->stackNeeded > MAX_STACK) return c->fail(HPT_ERR_UNSUPPORTED, "CommitScene: BVH deeper than the 64-entry traversal stack");
    c->accelCommitted = true; c->shadeTrisDirty = true;
    return HPT_OK;
  }
  // ---- top level over the instances' world boxes ----
  const size_t ni = c->insts.size();
  std::vector<Aabb> ib; std::vector<uint> liveInst;
  for (size_t i = 0; i < ni; i++) {
    const Geom& g = c->geoms[c->insts[i].geomId];
    if (g.bvh.rootRef == REF_NONE) continue;               // empty mesh: never enters the TLAS
    Aabb w; w.reset();
    // a moving instance: every point travels on a segment between its two key positions, so the union of the two key boxes bounds it
    for (int key = 0; key < (c->insts[i].motion ? 2 : 1); key++) {
      const float* m = key ? c->insts[i].m1 : c->insts[i].m;
      for (int k = 0; k < 8; k++) {
        const float p[3] = { (k & 1) ? g.bvh.bounds.hi[0] : g.bvh.bounds.lo[0], (k & 2) ? g.bvh.bounds.hi[1] : g.bvh.bounds.lo[1], (k & 4) ? g.bvh.bounds.hi[2] : g.bvh.bounds.lo[2] };
        const float q[3] = { m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12], m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13], m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14] };
        w.grow(q);
      }
    }
    w.pad();
    ib.push_back(w); liveInst.push_back((uint)i);
  }
  Bvh2 tlas = Bvh2Builder::build(ib, 1, std::max(16, ceil_log2(ib.size() + 1) + 2), true);
  {                                                          // TLAS visits + the chance of entering each instance x its BLAS visits
    double v = sah_node_visits(tlas);
    const float a0 = tlas.bounds.halfArea();
    std::vector<float> blasVisits(c->geoms.size(), -1.0f);
    for (size_t k = 0; k < ib.size() && a0 > 0.0f; k++) {
      const uint g = c->insts[liveInst[k]].geomId;
      if (blasVisits[g] < 0.0f) blasVisits[g] = sah_node_visits(c->geoms[g].bvh);
      v += double(ib[k].halfArea()) / a0 * blasVisits[g];
    }
    c->sahVisits = (float)v;
    c->S.nodeMin = c->nodeMinOverride >= 0 ? (uint)c->nodeMinOverride : (c->sahVisits >= HEAVY_SAH_VISITS ? 16u : 0u);
  }
  // instance leaves refer to positions in `ib`; map them back to real instance ids
  auto fixInst = [&](uint ref) -> uint {
    if (ref != REF_NONE && (ref & REF_LEAF) && ((ref >> 28) & 7u) == 0u) return REF_LEAF | liveInst[ref & 0x0FFFFFFFu];
    return ref;
  };
  // ---- flatten: [TLAS nodes][BLAS 0 nodes][BLAS 1 nodes]...  /  [tris 0][tris 1]... ----
  std::vector<BvhNode> nodes(tlas.nodes);
  for (BvhNode& n : nodes) { n.ref0 = fixInst(n.ref0); n.ref1 = fixInst(n.ref1); }
  const uint rootRef = fixInst(tlas.rootRef);
  std::vector<BvhTri> tris;
  std::vector<uint> geomRoot(c->geoms.size(), REF_NONE);
  for (size_t gi = 0; gi < c->geoms.size(); gi++) {
    const Geom& g = c->geoms[gi];
    const uint nodeBase = (uint)nodes.size(), triBase = (uint)tris.size();
    for (BvhNode n : g.bvh.nodes) { n.ref0 = patchRef(n.ref0, nodeBase, triBase); n.ref1 = patchRef(n.ref1, nodeBase, triBase); nodes.push_back(n); }
    tris.insert(tris.end(), g.tris.begin(), g.tris.end());
    geomRoot[gi] = patchRef(g.bvh.rootRef, nodeBase, triBase);
  }
  if (tris.size() >= (size_t(1) << 28) || nodes.size() >= (size_t(1) << 31)) return c->fail(HPT_ERR_UNSUPPORTED, "CommitScene: scene too large for 28-bit triangle references");
  std::vector<BvhInst> dinst(std::max<size_t>(ni, 1));    // (never empty: an empty scene still hands valid pointers to the kernels)
  for (size_t i = 0; i < ni; i++) {
    inverse_rows(c->insts[i].m, dinst[i].row0, dinst[i].row1, dinst[i].row2);
    dinst[i].root = geomRoot[c->insts[i].geomId]; dinst[i].geomId = c->insts[i].geomId; dinst[i].pad0 = c->insts[i].motion ? 1u : 0u; dinst[i].pad1 = 0;
  }
  // the triangle sweep keeps its own copy of the instance records: {rows, first triangle record, geomId, 0, number of record pairs}
  const bool sweep = !c->anyMotion && ni >= 1 && (c->accelLayout == 3 || (c->accelLayout == 0 && instTris <= SWEEP_MAX_TRIS && ni <= SWEEP_MAX_INSTS));
  if (c->accelLayout == 3 && c->anyMotion) return c->fail(HPT_ERR_UNSUPPORTED, "CommitScene: the triangle sweep does not hold moving instances (two-level or single-level layout)");
  if (sweep) {
    std::vector<uint> geomTriBase(c->geoms.size(), 0u);
    std::vector<BvhTri> st;                                     // every mesh's records in primitive order
    for (size_t gi = 0; gi < c->geoms.size(); gi++) {
      const Geom& g = c->geoms[gi];
      geomTriBase[gi] = (uint)st.size();
      const size_t base = st.size();
      st.resize(base + g.tris.size());
      for (const BvhTri& t : g.tris) st[base + t.primId] = t;   // g.tris is a permutation of the mesh's primitives (BVH leaf order)
      if (st.size() & 1u) { BvhTri z; std::memset(&z, 0, sizeof(z)); z.primId = 0xFFFFFFFFu; st.push_back(z); }   // pairs: an all-zero triangle has det == 0 and is never hit
    }
    if (st.empty()) st.push_back(BvhTri());
    HIPCHK(c, c->dSweepTris.upload(st.data(), st.size()));
    std::vector<BvhInst> sw(dinst);
    for (size_t i = 0; i < ni; i++) { const uint g = c->insts[i].geomId; sw[i].root = geomTriBase[g]; sw[i].pad0 = 0; sw[i].pad1 = (uint)(c->geoms[g].tris.size() + 1) / 2u; }
    HIPCHK(c, c->dSweepInsts.upload(sw.data(), sw.size()));
    // the instances' padded world boxes (over their triangles' world-space vertices): traceSweep's wave-uniform skip
    std::vector<float> sb(8 * std::max<size_t>(ni, 1), 0.0f);
    for (size_t i = 0; i < ni; i++) {
      const Geom& g = c->geoms[c->insts[i].geomId];
      const float* m = c->insts[i].m;
      Aabb b; b.reset();
      for (size_t v = 0; v + 2 < g.pos.size(); v += 3) {
        const float* p = &g.pos[v];
        const float q[3] = { m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12], m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13], m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14] };
        b.grow(q);
      }
      if (g.pos.empty()) { for (int a = 0; a < 3; a++) { b.lo[a] = 1.0f; b.hi[a] = -1.0f; } } else b.pad();
      for (int a = 0; a < 3; a++) { sb[8 * i + a] = b.lo[a]; sb[8 * i + 4 + a] = b.hi[a]; }
    }
    HIPCHK(c, c->dSweepBoxes.upload(sb.data(), sb.size()));
  }
  c->S.sweep = sweep ? 1u : 0u; c->S.sweepInsts = sweep ? c->dSweepInsts.p : nullptr; c->S.sweepTris = sweep ? c->dSweepTris.p : nullptr; c->S.sweepBoxes = sweep ? (const float4*)c->dSweepBoxes.p : nullptr;
  if (nodes.empty()) nodes.push_back(BvhNode());           // keep the pointers valid
  if (tris.empty()) tris.push_back(BvhTri());
  HIPCHK(c, c->dNodes.upload(nodes.data(), nodes.size()));
  HIPCHK(c, c->dTris.upload(tris.data(), tris.size()));
  HIPCHK(c, c->dInsts.upload(dinst.data(), dinst.size()));
  c->S.nodes = c->dNodes.p; c->S.tris = c->dTris.p; c->S.insts = c->dInsts.p;
  c->S.rootRef = rootRef; c->S.numInsts = (uint)ni; c->S.flatMode = 0;
  c->nodes4Count = 0; c->stackNeeded4 = 0; c->S.nodes4 = nullptr; c->S.root4 = REF_NONE; c->S.statsWide = 0; c->S.megaWide = 0; c->S.nodeMin4 = 0;
  c->flatRefittable = false;
  c->tCommit[0] = float(now_ms() - tBuild0); c->tCommit[1] = 0.0f; c->tCommit[2] = 0.0f; c->tCommit[3] = 0.0f;
  c->stackNeeded = tlas.depth + 1u + maxBlasDepth + 1u;
  if (std::getenv("HPT_DEBUG_ACCEL")) std::fprintf(stderr, "[hydra_hip] two-level BVH: %zu instances, TLAS depth %u, deepest BLAS %u, stack %u (LDS part %d)\n", ni, tlas.depth, maxBlasDepth, c->stackNeeded, LDS_STACK);
  if (c->stackNeeded > MAX_STACK) return c->fail(HPT_ERR_UNSUPPORTED, "CommitScene: BVH deeper than the 64-entry traversal stack");
  c->accelCommitted = true; c->shadeTrisDirty = true;
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_commit_scene"); }

// HBM part of the traversal stacks for a grid of `lanes` lanes (only touched by lanes whose stack outgrows LDS_STACK)
// stack entries a megakernel traversal may need: with HPT_FLAT_WIDE the single-level walk uses the 4-wide tree when the scene has one
static uint megaStackNeeded(const hpt_ctx* c) { return ((HPT_FLAT_WIDE || c->S.megaWide != 0u) && c->S.flatMode != 0u && c->S.motion == 0u && c->nodes4Count != 0u) ? std::max(c->stackNeeded, c->stackNeeded4) : c->stackNeeded; }
static hipError_t ensureStackOverflow(hpt_ctx* c, size_t lanes)
{
  const uint need = std::max(c->stackNeeded, c->stackNeeded4);                  // either tree may be walked
  const size_t extra = need > (uint)LDS_STACK ? need - LDS_STACK : 1;
  return c->dStackOvf.alloc(extra * lanes);
}

static int ray_query(hpt_ctx* c, const float* posNear, const float* dirFar, uint32_t n, void* out, int any, float time = 0.0f)
{
  if (!c || !posNear || !dirFar || !out) return HPT_ERR_ARG;
  if (!c->accelCommitted) return c->fail(HPT_ERR_STATE, "RayQuery before CommitScene");
  if (n == 0) return HPT_OK;
  (void)hipSetDevice(c->device);
  DevBuf<float4> dp, dd; DevBuf<uint> dout;
  const size_t outWords = any ? n : (size_t)n * 8;
  HIPCHK(c, dp.upload((const float4*)posNear, n));
  HIPCHK(c, dd.upload((const float4*)dirFar, n));
  HIPCHK(c, dout.alloc(outWords));
  const uint blocks = (n + 255) / 256;
  HIPCHK(c, ensureStackOverflow(c, (size_t)blocks * 256));
  if (c->S.sweep)        rayQueryKernel<false, false, true><<<dim3(blocks), dim3(256), 0, 0>>>(c->S, dp.p, dd.p, n, dout.p, any, c->dStackOvf.p);
  else if (c->S.flatMode && c->anyMotion) rayQueryKernel<true, true><<<dim3(blocks), dim3(256), 0, 0>>>(c->S, dp.p, dd.p, n, dout.p, any, c->dStackOvf.p, time);
  else if (c->S.flatMode) rayQueryKernel<true><<<dim3(blocks), dim3(256), 0, 0>>>(c->S, dp.p, dd.p, n, dout.p, any, c->dStackOvf.p);
  else if (c->anyMotion) rayQueryKernel<false, true><<<dim3(blocks), dim3(256), 0, 0>>>(c->S, dp.p, dd.p, n, dout.p, any, c->dStackOvf.p, time);
  else                   rayQueryKernel<false><<<dim3(blocks), dim3(256), 0, 0>>>(c->S, dp.p, dd.p, n, dout.p, any, c->dStackOvf.p);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpy(out, dout.p, outWords * 4, hipMemcpyDeviceToHost));
  dp.release(); dd.release(); dout.release();
  return HPT_OK;
}

extern "C" int hpt_ray_query_nearest(hpt_ctx* c, const float* p, const float* d, uint32_t n, hpt_hit* out) try { return ray_query(c, p, d, n, out, 0); }
catch (...) { return hptGuard(c, "hpt_ray_query_nearest"); }
extern "C" int hpt_ray_query_any(hpt_ctx* c, const float* p, const float* d, uint32_t n, uint32_t* out) try { return ray_query(c, p, d, n, out, 1); }
catch (...) { return hptGuard(c, "hpt_ray_query_any"); }
extern "C" int hpt_ray_query_nearest_motion(hpt_ctx* c, const float* p, const float* d, uint32_t n, float time, hpt_hit* out) try { return ray_query(c, p, d, n, out, 0, time); }
catch (...) { return hptGuard(c, "hpt_ray_query_nearest_motion"); }
extern "C" int hpt_ray_query_any_motion(hpt_ctx* c, const float* p, const float* d, uint32_t n, float time, uint32_t* out) try { return ray_query(c, p, d, n, out, 1, time); }
catch (...) { return hptGuard(c, "hpt_ray_query_any_motion"); }

// ---- scene tables -----------------------------------------------------------------------------------------------------------------
static bool lean_materials(const MaterialRec* m, size_t n)
{
  for (size_t i = 0; i < n; i++) if ((m[i].mtype != MAT_TYPE_GLTF && m[i].mtype != MAT_TYPE_LIGHT_SOURCE) || m[i].texid[1] != 0xFFFFFFFFu) return false;
  return true;
}
static int check_materials(hpt_ctx* c, const MaterialRec* m, size_t n, size_t numTex, size_t numMats, size_t numArrays1f)
{
  for (size_t i = 0; i < n; i++) {
    const uint t = m[i].mtype;
    if (t == MAT_TYPE_BLEND) {
      if (m[i].datai[0] >= numMats || m[i].datai[1] >= numMats) return c->fail(HPT_ERR_ARG, "blend material refers to a material that does not exist");
      if (m[i].texid[0] >= numTex) return c->fail(HPT_ERR_ARG, "material refers to a texture that does not exist");
      continue;
    }
    if (t == MAT_TYPE_THIN_FILM) {                            // every index filmArgs / filmReflTrans follow (hpt_film.h)
      auto bits = [](float f) { uint u; std::memcpy(&u, &f, 4); return u; };
      const uint layers = bits(m[i].data[FILM_LAYERS_COUNT]);
      if (layers < 1u || layers > 64u) return c->fail(HPT_ERR_ARG, "thin film: FILM_LAYERS_COUNT must be 1..64");
      if ((uint64_t)bits(m[i].data[FILM_ETA_OFFSET]) + layers > c->numFilmsEtaK || (uint64_t)bits(m[i].data[FILM_K_OFFSET]) + layers > c->numFilmsEtaK)
        return c->fail(HPT_ERR_ARG, "thin film: eta / k offsets reach past m_films_eta_k_vec");
      if ((uint64_t)bits(m[i].data[FILM_ETA_SPECID_OFFSET]) + layers > c->hFilmsSpecId.size() || (uint64_t)bits(m[i].data[FILM_K_SPECID_OFFSET]) + layers > c->hFilmsSpecId.size())
        return c->fail(HPT_ERR_ARG, "thin film: spectrum id offsets reach past m_films_spec_id_vec");
      for (uint l2 = 0; l2 < layers; l2++) for (int w = 0; w < 2; w++) {
        const uint id = c->hFilmsSpecId[bits(m[i].data[w ? FILM_K_SPECID_OFFSET : FILM_ETA_SPECID_OFFSET]) + l2];
        if (id != 0xFFFFFFFFu && id >= c->numSpectraHost) return c->fail(HPT_ERR_ARG, "thin film: layer refers to a spectrum that does not exist");
      }
      const bool tmap = bits(m[i].data[FILM_THICKNESS_MAP]) != 0u, pre = bits(m[i].data[FILM_PRECOMP_FLAG]) != 0u;
      if (tmap && m[i].texid[2] >= numTex) return c->fail(HPT_ERR_ARG, "thin film: the thickness map does not exist");
      if (m[i].texid[0] >= numTex) return c->fail(HPT_ERR_ARG, "material refers to a texture that does not exist");
      const uint64_t off = bits(m[i].data[FILM_PRECOMP_OFFSET]);
      if (pre && off > c->numPrecompFilms) return c->fail(HPT_ERR_ARG, "thin film: FILM_PRECOMP_OFFSET reaches past m_precomp_thin_films");
      continue;
    }
    if (t != MAT_TYPE_GLTF && t != MAT_TYPE_GLASS && t != MAT_TYPE_CONDUCTOR && t != MAT_TYPE_DIFFUSE && t != MAT_TYPE_DIELECTRIC && t != MAT_TYPE_PLASTIC && t != MAT_TYPE_LIGHT_SOURCE)
      return c->fail(HPT_ERR_UNSUPPORTED, "material type " + std::to_string(t) + " is not one of the reference's");
    if (t == MAT_TYPE_PLASTIC && (uint64_t)m[i].datai[0] + (uint64_t)MI_ROUGH_TRANSMITTANCE_RES > numArrays1f)
      return c->fail(HPT_ERR_ARG, "plastic material: transmittance table outside m_arrays1f");
    if (m[i].texid[1] != 0xFFFFFFFFu && m[i].texid[1] >= numTex) return c->fail(HPT_ERR_ARG, "material refers to a normal map that does not exist");
    if (m[i].texid[0] >= numTex) return c->fail(HPT_ERR_ARG, "material refers to a texture that does not exist");
    if ((m[i].cflags & FLAG_FOUR_TEXTURES) && (m[i].texid[2] >= numTex || m[i].texid[3] >= numTex)) return c->fail(HPT_ERR_ARG, "material refers to a texture that does not exist");
  }
  return HPT_OK;
}
// What the material table says about thin films: whether there is one, and whether each film's precomputed table has the size RGB / spectral
// rendering reads. The loader sizes a table by the mode it loads for (LoadThinFilmMaterial, integrator_pt_scene_mat.cpp:1147-1186): RGB
// 4 x FILM_ANGLE_RES x 3 (x FILM_THICKNESS_RES with a thickness map or a single layer), spectral 4 x FILM_ANGLE_RES x FILM_LENGTH_RES or no table
// (one film with a thickness map). The tables lie back to back in m_precomp_thin_films, so a table ends where the next one starts.
static void film_scan(hpt_ctx* c)
{
  auto bits = [](float f) { uint u; std::memcpy(&u, &f, 4); return u; };
  c->hasFilm = false; c->filmTablesRGB = true; c->filmTablesSpectral = true;
  std::set<uint64_t> starts;
  for (const MaterialRec& m : c->hMaterials) if (m.mtype == MAT_TYPE_THIN_FILM && bits(m.data[FILM_PRECOMP_FLAG]) != 0u) starts.insert(bits(m.data[FILM_PRECOMP_OFFSET]));
  starts.insert(c->numPrecompFilms);
  for (const MaterialRec& m : c->hMaterials) {
    if (m.mtype != MAT_TYPE_THIN_FILM) continue;
    c->hasFilm = true;
    const bool pre = bits(m.data[FILM_PRECOMP_FLAG]) != 0u, tmap = bits(m.data[FILM_THICKNESS_MAP]) != 0u;
    const uint layers = bits(m.data[FILM_LAYERS_COUNT]);
    uint64_t size = 0;
    if (pre) { const uint64_t off = bits(m.data[FILM_PRECOMP_OFFSET]); auto it = starts.upper_bound(off); size = (it == starts.end() ? c->numPrecompFilms : *it) - off; }
    if (size != 4ull * FILM_ANGLE_RES * 3ull * ((tmap || layers == 1u) ? (uint64_t)FILM_THICKNESS_RES : 1ull)) c->filmTablesRGB = false;
    if (pre ? size != 4ull * FILM_ANGLE_RES * FILM_LENGTH_RES : !(tmap && layers <= 2u)) c->filmTablesSpectral = false;
  }
}
// A blend refers to two other materials, possibly blends: the sampling loop of shadeVertex follows such references until it meets a
// leaf, so a reference cycle would never end on the device. Three-colour depth-first search over the blend nodes.
static int check_blend_graph(hpt_ctx* c, const std::vector<MaterialRec>& m)
{
  std::vector<unsigned char> state(m.size(), 0);           // 0 not seen, 1 on the current path, 2 done
  std::vector<std::pair<uint, int>> path;
  for (uint root = 0; root < (uint)m.size(); root++) {
    if (m[root].mtype != MAT_TYPE_BLEND || state[root] != 0) continue;
    path.clear(); path.push_back({ root, 0 }); state[root] = 1;
    while (!path.empty()) {
      auto& top = path.back();
      if (top.second == 2) { state[top.first] = 2; path.pop_back(); continue; }
      const uint child = m[top.first].datai[top.second++];
      if (child >= m.size()) return c->fail(HPT_ERR_ARG, "blend material refers to a material that does not exist");
      if (m[child].mtype != MAT_TYPE_BLEND) continue;
      if (state[child] == 1) return c->fail(HPT_ERR_ARG, "blend materials refer to each other in a cycle (material " + std::to_string(child) + ")");
      if (state[child] == 0) { state[child] = 1; path.push_back({ child, 0 }); }
    }
  }
  return HPT_OK;
}
static int check_lights(hpt_ctx* c, const LightRec* l, size_t n, size_t numTex, size_t numArrays1f)
{
  for (size_t i = 0; i < n; i++) {
    if (l[i].geomType == LIGHT_GEOM_ENV) {                                // a sampled environment map: its pdf table must lie inside m_arrays1f
      const uint64_t need = (uint64_t)l[i].pdfTableSizeX * l[i].pdfTableSizeY + 1u;
      if (l[i].pdfTableSizeX == 0 || l[i].pdfTableSizeY == 0 || (uint64_t)l[i].pdfTableOffset + need > numArrays1f || l[i].pdfTableSize < need)
        return c->fail(HPT_ERR_ARG, "environment light: pdf table outside m_arrays1f");
      if (l[i].texId != 0xFFFFFFFFu && l[i].texId >= numTex) return c->fail(HPT_ERR_ARG, "environment light refers to a texture that does not exist");
    }
    if (l[i].geomType < LIGHT_GEOM_RECT || l[i].geomType > LIGHT_GEOM_ENV) return c->fail(HPT_ERR_ARG, "bad light geomType");
    if (l[i].geomType != LIGHT_GEOM_ENV && l[i].texId != 0xFFFFFFFFu && l[i].texId >= numTex) return c->fail(HPT_ERR_ARG, "light refers to a projected texture that does not exist");
    if (l[i].iesId != 0xFFFFFFFFu && l[i].iesId >= numTex) return c->fail(HPT_ERR_ARG, "light refers to an IES texture that does not exist");
  }
  return HPT_OK;
}

// Every index the kernels follow without a bounds check of their own (an out-of-range one would fault on the device, which on this class of
// machine can take the whole node down): material ids per primitive and per remap target, vertex indices, remap-list and light ids.
static int check_tables(hpt_ctx* c, const hpt_scene_desc* d)
{
  if (d->numInsts && (!d->remapInst || !d->normMatrices)) return c->fail(HPT_ERR_ARG, "m_remapInst / m_normMatrices missing");
  if (d->numTris && (!d->triIndices || !d->matIdByPrimId)) return c->fail(HPT_ERR_ARG, "m_triIndices / m_matIdByPrimId missing");
  if (d->numVerts && !d->vData8f) return c->fail(HPT_ERR_ARG, "m_vData8f missing");
  if (d->numGeoms && !d->matVertOffset) return c->fail(HPT_ERR_ARG, "m_matVertOffset missing");
  for (uint g = 0; g < d->numGeoms; g++) {
    // per-mesh counts: given (geometry handed over with the tables), or the distance to the next mesh's offsets (geometry went through
    // AddGeom_Triangles3f: the Integrator keeps only m_matVertOffset)
    const uint64_t triOff = d->matVertOffset[2 * g + 0], vertOff = d->matVertOffset[2 * g + 1];
    const uint64_t triEnd = d->geomTriCount ? triOff + d->geomTriCount[g] : (g + 1 < d->numGeoms ? d->matVertOffset[2 * (g + 1) + 0] : d->numTris);
    const uint64_t vertEnd = d->geomVertCount ? vertOff + d->geomVertCount[g] : (g + 1 < d->numGeoms ? d->matVertOffset[2 * (g + 1) + 1] : d->numVerts);
    if (triEnd < triOff || vertEnd < vertOff || triEnd > d->numTris || vertEnd > d->numVerts) return c->fail(HPT_ERR_ARG, "m_matVertOffset: geometry " + std::to_string(g) + " reaches past the tables");
    for (uint64_t t = 3 * triOff; t < 3 * triEnd; t++)
      if (d->triIndices[t] >= vertEnd - vertOff) return c->fail(HPT_ERR_ARG, "m_triIndices: vertex index out of the mesh's range (geometry " + std::to_string(g) + ")");
  }
  for (uint t = 0; t < d->numTris; t++)
    if ((d->matIdByPrimId[t] & 0x00FFFFFFu) >= d->numMaterials) return c->fail(HPT_ERR_ARG, "m_matIdByPrimId: material id " + std::to_string(d->matIdByPrimId[t]) + " does not exist");
  // m_allRemapLists = the lists back to back, then one offset per list and the total (integrator_pt_scene.cpp:909-924)
  const uint size = d->allRemapListsSize, len = d->allRemapListsLen;
  int numLists = 0;
  if (len) {
    if (!d->allRemapLists || size >= len) return c->fail(HPT_ERR_ARG, "m_allRemapLists: size does not leave room for the offsets");
    numLists = (int)(len - size) - 1;
    for (int l = 0; l <= numLists; l++) {
      const int off = d->allRemapLists[size + (uint)l];
      if (off < 0 || (uint)off > size || (off & 1) || (l > 0 && off < d->allRemapLists[size + (uint)l - 1])) return c->fail(HPT_ERR_ARG, "m_allRemapLists: bad offset table");
    }
    for (uint i = 1; i < size; i += 2)
      if (d->allRemapLists[i] < 0 || (uint)d->allRemapLists[i] >= d->numMaterials) return c->fail(HPT_ERR_ARG, "m_allRemapLists: remap target " + std::to_string(d->allRemapLists[i]) + " does not exist");
  }
  for (uint i = 0; i < d->numInsts; i++) {
    const int list = d->remapInst[2 * i + 0], light = d->remapInst[2 * i + 1];
    if (list < -1 || list >= numLists) return c->fail(HPT_ERR_ARG, "m_remapInst: remap list " + std::to_string(list) + " does not exist");
    if (light < -1 || light >= (int)d->numLights) return c->fail(HPT_ERR_ARG, "m_remapInst: light " + std::to_string(light) + " does not exist");
  }
  return HPT_OK;
}

extern "C" int hpt_upload_scene(hpt_ctx* c, const hpt_scene_desc* d)
try {
  if (!c || !d) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  const double t0 = now_ms();
  if (d->numTextures == 0 || !d->textures) return c->fail(HPT_ERR_ARG, "m_textures must at least hold the white dummy texture");
  // the film tables first: check_materials follows a film material's indices into them
  c->numFilmsEtaK = d->filmsEtaK ? d->numFilmsEtaK : 0u; c->numPrecompFilms = d->precompThinFilms ? d->numPrecompThinFilms : 0u;
  c->hFilmsSpecId.assign(d->filmsSpecId ? d->filmsSpecId : nullptr, d->filmsSpecId ? d->filmsSpecId + d->numFilmsSpecId : nullptr);
  c->numSpectraHost = (d->specValues && d->specOffsetSz) ? d->numSpectra : 0u;
  int rc = check_materials(c, (const MaterialRec*)d->materials, d->numMaterials, d->numTextures, d->numMaterials, d->numArrays1f); if (rc) return rc;
  c->leanMaterials = lean_materials((const MaterialRec*)d->materials, d->numMaterials);
  { std::vector<MaterialRec> hm((const MaterialRec*)d->materials, (const MaterialRec*)d->materials + d->numMaterials);
    rc = check_blend_graph(c, hm); if (rc) return rc;
    c->hMaterials.swap(hm); }
  film_scan(c);
  if (d->numArrays1f && !d->arrays1f) return c->fail(HPT_ERR_ARG, "m_arrays1f: count without data");
  rc = check_lights(c, (const LightRec*)d->lights, d->numLights, d->numTextures, d->numArrays1f); if (rc) return rc;
  rc = check_tables(c, d); if (rc) return rc;

  if (d->vPos4f) {                                                     // LoadSceneGeometry + LoadSceneInstances order
    hpt_clear_geom(c);
    for (uint g = 0; g < d->numGeoms; g++) {
      const uint triOff = d->matVertOffset[2 * g + 0], vertOff = d->matVertOffset[2 * g + 1];
      const uint32_t id = hpt_add_geom_triangles3f(c, d->vPos4f + 4 * (size_t)vertOff, d->geomVertCount[g], d->triIndices + 3 * (size_t)triOff,
                                                   3 * (size_t)d->geomTriCount[g], 0, 16);
      if (id == 0xFFFFFFFFu) return HPT_ERR_ARG;
    }
    hpt_clear_scene(c);
    for (uint i = 0; i < d->numInsts; i++) {
      const bool moving = d->instHasMotion && d->instMatricesMotion && d->instHasMotion[i] != 0;
      float two[32];
      if (moving) { std::memcpy(two, d->instMatrices + 16 * (size_t)i, 64); std::memcpy(two + 16, d->instMatricesMotion + 16 * (size_t)i, 64); }
      const uint32_t id = moving ? hpt_add_instance_motion(c, d->instGeomId[i], two, 2) : hpt_add_instance(c, d->instGeomId[i], d->instMatrices + 16 * (size_t)i);
      if (id == 0xFFFFFFFFu) return c->fail(HPT_ERR_ARG, "AddInstance: bad geomId");
    }
    rc = hpt_commit_scene(c, 0); if (rc) return rc;
  }
  if (!c->accelCommitted) return c->fail(HPT_ERR_STATE, "CommitDeviceData before CommitScene");
  if (d->numInsts != c->insts.size()) return c->fail(HPT_ERR_ARG, "scene tables and acceleration structure disagree on the instance count");

  HIPCHK(c, c->dTriIndices.upload(d->triIndices, 3 * (size_t)d->numTris));
  c->hTriIndices.assign(d->triIndices, d->triIndices + 3 * (size_t)d->numTris);
  HIPCHK(c, c->dVData.upload(d->vData8f, 8 * (size_t)d->numVerts));
  HIPCHK(c, c->dMatIdByPrim.upload(d->matIdByPrimId, d->numTris));
  HIPCHK(c, c->dMatVertOffset.upload(d->matVertOffset, 2 * (size_t)d->numGeoms));
  std::vector<float> nm(12 * (size_t)std::max(1u, d->numInsts), 0.0f);
  for (uint i = 0; i < d->numInsts; i++) {
    const float* m = d->normMatrices + 16 * (size_t)i;                 // rows of the upper 3x3 (column-major source)
    float* o = &nm[12 * (size_t)i];
    o[0] = m[0]; o[1] = m[4]; o[2] = m[8];  o[3] = 0.0f;
    o[4] = m[1]; o[5] = m[5]; o[6] = m[9];  o[7] = 0.0f;
    o[8] = m[2]; o[9] = m[6]; o[10] = m[10]; o[11] = 0.0f;
  }
  HIPCHK(c, c->dNormMat.upload(nm.data(), nm.size()));
  // motion blur: m_normMatrices[m_normMatrices2Offs + i] = transpose(inverse(matrix at the end of the motion)) (integrator_pt_scene.cpp:864-897)
  if (d->normMatrices2Offs != 0 && d->normMatrices2Offs != d->numInsts) return c->fail(HPT_ERR_ARG, "m_normMatrices2Offs must be 0 or the instance count");
  if ((d->normMatrices2Offs != 0) != c->anyMotion) return c->fail(HPT_ERR_ARG, "m_normMatrices2Offs and the instances added with AddInstanceMotion disagree");
  {
    std::vector<float> nm2(12 * (size_t)std::max(1u, d->numInsts), 0.0f);
    if (d->normMatrices2Offs) for (uint i = 0; i < d->numInsts; i++) {
      const float* m = d->normMatrices + 16 * ((size_t)d->normMatrices2Offs + i);
      float* o = &nm2[12 * (size_t)i];
      o[0] = m[0]; o[1] = m[4]; o[2] = m[8];  o[4] = m[1]; o[5] = m[5]; o[6] = m[9];  o[8] = m[2]; o[9] = m[6]; o[10] = m[10];
    }
    HIPCHK(c, c->dNormMat2.upload(nm2.data(), nm2.size()));
  }
  c->S.normMat2 = c->dNormMat2.p; c->S.motion = d->normMatrices2Offs ? 1u : 0u; 
  HIPCHK(c, c->dRemapInst.upload(d->remapInst, 2 * (size_t)d->numInsts));
  {
    std::vector<int> rl(d->allRemapLists ? std::vector<int>(d->allRemapLists, d->allRemapLists + d->allRemapListsLen) : std::vector<int>());
    if (rl.empty()) rl.push_back(0);
    // RemapMaterialId (integrator_pt_mat.cpp:530-573) bisects over the list's INT count while indexing PAIRS, so its probes reach up to one
    // list length past the list's end (for the last list: past the array, in the reference too). Zero padding keeps those probes in bounds
    // and makes what they read the same here and in the checker.
    rl.resize(rl.size() + (size_t)d->allRemapListsSize + 2, 0);
    HIPCHK(c, c->dRemapLists.upload(rl.data(), rl.size()));
  }
  HIPCHK(c, c->dMaterials.upload((const MaterialRec*)d->materials, d->numMaterials));
  {
    std::vector<LightRec> ll((const LightRec*)d->lights, (const LightRec*)d->lights + d->numLights);
    if (ll.empty()) ll.resize(1);
    HIPCHK(c, c->dLights.upload(ll.data(), ll.size()));
  }
  for (void* p : c->texData) if (p) (void)hipFree(p);
  c->texData.clear(); c->hTextures.resize(d->numTextures);
  for (uint i = 0; i < d->numTextures; i++) {
    const hpt_texture_desc& t = d->textures[i];
    if (t.width == 0 || t.height == 0 || !t.data || t.format > 2) return c->fail(HPT_ERR_ARG, "bad texture descriptor");
    const size_t bytes = (size_t)t.width * t.height * (t.format == 1 ? 16 : 4);
    void* dp = nullptr;
    HIPCHK(c, hipMalloc(&dp, bytes));
    c->texData.push_back(dp);
    HIPCHK(c, hipMemcpy(dp, t.data, bytes, hipMemcpyHostToDevice));
    TexRec& r = c->hTextures[i];
    r.w = t.width; r.h = t.height; r.format = t.format; r.flags = t.flags; r.addrU = t.addressU; r.addrV = t.addressV; r.filter = t.filter; r.pad = 0;
    r.data = dp; r.diffOffset = ~0ull; r.diffW = r.diffH = r.diffChannels = 0; r.pad2 = 0;
  }
  HIPCHK(c, c->dTextures.upload(c->hTextures.data(), c->hTextures.size()));
  c->gradSize = 0;

  DevScene& S = c->S;
  S.triIndices = c->dTriIndices.p; S.vData8f = c->dVData.p; S.matIdByPrimId = c->dMatIdByPrim.p; S.matVertOffset = c->dMatVertOffset.p;
  S.normMat = c->dNormMat.p; S.remapInst = c->dRemapInst.p; S.allRemapLists = c->dRemapLists.p; S.allRemapListsSize = d->allRemapListsSize;
  S.numLights = d->numLights; S.materials = c->dMaterials.p; S.lights = c->dLights.p; S.textures = c->dTextures.p;
  c->numArrays1f = d->numArrays1f;
  { std::vector<float> a(d->numArrays1f ? d->arrays1f : nullptr, d->numArrays1f ? d->arrays1f + d->numArrays1f : nullptr); if (a.empty()) a.push_back(0.0f);
    HIPCHK(c, c->dArrays1f.upload(a.data(), a.size())); }
  S.arrays1f = c->dArrays1f.p;
  { std::vector<float> ek(d->filmsEtaK ? d->filmsEtaK : nullptr, d->filmsEtaK ? d->filmsEtaK + d->numFilmsEtaK : nullptr), pf(d->precompThinFilms ? d->precompThinFilms : nullptr, d->precompThinFilms ? d->precompThinFilms + d->numPrecompThinFilms : nullptr);
    std::vector<uint> si(c->hFilmsSpecId);
    if (ek.empty()) ek.push_back(0.0f); if (pf.empty()) pf.push_back(0.0f); if (si.empty()) si.push_back(0xFFFFFFFFu);
    HIPCHK(c, c->dFilmsEtaK.upload(ek.data(), ek.size())); HIPCHK(c, c->dPrecompFilms.upload(pf.data(), pf.size())); HIPCHK(c, c->dFilmsSpecId.upload(si.data(), si.size()));
    S.filmsEtaK = c->dFilmsEtaK.p; S.precompThinFilms = c->dPrecompFilms.p; S.filmsSpecId = c->dFilmsSpecId.p; }
  // ---- spectral tables (all optional: RGB rendering does not read them) ----
  {
    const uint WAVES = 471u;                                   // LAMBDA_MAX - LAMBDA_MIN + 1 samples per spectrum (Spectrum::ResampleUniform)
    c->spectralOk = d->specValues && d->specOffsetSz && d->numSpectra > 0 && d->cieXYZ && d->numCieXYZ >= WAVES;
    c->spectralWhyNot = c->spectralOk ? "" : "the scene came without m_spec_values / m_spec_offset_sz / m_cie_xyz";
    std::vector<uint> so(2 * std::max(1u, d->numSpectra), 0u);
    for (uint i = 0; i < d->numSpectra && d->specOffsetSz; i++) {
      so[2 * i] = d->specOffsetSz[2 * i]; so[2 * i + 1] = d->specOffsetSz[2 * i + 1];
      if (so[2 * i] == 0xFFFFFFFFu) {                            // a spectrum given by textures (lambda_ref_ids): its bands must come with it
        so[2 * i] = 0;
        const bool bands = d->specTexOffsetSz && d->specTexIdsWavelengths && d->specTexOffsetSz[2 * i + 1] >= 2u;
        if (!bands && c->spectralOk) { c->spectralOk = false; c->spectralWhyNot = "spectrum " + std::to_string(i) + " is given by textures, but m_spec_tex_ids_wavelengths / m_spec_tex_offset_sz did not come with the scene"; }
        continue;
      }
      if ((uint64_t)so[2 * i] + std::max(so[2 * i + 1], WAVES - 1u) > d->numSpecValues) return c->fail(HPT_ERR_ARG, "m_spec_offset_sz: spectrum " + std::to_string(i) + " reaches past m_spec_values");
    }
    {
      std::vector<uint> tos(2 * std::max(1u, d->numSpectra), 0u), tiw(2 * std::max(1u, d->specTexIdsWavelengths ? d->numSpecTexBands : 0u), 0u);
      if (d->specTexIdsWavelengths && d->numSpecTexBands) std::memcpy(tiw.data(), d->specTexIdsWavelengths, 8 * (size_t)d->numSpecTexBands);
      for (uint i = 0; i < d->numSpectra && d->specTexOffsetSz && d->specTexIdsWavelengths; i++) {
        const uint off = d->specTexOffsetSz[2 * i], sz = d->specTexOffsetSz[2 * i + 1];
        if (sz == 0u) continue;                                  // ({0xFFFFFFFF, 0}: a tabulated spectrum)
        if (sz < 2u || (uint64_t)off + sz > d->numSpecTexBands) return c->fail(HPT_ERR_ARG, "m_spec_tex_offset_sz: spectrum " + std::to_string(i) + " reaches past m_spec_tex_ids_wavelengths (or holds fewer than two bands)");
        for (uint k2 = 0; k2 < sz; k2++) {
          // (LoadSpectralTextures resolves the bands' texture ids only when it loads for spectral rendering: a scene loaded for RGB keeps the XML's)
          if (d->specTexIdsWavelengths[2 * (off + k2)] >= d->numTextures) { if (c->spectralOk) { c->spectralOk = false; c->spectralWhyNot = "m_spec_tex_ids_wavelengths refers to a texture that is not in m_textures (the scene was loaded for RGB rendering)"; } tiw[2 * (off + k2)] = 0u; }
          if (k2 && d->specTexIdsWavelengths[2 * (off + k2) + 1] <= d->specTexIdsWavelengths[2 * (off + k2 - 1) + 1]) return c->fail(HPT_ERR_ARG, "m_spec_tex_ids_wavelengths: the bands of a spectrum must ascend in wavelength");
        }
        tos[2 * i] = off; tos[2 * i + 1] = sz;
      }
      HIPCHK(c, c->dSpecTexOffsetSz.upload(tos.data(), tos.size())); HIPCHK(c, c->dSpecTexIdsWavelengths.upload(tiw.data(), tiw.size()));
      S.specTexOffsetSz = c->dSpecTexOffsetSz.p; S.specTexIdsWavelengths = c->dSpecTexIdsWavelengths.p;
    }
    auto specIdOk = [&](uint id) { return id == 0xFFFFFFFFu || id < d->numSpectra; };
    const MaterialRec* mm = (const MaterialRec*)d->materials;
    for (uint i = 0; i < d->numMaterials; i++)
      for (int k2 = 0; k2 < 4; k2++) if (d->specValues && !specIdOk(mm[i].spdid[k2])) return c->fail(HPT_ERR_ARG, "material " + std::to_string(i) + " refers to a spectrum that does not exist");
    // (every material type of the RGB kernels has its branch in the spectral kernel, blends and normal maps included.) Which instantiations a
    // spectral call needs (hpt_spectral.hip: SCOPE) is decided by the materials a hit can REACH: m_matIdByPrimId through the remap list of every
    // instance of the mesh (RemapMaterialId), then through blends - an entry of the library nothing resolves to (the reference's spectral fixture
    // keeps an unused legacy material 0) does not widen the kernel
    {
      std::vector<char> reached(d->numMaterials, 0);
      std::set<std::pair<uint, int>> seen;
      for (uint inst = 0; inst < d->numInsts; inst++) {
        const uint g = d->instGeomId[inst];
        const int list = d->remapInst[2 * inst + 0];
        if (g >= d->numGeoms || !seen.insert(std::make_pair(g, list)).second) continue;
        int rOff = 0, rSize = 0;
        if (list >= 0 && d->allRemapLists && (uint)list + 1 + d->allRemapListsSize < d->allRemapListsLen) {
          rOff = d->allRemapLists[d->allRemapListsSize + list]; rSize = (d->allRemapLists[d->allRemapListsSize + list + 1] - rOff) / 2;
        }
        const uint t0 = d->matVertOffset[2 * g];
        const uint t1 = std::min(d->geomTriCount ? t0 + d->geomTriCount[g] : (g + 1 < d->numGeoms ? d->matVertOffset[2 * (g + 1)] : d->numTris), d->numTris);
        for (uint t = t0; t < t1; t++) {
          uint id = d->matIdByPrimId[t];
          if (list >= 0 && rSize > 0) {
            // the device's RemapMaterialId exactly (hpt_device.h: remapMaterialId = integrator_pt_mat.cpp:530-573): a bisection over the list's INT
            // count that indexes PAIRS, so its probes can land in the next list or in the zero padding behind the table - 'reached' must follow it
            auto at = [&](long k) -> int { return (k >= 0 && (size_t)k < d->allRemapListsLen) ? d->allRemapLists[k] : 0; };
            const int rInts = 2 * rSize;
            int low = 0, high = rInts - 1;
            while (low <= high) { const int mid = low + ((high - low) / 2); if ((uint)at((long)rOff + mid * 2) >= id) high = mid - 1; else low = mid + 1; }
            if (high + 1 < rInts && (uint)at((long)rOff + (high + 1) * 2) == id) id = (uint)at((long)rOff + (high + 1) * 2 + 1);
          }
          id &= 0x00FFFFFFu;
          if (id < d->numMaterials) reached[id] = 1;
        }
      }
      c->spectralGltfMats = c->spectralHeavyMats = false;
      uint typeMask = 0u; bool rich = false;                    // the reachable material types, for the choice between MODE 0 and MODE 7 (hpt_decl.h)
      for (uint i = 0; i < d->numMaterials; i++) {
        if (!reached[i]) continue;
        if (mm[i].mtype < 32u) typeMask |= 1u << mm[i].mtype;
        if (mm[i].mtype == MAT_TYPE_BLEND || mm[i].mtype == MAT_TYPE_PLASTIC || (mm[i].mtype != MAT_TYPE_LIGHT_SOURCE && mm[i].texid[1] != 0xFFFFFFFFu)) rich = true;
        if (mm[i].mtype == MAT_TYPE_GLTF) c->spectralGltfMats = true;
        if (mm[i].mtype == MAT_TYPE_GLASS || mm[i].mtype == MAT_TYPE_BLEND || (mm[i].mtype != MAT_TYPE_LIGHT_SOURCE && mm[i].texid[1] != 0xFFFFFFFFu)) c->spectralHeavyMats = true;   // (a reached blend: whatever its leaves are)
      }
      c->fewMaterialTypes = !rich && __builtin_popcount(typeMask) <= 3;
    }
    const LightRec* ll2 = (const LightRec*)d->lights;
    for (uint i = 0; i < d->numLights; i++) {
      if (d->specValues && !specIdOk(ll2[i].specId)) return c->fail(HPT_ERR_ARG, "light " + std::to_string(i) + " refers to a spectrum that does not exist");
    }
    for (int k2 = 0; k2 < 3; k2++) if (d->camResponseSpectrumId[k2] >= (int)d->numSpectra) return c->fail(HPT_ERR_ARG, "m_camResponseSpectrumId refers to a spectrum that does not exist");
    std::vector<float> sv(d->specValues ? std::vector<float>(d->specValues, d->specValues + d->numSpecValues) : std::vector<float>());
    if (sv.size() < WAVES) sv.resize(WAVES, 0.0f);             // (a spectrum id that stands for textures reads as offset 0 wherever a tabulated one is looked up)
    std::vector<float4> cie(std::max(d->numCieXYZ, 1u), make_float4(0, 0, 0, 0));
    for (uint i = 0; i < d->numCieXYZ && d->cieXYZ; i++) cie[i] = make_float4(d->cieXYZ[4 * i], d->cieXYZ[4 * i + 1], d->cieXYZ[4 * i + 2], d->cieXYZ[4 * i + 3]);
    HIPCHK(c, c->dSpecValues.upload(sv.data(), sv.size()));
    HIPCHK(c, c->dSpecOffsetSz.upload(so.data(), so.size()));
    HIPCHK(c, c->dCieXYZ.upload(cie.data(), cie.size()));
    S.specValues = c->dSpecValues.p; S.specOffsetSz = c->dSpecOffsetSz.p; S.cieXYZ = c->dCieXYZ.p;
    S.numCieXYZ = d->cieXYZ ? d->numCieXYZ : 0u; S.numSpectra = d->numSpectra;
    for (int k2 = 0; k2 < 3; k2++) S.camResponseSpectrumId[k2] = d->specValues ? d->camResponseSpectrumId[k2] : -1;
    S.camResponseType = d->camResponseType;
  }
  c->hLightGeom.resize(d->numLights);
  for (uint i = 0; i < d->numLights; i++) c->hLightGeom[i] = ((const LightRec*)d->lights)[i].geomType;
  // a new scene invalidates the environment ids of the previous UpdateMembersPlainData until the next one
  S.envTexId = S.envLightId = S.envCamBackId = 0xFFFFFFFFu; S.envEnableSam = 0;
  c->sceneUploaded = true; c->shadeTrisDirty = true;
  c->tPathTrace[1] = c->tNaive[1] = c->tDR[1] = float(now_ms() - t0);   // host -> device time of the scene commit
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_upload_scene"); }

extern "C" int hpt_update_params(hpt_ctx* c, const hpt_params* p)
try {
  if (!c || !p) return HPT_ERR_ARG;
  if (p->spectralMode > 1) return c->fail(HPT_ERR_ARG, "bad spectral mode");
  if (p->spectralMode == 1 && !c->sceneUploaded) return c->fail(HPT_ERR_STATE, "UpdateMembersPlainData with m_spectral_mode before CommitDeviceData");
  if (p->spectralMode == 1 && !c->spectralOk) return c->fail(HPT_ERR_UNSUPPORTED, "spectral rendering: " + c->spectralWhyNot);
  if (p->spectralMode == 1 && p->envSpecIdPlus1 != 0u && p->envSpecIdPlus1 - 1u >= c->S.numSpectra) return c->fail(HPT_ERR_ARG, "m_envSpecId refers to a spectrum that does not exist");
  if (p->winWidth <= 0 || p->winHeight <= 0 || p->fbWidth <= 0 || p->fbHeight <= 0 || p->winWidth > 65535 || p->winHeight > 65535) return c->fail(HPT_ERR_ARG, "bad viewport");
  if (p->tileSize != 1 && p->tileSize != 2 && p->tileSize != 4 && p->tileSize != 8) return c->fail(HPT_ERR_ARG, "bad tile size");
  // kernel_PackXY tiles the window without a remainder (integrator_rt.cpp:13-31); SetViewport only ever picks a tile size that divides both
  // sides (integrator_pt.h:379-389). Anything else would index past W*H in PackXY and in every kernel that reads m_packedXY.
  if (p->winWidth % (int)p->tileSize != 0 || p->winHeight % (int)p->tileSize != 0)
    return c->fail(HPT_ERR_ARG, "tile size must divide the viewport's width and height (SetViewport falls back to 4 / 2 / 1, integrator_pt.h:379-389)");
  DevScene& S = c->S;
  std::memcpy(S.projInv, p->projInv, 64); std::memcpy(S.worldViewInv, p->worldViewInv, 64);
  S.winStartX = p->winStartX; S.winStartY = p->winStartY; S.winWidth = p->winWidth; S.winHeight = p->winHeight; S.fbWidth = p->fbWidth; S.fbHeight = p->fbHeight;
  S.traceDepth = p->traceDepth; S.integratorType = p->integratorType; S.renderLayer = p->renderLayer; S.tileSize = p->tileSize;
  S.exposureMult = p->exposureMult; S.camLensRadius = p->camLensRadius; S.camTargetDist = p->camTargetDist;
  S.spectralMode = p->spectralMode;
  S.envSpecId = p->envSpecIdPlus1 - 1u; S.envSpecMult = p->envSpecMult;   // (0 -> 0xFFFFFFFF: none)
  std::memcpy(S.camRespoceRGB, p->camRespoceRGB, 16); std::memcpy(S.envColor, p->envColor, 16);
  // the environment map (integrator_pt_scene.cpp:441-478): ids are checked against what CommitDeviceData uploaded
  if ((p->envTexId != 0xFFFFFFFFu || p->envCamBackId != 0xFFFFFFFFu || p->envLightId != 0xFFFFFFFFu) && !c->sceneUploaded)
    return c->fail(HPT_ERR_STATE, "UpdateMembersPlainData with an environment map before CommitDeviceData");
  if (p->envTexId != 0xFFFFFFFFu && p->envTexId >= c->hTextures.size()) return c->fail(HPT_ERR_ARG, "m_envTexId refers to a texture that does not exist");
  if (p->envCamBackId != 0xFFFFFFFFu && p->envCamBackId >= c->hTextures.size()) return c->fail(HPT_ERR_ARG, "m_envCamBackId refers to a texture that does not exist");
  if (p->envLightId != 0xFFFFFFFFu && (p->envLightId >= c->S.numLights || !c->envLightOk(p->envLightId))) return c->fail(HPT_ERR_ARG, "m_envLightId is not a LIGHT_GEOM_ENV light with a pdf table");
  if (p->envEnableSam != 0 && p->envLightId == 0xFFFFFFFFu) return c->fail(HPT_ERR_ARG, "m_envEnableSam without m_envLightId");
  S.envTexId = p->envTexId; S.envLightId = p->envLightId; S.envCamBackId = p->envCamBackId; S.envEnableSam = p->envEnableSam;
  std::memcpy(S.envSamRow0, p->envSamRow0, 16); std::memcpy(S.envSamRow1, p->envSamRow1, 16);
  c->paramsSet = true;
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_update_params"); }

// SetLines / SetPhysSize + m_enableOpticSim (integrator_pt.h:353-362, integrator_pt_scene.cpp:714-720, 1078-1141): n = 0 switches it off
extern "C" int hpt_set_optics(hpt_ctx* c, const float* lines4, uint32_t n, float physSizeX, float physSizeY)
try {
  if (!c || (n && !lines4)) return HPT_ERR_ARG;
  if (n > 64) return c->fail(HPT_ERR_ARG, "SetLines: more than 64 lens interfaces");
  for (uint32_t i = 0; i < n; i++) if (!(lines4[4 * i + 3] >= 0.0f)) return c->fail(HPT_ERR_ARG, "SetLines: negative aperture radius");
  (void)hipSetDevice(c->device);
  std::vector<float4> l(std::max<uint32_t>(n, 1u), make_float4(0, 0, 0, 0));
  for (uint32_t i = 0; i < n; i++) l[i] = make_float4(lines4[4 * i], lines4[4 * i + 1], lines4[4 * i + 2], lines4[4 * i + 3]);
  HIPCHK(c, c->dLensLines.upload(l.data(), l.size()));
  c->S.lensLines = c->dLensLines.p; c->S.lensCount = n; c->S.physSize[0] = physSizeX; c->S.physSize[1] = physSizeY; c->S.padLens = 0;
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_set_optics"); }

extern "C" int hpt_update_materials(hpt_ctx* c, size_t first, size_t count, const void* mats)
try {
  if (!c || !mats) return HPT_ERR_ARG;
  if (first + count > c->dMaterials.n) return c->fail(HPT_ERR_ARG, "Update_m_materials: range out of bounds");
  int rc = check_materials(c, (const MaterialRec*)mats, count, c->hTextures.size(), c->dMaterials.n, c->numArrays1f); if (rc) return rc;
  if (c->hMaterials.size() == c->dMaterials.n) {                                        // the blend graph of the table as it will be after the update
    std::vector<MaterialRec> hm(c->hMaterials);
    std::memcpy(hm.data() + first, mats, count * sizeof(MaterialRec));
    rc = check_blend_graph(c, hm); if (rc) return rc;
    c->hMaterials.swap(hm);
    film_scan(c);
  }
  if (!lean_materials((const MaterialRec*)mats, count)) c->leanMaterials = false;      // (an update can only widen the set of BSDFs in use)
  for (size_t i = 0; i < count; i++) {
    const MaterialRec& m = ((const MaterialRec*)mats)[i];
    if (m.mtype == MAT_TYPE_GLTF) c->spectralGltfMats = true;
    if (m.mtype == MAT_TYPE_GLASS || m.mtype == MAT_TYPE_BLEND || (m.mtype != MAT_TYPE_LIGHT_SOURCE && m.texid[1] != 0xFFFFFFFFu)) c->spectralHeavyMats = true;
  }
  c->fewMaterialTypes = false;                                 // (an update may bring in any type)
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipMemcpy(c->dMaterials.p + first, mats, count * sizeof(MaterialRec), hipMemcpyHostToDevice));
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_update_materials"); }
extern "C" int hpt_update_lights(hpt_ctx* c, size_t first, size_t count, const void* lights)
try {
  if (!c || !lights) return HPT_ERR_ARG;
  if (first + count > c->S.numLights) return c->fail(HPT_ERR_ARG, "Update_m_lights: range out of bounds");
  int rc = check_lights(c, (const LightRec*)lights, count, c->hTextures.size(), c->numArrays1f); if (rc) return rc;
  for (size_t i = 0; i < count; i++) c->hLightGeom[first + i] = ((const LightRec*)lights)[i].geomType;
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipMemcpy(c->dLights.p + first, lights, count * sizeof(LightRec), hipMemcpyHostToDevice));
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_update_lights"); }

// Update_m_matIdOffsets (integrator_pt.h:470): m_matVertOffset changed on the host
extern "C" int hpt_update_mat_id_offsets(hpt_ctx* c, const uint32_t* mvo, size_t numGeoms)
try {
  if (!c || !mvo) return HPT_ERR_ARG;
  if (!c->sceneUploaded) return c->fail(HPT_ERR_STATE, "Update_m_matIdOffsets before CommitDeviceData");
  if (2 * numGeoms != c->dMatVertOffset.n) return c->fail(HPT_ERR_ARG, "Update_m_matIdOffsets: geometry count differs from the committed scene");
  const size_t numTris = c->dMatIdByPrim.n, numVerts = c->dVData.n / 8;
  for (size_t g = 0; g < numGeoms; g++) {                     // every mesh's range must stay inside the uploaded tables (the kernels index them unchecked)
    const size_t nt = g < c->geoms.size() ? c->geoms[g].idx.size() / 3 : 0;
    if (mvo[2 * g] > numTris || mvo[2 * g + 1] > numVerts || (size_t)mvo[2 * g] + nt > numTris)
      return c->fail(HPT_ERR_ARG, "Update_m_matIdOffsets: geometry " + std::to_string(g) + " reaches past the tables");
    for (size_t t = 3 * (size_t)mvo[2 * g]; t < 3 * ((size_t)mvo[2 * g] + nt) && t < c->hTriIndices.size(); t++)
      if ((size_t)c->hTriIndices[t] + mvo[2 * g + 1] >= numVerts)
        return c->fail(HPT_ERR_ARG, "Update_m_matIdOffsets: geometry " + std::to_string(g) + " reaches past the tables (vertex index)");
  }
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipMemcpy(c->dMatVertOffset.p, mvo, 2 * numGeoms * sizeof(uint32_t), hipMemcpyHostToDevice));
  c->shadeTrisDirty = true;                                   // the shading records were gathered through the old offsets
  // the meshes now point at other triangle ranges, i.e. at other materials: what a hit can REACH was scanned at hpt_upload_scene through the
  // old offsets, so the kernels chosen by it widen to the ones that hold every branch (a scope-0 spectral kernel has no glass branch)
  c->spectralGltfMats = c->spectralHeavyMats = true; c->fewMaterialTypes = false;
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_update_mat_id_offsets"); }

extern "C" int hpt_pack_xy(hpt_ctx* c, uint32_t tidX, uint32_t tidY)
try {
  if (!c) return HPT_ERR_ARG;
  if (!c->paramsSet) return c->fail(HPT_ERR_STATE, "PackXYBlock before UpdateMembersPlainData");
  if ((int)tidX != c->S.winWidth || (int)tidY != c->S.winHeight) return c->fail(HPT_ERR_ARG, "PackXYBlock: size differs from the viewport");
  (void)hipSetDevice(c->device);
  const size_t n = (size_t)tidX * tidY;
  if (c->S.tileSize == 0u || tidX % c->S.tileSize != 0u || tidY % c->S.tileSize != 0u) return c->fail(HPT_ERR_ARG, "PackXYBlock: tile size does not divide the viewport");
  HIPCHK(c, c->dPackedXY.alloc(n));
  HIPCHK(c, hipMemsetAsync(c->dPackedXY.p, 0, n * sizeof(uint), nullptr));
  packXYKernel<<<dim3((tidX + 15) / 16, (tidY + 15) / 16), dim3(16, 16), 0, 0>>>(c->dPackedXY.p, (int)tidX, (int)tidY, c->S.tileSize);
  HIPCHK(c, hipGetLastError());
  c->packedCount = (uint)n;
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_pack_xy"); }
extern "C" int hpt_get_packed_xy(hpt_ctx* c, uint32_t* out, uint32_t count)
try {
  if (!c || !out || count > c->packedCount) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipMemcpy(out, c->dPackedXY.p, (size_t)count * 4, hipMemcpyDeviceToHost));
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_get_packed_xy"); }
extern "C" int hpt_init_random_gens_from(hpt_ctx* c, uint32_t n, uint32_t firstSeed)
try {
  if (!c || n == 0) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  HIPCHK(c, c->dGens.alloc(n));
  initRandomGensKernel<<<dim3((n + 255) / 256), dim3(256), 0, 0>>>(c->dGens.p, n, firstSeed);
  HIPCHK(c, hipGetLastError());
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_init_random_gens_from"); }
extern "C" int hpt_init_random_gens(hpt_ctx* c, uint32_t n) try { return hpt_init_random_gens_from(c, n, 0u); }
catch (...) { return hptGuard(c, "hpt_init_random_gens"); }
extern "C" int hpt_get_random_gens(hpt_ctx* c, uint32_t* out, uint32_t count)
try {
  if (!c || !out || count > c->dGens.n) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  HIPCHK(c, hipMemcpy(out, c->dGens.p, (size_t)count * 8, hipMemcpyDeviceToHost));
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_get_random_gens"); }
extern "C" int hpt_set_random_gens(hpt_ctx* c, const uint32_t* in, uint32_t count)
try {
  if (!c || !in) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  HIPCHK(c, c->dGens.alloc(count));
  HIPCHK(c, hipMemcpy(c->dGens.p, in, (size_t)count * 8, hipMemcpyHostToDevice));
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_set_random_gens"); }

// ---- the hot path -------------------------------------------------------------------------------------------------------------------
static int gridBlocks(hpt_ctx* c, bool dr, bool fullMaterials = false)
{
  int bpc = c->blocksPerCU;
  if (bpc <= 0) bpc = fullMaterials ? HPT_FULL_WAVES : 4;   // = the kernel's __launch_bounds__; measured for DR: 2 -> 185, 3 -> 206, 4 -> 259 Mpaths/s
  return c->numCUs * bpc;
}

static bool useWavefront(hpt_ctx* c, bool naive, bool dr, bool stats, uint tidCount);
static bool wfWide(const hpt_ctx* c);
static int launch_wavefront(hpt_ctx* c, const Job& job, hipStream_t st, bool dr, int specScope = -1);
static int launch_stream(hpt_ctx* c, const Job& job, hipStream_t st);

// DEEP: the scene's BVH can need more than LDS_STACK stack entries, so pushes / pops check for the HBM overflow part
// FLAT: single-level world-space BVH (static scenes within FLAT_TRI_BUDGET) vs two-level TLAS/BLAS
template <bool STATS, bool DR, int NAIVE>
static void launchPT(const DevScene& S, const Job& job, int blocks, hipStream_t st, bool deep)
{
  if (S.sweep) pathTraceKernel<STATS, DR, NAIVE, false, false, false, true><<<dim3(blocks), dim3(256), 0, st>>>(S, job);
  else if (S.flatMode) {
    if (deep) pathTraceKernel<STATS, DR, NAIVE, true, true><<<dim3(blocks), dim3(256), 0, st>>>(S, job);
    else      pathTraceKernel<STATS, DR, NAIVE, false, true><<<dim3(blocks), dim3(256), 0, st>>>(S, job);
  } else {
    if (deep) pathTraceKernel<STATS, DR, NAIVE, true, false><<<dim3(blocks), dim3(256), 0, st>>>(S, job);
    else      pathTraceKernel<STATS, DR, NAIVE, false, false><<<dim3(blocks), dim3(256), 0, st>>>(S, job);
  }
}

template <bool DR, bool LEAN>
static void launchBlock(const DevScene& S, const Job& job, int blocks, hipStream_t st, bool deep, uint refillBelow, uint nodeMin, bool wide = false)
{
  const dim3 g(blocks), b(256);
  if (wide && LEAN) { if (deep) pathTraceBlockKernel<DR, true, true, true, true><<<g, b, 0, st>>>(S, job, refillBelow, nodeMin); else pathTraceBlockKernel<DR, true, false, true, true><<<g, b, 0, st>>>(S, job, refillBelow, nodeMin); }
  else if (S.flatMode) { if (deep) pathTraceBlockKernel<DR, LEAN, true, true><<<g, b, 0, st>>>(S, job, refillBelow, nodeMin); else pathTraceBlockKernel<DR, LEAN, false, true><<<g, b, 0, st>>>(S, job, refillBelow, nodeMin); }
  else            { if (deep) pathTraceBlockKernel<DR, LEAN, true, false><<<g, b, 0, st>>>(S, job, refillBelow, nodeMin); else pathTraceBlockKernel<DR, LEAN, false, false><<<g, b, 0, st>>>(S, job, refillBelow, nodeMin); }
}

template <int WIDE>
static void launchSpectral(const DevScene& S, const Job& job, int blocks, hipStream_t st, bool deep)
{
  const dim3 sg(blocks), sb(256);
  if (S.motion) {                                              // moving instances (never a sweep scene)
    if (S.flatMode) { if (deep) pathTraceSpectralKernel<true, true, false, true, WIDE><<<sg, sb, 0, st>>>(S, job); else pathTraceSpectralKernel<false, true, false, true, WIDE><<<sg, sb, 0, st>>>(S, job); }
    else            { if (deep) pathTraceSpectralKernel<true, false, false, true, WIDE><<<sg, sb, 0, st>>>(S, job); else pathTraceSpectralKernel<false, false, false, true, WIDE><<<sg, sb, 0, st>>>(S, job); }
  }
  else if (S.sweep)     pathTraceSpectralKernel<false, false, true, false, WIDE><<<sg, sb, 0, st>>>(S, job);
  else if (S.flatMode)  { if (deep) pathTraceSpectralKernel<true, true, false, false, WIDE><<<sg, sb, 0, st>>>(S, job); else pathTraceSpectralKernel<false, true, false, false, WIDE><<<sg, sb, 0, st>>>(S, job); }
  else                  { if (deep) pathTraceSpectralKernel<true, false, false, false, WIDE><<<sg, sb, 0, st>>>(S, job); else pathTraceSpectralKernel<false, false, false, false, WIDE><<<sg, sb, 0, st>>>(S, job); }
}

template <int SCOPE>
static void launchSpectralBlock(const DevScene& S, const Job& job, int blocks, hipStream_t st, bool deep, bool wide, uint refillBelow, uint nodeMin)
{
  const dim3 g(blocks), b(256);
  if (wide)            { if (deep) pathTraceBlockSpectralKernel<true, true, true, SCOPE><<<g, b, 0, st>>>(S, job, refillBelow, nodeMin); else pathTraceBlockSpectralKernel<false, true, true, SCOPE><<<g, b, 0, st>>>(S, job, refillBelow, nodeMin); }
  else if (S.flatMode) { if (deep) pathTraceBlockSpectralKernel<true, true, false, SCOPE><<<g, b, 0, st>>>(S, job, refillBelow, nodeMin); else pathTraceBlockSpectralKernel<false, true, false, SCOPE><<<g, b, 0, st>>>(S, job, refillBelow, nodeMin); }
  else                 { if (deep) pathTraceBlockSpectralKernel<true, false, false, SCOPE><<<g, b, 0, st>>>(S, job, refillBelow, nodeMin); else pathTraceBlockSpectralKernel<false, false, false, SCOPE><<<g, b, 0, st>>>(S, job, refillBelow, nodeMin); }
}

// the MOTION variants: moving instances (two-level layout only), every BSDF branch
template <int MODE>
static void launchPTMotion(const DevScene& S, const Job& job, int blocks, hipStream_t st, bool deep)
{
  if (S.flatMode) {
    if (deep) pathTraceKernel<false, false, MODE, true, true, true><<<dim3(blocks), dim3(256), 0, st>>>(S, job);
    else      pathTraceKernel<false, false, MODE, false, true, true><<<dim3(blocks), dim3(256), 0, st>>>(S, job);
  } else {
    if (deep) pathTraceKernel<false, false, MODE, true, false, true><<<dim3(blocks), dim3(256), 0, st>>>(S, job);
    else      pathTraceKernel<false, false, MODE, false, false, true><<<dim3(blocks), dim3(256), 0, st>>>(S, job);
  }
}

// DevScene::shadeTris for this launch: gltf / emissive scenes on the single-level layout get one 64-byte record of shading data per triangle
// record (rebuilt on the device after a commit or a table upload); everything else leaves the pointer null and gathers through the index chain
static int ensureShadeTris(hpt_ctx* c, hipStream_t st)
{
  const bool want = c->shadeTrisEnabled && c->S.flatMode != 0u && c->S.motion == 0u && c->leanMaterials && !c->forceFull && c->S.lensCount == 0u &&
                    c->instTris > 0 && c->instTris < (size_t(1) << 28) && c->sceneUploaded && c->accelCommitted;
  if (!want) { c->S.shadeTris = nullptr; return HPT_OK; }
  if (c->shadeTrisDirty || c->dShadeTris.n < 4 * c->instTris) {
    HIPCHK(c, c->dShadeTris.alloc(4 * c->instTris));
    DevScene Sb = c->S; Sb.shadeTris = nullptr;
    const uint n = (uint)c->instTris;
    buildShadeTrisKernel<<<dim3((n + 255u) / 256u), dim3(256), 0, st>>>(Sb, n, c->dShadeTris.p);
    HIPCHK(c, hipGetLastError());
    c->shadeTrisDirty = false;
  }
  c->S.shadeTris = c->dShadeTris.p;
  return HPT_OK;
}

static int launch_path_trace(hpt_ctx* c, Job& job, bool naive, bool dr, hipStream_t st)
{
  const bool inRays = job.inRayPos != nullptr;
  if (!c->sceneUploaded || !c->paramsSet) return c->fail(HPT_ERR_STATE, "PathTraceBlock before CommitDeviceData / UpdateMembersPlainData");
  if (!inRays && c->packedCount != (uint)(c->S.winWidth * c->S.winHeight)) return c->fail(HPT_ERR_STATE, "PathTraceBlock before PackXYBlock");
  job.tidStride = c->tidStride > 1 ? c->tidStride : 1u;
  job.tidChunk = (job.tidStride > 1 && c->tidChunk > 0) ? c->tidChunk : 0x40000000u;
  job.tidEnd = inRays ? job.tidBegin + job.tidCount : c->packedCount;
  if (inRays) { job.tidStride = 1; job.tidChunk = 0x40000000u; }
  if (!inRays && job.tidStride == 1 && (size_t)job.tidBegin + job.tidCount > c->packedCount) return c->fail(HPT_ERR_ARG, "PathTraceBlock: tid range exceeds the viewport");
  if (c->dGens.n < (inRays ? (size_t)job.tidEnd : (size_t)c->packedCount)) return c->fail(HPT_ERR_STATE, "PathTraceBlock: m_randomGens smaller than the thread range (InitRandomGens)");
  if (job.channels < 1 || (job.channels > 4 && c->S.spectralMode == 0u)) return c->fail(HPT_ERR_UNSUPPORTED, "PathTraceBlock: channels must be 1..4 (more are the wavelength layers of spectral rendering)");
  // two channels: kernel_ContributeToImage writes three components at pixel * channels (integrator_pt.cpp:636-641), i.e. into the next pixel and,
  // for the last one, past the buffer - refused rather than restated
  if (job.channels == 2) return c->fail(HPT_ERR_UNSUPPORTED, "PathTraceBlock: a two-channel framebuffer cannot hold the three components written per pixel (1, 3 or 4 channels)");
  if (dr && (c->S.traceDepth == 0 || c->S.traceDepth > 16)) return c->fail(HPT_ERR_ARG, "PathTraceDR: trace depth must be 1..16");
  if (dr && c->S.lensCount) return c->fail(HPT_ERR_UNSUPPORTED, "PathTraceDR: the lens simulation is not differentiated");
  if (dr && c->S.motion) return c->fail(HPT_ERR_UNSUPPORTED, "PathTraceDR: motion blur is not differentiated");
  // the DR kernels add the constant m_envColor unweighted; the reference's replay evaluates EnvironmentColor() with the map and the
  // env-sampling MIS weight (integrator_dr.cpp:1077-1098): such scenes are refused rather than differentiated differently
  if (dr && (c->S.envTexId != 0xFFFFFFFFu || c->S.envEnableSam != 0u || c->S.envCamBackId != 0xFFFFFFFFu))
    return c->fail(HPT_ERR_UNSUPPORTED, "PathTraceDR: environment maps, their sampling and camera back plates are not differentiated (constant m_envColor only)");
  if (dr && !c->leanMaterials) return c->fail(HPT_ERR_UNSUPPORTED, "PathTraceDR: gltf and emissive materials without normal maps only (what the reference's replay differentiates, integrator_dr.cpp:461-612)");
  // never more lanes than pixels: with fewer, the hardware's round-robin block placement spreads them evenly over the CUs, whereas a
  // full grid would let whichever waves ask first take all the work (a small multi-GPU share of a frame)
  if (c->S.spectralMode != 0u) {                               // four wavelengths per path: its own (plain) kernel
    if (dr) return c->fail(HPT_ERR_UNSUPPORTED, "spectral rendering: not the differentiable integrator");
    if (inRays && job.channels != 1u && job.channels != 4u) return c->fail(HPT_ERR_ARG, "PathTraceFromInputRays in spectral mode: 1 or 4 channels (kernel_CopyColorToOutput)");
    job.naive = naive ? 1u : 0u;
    if (c->hasFilm && !c->filmTablesSpectral) return c->fail(HPT_ERR_ARG, "thin film: m_precomp_thin_films was precomputed for RGB rendering (LoadScene sizes the tables by m_spectral_mode)");
    job.gens = c->dGens.p; job.packedXY = c->dPackedXY.p; job.packedCount = c->packedCount;
    // the scope the scene needs (hpt_spectral.hip: SCOPE): the narrowest instantiations that hold it
    bool heavy = c->S.lensCount != 0u || c->S.envTexId != 0xFFFFFFFFu || c->S.envCamBackId != 0xFFFFFFFFu || c->spectralHeavyMats;
    for (const uint g : c->hLightGeom) heavy = heavy || g == LIGHT_GEOM_ENV;
    const int scope = (heavy && c->fewMaterialTypes) ? 3 : (heavy ? 2 : (c->spectralGltfMats ? 1 : 0));
    // Schedule: scenes whose rays walk a real tree (SAH estimate >= 8: the reference's spectral fixture, interiors) run the block-local kernel -
    // persistent blocks on the work queue, rays repacked per block, the 4-wide tree on heavy scenes; the others, input-ray batches, moving
    // instances and sweep scenes keep the one-thread-per-pixel kernel. hpt_set_schedule(1) / (3) force either.
    const bool specBlock = !inRays && c->S.motion == 0u && c->S.sweep == 0u && (c->S.traceDepth > 0u || naive) &&
                           (c->schedule == 3 || (c->schedule == 0 && c->sahVisits >= BLOCK_SAH_VISITS));
    // ... and heavy scenes in calls with enough pixels the wavefront schedule (wfShadeSpecKernel + the shared trace kernel), as their RGB rendering does
    const bool specWave = !inRays && !naive && c->S.motion == 0u && c->S.sweep == 0u && c->S.traceDepth > 0u && !c->instrument &&
                          (c->schedule == 2 || (c->schedule == 0 && c->sahVisits >= HEAVY_SAH_VISITS && job.tidCount >= WF_AUTO_PIXELS));
    if (specWave) {
      c->lastSchedule = 2; c->lastWide = wfWide(c) ? 1u : 0u; c->lastDeep = (c->lastWide ? c->stackNeeded4 : c->stackNeeded) > (uint)LDS_STACK ? 1u : 0u; c->lastShadeRecords = 0u;
      return launch_wavefront(c, job, st, false, scope == 0 ? 1 : scope);
    }
    if (specBlock) {
      const bool bwide = c->S.megaWide != 0u && c->S.flatMode != 0u && c->nodes4Count != 0u;
      const bool bdeep = (bwide ? std::max(c->stackNeeded, c->stackNeeded4) : c->stackNeeded) > (uint)LDS_STACK;
      const int bpc = c->blocksPerCU > 0 ? c->blocksPerCU : (scope == 2 ? HPT_SPEC_WIDE_WAVES : HPT_SPEC_WAVES);
      const int blocks = (int)std::min<size_t>((size_t)c->numCUs * bpc, ((size_t)job.tidCount + 255) / 256);
      HIPCHK(c, c->dQueue.alloc(1));
      HIPCHK(c, hipMemsetAsync(c->dQueue.p, 0, 4, st));
      job.queue = c->dQueue.p;
      HIPCHK(c, ensureStackOverflow(c, (size_t)blocks * 256));
      job.stackOverflow = c->dStackOvf.p; job.gridLanes = (uint)blocks * 256u;
      const uint bNodeMin = bwide ? std::max(c->bwNodeMin, 16u) : c->bwNodeMin;
      c->lastSchedule = 3; c->lastWide = bwide ? 1u : 0u; c->lastDeep = bdeep ? 1u : 0u; c->lastShadeRecords = 0u;
      HIPCHK(c, hipEventRecord(c->ev0, st));
      if (scope == 3) launchSpectralBlock<3>(c->S, job, blocks, st, bdeep, bwide, c->bwRefillBelow, bNodeMin);
      else if (scope == 2) launchSpectralBlock<2>(c->S, job, blocks, st, bdeep, bwide, c->bwRefillBelow, bNodeMin);
      else if (scope == 1) launchSpectralBlock<1>(c->S, job, blocks, st, bdeep, bwide, c->bwRefillBelow, bNodeMin);
      else launchSpectralBlock<0>(c->S, job, blocks, st, bdeep, bwide, c->bwRefillBelow, bNodeMin);
      HIPCHK(c, hipGetLastError());
      HIPCHK(c, hipEventRecord(c->ev1, st));
      return HPT_OK;
    }
    const int sblocks = (int)(((size_t)job.tidCount + 255) / 256);
    HIPCHK(c, ensureStackOverflow(c, (size_t)sblocks * 256));
    job.stackOverflow = c->dStackOvf.p; job.gridLanes = (uint)sblocks * 256u;
    const bool sdeep = megaStackNeeded(c) > (uint)LDS_STACK;
    c->lastSchedule = 1; c->lastWide = (c->S.megaWide != 0u && c->S.flatMode != 0u && c->nodes4Count != 0u && c->S.motion == 0u) ? 1u : 0u; c->lastDeep = sdeep ? 1u : 0u; c->lastShadeRecords = 0u;
    HIPCHK(c, hipEventRecord(c->ev0, st));
    if (scope == 3) launchSpectral<3>(c->S, job, sblocks, st, sdeep);
    else if (scope == 2) launchSpectral<2>(c->S, job, sblocks, st, sdeep); else if (scope == 1) launchSpectral<1>(c->S, job, sblocks, st, sdeep); else launchSpectral<0>(c->S, job, sblocks, st, sdeep);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev1, st));
    return HPT_OK;
  }
  { const int rcS = ensureShadeTris(c, st); if (rcS != HPT_OK) return rcS; }
  const bool motion = c->S.motion != 0;
  const bool film = c->hasFilm;                                // MODE 4 / 5 / 6 kernels, wfShadeKernel<.., FILM>
  if (film && !c->filmTablesRGB) return c->fail(HPT_ERR_ARG, "thin film: m_precomp_thin_films holds no RGB table for a film (LoadScene precomputes every film in RGB mode, sized by its thickness map)");
  if (film && c->instrument && !dr) return c->fail(HPT_ERR_UNSUPPORTED, "thin films: the instrumented probe has no film variant");
  const bool fullMaterials = film || motion || !dr && !(c->leanMaterials && !c->forceFull && c->S.lensCount == 0u && !naive && !inRays && !(c->instrument && !dr));   // MODE 0 / 1 / 2 / STATS kernels
  // MODE 7: the kernel with every BSDF branch built for one more wave per SIMD, for scenes with few material types (hpt_decl.h)
  const bool mode7 = fullMaterials && c->fewMaterialTypes && !c->forceFull && !film && !motion && !dr && !inRays && !naive && !(c->instrument && !dr);
  const int blocks = (int)std::min<size_t>((size_t)gridBlocks(c, dr, fullMaterials && !mode7), ((size_t)job.tidCount + 255) / 256);
  HIPCHK(c, c->dQueue.alloc(1));
  HIPCHK(c, hipMemsetAsync(c->dQueue.p, 0, 4, st));
  job.queue = c->dQueue.p;
  job.gens = c->dGens.p; job.packedXY = c->dPackedXY.p; job.packedCount = c->packedCount;
  job.counters = nullptr;
  job.drSkipNonFinite = c->drSkipNonFinite ? 1u : 0u;
  const bool stats = c->instrument;                            // (the DR probe exists as a megakernel only: hpt_kernels.hip, group 15)
  c->lastSchedule = 1;
  c->lastShadeRecords = c->S.shadeTris != nullptr ? 1u : 0u;
  // schedule 4 (experimental, never chosen automatically): the block-owned streaming form of the wavefront schedule - heavy static scenes with gltf / emissive materials
  if (c->schedule == 4 && !inRays && !naive && !dr && !stats && !motion && !film && c->leanMaterials && !c->forceFull && c->S.lensCount == 0u && wfWide(c) && c->S.traceDepth > 0u) {
    c->lastSchedule = 4; c->lastWide = 1u; c->lastDeep = c->stackNeeded4 > (uint)LDS_STACK ? 1u : 0u;
    return launch_stream(c, job, st);
  }
  if (!inRays && useWavefront(c, naive, dr, stats && c->schedule != 2, job.tidCount)) {
    c->lastSchedule = 2; c->lastWide = (wfWide(c) && !c->instrument) ? 1u : 0u; c->lastDeep = (c->lastWide ? c->stackNeeded4 : c->stackNeeded) > (uint)LDS_STACK ? 1u : 0u;
    return launch_wavefront(c, job, st, dr);
  }
  c->lastWide = (!stats && !motion && (HPT_FLAT_WIDE || c->S.megaWide != 0u) && c->S.flatMode != 0u && c->nodes4Count != 0u) ? 1u : 0u;
  c->lastDeep = megaStackNeeded(c) > (uint)LDS_STACK ? 1u : 0u;
  if (stats) { HIPCHK(c, c->dCounters.alloc(1)); HIPCHK(c, hipMemsetAsync(c->dCounters.p, 0, sizeof(Counters), st)); job.counters = c->dCounters.p; }
  if (dr) {
    job.recordLanes = (uint)blocks * 256u;
    HIPCHK(c, c->dRecord.alloc((size_t)job.recordLanes * REC_FIELDS * (c->S.traceDepth + 1)));
    job.record = c->dRecord.p;
  }
  HIPCHK(c, ensureStackOverflow(c, (size_t)blocks * 256));
  job.stackOverflow = c->dStackOvf.p; job.gridLanes = (uint)blocks * 256u;
  HIPCHK(c, hipEventRecord(c->ev0, st));
  const bool deep = megaStackNeeded(c) > (uint)LDS_STACK;
  // schedule 3: the megakernel with block-local ray repacking (hpt_block.hip) - PathTrace and PathTraceDR on the BVH2 of either layout
  // automatic: gltf / emissive scenes (and PathTraceDR) whose rays walk a real tree - the test_228 class (SAH estimate 10: 696 -> 735 Mpaths/s at 1024^2,
  // 527 -> 630 at 512^2, PathTraceDR 428 -> 453) and heavy scenes in calls too small for the wavefront schedule (1 M triangles at 640 x 360: 145 -> 160);
  // the light fixtures with every BSDF branch stay on the plain megakernel (typed_materials, estimate 4.8: 1186 vs 1133) - profiles/bw_check.py, bw_heavy.py
  const bool bwLean = dr || (c->leanMaterials && !c->forceFull && c->S.lensCount == 0u);
  const bool bwAuto = c->schedule == 0 && bwLean && c->sahVisits >= BLOCK_SAH_VISITS;
  const bool blockLocal = (c->schedule == 3 || bwAuto) && !naive && !inRays && !motion && !film && !stats && c->S.sweep == 0u;
  if (blockLocal) {
    const bool lean = dr || (c->leanMaterials && !c->forceFull && c->S.lensCount == 0u);
    const bool bwide = lean && (c->bwWide == 1 || (c->bwWide < 0 && c->S.megaWide != 0u)) && c->S.flatMode != 0u && c->nodes4Count != 0u && c->wideEnabled;      // heavy scenes: the 4-wide compressed tree, as the megakernel walks it
    const bool bdeep = (bwide ? std::max(c->stackNeeded, c->stackNeeded4) : c->stackNeeded) > (uint)LDS_STACK;
    const uint bNodeMin = bwide ? std::max(c->bwNodeMin, 16u) : c->bwNodeMin;                           // (a 4-wide visit is three times the work: the vote pays earlier)
    c->lastSchedule = 3; c->lastWide = bwide ? 1u : 0u; c->lastDeep = bdeep ? 1u : 0u;
    if (dr) launchBlock<true, true>(c->S, job, blocks, st, bdeep, c->bwRefillBelow, bNodeMin, bwide);
    else if (lean) launchBlock<false, true>(c->S, job, blocks, st, bdeep, c->bwRefillBelow, bNodeMin, bwide);
    else launchBlock<false, false>(c->S, job, std::min(blocks, c->numCUs * (c->blocksPerCU > 0 ? c->blocksPerCU : HPT_BW_FULL_WAVES)), st, bdeep, c->bwRefillBelow, c->bwNodeMin);
  }
  else if (motion && film) {
    if (inRays) launchPTMotion<6>(c->S, job, blocks, st, deep); else if (naive) launchPTMotion<5>(c->S, job, blocks, st, deep); else launchPTMotion<4>(c->S, job, blocks, st, deep);
  }
  else if (motion) {
    if (inRays) launchPTMotion<2>(c->S, job, blocks, st, deep); else if (naive) launchPTMotion<1>(c->S, job, blocks, st, deep); else launchPTMotion<0>(c->S, job, blocks, st, deep);
  }
  else if (dr) {
    // the DR kernel's waves are emptier than the forward kernel's (lane utilisation 0.25), so leaving the inner-node loop when fewer than four
    // lanes still walk pays even on light scenes: test_228 class 346 -> 364 Mpaths/s (forward: 698 -> 699; profiles/vote_medium.sh)
    DevScene Sd = c->S;
    if (c->nodeMinOverride < 0 && Sd.nodeMin < 4u) Sd.nodeMin = 4u;
    if (stats) {                                                 // the counting probe follows the walk an uninstrumented call would do (see the forward probe below)
      Sd.statsWide = (wfWide(c) && (c->statsWide || useWavefront(c, naive, dr, false, job.tidCount))) ? 1u : 0u;
      launchPT<true, true, 0>(Sd, job, blocks, st, deep || (Sd.statsWide && c->stackNeeded4 > (uint)LDS_STACK));
    } else launchPT<false, true, 0>(Sd, job, blocks, st, deep);
  }
  else if (film) {
    if (inRays) launchPT<false, false, 6>(c->S, job, blocks, st, deep); else if (naive) launchPT<false, false, 5>(c->S, job, blocks, st, deep); else launchPT<false, false, 4>(c->S, job, blocks, st, deep);
  }
  else if (inRays) launchPT<false, false, 2>(c->S, job, blocks, st, deep);
  else if (naive)  launchPT<false, false, 1>(c->S, job, blocks, st, deep);
  else if (stats) {
    // the counting probe follows the walk an uninstrumented call would do: where that is the wavefront trace kernel on the 4-wide tree, the
    // probe's single-level traversal walks that tree too (node visits = 64-byte lines of the tree actually used)
    DevScene Sp = c->S;
    Sp.statsWide = (wfWide(c) && (c->statsWide || useWavefront(c, naive, dr, false, job.tidCount))) ? 1u : 0u;   // "stats_wide": the caller says the measured call ran there
    launchPT<true, false, 0>(Sp, job, blocks, st, deep || (Sp.statsWide && c->stackNeeded4 > (uint)LDS_STACK));
  }
  else if (c->leanMaterials && !c->forceFull && c->S.lensCount == 0u) launchPT<false, false, 3>(c->S, job, blocks, st, deep);
  else if (mode7)  launchPT<false, false, 7>(c->S, job, blocks, st, deep);
  else             launchPT<false, false, 0>(c->S, job, blocks, st, deep);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->ev1, st));
  return HPT_OK;
}


// ---- wavefront schedule ---------------------------------------------------------------------------------------------------------------
// Heavy scenes (HEAVY_SAH_VISITS) are rendered by the shade / trace kernel pair; on light ones the path state's trip through HBM costs more
// than the ray replacement gains (measured crossover: see DESIGN.md, "two schedules").
static const uint   WF_POOL_MAX = 1u << 22;        // pool slots (pixels in flight) per batch: 4M x 148 B = 620 MB
static const uint   WF_CHECK = 8;                  // progress word copied back every WF_CHECK shade passes
static const uint   WF_RING = 8;                   // ... and at most WF_RING such checkpoints in flight
static const uint   WF_GROUPS_AUTO = 1;            // pixel groups (streams) per call; measured 1M triangles: 1 -> 224, 2 -> 223, 3 -> 204 Mpaths/s

static bool useWavefront(hpt_ctx* c, bool naive, bool dr, bool stats, uint tidCount)
{
  if (naive || stats) return false;                // those variants exist as megakernels only
  if (c->schedule == 1 || c->schedule == 3) return false;
  if (c->schedule == 2) return true;                 // (4, where its kernel does not apply: the automatic choice)
  // ... and only for calls with enough pixels to keep the trace kernel's lanes supplied with replacement rays: measured on the 1M-triangle
  // scene (profiles/share.sh, 2.07 M / 1.04 M / 518 K / 259 K pixels per call): wavefront 226 / 199 / 162 / 108 vs megakernel 171 / 161 / 159 / 146 Mpaths/s
  return c->sahVisits >= HEAVY_SAH_VISITS && tidCount >= WF_AUTO_PIXELS;
}

static bool wfWide(const hpt_ctx* c) { return c->wideEnabled && c->S.flatMode != 0u && c->S.motion == 0u && c->nodes4Count != 0u; }
static uint wfStackNeeded(const hpt_ctx* c) { return std::max(std::max(c->stackNeeded, wfWide(c) ? c->stackNeeded4 : 0u), 1u); }   // entries a suspended ray may have to save

template <bool STATS>
static void launchWfTrace(hpt_ctx* c, const DevScene& S, const WfPool& P, uint iter, int blocks, hipStream_t st, bool deep, uint* ovf)
{
  const uint lanes = (uint)blocks * 256u; Counters* cn = c->dCounters.p;
  const uint grace = c->wfGrace;
  if (c->S.motion != 0u && !STATS) {                           // moving instances: the rays carry their path's time
    if (c->S.flatMode) {
      if (deep) wfTraceKernel<true, true, false, true><<<dim3(blocks), dim3(256), 0, st>>>(S, P, iter, c->wfRefillBelow, grace, ovf, lanes, cn);
      else      wfTraceKernel<false, true, false, true><<<dim3(blocks), dim3(256), 0, st>>>(S, P, iter, c->wfRefillBelow, grace, ovf, lanes, cn);
    } else {
      if (deep) wfTraceKernel<true, false, false, true><<<dim3(blocks), dim3(256), 0, st>>>(S, P, iter, c->wfRefillBelow, grace, ovf, lanes, cn);
      else      wfTraceKernel<false, false, false, true><<<dim3(blocks), dim3(256), 0, st>>>(S, P, iter, c->wfRefillBelow, grace, ovf, lanes, cn);
    }
    return;
  }
  if (wfWide(c)) {                                              // the 4-wide compressed tree (static single-level scenes)
    if (c->stackNeeded4 > (uint)LDS_STACK) wfTraceKernel<true, true, STATS, false, true><<<dim3(blocks), dim3(256), 0, st>>>(S, P, iter, c->wfRefillBelow, grace, ovf, lanes, cn);
    else                                   wfTraceKernel<false, true, STATS, false, true><<<dim3(blocks), dim3(256), 0, st>>>(S, P, iter, c->wfRefillBelow, grace, ovf, lanes, cn);
    return;
  }
  if (c->S.flatMode) {
    if (deep) wfTraceKernel<true, true, STATS><<<dim3(blocks), dim3(256), 0, st>>>(S, P, iter, c->wfRefillBelow, grace, ovf, lanes, cn);
    else      wfTraceKernel<false, true, STATS><<<dim3(blocks), dim3(256), 0, st>>>(S, P, iter, c->wfRefillBelow, grace, ovf, lanes, cn);
  } else {
    if (deep) wfTraceKernel<true, false, STATS><<<dim3(blocks), dim3(256), 0, st>>>(S, P, iter, c->wfRefillBelow, grace, ovf, lanes, cn);
    else      wfTraceKernel<false, false, STATS><<<dim3(blocks), dim3(256), 0, st>>>(S, P, iter, c->wfRefillBelow, grace, ovf, lanes, cn);
  }
}

static int launch_wavefront(hpt_ctx* c, const Job& job, hipStream_t st, bool dr, int specScope)   // specScope >= 0: spectral rendering (wfShadeSpecKernel<scope>)
{
  const bool spec = specScope >= 0;
  DevScene Sw = c->S;
  if (spec) Sw.shadeTris = nullptr;                  // the spectral shade kernel fetches its surface through the primitive id: the trace pass must report that
  auto launchShade = [&](const dim3 sg, hipStream_t gs, const WfPool& P, const WfJob& wj) {
    if (spec) {
      if (specScope == 3) wfShadeSpecKernel<3><<<sg, dim3(256), 0, gs>>>(Sw, P, wj);
      else if (specScope == 2) wfShadeSpecKernel<2><<<sg, dim3(256), 0, gs>>>(Sw, P, wj);
      else wfShadeSpecKernel<1><<<sg, dim3(256), 0, gs>>>(Sw, P, wj);
    }
    else if (dr)               wfShadeKernel<true, true><<<sg, dim3(256), 0, gs>>>(c->S, P, wj);
    else if (c->hasFilm)       { if (c->S.motion) wfShadeKernel<false, false, true, true><<<sg, dim3(256), 0, gs>>>(c->S, P, wj); else wfShadeKernel<false, false, false, true><<<sg, dim3(256), 0, gs>>>(c->S, P, wj); }
    else if (c->S.motion)      wfShadeKernel<false, false, true><<<sg, dim3(256), 0, gs>>>(c->S, P, wj);
    else if (c->leanMaterials && !c->forceFull && c->S.lensCount == 0u) wfShadeKernel<false, true><<<sg, dim3(256), 0, gs>>>(c->S, P, wj);
    else                       wfShadeKernel<false, false><<<sg, dim3(256), 0, gs>>>(c->S, P, wj);
  };
  // ---- groups: contiguous runs of work items, whole 256-item blocks each ----
  uint nGroups = c->wfGroupCount > 0 ? (uint)c->wfGroupCount : WF_GROUPS_AUTO;
  nGroups = std::max(nGroups, (job.tidCount + WF_POOL_MAX - 1) / WF_POOL_MAX);
  nGroups = std::min(nGroups, std::max(1u, job.tidCount / 4096u));             // tiny calls: one group
  nGroups = std::min(nGroups, 64u);
  const uint per = (((job.tidCount + nGroups - 1) / nGroups) + 255u) & ~255u;
  while (c->wfGroups.size() < nGroups) {
    hpt_ctx::WfGroup* g = new hpt_ctx::WfGroup();
    c->wfGroups.push_back(g);
    HIPCHK(c, hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    HIPCHK(c, hipHostMalloc((void**)&g->progress, WF_RING * sizeof(uint)));
    for (auto& e : g->ev) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&g->done, hipEventDisableTiming));
  }
  if (!c->wfFork) HIPCHK(c, hipEventCreateWithFlags(&c->wfFork, hipEventDisableTiming));

  int bpc = c->wfBlocksPerCU > 0 ? c->wfBlocksPerCU : HPT_WF_WAVES;
  const int traceBlocks = c->numCUs * bpc;
  // HBM part of the traversal stacks: one slice per group (their trace kernels run concurrently)
  HIPCHK(c, ensureStackOverflow(c, (size_t)traceBlocks * 256 * nGroups));
  const size_t ovfPerGroup = c->dStackOvf.n / nGroups;
  const bool deep = c->stackNeeded > (uint)LDS_STACK;
  const bool stats = c->instrument;
  if (stats) { HIPCHK(c, c->dCounters.alloc(1)); HIPCHK(c, hipMemsetAsync(c->dCounters.p, 0, sizeof(Counters), st)); }

  WfJob wj;
  wj.tidBegin = job.tidBegin; wj.tidChunk = job.tidChunk; wj.tidStride = job.tidStride; wj.tidEnd = job.tidEnd;
  wj.passNum = job.passNum; wj.channels = job.channels; wj.outColor = job.outColor; wj.gens = job.gens; wj.packedXY = job.packedXY;
  wj.drSkipNonFinite = job.drSkipNonFinite;
  wj.refImg = job.refImg; wj.data = job.data; wj.grad = job.grad; wj.lossAccum = job.lossAccum; wj.record = nullptr;
  // every path takes at most traceDepth shade passes after the one that generated it; the pass that ends it (or the next one, when a
  // shadow ray was outstanding) also generates the pixel's next path
  // safety net only (the loop ends when a round queues no ray): pixels whose rays were suspended sit rounds out, so there is no tight bound
  const unsigned long long iterCap = c->wfIterCap ? c->wfIterCap : 64ull * ((unsigned long long)job.passNum * (c->S.traceDepth + 2ull) + 4ull);
  c->lastWfIters = 0;
  unsigned long long capLeft = 0;                  // rays still queued when a group ran into iterCap
  if (dr) { HIPCHK(c, c->dLossAcc.alloc(1)); HIPCHK(c, hipMemsetAsync(c->dLossAcc.p, 0, sizeof(double), st)); }
  HIPCHK(c, hipEventRecord(c->ev0, st));
  HIPCHK(c, hipEventRecord(c->wfFork, st));

  std::vector<WfPool> pools(nGroups);
  uint live = 0;
  for (uint gi = 0; gi < nGroups; gi++) {
    hpt_ctx::WfGroup& g = *c->wfGroups[gi];
    g.itemBase = std::min(gi * per, job.tidCount); g.itemCount = std::min(per, job.tidCount - g.itemBase);
    g.it = 0; g.checkpoints = 0; g.finished = g.itemCount == 0;
    if (g.finished) continue;
    live++;
    for (int i = 0; i < 8; i++) HIPCHK(c, g.f4[i].alloc(g.itemCount));
    for (int i = 0; i < 3; i++) HIPCHK(c, g.u[i].alloc(g.itemCount));
    const size_t maxSusp = (size_t)traceBlocks * 256, suspWords = (WF_SUSP_WORDS + (size_t)wfStackNeeded(c)) * maxSusp;
    HIPCHK(c, g.u[3].alloc(2 * (size_t)g.itemCount + maxSusp)); HIPCHK(c, g.u[5].alloc(2 * (size_t)g.itemCount + maxSusp));
    HIPCHK(c, g.u[4].alloc(2 * WF_CTR_WORDS));
    HIPCHK(c, g.u[6].alloc(g.itemCount));
    HIPCHK(c, g.u[7].alloc(suspWords)); HIPCHK(c, g.u[8].alloc(suspWords));
    if (dr) { HIPCHK(c, g.rec.alloc((size_t)g.itemCount * REC_FIELDS * (c->S.traceDepth + 1))); HIPCHK(c, g.lossSlot.alloc(g.itemCount)); }
    if (c->S.motion) HIPCHK(c, g.time.alloc(g.itemCount));
    if (spec) { HIPCHK(c, g.waves.alloc(g.itemCount)); HIPCHK(c, g.fb.alloc(g.itemCount)); }
    WfPool& P = pools[gi];
    P.waves = spec ? g.waves.p : nullptr; P.fb = spec ? g.fb.p : nullptr;
    P.rayO = g.f4[0].p; P.rayD = g.f4[1].p; P.thr = g.f4[2].p; P.acc = g.f4[3].p;
    P.shO = g.f4[4].p; P.shD = g.f4[5].p; P.contrib = g.f4[6].p; P.hit = g.f4[7].p;
    P.hitInst = g.u[0].p; P.occl = g.u[1].p; P.lossSlot = dr ? g.lossSlot.p : nullptr; P.time = c->S.motion ? g.time.p : nullptr; P.status = g.u[2].p; P.rayQ[0] = g.u[3].p; P.rayQ[1] = g.u[5].p; P.ctr = g.u[4].p; P.inflight = g.u[6].p;
    P.susp[0] = g.u[7].p; P.susp[1] = g.u[8].p; P.maxSusp = (uint)maxSusp; P.suspStack = wfStackNeeded(c);
    HIPCHK(c, hipStreamWaitEvent(g.stream, c->wfFork, 0));
    HIPCHK(c, hipMemsetAsync(P.ctr, 0, 2 * WF_CTR_WORDS * sizeof(uint), g.stream));
    wfInitKernel<<<dim3((g.itemCount + 255u) / 256u), dim3(256), 0, g.stream>>>(P, g.itemCount, job.passNum);
  }
  // ---- rounds: one shade + trace pass per live group, groups interleaved so that their kernels overlap on the device ----
  while (live > 0) {
    for (uint gi = 0; gi < nGroups; gi++) {
      hpt_ctx::WfGroup& g = *c->wfGroups[gi];
      if (g.finished) continue;
      const WfPool& P = pools[gi];
      wj.itemBase = g.itemBase; wj.itemCount = g.itemCount; wj.iter = (uint)g.it; wj.record = g.rec.p;
      const dim3 sg((g.itemCount + 255u) / 256u);
      launchShade(sg, g.stream, P, wj);
      if ((g.it % WF_CHECK) == WF_CHECK - 1) {
        const uint slot = g.checkpoints % WF_RING;
        if (g.checkpoints >= WF_RING) {                                     // oldest checkpoint of the ring: wait for it, then look at it
          HIPCHK(c, hipEventSynchronize(g.ev[slot]));
          if (g.progress[slot] == 0u) g.finished = true;
        }
        if (!g.finished) {
          HIPCHK(c, hipMemcpyAsync(&g.progress[slot], P.ctr + WF_CTR_WORDS * (wj.iter & 1u), sizeof(uint), hipMemcpyDeviceToHost, g.stream));
          HIPCHK(c, hipEventRecord(g.ev[slot], g.stream));
          g.checkpoints++;
          for (uint k = 1; k < WF_RING && k < g.checkpoints; k++) {         // newer checkpoints that have already landed
            const uint sl = (g.checkpoints - 1 - k) % WF_RING;
            if (hipEventQuery(g.ev[sl]) == hipSuccess && g.progress[sl] == 0u) { g.finished = true; break; }
          }
        }
      }
      if (!g.finished) {
        uint* ovf = c->dStackOvf.p + gi * ovfPerGroup;
        if (stats) launchWfTrace<true>(c, Sw, P, wj.iter, traceBlocks, g.stream, deep, ovf); else launchWfTrace<false>(c, Sw, P, wj.iter, traceBlocks, g.stream, deep, ovf);
        g.it++;
        if (gi == 0) c->lastWfIters++;
        if (g.it >= iterCap) {
          // The safety net tripped. One more shade pass tells whether anything is left: it queues a ray for every path still alive and the
          // rays the last trace pass suspended are already counted in the same word. Work left = an incomplete frame: say so.
          wj.iter = (uint)g.it;
          launchShade(sg, g.stream, P, wj);
          uint left = 0;
          HIPCHK(c, hipMemcpyAsync(&left, P.ctr + WF_CTR_WORDS * (wj.iter & 1u), sizeof(uint), hipMemcpyDeviceToHost, g.stream));
          HIPCHK(c, hipStreamSynchronize(g.stream));
          if (left != 0u) capLeft += left;
          g.finished = true;
        }
      }
      if (g.finished) {
        live--;
        if (dr) wfLossReduceKernel<<<dim3(std::min((g.itemCount + 255u) / 256u, 1024u)), dim3(256), 0, g.stream>>>(pools[gi].lossSlot, g.itemCount, c->dLossAcc.p);
        HIPCHK(c, hipEventRecord(g.done, g.stream));
        HIPCHK(c, hipStreamWaitEvent(st, g.done, 0));
      }
    }
    HIPCHK(c, hipGetLastError());
  }
  if (dr) wfLossFinishKernel<<<dim3(1), dim3(1), 0, st>>>(c->dLossAcc.p, job.lossAccum);
  HIPCHK(c, hipEventRecord(c->ev1, st));
  if (capLeft != 0ull)
    return c->fail(HPT_ERR_STATE, "wavefront schedule: stopped after " + std::to_string(iterCap) + " rounds with " + std::to_string(capLeft) +
                   " rays still queued - the frame is incomplete (render it with hpt_set_schedule(ctx, 1, ...) or hpt_set_option(\"wf_grace\", 0))");
  return HPT_OK;
}

// ---- schedule 4: block-owned streaming (hpt_stream.hip) -----------------------------------------------------------------------------------
static int launch_stream(hpt_ctx* c, const Job& job, hipStream_t st)
{
  while (c->wfGroups.size() < 1) {
    hpt_ctx::WfGroup* g = new hpt_ctx::WfGroup();
    c->wfGroups.push_back(g);
    HIPCHK(c, hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    HIPCHK(c, hipHostMalloc((void**)&g->progress, WF_RING * sizeof(uint)));
    for (auto& e : g->ev) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&g->done, hipEventDisableTiming));
  }
  hpt_ctx::WfGroup& g = *c->wfGroups[0];
  const int bpc = c->wfBlocksPerCU > 0 ? c->wfBlocksPerCU : HPT_STREAM_WAVES;
  const uint blocks = (uint)std::max(1, std::min(c->numCUs * bpc, (int)((job.tidCount + 255u) / 256u)));
  uint spb = (job.tidCount + blocks - 1u) / blocks;
  spb = (spb + 255u) & ~255u;                                   // slots per block: whole 256-slot steps of the shade loop
  const uint n = job.tidCount;
  for (int i = 0; i < 8; i++) HIPCHK(c, g.f4[i].alloc(n));
  for (int i = 0; i < 3; i++) HIPCHK(c, g.u[i].alloc(n));
  HIPCHK(c, g.u[6].alloc(n));
  HIPCHK(c, g.u[3].alloc((size_t)blocks * 2u * spb));            // the blocks' ray queues
  HIPCHK(c, ensureStackOverflow(c, (size_t)blocks * 256));
  WfPool P; std::memset(&P, 0, sizeof(P));
  P.rayO = g.f4[0].p; P.rayD = g.f4[1].p; P.thr = g.f4[2].p; P.acc = g.f4[3].p;
  P.shO = g.f4[4].p; P.shD = g.f4[5].p; P.contrib = g.f4[6].p; P.hit = g.f4[7].p;
  P.hitInst = g.u[0].p; P.occl = g.u[1].p; P.status = g.u[2].p; P.inflight = g.u[6].p;
  WfJob wj; std::memset(&wj, 0, sizeof(wj));
  wj.itemBase = 0; wj.itemCount = n; wj.tidBegin = job.tidBegin; wj.tidChunk = job.tidChunk; wj.tidStride = job.tidStride; wj.tidEnd = job.tidEnd;
  wj.passNum = job.passNum; wj.channels = job.channels; wj.outColor = job.outColor; wj.gens = job.gens; wj.packedXY = job.packedXY;
  HIPCHK(c, hipEventRecord(c->ev0, st));
  wfInitKernel<<<dim3((n + 255u) / 256u), dim3(256), 0, st>>>(P, n, job.passNum);
  if (c->stackNeeded4 > (uint)LDS_STACK) streamKernel<true><<<dim3(blocks), dim3(256), 0, st>>>(c->S, P, wj, spb, c->wfRefillBelow, g.u[3].p, c->dStackOvf.p, blocks * 256u);
  else                                   streamKernel<false><<<dim3(blocks), dim3(256), 0, st>>>(c->S, P, wj, spb, c->wfRefillBelow, g.u[3].p, c->dStackOvf.p, blocks * 256u);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->ev1, st));
  c->lastWfIters = 0;
  return HPT_OK;
}

extern "C" int hpt_path_trace_block_dev(hpt_ctx* c, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* outDev, uint32_t passNum, int naive, void* stream)
try {
  if (!c || !outDev) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (tidCount == 0 || passNum == 0) return HPT_OK;
  Job job; std::memset(&job, 0, sizeof(job));
  job.tidBegin = tidBegin; job.tidCount = tidCount; job.passNum = passNum; job.channels = channels; job.outColor = outDev;
  return launch_path_trace(c, job, naive != 0, false, (hipStream_t)stream);
}
catch (...) { return hptGuard(c, "hpt_path_trace_block_dev"); }

static int path_trace_host(hpt_ctx* c, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out, uint32_t passNum, int naive)
{
  if (!c || !out) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (!c->paramsSet) return c->fail(HPT_ERR_STATE, "PathTraceBlock before UpdateMembersPlainData");
  float* slots = naive ? c->tNaive : c->tPathTrace;
  const size_t n = (size_t)c->S.winWidth * c->S.winHeight * channels;
  const double t0 = now_ms();
  HIPCHK(c, c->dFrame.alloc(n));
  HIPCHK(c, hipMemcpy(c->dFrame.p, out, n * 4, hipMemcpyHostToDevice));            // the callee ACCUMULATES into the caller's buffer
  const double t1 = now_ms();
  int rc = hpt_path_trace_block_dev(c, tidBegin, tidCount, channels, c->dFrame.p, passNum, naive, nullptr);
  if (rc) return rc;
  HIPCHK(c, hipDeviceSynchronize());
  const double t2 = now_ms();
  HIPCHK(c, hipMemcpy(out, c->dFrame.p, n * 4, hipMemcpyDeviceToHost));
  const double t3 = now_ms();
  float kms = 0.0f;
  if (tidCount && passNum) (void)hipEventElapsedTime(&kms, c->ev0, c->ev1);
  c->lastKernelMs = kms;
  slots[0] = kms; slots[1] = float(t1 - t0); slots[2] = float(t3 - t2); slots[3] = float((t2 - t1) - kms);
  return HPT_OK;
}

// PathTraceFromInputRaysBlock (integrator_pt.h:261, integrator_pt_host.cpp:92-103): device and host-pointer forms
extern "C" int hpt_path_trace_from_input_rays_block_dev(hpt_ctx* c, uint32_t tid, uint32_t channels, const float* rayPosDev, const float* rayDirDev, float* outDev, uint32_t passNum, void* stream)
try {
  if (!c || !rayPosDev || !rayDirDev || !outDev) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (tid == 0 || passNum == 0) return HPT_OK;
  Job job; std::memset(&job, 0, sizeof(job));
  job.tidBegin = 0; job.tidCount = tid; job.passNum = passNum; job.channels = channels; job.outColor = outDev;
  job.inRayPos = (const float4*)rayPosDev; job.inRayDir = (const float4*)rayDirDev;
  return launch_path_trace(c, job, false, false, (hipStream_t)stream);
}
catch (...) { return hptGuard(c, "hpt_path_trace_from_input_rays_block_dev"); }
extern "C" int hpt_path_trace_from_input_rays_block(hpt_ctx* c, uint32_t tid, uint32_t channels, const float* rayPos, const float* rayDir, float* out, uint32_t passNum)
try {
  if (!c || !rayPos || !rayDir || !out) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (tid == 0 || passNum == 0) return HPT_OK;
  DevBuf<float> dp, dd, dout;
  const size_t n = (size_t)tid * channels;
  const double t0 = now_ms();
  HIPCHK(c, dp.upload(rayPos, (size_t)tid * 4)); HIPCHK(c, dd.upload(rayDir, (size_t)tid * 4)); HIPCHK(c, dout.upload(out, n));
  const double t1 = now_ms();
  int rc = hpt_path_trace_from_input_rays_block_dev(c, tid, channels, dp.p, dd.p, dout.p, passNum, nullptr);
  if (rc == HPT_OK) { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) rc = c->hipFail(e, "hipDeviceSynchronize"); }
  const double t2 = now_ms();
  if (rc == HPT_OK) { hipError_t e = hipMemcpy(out, dout.p, n * sizeof(float), hipMemcpyDeviceToHost); if (e != hipSuccess) rc = c->hipFail(e, "hipMemcpy"); }
  const double t3 = now_ms();
  if (rc == HPT_OK) {                                                      // fromRaysPtTime (integrator_pt_host.cpp:92-103) in GetExecutionTime's four slots
    float kms = 0.0f; (void)hipEventElapsedTime(&kms, c->ev0, c->ev1);
    c->lastKernelMs = kms;
    c->tFromRays[0] = kms; c->tFromRays[1] = float(t1 - t0); c->tFromRays[2] = float(t3 - t2); c->tFromRays[3] = float((t2 - t1) - kms);
  }
  dp.release(); dd.release(); dout.release();
  return rc;
}
catch (...) { return hptGuard(c, "hpt_path_trace_from_input_rays_block"); }

extern "C" int hpt_path_trace_block(hpt_ctx* c, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out, uint32_t passNum)
try { return path_trace_host(c, tidBegin, tidCount, channels, out, passNum, 0); }
catch (...) { return hptGuard(c, "hpt_path_trace_block"); }
extern "C" int hpt_naive_path_trace_block(hpt_ctx* c, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out, uint32_t passNum)
try { return path_trace_host(c, tidBegin, tidCount, channels, out, passNum, 1); }
catch (...) { return hptGuard(c, "hpt_naive_path_trace_block"); }

// ---- differentiable rendering ---------------------------------------------------------------------------------------------------------
extern "C" int hpt_reset_diff_tex(hpt_ctx* c)
try {
  if (!c) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  for (TexRec& t : c->hTextures) { t.diffOffset = ~0ull; t.diffW = t.diffH = t.diffChannels = 0; }
  c->gradSize = 0;
  if (!c->hTextures.empty()) HIPCHK(c, c->dTextures.upload(c->hTextures.data(), c->hTextures.size()));
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_reset_diff_tex"); }

extern "C" int hpt_put_diff_tex2d(hpt_ctx* c, uint32_t texId, uint32_t w, uint32_t h, uint32_t channels, uint64_t* outOffset, uint64_t* outSize)
try {
  if (!c || !outOffset || !outSize) return HPT_ERR_ARG;
  if (texId >= c->hTextures.size()) {                                     // reference: prints and returns (size_t(-1), 0) (integrator_dr.cpp:35-39)
    *outOffset = ~0ull; *outSize = 0;
    return c->fail(HPT_ERR_ARG, "[IntegratorDR::PutDiffTex2D]: bad tex id = " + std::to_string(texId));
  }
  if (channels != 1 && channels != 4) return c->fail(HPT_ERR_ARG, "PutDiffTex2D: channels must be 1 or 4");
  if (channels == 4 && (c->gradSize % 4) != 0) return c->fail(HPT_ERR_ARG, "PutDiffTex2D: 4-channel textures must start at a multiple of 4 floats");
  (void)hipSetDevice(c->device);
  TexRec& t = c->hTextures[texId];
  t.diffOffset = c->gradSize; t.diffW = w; t.diffH = h; t.diffChannels = channels;
  const uint64_t sz = (uint64_t)w * h * channels;
  if (c->gradSize + sz > (uint64_t(1) << 31)) return c->fail(HPT_ERR_UNSUPPORTED, "PutDiffTex2D: more than 2^31 parameters (the gradient scatter indexes a_dataGrad with 31 bits)");
  *outOffset = c->gradSize; *outSize = sz;
  c->gradSize += sz;
  HIPCHK(c, c->dTextures.upload(c->hTextures.data(), c->hTextures.size()));
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_put_diff_tex2d"); }

extern "C" int hpt_path_trace_dr_dev(hpt_ctx* c, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* outDev, uint32_t passNum,
                                     const float* refDev, const float* dataDev, float* gradDev, size_t gradSize, float* lossDev, void* stream)
try {
  if (!c || !outDev || !refDev || !dataDev || !gradDev || !lossDev) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (gradSize < c->gradSize) return c->fail(HPT_ERR_ARG, "PathTraceDR: a_gradSize smaller than the registered differentiable textures");
  if (channels < 3) return c->fail(HPT_ERR_ARG, "PathTraceDR: channels must be 3 or 4");
  if (tidCount == 0 || passNum == 0) return HPT_OK;
  Job job; std::memset(&job, 0, sizeof(job));
  job.tidBegin = tidBegin; job.tidCount = tidCount; job.passNum = passNum; job.channels = channels; job.outColor = outDev;
  job.refImg = refDev; job.data = dataDev; job.grad = gradDev; job.lossAccum = lossDev;
  return launch_path_trace(c, job, false, true, (hipStream_t)stream);
}
catch (...) { return hptGuard(c, "hpt_path_trace_dr_dev"); }

extern "C" int hpt_path_trace_dr(hpt_ctx* c, uint32_t tidBegin, uint32_t tidCount, uint32_t channels, float* out, uint32_t passNum,
                                 const float* refImg, const float* data, float* dataGrad, size_t gradSize, float* outLoss)
try {
  if (!c || !out || !refImg || !data || !dataGrad) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (!c->paramsSet) return c->fail(HPT_ERR_STATE, "PathTraceDR before UpdateMembersPlainData");
  const size_t n = (size_t)c->S.winWidth * c->S.winHeight * channels;
  const double t0 = now_ms();
  HIPCHK(c, c->dFrame.alloc(n)); HIPCHK(c, c->dRef.alloc(n)); HIPCHK(c, c->dData.alloc(gradSize)); HIPCHK(c, c->dGrad.alloc(gradSize)); HIPCHK(c, c->dLoss.alloc(1));
  HIPCHK(c, hipMemcpy(c->dFrame.p, out, n * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->dRef.p, refImg, n * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->dData.p, data, gradSize * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemset(c->dGrad.p, 0, gradSize * 4));                       // memset(a_dataGrad, 0, ...) (integrator_dr.cpp:1139)
  HIPCHK(c, hipMemset(c->dLoss.p, 0, 4));
  const double t1 = now_ms();
  int rc = hpt_path_trace_dr_dev(c, tidBegin, tidCount, channels, c->dFrame.p, passNum, c->dRef.p, c->dData.p, c->dGrad.p, gradSize, c->dLoss.p, nullptr);
  if (rc) return rc;
  HIPCHK(c, hipDeviceSynchronize());
  const double t2 = now_ms();
  float lossSum = 0.0f;
  HIPCHK(c, hipMemcpy(out, c->dFrame.p, n * 4, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(dataGrad, c->dGrad.p, gradSize * 4, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(&lossSum, c->dLoss.p, 4, hipMemcpyDeviceToHost));
  const double t3 = now_ms();
  if (outLoss) *outLoss = lossSum / float(c->S.winWidth * c->S.winHeight);  // avgLoss /= W*H (integrator_dr.cpp:1206)
  float kms = 0.0f;
  if (tidCount && passNum) (void)hipEventElapsedTime(&kms, c->ev0, c->ev1);
  c->lastKernelMs = kms;
  c->tDR[0] = kms; c->tDR[1] = float(t1 - t0); c->tDR[2] = float(t3 - t2); c->tDR[3] = float((t2 - t1) - kms);
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_path_trace_dr"); }

extern "C" int hpt_adam_step_dev(hpt_ctx* c, float* state, const float* grad, float* momentum, float* gsq, size_t n, int iter, void* stream)
try {
  if (!c || !state || !grad || !momentum || !gsq) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (n == 0) return HPT_OK;
  const float gamma = 0.25f / float(iter / 100 + 1);
  const int blocks = (int)std::min<size_t>((n + 255) / 256, (size_t)c->numCUs * 8);
  adamStepKernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(state, grad, momentum, gsq, n, gamma);
  HIPCHK(c, hipGetLastError());
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_adam_step_dev"); }

extern "C" int hpt_image2d4f_regularizer_dev(hpt_ctx* c, int w, int h, const float* data, float* grad, void* stream)
try {
  if (!c || !data || !grad || w < 0 || h < 0) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (w < 3 || h < 3) return HPT_OK;                                  // no interior pixel: the loss is identically zero
  image2D4fRegularizerKernel<<<dim3((w + 15) / 16, (h + 15) / 16), dim3(16, 16), 0, (hipStream_t)stream>>>(w, h, (const float4*)data, (float4*)grad);
  HIPCHK(c, hipGetLastError());
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_image2d4f_regularizer_dev"); }

extern "C" int hpt_image2d4f_regularizer(hpt_ctx* c, int w, int h, const float* data, float* grad)
try {
  if (!c || !data || !grad || w < 0 || h < 0) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  const size_t n = (size_t)w * h * 4;
  if (n == 0) return HPT_OK;
  DevBuf<float> dd, dg;
  HIPCHK(c, dd.upload(data, n)); HIPCHK(c, dg.upload(grad, n));
  int rc = hpt_image2d4f_regularizer_dev(c, w, h, dd.p, dg.p, nullptr);
  if (rc == HPT_OK) { hipError_t e = hipMemcpy(grad, dg.p, n * sizeof(float), hipMemcpyDeviceToHost); if (e != hipSuccess) rc = c->hipFail(e, "hipMemcpy"); }
  dd.release(); dg.release();
  return rc;
}
catch (...) { return hptGuard(c, "hpt_image2d4f_regularizer"); }

// ---- multi-GPU: RCCL collectives behind the C ABI (one context = one GPU = one rank) ---------------------------------------------------
// The path shards without any data-path exchange; what has to cross xGMI is one reduce(SUM) of the framebuffer per frame and,
// for PathTraceDR, one all_reduce(SUM) of a_dataGrad (and the loss) per optimisation iteration. librccl is dlopen'ed here instead of being
// linked, so that a process which already carries an RCCL (PyTorch bundles one) keeps using that copy.
namespace {
struct Id128 { char internal[128]; };         // ncclUniqueId (rccl.h:40-43), passed by value
struct RcclApi
{
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
const int RCCL_FLOAT32 = 7, RCCL_SUM = 0;       // ncclFloat32, ncclSum (rccl.h:448-466)

int loadRccl(hpt_ctx* c)
{
  if (g_rccl.AllReduce) return HPT_OK;
  void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return c->fail(HPT_ERR_UNSUPPORTED, std::string("hpt_comm: cannot load librccl: ") + dlerror());
  c->rcclLib = lib;
  *(void**)&g_rccl.GetUniqueId = dlsym(lib, "ncclGetUniqueId");
  *(void**)&g_rccl.CommInitRank = dlsym(lib, "ncclCommInitRank");
  *(void**)&g_rccl.CommDestroy = dlsym(lib, "ncclCommDestroy");
  *(void**)&g_rccl.Reduce = dlsym(lib, "ncclReduce");
  *(void**)&g_rccl.AllReduce = dlsym(lib, "ncclAllReduce");
  *(void**)&g_rccl.GetErrorString = dlsym(lib, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.Reduce || !g_rccl.AllReduce) { g_rccl = RcclApi(); return c->fail(HPT_ERR_UNSUPPORTED, "hpt_comm: librccl lacks the expected entry points"); }
  return HPT_OK;
}
int rcclFail(hpt_ctx* c, int r, const char* what) { return c->fail(HPT_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error")); }
} // namespace

extern "C" int hpt_comm_get_unique_id(hpt_ctx* c, void* id128)
try {
  if (!c || !id128) return HPT_ERR_ARG;
  int rc = loadRccl(c); if (rc) return rc;
  const int r = g_rccl.GetUniqueId(id128);
  return r ? rcclFail(c, r, "ncclGetUniqueId") : HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_comm_get_unique_id"); }
extern "C" int hpt_comm_init(hpt_ctx* c, int nranks, int rank, const void* id128)
try {
  if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return HPT_ERR_ARG;
  if (c->comm) return c->fail(HPT_ERR_STATE, "hpt_comm_init: communicator already initialised");
  int rc = loadRccl(c); if (rc) return rc;
  (void)hipSetDevice(c->device);
  Id128 id; std::memcpy(&id, id128, sizeof(id));
  const int r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
  if (r) { c->comm = nullptr; return rcclFail(c, r, "ncclCommInitRank"); }
  c->commRanks = nranks; c->commRank = rank;
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_comm_init"); }
extern "C" int hpt_comm_destroy(hpt_ctx* c)
try {
  if (!c) return HPT_ERR_ARG;
  if (c->comm) { (void)hipSetDevice(c->device); (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; c->commRanks = 0; }
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_comm_destroy"); }
extern "C" int hpt_reduce_framebuffer(hpt_ctx* c, float* frameDev, size_t count, int root, void* stream)
try {
  if (!c || !frameDev) return HPT_ERR_ARG;
  if (!c->comm) return c->fail(HPT_ERR_STATE, "hpt_reduce_framebuffer before hpt_comm_init");
  (void)hipSetDevice(c->device);
  const int r = g_rccl.Reduce(frameDev, frameDev, count, RCCL_FLOAT32, RCCL_SUM, root, c->comm, (hipStream_t)stream);
  return r ? rcclFail(c, r, "ncclReduce") : HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_reduce_framebuffer"); }
extern "C" int hpt_allreduce_grad(hpt_ctx* c, float* gradDev, size_t count, void* stream)
try {
  if (!c || !gradDev) return HPT_ERR_ARG;
  if (!c->comm) return c->fail(HPT_ERR_STATE, "hpt_allreduce_grad before hpt_comm_init");
  (void)hipSetDevice(c->device);
  const int r = g_rccl.AllReduce(gradDev, gradDev, count, RCCL_FLOAT32, RCCL_SUM, c->comm, (hipStream_t)stream);
  return r ? rcclFail(c, r, "ncclAllReduce") : HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_allreduce_grad"); }

// ---- timing / instrumentation --------------------------------------------------------------------------------------------------------
extern "C" int hpt_get_execution_time(hpt_ctx* c, const char* name, float out[4])
try {
  if (!c || !name || !out) return HPT_ERR_ARG;
  const std::string n(name);
  const float* src = nullptr;
  if (n == "PathTrace" || n == "PathTraceBlock") src = c->tPathTrace;                       // integrator_pt_lgt.cpp:241-251
  else if (n == "NaivePathTrace" || n == "NaivePathTraceBlock") src = c->tNaive;
  else if (n == "PathTraceFromInputRays" || n == "PathTraceFromInputRaysBlock") src = c->tFromRays;
  else if (n == "PathTraceDR" || n == "PathTraceDRBlock") src = c->tDR;                     // integrator_dr2.cpp:82-88
  if (!src) return HPT_OK;                                                                 // unknown names leave `out` untouched, as the reference does
  for (int i = 0; i < 4; i++) out[i] = src[i];
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_get_execution_time"); }
extern "C" int hpt_set_instrumentation(hpt_ctx* c, int enabled) try { if (!c) return HPT_ERR_ARG; c->instrument = enabled != 0; return HPT_OK; }
catch (...) { return hptGuard(c, "hpt_set_instrumentation"); }
extern "C" int hpt_get_counters(hpt_ctx* c, uint64_t out[16])
try {
  if (!c || !out) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (!c->dCounters.p) { for (int i = 0; i < 16; i++) out[i] = 0; return HPT_OK; }
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(out, c->dCounters.p, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_get_counters"); }
extern "C" int hpt_get_dr_counters(hpt_ctx* c, uint64_t out[16])
try {
  if (!c || !out) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (!c->dCounters.p) { for (int i = 0; i < 16; i++) out[i] = 0; return HPT_OK; }
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, hipMemcpy(out, (const uint64_t*)c->dCounters.p + 16, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_get_dr_counters"); }
extern "C" int hpt_set_tid_interleave(hpt_ctx* c, uint32_t chunk, uint32_t stride)
try {
  if (!c || (stride > 1 && (chunk == 0 || chunk % 64 != 0))) return HPT_ERR_ARG;
  c->tidChunk = chunk; c->tidStride = stride ? stride : 1;
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_set_tid_interleave"); }
extern "C" int hpt_set_launch_config(hpt_ctx* c, int blocksPerCU) try { if (!c || blocksPerCU < 0 || blocksPerCU > 8) return HPT_ERR_ARG; c->blocksPerCU = blocksPerCU; return HPT_OK; }
catch (...) { return hptGuard(c, "hpt_set_launch_config"); }
extern "C" int hpt_set_schedule(hpt_ctx* c, int schedule, int refillBelow, int traceBlocksPerCU, int groups)
try {
  if (!c || schedule < 0 || schedule > 4 || refillBelow < 0 || refillBelow > 64 || traceBlocksPerCU < 0 || traceBlocksPerCU > 8 || groups < 0 || groups > 64) return HPT_ERR_ARG;
  c->wfGroupCount = groups;
  c->schedule = schedule;
  if (refillBelow > 0) c->wfRefillBelow = (uint)refillBelow;
  c->wfBlocksPerCU = traceBlocksPerCU;
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_set_schedule"); }
extern "C" int hpt_set_option(hpt_ctx* c, const char* name, int value)
try {
  if (!c || !name || value < 0) return HPT_ERR_ARG;
  const std::string k(name);
  if (k == "wf_grace") c->wfGrace = (uint)value;                                      // trips after the queue ran dry before a trace wave suspends its rays (0: never)
  else if (k == "node_min") { c->nodeMinOverride = value & 63; c->accelCommitted = false; c->flatRefittable = false; }
  else if (k == "refit") c->refitEnabled = value != 0;                                  // 0: UpdateInstance / UpdateGeom + CommitScene always rebuild the tree on the host   // voted exit of the inner-node loop; takes effect at the next CommitScene
  else if (k == "dbg_no_normal_lerp") { if (c->S.motion) c->S.motion = value ? 3u : 1u; }   // diagnostic: moving instances without the reference's normal interpolation (bit 1 of DevScene::motion)
  else if (k == "dbg_wf_iter_cap") c->wfIterCap = (uint)value;                         // diagnostic: make the wavefront loop's safety net reachable in a test
  else if (k == "dr_skip_nonfinite") c->drSkipNonFinite = value != 0;                  // PathTraceDR: drop samples whose radiance is not finite (default 0: PixelLossPT as in the reference)
  else if (k == "shade_records") c->shadeTrisEnabled = value != 0;                     // 0: no DevScene::shadeTris (A/B, diagnosis)
  else if (k == "build_threads") c->buildThreads = std::min(value, 64);                // host threads CommitScene builds its trees with (0: the usable cores, at most 16)
  else if (k == "stats_wide") c->statsWide = value != 0;
  else if (k == "bw_refill_below") { if (value < 1 || value > 64) return c->fail(HPT_ERR_ARG, "bw_refill_below: 1..64"); c->bwRefillBelow = (uint)value; }
  else if (k == "bw_wide") { if (value < -1 || value > 1) return c->fail(HPT_ERR_ARG, "bw_wide: -1 automatic, 0, 1"); c->bwWide = value; }
  else if (k == "bw_node_min") { if (value < 0 || value > 64) return c->fail(HPT_ERR_ARG, "bw_node_min: 0..64"); c->bwNodeMin = (uint)value; }
  else if (k == "device_build") { if (value < -1 || value > 1) return c->fail(HPT_ERR_ARG, "device_build: -1 by CommitScene's options, 0 never, 1 always"); c->deviceBuild = value; c->accelCommitted = false; c->flatRefittable = false; }
  else if (k == "wide_nodes") { c->wideEnabled = value != 0; c->S.megaWide = (c->wideEnabled && c->nodes4Count != 0u && c->S.flatMode != 0u && c->sahVisits >= HEAVY_SAH_VISITS) ? 1u : 0u; }   // both users of the tree, at once                             // 0: the wavefront trace kernel walks the BVH2 instead of the 4-wide compressed tree (A/B, diagnosis)
  else if (k == "force_full_materials") c->forceFull = value != 0;                     // diagnostic: never pick the lean (gltf + emissive) kernels
  else return c->fail(HPT_ERR_ARG, "hpt_set_option: unknown option " + k);
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_set_option"); }
extern "C" int hpt_get_accel_info(hpt_ctx* c, float out[4])
try {
  if (!c || !out) return HPT_ERR_ARG;
  out[0] = c->sahVisits; out[1] = (float)c->instTris; out[2] = (float)c->insts.size(); out[3] = c->S.sweep ? 2.0f : (c->S.flatMode ? 1.0f : 0.0f);
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_get_accel_info"); }
extern "C" int hpt_get_commit_time(hpt_ctx* c, float out[4])
try {
  if (!c || !out) return HPT_ERR_ARG;
  for (int i = 0; i < 4; i++) out[i] = c->tCommit[i];
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_get_commit_time"); }
extern "C" int hpt_get_schedule(hpt_ctx* c, int* lastSchedule, uint32_t* lastIterations)
try {
  if (!c) return HPT_ERR_ARG;
  if (lastSchedule) *lastSchedule = (int)c->lastSchedule;
  if (lastIterations) *lastIterations = c->lastWfIters;
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_get_schedule"); }
extern "C" int hpt_get_last_launch(hpt_ctx* c, uint32_t out[4])
try {
  if (!c || !out) return HPT_ERR_ARG;
  out[0] = c->lastSchedule; out[1] = c->lastWide; out[2] = c->lastShadeRecords; out[3] = c->lastDeep;
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_get_last_launch"); }
extern "C" int hpt_set_accel_layout(hpt_ctx* c, int layout)
try {
  if (!c || layout < 0 || layout > 3) return HPT_ERR_ARG;
  c->accelLayout = layout; c->accelCommitted = false; c->flatRefittable = false;
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_set_accel_layout"); }
extern "C" int hpt_last_kernel_ms(hpt_ctx* c, float* ms)
try {
  if (!c || !ms) return HPT_ERR_ARG;
  (void)hipSetDevice(c->device);
  float k = 0.0f;
  hipError_t e = hipEventSynchronize(c->ev1);
  if (e == hipSuccess) e = hipEventElapsedTime(&k, c->ev0, c->ev1);
  if (e != hipSuccess) { *ms = c->lastKernelMs; return HPT_OK; }
  *ms = c->lastKernelMs = k;
  return HPT_OK;
}
catch (...) { return hptGuard(c, "hpt_last_kernel_ms"); }
