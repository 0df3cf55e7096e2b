// Plain-data layouts shared by the host side and the gfx950 kernels.
//
// Material / LightSource keep the reference's 320-byte records (include/cmaterial.h:187-203,
// include/clight.h:19-56) so that Integrator::m_materials / m_lights upload with one memcpy and the
// Update_m_materials / Update_m_lights partial-update hooks stay trivial.
#pragma once
#include <stdint.h>
#include <math.h>

namespace hpt {

typedef unsigned int uint;

struct MaterialRec
{
  uint  mtype, cflags, lightId, nonlinear;
  uint  texid[4], spdid[4], datai[4];
  float colors[4][4];
  float row0[4][4];
  float row1[4][4];
  float data[16];
};
static_assert(sizeof(MaterialRec) == 320, "Material record must stay 320 bytes");

struct LightRec
{
  float matrix[16], iesMatrix[16];               // column-major
  float samplerRow0[4], samplerRow1[4], samplerRow0Inv[4], samplerRow1Inv[4];
  float pos[4], intensity[4], norm[4];
  float size[2]; float pdfA; uint geomType;
  uint  distType, flags, pdfTableOffset, pdfTableSize;
  uint  specId, texId, iesId; float mult;
  uint  pdfTableSizeX, pdfTableSizeY, camBackTexId; float lightCos1;
  float lightCos2; uint matId; float dummy2, dummy3;
};
static_assert(sizeof(LightRec) == 320, "LightSource record must stay 320 bytes");

// ---- BVH2 in HBM ------------------------------------------------------------------------------------------------
// One 64-byte node holds BOTH child boxes, so one visit = four 16-byte loads from one 64-byte line:
//   q0 = (lo0.x lo0.y lo0.z hi0.x)  q1 = (hi0.y hi0.z lo1.x lo1.y)  q2 = (lo1.z hi1.x hi1.y hi1.z)  q3 = (ref0 ref1 - -)
// A child reference is 32 bits:
//   bit 31 clear : inner node, value = node index (TLAS and all BLAS share one array)
//   bit 31 set   : leaf; bits 30..28 = triangle count (1..4) and bits 27..0 = first triangle (global, BVH order);
//                  count 0 = instance leaf, bits 27..0 = instance id; 0xFFFFFFFF = "leave instance" stack marker.
static const uint REF_LEAF    = 0x80000000u;
static const uint REF_RESTORE = 0xFFFFFFFFu;
static const uint REF_NONE    = 0xFFFFFFFEu;   // empty scene
#ifndef HPT_BVH_LEAF_MAX
#define HPT_BVH_LEAF_MAX 2
#endif
static const int  BVH_LEAF_MAX = HPT_BVH_LEAF_MAX;   // measured on MI355X: 4 -> 1371, 2 -> 1800, 1 -> 1676 Mpaths/s (Cornell, round 1); 1M triangles on the 4-wide tree: 1 / 2 / 3 / 4 -> 280 / 288 / 281 / 273

struct BvhNode { float q[12]; uint ref0, ref1, pad0, pad1; };   // q = child 0 {lo.x hi.x lo.y hi.y lo.z hi.z}, child 1 {same}: (lo, hi) pairs feed v_pk_* slab tests
static_assert(sizeof(BvhNode) == 64, "BVH2 node must be one 64-byte line");

// 64-byte 4-wide node for the heavy-scene trace kernel (single-level layout): up to four children, their boxes as 8-bit offsets in the node's
// own frame. org = lower corner of the children's union; per axis a power-of-two scale 2^(b - 127) (b = byte a of `exps`, so the float is b << 23);
// q[0..2] = lower bounds x / y / z, q[3..5] = upper bounds, byte c of each word = child c; a bound decodes as fma(q, scale, org) in float - the
// quantiser (quantizeNode4, below) rounds outwards and CHECKS the decoded value with the same fma, so a decoded box always contains the (padded)
// BVH2 child box it came from. Boxes only cull, hits are decided by the exact triangle test: the compressed tree returns bit for bit what the
// BVH2 returns. One node = one 64-byte line = four 16-byte loads, half the lines and half the dependent steps of the BVH2 walk.
struct BvhNode4 { float org[3]; uint exps; uint q[6]; uint pad[2]; uint ref[4]; };   // exps byte 3: bit c set = child c exists; ref as in BvhNode (inner: BvhNode4 index)
static_assert(sizeof(BvhNode4) == 64, "4-wide node must be one 64-byte line");

#if defined(__HIPCC__)
#define HPT_HD __host__ __device__ inline
#else
#define HPT_HD inline
#endif
HPT_HD float hptBitsToFloat(uint b) { union { uint u; float f; } c; c.u = b; return c.f; }
HPT_HD uint  hptFloatToBits(float f) { union { uint u; float f; } c; c.f = f; return c.u; }
// lo[c][a], hi[c][a]: the boxes of the children with bit c set in `valid` (the others are ignored); refs are not touched
HPT_HD void quantizeNode4(const float lo[4][3], const float hi[4][3], uint valid, BvhNode4& out)
{
  float org[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, top[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
  for (int c = 0; c < 4; c++) if (valid & (1u << c)) for (int a = 0; a < 3; a++) { org[a] = lo[c][a] < org[a] ? lo[c][a] : org[a]; top[a] = hi[c][a] > top[a] ? hi[c][a] : top[a]; }
  uint exps = valid << 24;
  for (int w = 0; w < 6; w++) out.q[w] = 0u;
  for (int a = 0; a < 3; a++) {
    out.org[a] = org[a];
    const float x = (top[a] - org[a]) * (1.0f / 254.0f);              // scale must exceed extent / 255: take the next power of two above extent / 254
    uint b = ((hptFloatToBits(x) >> 23) & 0xFFu) + 1u;
    if (b < 1u) b = 1u;
    if (b > 254u) b = 254u;
    for (int attempt = 0; attempt < 4; attempt++) {
      const float s = hptBitsToFloat(b << 23);
      bool ok = true;
      uint wlo = 0u, whi = 0u;
      for (int c = 0; c < 4; c++) {
        if (!(valid & (1u << c))) continue;
        float fl = floorf((lo[c][a] - org[a]) / s), fh = ceilf((hi[c][a] - org[a]) / s);
        fl = fl < 0.0f ? 0.0f : (fl > 255.0f ? 255.0f : fl);
        fh = fh < 0.0f ? 0.0f : fh;
        while (fl > 0.0f && fmaf(fl, s, org[a]) > lo[c][a]) fl -= 1.0f;        // decoded lower bound must not exceed the true one (org itself never does)
        while (fh <= 255.0f && fmaf(fh, s, org[a]) < hi[c][a]) fh += 1.0f;     // decoded upper bound must reach the true one
        if (fh > 255.0f) { ok = false; break; }
        wlo |= (uint)fl << (8 * c); whi |= (uint)fh << (8 * c);
      }
      if (ok) { out.q[a] = wlo; out.q[3 + a] = whi; break; }
      b = b < 254u ? b + 1u : b;                                         // coarser grid and again (at most a step or two: rounding at the top end)
      if (attempt == 3) { out.q[a] = 0u; out.q[3 + a] = 0xFFFFFFFFu; }   // cannot happen for finite boxes; the whole node range stays conservative
    }
    exps |= b << (8 * a);
  }
  out.exps = exps;
}

// 64 bytes of shading data per triangle record of the single-level layout, in the records' order: what kernel_RayTrace2 gathers through five
// dependent fetches (instance -> mesh offsets -> three indices -> three vertices, primitive -> material id -> remap list) in ONE line:
//   q0 = (nA.xyz, txA)  q1 = (nB.xyz, txB)  q2 = (nC.xyz, txC)  q3 = (tyA, tyB, tyC, material id after the instance's remap list)
// - the values themselves, so the interpolation is the same arithmetic on the same numbers. Built on the device (buildShadeTrisKernel).

// 48-byte triangle: v0, e1 = v1-v0, e2 = v2-v0 (what Moeller-Trumbore consumes), primId in the spare lane
struct BvhTri { float v0[3]; uint primId; float e1[3]; uint instId; float e2[3]; uint pad1; };   // instId: flat (single-level) mode only
static_assert(sizeof(BvhTri) == 48, "triangle record must be 48 bytes");

// 64-byte instance record: world->object rows (3x4), BLAS root reference, mesh id
struct BvhInst { float row0[4], row1[4], row2[4]; uint root, geomId, pad0, pad1; };   // pad0 = 1: a moving instance (DevScene::instMotion)
static_assert(sizeof(BvhInst) == 64, "instance record must be 64 bytes");

struct TexRec
{
  uint w, h, format, flags, addrU, addrV, filter, pad;
  const void* data;       // device pointer
  // differentiable-texture binding (IntegratorDR::TexInfo, diff_render/integrator_dr.h:56-64); offset = ~0 when unbound
  unsigned long long diffOffset; uint diffW, diffH, diffChannels, pad2;
};

// Everything a kernel needs, passed by value as the kernel argument (lands in SGPRs via the kernarg segment).
struct DevScene
{
  const BvhNode* nodes;
  const BvhTri*  tris;
  const BvhInst* insts;
  const BvhTri*  sweepTris;       // sweep scenes: triangle records per mesh in PRIMITIVE order (traceSweep's tie rule relies on it), one spare record at the end
  const BvhInst* sweepInsts;      // sweep scenes: the instance records with root = first triangle record, pad1 = triangle count
  uint           rootRef;
  uint           numInsts;
  uint           nodeMin;         // traversal leaves its inner-node loop when fewer lanes than this still hold an inner node (0: never)
  uint           flatMode;        // 1: one world-space BVH2 over all instanced triangles (leaf triangles carry their instance id)
  const BvhNode4* nodes4;         // single-level layout, static scenes: the same tree collapsed to 4-wide compressed nodes (wfTraceKernel<WIDE>), or null
  uint           root4;           // its root (a BvhNode4 index)
  const float4*  shadeTris;       // single-level layout, gltf / emissive scenes: 64 B of shading data per triangle record (same index as `tris`): see ShadeTri
  uint           megaWide;        // the megakernel's single-level traversal walks nodes4 too (heavy scenes: set with nodeMin4)
  uint           statsWide;       // instrumented megakernel only: count the walk over nodes4 (what the wavefront trace kernel does on this scene)
  uint           nodeMin4;        // nodeMin for the walk over nodes4 (a 4-wide visit is three times the work of a BVH2 one: the vote pays earlier)

  const uint*    triIndices;      // m_triIndices
  const float*   vData8f;         // m_vData8f (8 floats per vertex)
  const uint*    matIdByPrimId;   // m_matIdByPrimId
  const uint*    matVertOffset;   // m_matVertOffset (2 per geom)
  const float*   normMat;         // 12 floats per instance: rows of the upper 3x3 of m_normMatrices (padded to float4)
  const int*     remapInst;       // m_remapInst (2 per instance)
  const int*     allRemapLists;   // m_allRemapLists
  uint           allRemapListsSize;
  uint           numLights;
  const MaterialRec* materials;
  const LightRec*    lights;
  const TexRec*      textures;
  const float*       arrays1f;    // m_arrays1f: pdf table of the sampled environment map
  // motion blur (integrator_pt_scene.cpp:848-897, EmbreeRT.cpp:264-292): instances with two key transforms, interpolated per ray at its time
  const float*       instMotion;  // 24 floats per instance: object->world rows (3x4) at time 0, then at time 1 (only read for BvhInst::pad0 != 0)
  const float*       normMat2;    // 12 floats per instance: rows of the upper 3x3 of m_normMatrices[m_normMatrices2Offs + i]
  uint               motion;      // m_normMatrices2Offs != 0: a time is drawn per path and normals are interpolated
  uint               sweep;       // 1: the scene is small enough for the wave-uniform triangle sweep (traceSweep); BvhInst::root / pad1 = the instance's triangle range
  // lens simulation (integrator_pt.cpp:78-104, 806-938): m_lines as {curvatureRadius, thickness, eta, apertureRadius}, film side first
  const float4*      lensLines;   // lensCount entries; lensCount = 0: m_enableOpticSim off
  uint               lensCount;
  float              physSize[2]; // m_physSize
  uint               padLens;

  // spectral rendering (m_spectral_mode, hpt_spectral.hip): every spectrum resampled at 1 nm from LAMBDA_MIN (m_spec_values), {offset, size} per spectrum
  // id (m_spec_offset_sz), the CIE 1931 observer at 1 nm (m_cie_xyz), the camera's response spectra (m_camResponseSpectrumId, -1: none)
  const float*       specValues;
  const uint*        specOffsetSz;
  const float4*      cieXYZ;
  uint               numCieXYZ, numSpectra;
  int                camResponseSpectrumId[3];
  uint               camResponseType, spectralMode;

  // plain-data members (UpdateMembersPlainData)
  float projInv[16], worldViewInv[16];
  int   winStartX, winStartY, winWidth, winHeight, fbWidth, fbHeight;
  uint  traceDepth, integratorType, renderLayer, tileSize;
  float exposureMult, camLensRadius, camTargetDist;
  float camRespoceRGB[4], envColor[4];
  uint  envTexId, envLightId, envCamBackId, envEnableSam;   // m_envTexId, m_envLightId, m_envCamBackId, m_envEnableSam (0xFFFFFFFF: none)
  float envSamRow0[4], envSamRow1[4];
  uint  envSpecId; float envSpecMult;                        // m_envSpecId (0xFFFFFFFF: none), m_envSpecMult: the environment's spectrum in spectral mode
  // thin films (integrator_pt.h:588-590, hpt_film.h; last, so that no other member moves): eta then k per layer, their spectrum ids, the loader's reflectance / transmittance tables
  const float*       filmsEtaK;
  const uint*        filmsSpecId;
  const float*       precompThinFilms;
  // spectra given by textures (m_spec_tex_ids_wavelengths, m_spec_tex_offset_sz: uint2 each), read by the spectral kernel's colour lookup
  const uint*        specTexIdsWavelengths;
  const uint*        specTexOffsetSz;
  // sweep scenes: the instances' padded world boxes, {lo.xyz, -} {hi.xyz, -} per instance: a wave skips an instance none of its rays can reach (traceSweep)
  const float4*      sweepBoxes;
};

struct Counters { unsigned long long v[32]; };  // rays, nodes, tris, surfaceHits, shadowRays, paths, instEnter, texFetch,
                                                // then wave-cycles (s_memtime) of: queue+regen, closest-hit traversal, shading, shadow traversal, path end; loop trips;
                                                // [16..23] PathTraceDR probe: records stored, records with a parameter texture, wave-cycles of the record stores, of the
                                                // reverse sweeps, wave-trips with a sweep, lanes in those sweeps, atomic wave-instructions, bounces walked by sweeps, [24] records written to HBM

} // namespace hpt
