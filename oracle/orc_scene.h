// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).  PARITY UNPINNED.
//
// Scene container, ray/scene queries and texture fetch for the CPU restatement.
//   * ray queries restate the ISceneObject contract (external/CrossRT/CrossRT.h:157-176) with the
//     two-level semantics of the Embree backend (external/CrossRT/EmbreeRT.cpp:138-190, 242-292,
//     310-484): per-mesh geometry in object space, instances with column-major 4x4 matrices, the ray is
//     taken to object space with the inverse matrix and t is shared between the two spaces.
//     Embree itself (3.13.5, prebuilt, absent) is replaced by a plain median-split BVH + a brute-force
//     mode; the closest hit is defined independently of the acceleration structure:
//     min t, ties broken by (instId, primId).
//   * texture fetch restates the only in-tree definition of LiteImage's bilinear sampler,
//     IntegratorDR::Tex2DFetchAD / bilinearOffsets (diff_render/integrator_dr.cpp:60-161).
#pragma once
#include "orc_math.h"
#include "orc_api.h"
#include <vector>
#include <cfloat>

namespace orc {

// ---- records shared with the reference's binary layout ------------------------------------------------------
struct Material            // include/cmaterial.h:187-203 (320 bytes)
{
  uint  mtype, cflags, lightId, nonlinear;
  uint  texid[4], spdid[4], datai[4];
  f4    colors[4];
  f4    row0[4];
  f4    row1[4];
  float data[16];
};
static_assert(sizeof(Material) == 320, "Material must be 320 bytes");

struct LightSource         // include/clight.h:19-56 (320 bytes)
{
  m4    matrix, iesMatrix;
  f4    samplerRow0, samplerRow1, samplerRow0Inv, samplerRow1Inv;
  f4    pos, intensity, norm;
  f2    size; float pdfA; uint geomType;
  uint  distType, flags, pdfTableOffset, pdfTableSize;
  uint  specId, texId, iesId; float mult;
  uint  pdfTableSizeX, pdfTableSizeY, camBackTexId; float lightCos1;
  float lightCos2; uint matId; float dummy2, dummy3;
};
static_assert(sizeof(LightSource) == 320, "LightSource must be 320 bytes");

struct Texture
{
  uint w, h, format, flags, addrU, addrV, filter;
  std::vector<uint32_t> ldr;   // format 0
  std::vector<float>    hdr;   // format 1 (4 floats / texel) or 2 (1 float / texel)
};

// ---- simple BVH ---------------------------------------------------------------------------------------------
struct Box { f3 lo, hi; };
static inline Box box_empty() { Box b = { mk3(FLT_MAX, FLT_MAX, FLT_MAX), mk3(-FLT_MAX, -FLT_MAX, -FLT_MAX) }; return b; }
static inline void box_grow(Box& b, f3 p)
{
  b.lo = mk3(std::min(b.lo.x, p.x), std::min(b.lo.y, p.y), std::min(b.lo.z, p.z));
  b.hi = mk3(std::max(b.hi.x, p.x), std::max(b.hi.y, p.y), std::max(b.hi.z, p.z));
}
static inline void box_merge(Box& b, const Box& o) { box_grow(b, o.lo); box_grow(b, o.hi); }

struct BvhNode { Box box; int left, right, first, count; };

struct SimpleBvh
{
  std::vector<BvhNode> nodes;
  std::vector<uint>    items;    // item ids in leaf order

  void build(const std::vector<Box>& itemBoxes)
  {
    nodes.clear(); items.resize(itemBoxes.size());
    for (size_t i = 0; i < items.size(); i++) items[i] = (uint)i;
    if (items.empty()) return;
    nodes.reserve(items.size() * 2);
    build_rec(itemBoxes, 0, (int)items.size());
  }

private:
  int build_rec(const std::vector<Box>& ib, int first, int count)
  {
    BvhNode n; n.box = box_empty(); n.left = n.right = -1; n.first = first; n.count = count;
    Box cb = box_empty();
    for (int i = first; i < first + count; i++) {
      box_merge(n.box, ib[items[i]]);
      const Box& b = ib[items[i]];
      box_grow(cb, mk3(0.5f * (b.lo.x + b.hi.x), 0.5f * (b.lo.y + b.hi.y), 0.5f * (b.lo.z + b.hi.z)));
    }
    // conservative padding so that the slab test can never cull a primitive the exact test accepts
    const float ex = std::max(std::max(n.box.hi.x - n.box.lo.x, n.box.hi.y - n.box.lo.y), n.box.hi.z - n.box.lo.z);
    const float mag = std::max(std::max(std::abs(n.box.lo.x), std::abs(n.box.hi.x)),
                      std::max(std::max(std::abs(n.box.lo.y), std::abs(n.box.hi.y)), std::max(std::abs(n.box.lo.z), std::abs(n.box.hi.z))));
    const float pad = 1e-5f * std::max(ex, mag) + 1e-30f;
    n.box.lo = n.box.lo - mk3(pad, pad, pad);
    n.box.hi = n.box.hi + mk3(pad, pad, pad);

    const int id = (int)nodes.size();
    nodes.push_back(n);
    if (count <= 4) return id;

    const f3 ext = cb.hi - cb.lo;
    int axis = 0;
    if (ext.y > ext.x) axis = 1;
    if (ext.z > (axis == 0 ? ext.x : ext.y)) axis = 2;
    const int mid = first + count / 2;
    std::nth_element(items.begin() + first, items.begin() + mid, items.begin() + first + count,
                     [&](uint a, uint b) {
                       const Box& ba = ib[a]; const Box& bb = ib[b];
                       const float ca = axis == 0 ? ba.lo.x + ba.hi.x : (axis == 1 ? ba.lo.y + ba.hi.y : ba.lo.z + ba.hi.z);
                       const float cb2 = axis == 0 ? bb.lo.x + bb.hi.x : (axis == 1 ? bb.lo.y + bb.hi.y : bb.lo.z + bb.hi.z);
                       return ca < cb2;
                     });
    const int l = build_rec(ib, first, mid - first);
    const int r = build_rec(ib, mid, first + count - mid);
    nodes[id].left = l; nodes[id].right = r; nodes[id].count = 0;
    return id;
  }
};

static inline bool ray_box(const Box& b, f3 o, f3 d, float tnear, float tfar)
{
  float t0 = tnear, t1 = tfar;
  const float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
  const float lo[3] = { b.lo.x, b.lo.y, b.lo.z }, hi[3] = { b.hi.x, b.hi.y, b.hi.z };
  for (int a = 0; a < 3; a++) {
    if (dd[a] == 0.0f) {
      if (oo[a] < lo[a] || oo[a] > hi[a]) return false;
    } else {
      const float inv = 1.0f / dd[a];
      float ta = (lo[a] - oo[a]) * inv, tb = (hi[a] - oo[a]) * inv;
      if (ta > tb) std::swap(ta, tb);
      // widen by a couple of ulps: the test only has to be conservative
      ta -= std::abs(ta) * 4e-7f; tb += std::abs(tb) * 4e-7f;
      t0 = std::max(t0, ta); t1 = std::min(t1, tb);
      if (t0 > t1) return false;
    }
  }
  return true;
}

// The exact ray/triangle predicate both the oracle and the HIP traversal implement, operation for
// operation: Moeller-Trumbore, no back-face culling (Embree default), closed barycentric domain,
// closed [tnear, tfar]. u weights vertex B, v weights vertex C.
static inline bool ray_tri(f3 o, f3 d, f3 A, f3 B, f3 C, float tnear, float tfar, float* t, float* u, float* v)
{
  const f3 e1 = B - A, e2 = C - A;
  const f3 pvec = cross(d, e2);
  const float det = dot(e1, pvec);
  if (det == 0.0f) return false;
  const float inv = 1.0f / det;
  const f3 tvec = o - A;
  const float uu = dot(tvec, pvec) * inv;
  const f3 qvec = cross(tvec, e1);
  const float vv = dot(d, qvec) * inv;
  const float tt = dot(e2, qvec) * inv;
  if (!(uu >= 0.0f) || !(vv >= 0.0f) || !(uu + vv <= 1.0f)) return false;
  if (!(tt >= tnear) || !(tt <= tfar)) return false;
  *t = tt; *u = uu; *v = vv;
  return true;
}

struct Scene
{
  // flat buffers with the reference's member names (integrator_pt.h:472-500)
  std::vector<f4>       vPos4f;
  std::vector<f4>       vData8f;          // 2 x float4 per vertex
  std::vector<uint>     triIndices, matIdByPrimId;
  std::vector<uint>     matVertOffset;    // 2 per geom
  std::vector<uint>     geomTriCount, geomVertCount;
  std::vector<uint>     instGeomId;
  std::vector<m4>       instMatrices, instMatricesInv, normMatrices;
  std::vector<m4>       instMatricesMotion, normMatrices2;   // motion blur: the key at time 1, m_normMatrices[m_normMatrices2Offs + i]
  std::vector<uint>     instHasMotion;
  bool                  motion = false;                       // m_normMatrices2Offs != 0
  std::vector<int>      remapInst;        // 2 per instance
  std::vector<int>      allRemapLists;
  uint                  allRemapListsSize = 0;
  std::vector<Material> materials;
  std::vector<LightSource> lights;
  std::vector<Texture>  textures;
  std::vector<float>    arrays1f;        // m_arrays1f
  // spectral rendering (integrator_pt.h: m_spec_values, m_spec_offset_sz, m_cie_xyz, m_camResponseSpectrumId, m_camResponseType)
  std::vector<float>    specValues;
  std::vector<uint>     specOffsetSz;    // uint2 per spectrum
  std::vector<f4>       cieXYZ;
  int                   camResponseSpectrumId[3] = { -1, -1, -1 };
  uint                  camResponseType = 0;
  // thin films (integrator_pt.h:587-590)
  std::vector<float>    filmsThickness, filmsEtaK, precompThinFilms;
  std::vector<uint>     filmsSpecId;
  std::vector<uint>     specTexIdsWavelengths, specTexOffsetSz;   // uint2 each (integrator_pt.h: m_spec_tex_ids_wavelengths, m_spec_tex_offset_sz)

  std::vector<SimpleBvh> blas;            // per geom
  SimpleBvh              tlas;

  void load(const orc_scene_desc* s)
  {
    vPos4f.assign((const f4*)s->vPos4f, (const f4*)s->vPos4f + s->numVerts);
    vData8f.assign((const f4*)s->vData8f, (const f4*)s->vData8f + 2 * (size_t)s->numVerts);
    triIndices.assign(s->triIndices, s->triIndices + 3 * (size_t)s->numTris);
    matIdByPrimId.assign(s->matIdByPrimId, s->matIdByPrimId + s->numTris);
    matVertOffset.assign(s->matVertOffset, s->matVertOffset + 2 * (size_t)s->numGeoms);
    geomTriCount.assign(s->geomTriCount, s->geomTriCount + s->numGeoms);
    geomVertCount.assign(s->geomVertCount, s->geomVertCount + s->numGeoms);
    instGeomId.assign(s->instGeomId, s->instGeomId + s->numInsts);
    instMatrices.assign((const m4*)s->instMatrices, (const m4*)s->instMatrices + s->numInsts);
    normMatrices.assign((const m4*)s->normMatrices, (const m4*)s->normMatrices + s->numInsts);
    motion = s->normMatrices2Offs != 0;
    normMatrices2.clear(); instMatricesMotion.clear(); instHasMotion.assign(s->numInsts, 0u);
    if (motion) normMatrices2.assign((const m4*)s->normMatrices + s->normMatrices2Offs, (const m4*)s->normMatrices + s->normMatrices2Offs + s->numInsts);
    if (s->instMatricesMotion && s->instHasMotion) {
      instMatricesMotion.assign((const m4*)s->instMatricesMotion, (const m4*)s->instMatricesMotion + s->numInsts);
      instHasMotion.assign(s->instHasMotion, s->instHasMotion + s->numInsts);
    }
    remapInst.assign(s->remapInst, s->remapInst + 2 * (size_t)s->numInsts);
    if (s->allRemapLists) allRemapLists.assign(s->allRemapLists, s->allRemapLists + s->allRemapListsLen);
    // RemapMaterialId's bisection probes up to one list length past a list's end (see orc_pathtrace.cpp): zero padding keeps the probes of the
    // last list inside the vector
    allRemapLists.resize(allRemapLists.size() + (size_t)s->allRemapListsSize + 2, 0);
    allRemapListsSize = s->allRemapListsSize;
    materials.assign((const Material*)s->materials, (const Material*)s->materials + s->numMaterials);
    lights.assign((const LightSource*)s->lights, (const LightSource*)s->lights + s->numLights);
    textures.resize(s->numTextures);
    for (uint i = 0; i < s->numTextures; i++) {
      const orc_texture_desc& d = s->textures[i];
      Texture& t = textures[i];
      t.w = d.width; t.h = d.height; t.format = d.format; t.flags = d.flags;
      t.addrU = d.addressU; t.addrV = d.addressV; t.filter = d.filter;
      const size_t n = (size_t)d.width * d.height;
      if (d.format == 0) t.ldr.assign((const uint32_t*)d.data, (const uint32_t*)d.data + n);
      else if (d.format == 1) t.hdr.assign((const float*)d.data, (const float*)d.data + 4 * n);
      else t.hdr.assign((const float*)d.data, (const float*)d.data + n);
    }
    arrays1f.clear();
    if (s->arrays1f && s->numArrays1f) arrays1f.assign(s->arrays1f, s->arrays1f + s->numArrays1f);
    specValues.clear(); specOffsetSz.clear(); cieXYZ.clear();
    if (s->specValues && s->numSpecValues) specValues.assign(s->specValues, s->specValues + s->numSpecValues);
    if (s->specOffsetSz && s->numSpectra) specOffsetSz.assign(s->specOffsetSz, s->specOffsetSz + 2 * (size_t)s->numSpectra);
    if (s->cieXYZ && s->numCieXYZ) cieXYZ.assign((const f4*)s->cieXYZ, (const f4*)s->cieXYZ + s->numCieXYZ);
    cieXYZ.resize(std::max<size_t>(cieXYZ.size(), 472), mk4(0, 0, 0, 0));   // SpectrumToXYZ reads entry `offset` before testing offset >= 471
    for (int k = 0; k < 3; k++) camResponseSpectrumId[k] = s->specValues ? s->camResponseSpectrumId[k] : -1;
    camResponseType = s->camResponseType;
    filmsThickness.clear(); filmsEtaK.clear(); precompThinFilms.clear(); filmsSpecId.clear();
    if (s->filmsThickness && s->numFilmsThickness) filmsThickness.assign(s->filmsThickness, s->filmsThickness + s->numFilmsThickness);
    if (s->filmsSpecId && s->numFilmsSpecId) filmsSpecId.assign(s->filmsSpecId, s->filmsSpecId + s->numFilmsSpecId);
    if (s->filmsEtaK && s->numFilmsEtaK) filmsEtaK.assign(s->filmsEtaK, s->filmsEtaK + s->numFilmsEtaK);
    if (s->precompThinFilms && s->numPrecompThinFilms) precompThinFilms.assign(s->precompThinFilms, s->precompThinFilms + s->numPrecompThinFilms);
    specTexIdsWavelengths.clear(); specTexOffsetSz.clear();
    if (s->specTexIdsWavelengths && s->numSpecTexBands) specTexIdsWavelengths.assign(s->specTexIdsWavelengths, s->specTexIdsWavelengths + 2 * (size_t)s->numSpecTexBands);
    if (s->specTexOffsetSz && s->numSpectra) specTexOffsetSz.assign(s->specTexOffsetSz, s->specTexOffsetSz + 2 * (size_t)s->numSpectra);
    instMatricesInv.resize(instMatrices.size());
    for (size_t i = 0; i < instMatrices.size(); i++) instMatricesInv[i] = affine_inverse(instMatrices[i]);
    build_accel();
  }

  f3 vert(uint geom, uint prim, int k) const
  {
    const uint triOffset = matVertOffset[2 * geom + 0], vertOffset = matVertOffset[2 * geom + 1];
    return xyz(vPos4f[vertOffset + triIndices[(triOffset + prim) * 3 + k]]);
  }

  void build_accel()
  {
    blas.resize(geomTriCount.size());
    for (size_t g = 0; g < blas.size(); g++) {
      std::vector<Box> boxes(geomTriCount[g]);
      for (uint p = 0; p < geomTriCount[g]; p++) {
        Box b = box_empty();
        for (int k = 0; k < 3; k++) box_grow(b, vert((uint)g, p, k));
        boxes[p] = b;
      }
      blas[g].build(boxes);
    }
    std::vector<Box> ib(instGeomId.size());
    for (size_t i = 0; i < ib.size(); i++) {
      Box w = box_empty();
      const SimpleBvh& b = blas[instGeomId[i]];
      if (!b.nodes.empty()) {
        const Box& r = b.nodes[0].box;
        for (int c = 0; c < 8; c++) {
          const f3 p = mk3((c & 1) ? r.hi.x : r.lo.x, (c & 2) ? r.hi.y : r.lo.y, (c & 4) ? r.hi.z : r.lo.z);
          box_grow(w, mul4x3(instMatrices[i], p));
          if (instHasMotion[i]) box_grow(w, mul4x3(instMatricesMotion[i], p));       // points move on segments: the two key boxes bound the sweep
        }
        if (instHasMotion[i]) { const f3 e = (w.hi - w.lo) * 1e-5f + mk3(1e-6f, 1e-6f, 1e-6f); w.lo = w.lo - e; w.hi = w.hi + e; }
      } else { w.lo = w.hi = mk3(0, 0, 0); }
      ib[i] = w;
    }
    tlas.build(ib);
  }

  // ---- closest hit ------------------------------------------------------------------------------------------
  struct Best { float t; uint prim, inst; float u, v; bool hit; };

  static inline bool better(float t, uint inst, uint prim, const Best& b)
  {
    if (!b.hit) return true;
    if (t < b.t) return true;
    if (t > b.t) return false;
    if (inst != b.inst) return inst < b.inst;
    return prim < b.prim;
  }

  // A moving instance (AddInstanceMotion, EmbreeRT.cpp:264-292): object->world interpolated linearly between the two keys at the ray's time,
  // then inverted for this ray - cofactors in double, the translation subtracted first (the same arithmetic as the device's)
  void to_object_space_motion(uint inst, float time, f3 wo, f3 wd, f3* o, f3* d) const
  {
    const m4& m0 = instMatrices[inst]; const m4& m1 = instMatricesMotion[inst];
    double a[12];
    const double t = (double)time;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) { const double k0 = (double)m0.c[c][r], k1 = (double)m1.c[c][r]; a[4 * r + c] = k0 + t * (k1 - k0); }
    const double c00 = a[5] * a[10] - a[6] * a[9], c01 = a[2] * a[9] - a[1] * a[10], c02 = a[1] * a[6] - a[2] * a[5];
    const double c10 = a[6] * a[8] - a[4] * a[10], c11 = a[0] * a[10] - a[2] * a[8], c12 = a[2] * a[4] - a[0] * a[6];
    const double c20 = a[4] * a[9] - a[5] * a[8],  c21 = a[1] * a[8] - a[0] * a[9],  c22 = a[0] * a[5] - a[1] * a[4];
    const double det = a[0] * c00 + a[1] * c10 + a[2] * c20;
    const double id = 1.0 / det;
    const double px = (double)wo.x - a[3], py = (double)wo.y - a[7], pz = (double)wo.z - a[11];
    const double dx = (double)wd.x, dy = (double)wd.y, dz = (double)wd.z;
    *o = mk3((float)((c00 * px + c01 * py + c02 * pz) * id), (float)((c10 * px + c11 * py + c12 * pz) * id), (float)((c20 * px + c21 * py + c22 * pz) * id));
    *d = mk3((float)((c00 * dx + c01 * dy + c02 * dz) * id), (float)((c10 * dx + c11 * dy + c12 * dz) * id), (float)((c20 * dx + c21 * dy + c22 * dz) * id));
  }

  void intersect_instance(uint inst, f3 o, f3 d, float tnear, float tfar, Best& best, bool anyHit, bool brute, float time = 0.0f) const
  {
    const uint g = instGeomId[inst];
    const m4& inv = instMatricesInv[inst];
    f3 lo = mul4x3(inv, o), ld = mul3x3(inv, d);
    if (instHasMotion[inst]) to_object_space_motion(inst, time, o, d, &lo, &ld);
    const SimpleBvh& b = blas[g];
    if (brute) {
      for (uint p = 0; p < geomTriCount[g]; p++) {
        float t, u, v;
        const float far_ = best.hit ? best.t : tfar;
        if (ray_tri(lo, ld, vert(g, p, 0), vert(g, p, 1), vert(g, p, 2), tnear, far_, &t, &u, &v) && better(t, inst, p, best)) {
          best.t = t; best.prim = p; best.inst = inst; best.u = u; best.v = v; best.hit = true;
          if (anyHit) return;
        }
      }
      return;
    }
    if (b.nodes.empty()) return;
    int stack[96]; int sp = 0; stack[sp++] = 0;
    while (sp > 0) {
      const BvhNode& n = b.nodes[stack[--sp]];
      const float far_ = best.hit ? best.t : tfar;
      if (!ray_box(n.box, lo, ld, tnear, far_)) continue;
      if (n.count > 0) {
        for (int i = n.first; i < n.first + n.count; i++) {
          const uint p = b.items[i];
          float t, u, v;
          const float far2 = best.hit ? best.t : tfar;
          if (ray_tri(lo, ld, vert(g, p, 0), vert(g, p, 1), vert(g, p, 2), tnear, far2, &t, &u, &v) && better(t, inst, p, best)) {
            best.t = t; best.prim = p; best.inst = inst; best.u = u; best.v = v; best.hit = true;
            if (anyHit) return;
          }
        }
      } else { stack[sp++] = n.left; stack[sp++] = n.right; }
    }
  }

  Best trace(f3 o, f3 d, float tnear, float tfar, bool anyHit, bool brute, float time = 0.0f) const
  {
    Best best; best.hit = false; best.t = tfar; best.prim = best.inst = 0xFFFFFFFFu; best.u = best.v = 0.0f;
    if (brute) {
      for (uint i = 0; i < instGeomId.size(); i++) { intersect_instance(i, o, d, tnear, tfar, best, anyHit, true, time); if (anyHit && best.hit) break; }
      return best;
    }
    if (tlas.nodes.empty()) return best;
    int stack[96]; int sp = 0; stack[sp++] = 0;
    while (sp > 0) {
      const BvhNode& n = tlas.nodes[stack[--sp]];
      const float far_ = best.hit ? best.t : tfar;
      if (!ray_box(n.box, o, d, tnear, far_)) continue;
      if (n.count > 0) {
        for (int i = n.first; i < n.first + n.count; i++) {
          intersect_instance(tlas.items[i], o, d, tnear, tfar, best, anyHit, false, time);
          if (anyHit && best.hit) return best;
        }
      } else { stack[sp++] = n.left; stack[sp++] = n.right; }
    }
    return best;
  }

  // RayQuery_NearestHit (EmbreeRT.cpp:310-362): coords[0] = v (weight of C), coords[1] = u (weight of B)
  orc_hit nearest_hit(f4 posNear, f4 dirFar, bool brute = false, float time = 0.0f) const
  {
    const Best b = trace(xyz(posNear), xyz(dirFar), posNear.w, dirFar.w, false, brute, time);
    orc_hit h;
    if (b.hit) {
      h.t = b.t; h.primId = b.prim; h.instId = b.inst; h.geomId = instGeomId[b.inst];
      h.coords[0] = b.v; h.coords[1] = b.u; h.coords[2] = 1.0f - b.v - b.u; h.coords[3] = 0.0f;
    } else {
      h.t = dirFar.w; h.primId = h.instId = h.geomId = 0xFFFFFFFFu;
      h.coords[0] = h.coords[1] = h.coords[2] = h.coords[3] = 0.0f;
    }
    return h;
  }
  // RayQuery_AnyHit (EmbreeRT.cpp:364-392)
  bool any_hit(f4 posNear, f4 dirFar, bool brute = false, float time = 0.0f) const
  {
    return trace(xyz(posNear), xyz(dirFar), posNear.w, dirFar.w, true, brute, time).hit;
  }

  // ---- textures ---------------------------------------------------------------------------------------------
  static inline int wrap_i(int p, int n) { int r = p % n; return r < 0 ? r + n : r; }

  f4 texel(const Texture& t, int off) const
  {
    if (t.format == 0) {
      const uint32_t v = t.ldr[off];
      const float s = 1.0f / 255.0f;
      return mk4(float(v & 0xFF) * s, float((v >> 8) & 0xFF) * s, float((v >> 16) & 0xFF) * s, float(v >> 24) * s);
    }
    if (t.format == 1) return mk4(t.hdr[4 * off + 0], t.hdr[4 * off + 1], t.hdr[4 * off + 2], t.hdr[4 * off + 3]);
    const float v = t.hdr[off];
    return mk4(v, v, v, v);
  }

  // tap positions and weights of the bilinear fetch (integrator_dr.cpp:60-93, 110-128)
  struct Taps { int off[4]; float w[4]; };
  static Taps bilinear_taps(const Texture& t, f2 uv)
  {
    float ffx = uv.x * float(t.w) - 0.5f;
    float ffy = uv.y * float(t.h) - 0.5f;
    if (t.addrU == 2 && ffx < 0) ffx = 0.0f;
    if (t.addrV == 2 && ffy < 0) ffy = 0.0f;
    const int px = (int)ffx, py = (int)ffy;
    const float fx = std::abs(ffx - (float)px), fy = std::abs(ffy - (float)py);
    const float fx1 = 1.0f - fx, fy1 = 1.0f - fy;
    const int sx = (ffx > 0.0f) ? 1 : -1, sy = (ffy > 0.0f) ? 1 : -1;
    const int x0 = wrap_i(px, (int)t.w), x1 = wrap_i(px + sx, (int)t.w);
    const int y0 = wrap_i(py, (int)t.h), y1 = wrap_i(py + sy, (int)t.h);
    Taps r;
    r.off[0] = y0 * (int)t.w + x0; r.off[1] = y0 * (int)t.w + x1; r.off[2] = y1 * (int)t.w + x0; r.off[3] = y1 * (int)t.w + x1;
    r.w[0] = fx1 * fy1; r.w[1] = fx * fy1; r.w[2] = fx1 * fy; r.w[3] = fx * fy;
    return r;
  }

  f4 tex_sample(uint texId, f2 uv) const
  {
    const Texture& t = textures[texId];
    f4 res;
    if (t.filter == 0) {     // NEAREST
      int px = (int)std::floor(uv.x * float(t.w)), py = (int)std::floor(uv.y * float(t.h));
      px = (t.addrU == 2) ? std::min(std::max(px, 0), (int)t.w - 1) : wrap_i(px, (int)t.w);
      py = (t.addrV == 2) ? std::min(std::max(py, 0), (int)t.h - 1) : wrap_i(py, (int)t.h);
      res = texel(t, py * (int)t.w + px);
    } else {
      const Taps k = bilinear_taps(t, uv);
      const f4 a = texel(t, k.off[0]), b = texel(t, k.off[1]), c = texel(t, k.off[2]), d = texel(t, k.off[3]);
      res.x = a.x * k.w[0] + b.x * k.w[1] + c.x * k.w[2] + d.x * k.w[3];
      res.y = a.y * k.w[0] + b.y * k.w[1] + c.y * k.w[2] + d.y * k.w[3];
      res.z = a.z * k.w[0] + b.z * k.w[1] + c.z * k.w[2] + d.z * k.w[3];
      res.w = a.w * k.w[0] + b.w * k.w[1] + c.w * k.w[2] + d.w * k.w[3];
    }
    if (t.flags & 1u) { res.x = std::pow(res.x, 2.2f); res.y = std::pow(res.y, 2.2f); res.z = std::pow(res.z, 2.2f); }
    return res;
  }
};

} // namespace orc
