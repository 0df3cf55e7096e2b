// The trace phase of the block-local schedules (hpt_block.hip, and the spectral one in hpt_spectral.hip): the block's four waves drain the ray
// pool in LDS together, with ray replacement. pool: BW_POOL entries, field-major - o.xyz, tfar | d.xyz, kind (0 closest hit, 1 shadow);
// an entry is read once, by the lane that takes the ray, and then receives the ray's result: closest hit {t, u, v, primId, instId or
// 0xFFFFFFFF, triangle record}, shadow {occluded}. Resumable traversal as in wfTraceKernel: BVH2 of either layout, or the 4-wide compressed tree.
#pragma once
#include "hpt_decl.h"

namespace hpt {

static const uint BW_POOL = 512u;                    // rays per round: two per lane at most

template <bool DEEP, bool FLAT, bool WIDE>
HPT_DEV void blockTracePhase(const DevScene& S, const TravStack& stk, uint* pool, uint* poolHead, const uint total, const uint refillBelow, const uint nodeMin, const uint lane)
{
  bool has = false, isAny = false, found = false;
  uint slot = 0, cur = REF_NONE, curInst = 0xFFFFFFFFu; int sp = 0;
  V3 wo = v3(0, 0, 0), wd = v3(0, 0, 1), o = wo, d = wd, id = v3(0, 0, 0), oid = v3(0, 0, 0);
  float hitT = 0.0f, hitU = 0.0f, hitV = 0.0f; uint hitPrim = 0xFFFFFFFFu, hitInst = 0xFFFFFFFFu, hitSlot = 0xFFFFFFFFu;
  bool dry = false;                                                      // wave-uniform: the pool had nothing left at the last refill
#define HPT_PUSH(v) do { if (DEEP) stkPush(stk, sp, (v)); else stk.lds[sp * 256] = (v); sp++; } while (0)
#define HPT_POP()   do { sp--; cur = DEEP ? stkPop(stk, sp) : stk.lds[sp * 256]; } while (0)
  while (true) {
    // ---- refill the idle lanes ----
    if (!dry) {
      const unsigned long long mask = __ballot(!has);
      const uint n = (uint)__popcll(mask);
      if (n != 0u) {
        uint base = 0;
        if (lane == 0u) base = atomicAdd(poolHead, n);
        base = __shfl(base, 0);
        const uint e = base + mbcnt64(mask);
        if (!has && e < total) {
          wo = v3(__uint_as_float(pool[0 * BW_POOL + e]), __uint_as_float(pool[1 * BW_POOL + e]), __uint_as_float(pool[2 * BW_POOL + e]));
          hitT = __uint_as_float(pool[3 * BW_POOL + e]);
          wd = v3(__uint_as_float(pool[4 * BW_POOL + e]), __uint_as_float(pool[5 * BW_POOL + e]), __uint_as_float(pool[6 * BW_POOL + e]));
          isAny = pool[7 * BW_POOL + e] != 0u;
          slot = e; o = wo; d = wd; slabRay(wo, wd, id, oid);
          cur = WIDE ? S.root4 : S.rootRef; curInst = 0xFFFFFFFFu; sp = 0; found = false;
          hitPrim = 0xFFFFFFFFu; hitInst = 0xFFFFFFFFu; hitSlot = 0xFFFFFFFFu; hitU = 0.0f; hitV = 0.0f;
          has = cur != REF_NONE;
          if (!has) {                                                    // empty scene: every ray misses
            if (isAny) pool[0 * BW_POOL + e] = 0u; else { pool[4 * BW_POOL + e] = 0xFFFFFFFFu; }
          }
        }
        if (base + n >= total) dry = true;
      }
    }
    if (!__any(has)) break;
    // ---- traverse until this lane's ray is done, or the wave has thinned out and the pool can refill it ----
    if (has) {
      while (true) {
        while ((cur & REF_LEAF) == 0u) {
          if (WIDE) {                                                  // the 4-wide compressed tree of heavy single-level scenes (hpt_device.h: wideNodeStep)
            wideNodeStep<DEEP>(S, stk, oid, id, hitT, cur, sp);
            if (nodeMin != 0u && (uint)__popcll(__ballot((cur & REF_LEAF) == 0u)) < nodeMin) break;
            continue;
          }
          const float4* np = (const float4*)(S.nodes + cur);
          const float4 q0 = np[0], q1 = np[1], q2 = np[2];
          const uint4  q3 = ((const uint4*)np)[3];
          bool h0, h1; float t0n, t1n;
          nodeSlabs(q0, q1, q2, oid, id, 0.0f, hitT, h0, h1, t0n, t1n);
          if (h0 && h1) { const bool firstIs0 = t0n <= t1n; HPT_PUSH(firstIs0 ? q3.y : q3.x); cur = firstIs0 ? q3.x : q3.y; }
          else if (h0) cur = q3.x;
          else if (h1) cur = q3.y;
          else if (sp > 0) HPT_POP();
          else cur = REF_NONE;
          if (nodeMin != 0u && (uint)__popcll(__ballot((cur & REF_LEAF) == 0u)) < nodeMin) break;
        }
        const uint leaf = cur;
        bool done = (leaf == REF_NONE);
        if (!done && (leaf & REF_LEAF) != 0u) {
          const uint cnt = (leaf >> 28) & 7u;
          if (FLAT || (cnt >= 1u && cnt <= 4u)) {
            const uint first = leaf & 0x0FFFFFFFu;
            for (uint k = 0; k < cnt; k++) {
              const float4* tp = (const float4*)(S.tris + first + k);
              const float4 a = tp[0], b = tp[1], c = tp[2];
              uint inst = curInst;
              if (FLAT) {
                inst = __float_as_uint(b.w);
                if (inst != curInst) { toObjectSpace(S.insts, inst, wo, wd, o, d); curInst = inst; }
              }
              if (triangleTest(a, b, c, o, d, 0.0f, inst, hitT, hitPrim, hitInst, hitU, hitV, found)) hitSlot = first + k;
            }
            if (isAny && found) done = true;
            else if (sp > 0) HPT_POP(); else done = true;
          } else if (cnt == 0u) {
            const uint inst = cur & 0x0FFFFFFFu;
            toObjectSpace(S.insts, inst, wo, wd, o, d);
            slabRay(o, d, id, oid);
            curInst = inst;
            HPT_PUSH(REF_RESTORE);
            cur = S.insts[inst].root;
          } else {
            o = wo; d = wd; slabRay(o, d, id, oid); curInst = 0xFFFFFFFFu;
            if (sp > 0) HPT_POP(); else done = true;
          }
        }
        if (done) {
          // the ray's pool entry now carries its result (nobody reads the ray from it any more)
          if (isAny) pool[0 * BW_POOL + slot] = found ? 1u : 0u;
          else {
            pool[0 * BW_POOL + slot] = __float_as_uint(hitT); pool[1 * BW_POOL + slot] = __float_as_uint(hitU); pool[2 * BW_POOL + slot] = __float_as_uint(hitV);
            pool[3 * BW_POOL + slot] = hitPrim; pool[4 * BW_POOL + slot] = found ? hitInst : 0xFFFFFFFFu; pool[5 * BW_POOL + slot] = (FLAT && S.shadeTris != nullptr) ? hitSlot : 0xFFFFFFFFu;
          }
          has = false;
          break;
        }
        if (!dry && (uint)__popcll(__ballot(true)) < refillBelow) break;
      }
    }
  }
#undef HPT_PUSH
#undef HPT_POP
}

} // namespace hpt
