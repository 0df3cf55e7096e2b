// Host check of the folded box tests of the trace kernels (hydracore3_amd/csrc/hpt_device.h: slabRay, nodeSlabs, wideNodeStep): a plane's distance is
// computed as plane * id - origin * id with a reciprocal direction kept finite (and, in the 4-wide node, with the decode folded in:
// q * (scale * id) + (base * id - origin * id)). Boxes only cull, so the property that matters is one-sided: whenever the EXACT ray (float origin and
// direction, arithmetic in long double) meets the UNPADDED box at some t* in [tnear, best], the float test on the PADDED box (Aabb::pad: relative
// 1e-5) - resp. on the padded box quantised by quantizeNode4 - must answer "hit". Rays are aimed at points inside the box, on its faces, edges and
// corners, from near and far origins, with tiny and exactly-zero direction components; boxes are ordinary, flat, point-sized, far from the origin.
// The reciprocal is perturbed by an ulp either way (v_rcp_f32 is a 1-ulp approximation). `best` is the exact entry distance itself in half of the
// cases (a triangle lying in the box's entry face). The formulas are restated here line by line from hpt_device.h. Plain g++, no GPU.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <algorithm>
#include <cfloat>
struct float4 { float x, y, z, w; };
#include "../../hydracore3_amd/csrc/hpt_types.h"
#include "../../hydracore3_amd/csrc/bvh_build.h"
using namespace hpt;

static float clampFinite(float x) { return std::fmin(std::fmax(x, -1.0e30f), 1.0e30f); }

// nodeSlabs (one child), folded form
static bool slabsFolded(const float lo[3], const float hi[3], const float oid[3], const float id[3], float tnear, float best)
{
  float a0[3], a1[3];
  for (int a = 0; a < 3; a++) { a0[a] = std::fmaf(lo[a], id[a], -oid[a]); a1[a] = std::fmaf(hi[a], id[a], -oid[a]); }
  const float tn = std::fmax(std::fmax(std::fmin(a0[0], a1[0]), std::fmin(a0[1], a1[1])), std::fmax(std::fmin(a0[2], a1[2]), tnear));
  const float tf = std::fmin(std::fmin(std::fmax(a0[0], a1[0]), std::fmax(a0[1], a1[1])), std::fmin(std::fmax(a0[2], a1[2]), best));
  return tn * 0.999999f <= tf * 1.000001f;
}
// the form of rounds 1-2: (plane - origin) * id with the raw reciprocal
static bool slabsPlain(const float lo[3], const float hi[3], const float o[3], const float id[3], float tnear, float best)
{
  float a0[3], a1[3];
  for (int a = 0; a < 3; a++) { a0[a] = (lo[a] - o[a]) * id[a]; a1[a] = (hi[a] - o[a]) * id[a]; }
  const float tn = std::fmax(std::fmax(std::fmin(a0[0], a1[0]), std::fmin(a0[1], a1[1])), std::fmax(std::fmin(a0[2], a1[2]), tnear));
  const float tf = std::fmin(std::fmin(std::fmax(a0[0], a1[0]), std::fmax(a0[1], a1[1])), std::fmin(std::fmax(a0[2], a1[2]), best));
  return tn * 0.999999f <= tf * 1.000001f;
}
// wideNodeStep, child c of a quantised node
static bool wideFolded(const BvhNode4& nd, int c, const float oid[3], const float id[3], float tnear, float best)
{
  float tn = tnear, tf = 3.0e38f;
  for (int a = 0; a < 3; a++) {
    const float s = hptBitsToFloat(((nd.exps >> (8 * a)) & 0xFFu) << 23) * id[a];
    const float b = std::fmaf(nd.org[a], id[a], -oid[a]);
    const bool neg = id[a] < 0.0f;
    const uint wl = nd.q[a], wh = nd.q[3 + a];
    const uint nW = neg ? wh : wl, fW = neg ? wl : wh;
    const float n = std::fmaf(float((nW >> (8 * c)) & 0xFFu), s, b), f = std::fmaf(float((fW >> (8 * c)) & 0xFFu), s, b);
    tn = std::fmax(tn, n); tf = std::fmin(tf, f);
  }
  tf = std::fmin(tf * 1.0000021f, best * 1.0000021f);
  return tn <= tf && ((nd.exps >> (24 + c)) & 1u);
}

int main()
{
  std::mt19937 rng(20260);
  std::uniform_real_distribution<float> U(0.0f, 1.0f);
  auto sym = [&]() { return 2.0f * U(rng) - 1.0f; };
  long cases = 0, exactHits = 0, missFold = 0, missPlain = 0, missWide = 0, wideCases = 0;
  for (int iter = 0; iter < 1500000; iter++) {
    const int kind = iter % 10;
    // the box
    const float centreMag = kind == 3 ? 1.0e3f : (kind == 4 ? 3.0e5f : (kind == 8 ? 1.0e-3f : 10.0f));
    const float size = kind == 1 ? 1.0e-4f : (kind == 2 ? 50.0f : (kind == 8 ? 1.0e-2f : 0.5f));
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) {
      const float p = centreMag * sym(), e = (kind == 5 && a == iter % 3) ? 0.0f : (kind == 6 ? size * 1.0e-4f * U(rng) : size * U(rng));
      lo[a] = p; hi[a] = p + e;
    }
    // a target on / in the unpadded box
    float tgt[3];
    const int where = (iter / 10) % 4;                       // 0 inside, 1 a face, 2 an edge, 3 a corner
    for (int a = 0; a < 3; a++) tgt[a] = lo[a] + (hi[a] - lo[a]) * U(rng);
    for (int k = 0; k < where; k++) { const int a = (iter + k) % 3; tgt[a] = (rng() & 1u) ? lo[a] : hi[a]; }
    // the ray: origin near, far or very far; sometimes axis-parallel (an exactly zero component, the origin inside that slab), sometimes a tiny component
    const int okind = (iter / 40) % 5;
    const float dist = okind == 0 ? size * 0.5f : (okind == 1 ? 5.0f : (okind == 2 ? 300.0f : (okind == 3 ? 1.0e4f : 0.05f)));
    float o[3], d[3];
    for (int a = 0; a < 3; a++) o[a] = tgt[a] + dist * sym();
    const int dk = (iter / 200) % 6;
    if (dk == 1 || dk == 2) { const int a = iter % 3; o[a] = tgt[a]; }                       // -> d[a] = 0 exactly
    if (dk == 2) { const int a = (iter + 1) % 3; o[a] = tgt[a]; }
    if (dk == 3) { const int a = iter % 3; o[a] = tgt[a] + 1.0e-6f * dist * sym(); }         // tiny component
    double len = 0.0;
    for (int a = 0; a < 3; a++) { d[a] = tgt[a] - o[a]; len += (double)d[a] * d[a]; }
    if (len == 0.0) continue;
    for (int a = 0; a < 3; a++) d[a] = (float)(d[a] / std::sqrt(len));
    // exact interval of the float ray against the unpadded box
    long double tn = 0.0L, tf = 1.0e300L; bool miss = false;
    for (int a = 0; a < 3; a++) {
      if (d[a] == 0.0f) { if (o[a] < lo[a] || o[a] > hi[a]) miss = true; continue; }
      long double t0 = ((long double)lo[a] - o[a]) / d[a], t1 = ((long double)hi[a] - o[a]) / d[a];
      if (t0 > t1) std::swap(t0, t1);
      tn = std::max(tn, t0); tf = std::min(tf, t1);
    }
    cases++;
    if (miss || tn > tf) continue;                            // (the normalised direction no longer passes through the box: nothing to require)
    exactHits++;
    // best: far away, or the exact entry / a point of the interval (rounded UP to float: best >= t*)
    float best = 3.0e38f;
    const int bk = (iter / 7) % 4;
    if (bk == 1) best = std::nextafterf((float)tn, 3.0e38f);
    if (bk == 2) best = std::nextafterf((float)(tn + (tf - tn) * U(rng)), 3.0e38f);
    if (best < 0.0f) best = 0.0f;
    // the padded box, the ray's side of the test (reciprocal an ulp off either way)
    Aabb box; for (int a = 0; a < 3; a++) { box.lo[a] = lo[a]; box.hi[a] = hi[a]; }
    box.pad();
    float idRaw[3], id[3], oid[3];
    for (int a = 0; a < 3; a++) {
      float r = 1.0f / d[a];
      if (std::isfinite(r)) { const unsigned k = rng() % 3u; if (k == 1u) r = std::nextafterf(r, 3.0e38f); if (k == 2u) r = std::nextafterf(r, -3.0e38f); }
      idRaw[a] = r; id[a] = clampFinite(r); oid[a] = o[a] * id[a];
    }
    if (!slabsFolded(box.lo, box.hi, oid, id, 0.0f, best)) {
      if (missFold < 5) std::printf("FOLDED MISS iter %d: box [%g %g %g]-[%g %g %g] o (%g %g %g) d (%g %g %g) exact [%Lg, %Lg] best %g\n", iter, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], o[0], o[1], o[2], d[0], d[1], d[2], tn, tf, best);
      missFold++;
    }
    if (!slabsPlain(box.lo, box.hi, o, idRaw, 0.0f, best)) missPlain++;
    // the same box as child of a quantised 4-wide node among three random siblings
    {
      float l4[4][3], h4[4][3];
      const int me = iter % 4;
      for (int c = 0; c < 4; c++) for (int a = 0; a < 3; a++) {
        if (c == me) { l4[c][a] = box.lo[a]; h4[c][a] = box.hi[a]; continue; }
        const float ext = (hi[a] - lo[a]) + size;
        const float p = lo[a] + 3.0f * ext * sym(), e = ext * U(rng);
        Aabb sb; sb.lo[0] = sb.lo[1] = sb.lo[2] = p; sb.hi[0] = sb.hi[1] = sb.hi[2] = p + e; sb.pad();
        l4[c][a] = sb.lo[0]; h4[c][a] = sb.hi[0];
      }
      BvhNode4 nd; quantizeNode4(l4, h4, 0xFu, nd);
      wideCases++;
      if (!wideFolded(nd, me, oid, id, 0.0f, best)) {
        if (missWide < 5) std::printf("WIDE MISS iter %d: box [%g %g %g]-[%g %g %g] o (%g %g %g) d (%g %g %g) exact [%Lg, %Lg] best %g\n", iter, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], o[0], o[1], o[2], d[0], d[1], d[2], tn, tf, best);
        missWide++;
      }
    }
  }
  std::printf("slab tests: %ld rays, %ld exact hits; misses: folded BVH2 form %ld, folded 4-wide form %ld (of %ld), plain form of rounds 1-2 %ld\n", cases, exactHits, missFold, missWide, wideCases, missPlain);
  if (missFold == 0 && missWide == 0) std::printf("all conservative\n");
  return (missFold == 0 && missWide == 0) ? 0 : 1;
}
