// Host check of the 4-wide node quantiser (hydracore3_amd/csrc/hpt_types.h: quantizeNode4): for random child boxes - ordinary, flat, point-sized,
// far from the origin, tiny next to huge - the box DECODED the way the trace kernel decodes it (fma(byte, 2^(b - 127), org) in float) contains the
// box that went in, and is not looser than two grid steps per side. Plain g++, no GPU:  g++ -std=c++17 -O2 quantize_test.cpp -o quantize_test
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
struct float4 { float x, y, z, w; };
#include "../../hydracore3_amd/csrc/hpt_types.h"
using namespace hpt;

int main()
{
  std::mt19937 rng(12345);
  std::uniform_real_distribution<float> U(0.0f, 1.0f);
  long checked = 0, loose = 0;
  for (int iter = 0; iter < 200000; iter++) {
    const int kind = iter % 8;
    const float centre = kind == 3 ? 1.0e4f : (kind == 4 ? -3.0e6f : (kind == 7 ? 0.0f : 10.0f * (U(rng) - 0.5f)));
    const float size = kind == 1 ? 1.0e-6f : (kind == 2 ? 1.0e3f : (kind == 7 ? 1.0e-30f : 0.5f));
    float lo[4][3], hi[4][3];
    const uint valid = (iter % 5 == 0) ? 0x3u : ((iter % 7 == 0) ? 0x7u : 0xFu);
    for (int c = 0; c < 4; c++) for (int a = 0; a < 3; a++) {
      const float p = centre + size * (U(rng) - 0.5f), e = (kind == 5 && a == 1) ? 0.0f : (kind == 6 ? size * 1.0e-4f * U(rng) : size * U(rng));
      lo[c][a] = p; hi[c][a] = p + e;
    }
    BvhNode4 nd;
    quantizeNode4(lo, hi, valid, nd);
    if ((nd.exps >> 24) != valid) { std::printf("valid mask lost\n"); return 1; }
    for (int a = 0; a < 3; a++) {
      const float s = hptBitsToFloat(((nd.exps >> (8 * a)) & 0xFFu) << 23);
      float nlo = 3e38f, nhi = -3e38f;
      for (int c = 0; c < 4; c++) if (valid & (1u << c)) { nlo = std::fmin(nlo, lo[c][a]); nhi = std::fmax(nhi, hi[c][a]); }
      for (int c = 0; c < 4; c++) {
        if (!(valid & (1u << c))) continue;
        const float ql = float((nd.q[a] >> (8 * c)) & 0xFFu), qh = float((nd.q[3 + a] >> (8 * c)) & 0xFFu);
        const float dl = std::fmaf(ql, s, nd.org[a]), dh = std::fmaf(qh, s, nd.org[a]);
        checked++;
        if (!(dl <= lo[c][a]) || !(dh >= hi[c][a])) { std::printf("NOT CONSERVATIVE: iter %d axis %d child %d: [%g, %g] decoded as [%g, %g] (scale %g, org %g)\n", iter, a, c, lo[c][a], hi[c][a], dl, dh, s, nd.org[a]); return 1; }
        const float slack = 2.0f * s + 4.0f * 1.1920929e-7f * std::fmax(std::fabs(nlo), std::fabs(nhi));
        if (lo[c][a] - dl > slack || dh - hi[c][a] > slack) loose++;
        if (s * 255.0f < (nhi - nlo) * 0.999f) { std::printf("grid does not span the node: iter %d axis %d\n", iter, a); return 1; }
        if (s > 1.0e-30f && s * 255.0f > 4.1f * (nhi - nlo) + 1.0e-30f && (nhi - nlo) > 1.0e-35f && std::fabs(nlo) < 1.0e3f * (nhi - nlo)) { std::printf("grid much coarser than needed: iter %d axis %d: scale %g, extent %g\n", iter, a, s, nhi - nlo); return 1; }
      }
    }
  }
  std::printf("quantizeNode4: %ld bounds checked, all conservative, %ld looser than two grid steps\n", checked, loose);
  return loose == 0 ? 0 : 1;
}
