// JPEG texture files of LoadTextureAndMakeCombined (integrator_pt_scene_tex.cpp:24-33: ".jpg" / ".jpeg" through LiteImage::LoadImage<uint32_t>).
// LiteImage is absent from the reference tree, so this is the format itself (ITU-T T.81): 8-bit Huffman JPEG, baseline / extended sequential
// (SOF0 / SOF1) and progressive (SOF2), one component (grey) or three (YCbCr, JFIF), sampling factors 1 or 2, restart intervals. The arithmetic
// where the standard leaves a choice follows the IJG decoder's defaults - the "slow" 13-bit integer inverse DCT, triangle-filter ("fancy")
// chroma upsampling, 16-bit fixed-point YCbCr -> RGB - so that the texels equal what the common libjpeg-based readers return
// (tests/test_cpu.py compares with PIL, which wraps libjpeg). Header-only host C++17; both scene loaders use this one decoder (the Python one
// through hpt_decode_jpeg).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace hydra_hip {
namespace jpeg {

static const uint8_t ZIGZAG[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

struct Huff { bool set = false; int mincode[17], maxcode[18], valptr[17]; uint8_t vals[256]; };
struct Comp { int id = 0, h = 1, v = 1, tq = 0, bw = 0, bh = 0, dcPred = 0, w = 0, hgt = 0; std::vector<int16_t> coef; std::vector<uint8_t> plane; };

struct Decoder
{
  const uint8_t* d; size_t n, pos = 0; std::string err;
  uint32_t bitBuf = 0; int bitCnt = 0; bool hitMarker = false;
  uint16_t qt[4][64]; bool qtSet[4] = { false, false, false, false };
  Huff dc[4], ac[4];
  Comp comp[3]; int nComp = 0, W = 0, H = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, restart = 0; bool progressive = false, haveFrame = false;
  int eobrun = 0;

  Decoder(const uint8_t* data, size_t size) : d(data), n(size) {}
  bool fail(const char* m) { if (err.empty()) err = m; return false; }
  int u8() { return pos < n ? d[pos++] : 0; }
  int u16() { const int a = u8(); return (a << 8) | u8(); }

  // ---- entropy-coded segment: bits MSB first, 0xFF00 is a stuffed 0xFF, any other marker ends the data (zeros are fed from there on) ----
  void resetBits() { bitBuf = 0; bitCnt = 0; hitMarker = false; }
  int bit()
  {
    if (bitCnt == 0) {
      int b = 0;
      if (!hitMarker && pos < n) {
        b = d[pos++];
        if (b == 0xFF) { const int m = pos < n ? d[pos] : 0xD9; if (m == 0) pos++; else { hitMarker = true; pos--; b = 0; } }
      }
      bitBuf = (uint32_t)b; bitCnt = 8;
    }
    bitCnt--;
    return (int)((bitBuf >> bitCnt) & 1u);
  }
  int receive(int s) { int v = 0; for (int i = 0; i < s; i++) v = (v << 1) | bit(); return v; }
  static int extend(int v, int s) { return (s && v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; }      // T.81 F.12
  int decode(const Huff& h)                                                                           // T.81 F.16
  {
    int code = 0;
    for (int len = 1; len <= 16; len++) {
      code = (code << 1) | bit();
      if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
    }
    fail("jpeg: bad Huffman code"); return 0;
  }

  bool readDQT(int len)
  {
    while (len > 0) {
      const int pq = u8(), p = pq >> 4, t = pq & 15; len--;
      if (t > 3) return fail("jpeg: bad quantisation table id");
      for (int i = 0; i < 64; i++) { qt[t][ZIGZAG[i]] = (uint16_t)(p ? u16() : u8()); len -= p ? 2 : 1; }
      qtSet[t] = true;
    }
    return true;
  }
  bool readDHT(int len)
  {
    while (len > 0) {
      const int tc = u8(), cls = tc >> 4, t = tc & 15; len--;
      if (t > 3 || cls > 1) return fail("jpeg: bad Huffman table id");
      int counts[17]; int total = 0;
      for (int i = 1; i <= 16; i++) { counts[i] = u8(); total += counts[i]; } len -= 16;
      if (total > 256) return fail("jpeg: bad Huffman table");
      Huff& h = cls ? ac[t] : dc[t];
      for (int i = 0; i < total; i++) h.vals[i] = (uint8_t)u8(); len -= total;
      int code = 0, k = 0;
      for (int l = 1; l <= 16; l++) {                                                                 // T.81 C.2 / F.2.2.3
        h.valptr[l] = k; h.mincode[l] = code;
        code += counts[l]; k += counts[l];
        h.maxcode[l] = counts[l] ? code - 1 : -1;
        code <<= 1;
      }
      h.set = true;
    }
    return true;
  }
  bool readSOF(int marker)
  {
    progressive = marker == 0xC2;
    const int prec = u8(); H = u16(); W = u16(); nComp = u8();
    if (prec != 8) return fail("jpeg: only 8-bit samples are read");
    if (W <= 0 || H <= 0) return fail("jpeg: empty frame");
    if ((uint64_t)W * (uint64_t)H > (uint64_t(1) << 28)) return fail("jpeg: frames beyond 2^28 pixels are not read");   // (coefficients and planes are held whole)
    if (nComp != 1 && nComp != 3) return fail("jpeg: only grey and YCbCr images are read");
    hmax = vmax = 1;
    for (int i = 0; i < nComp; i++) {
      comp[i].id = u8(); const int hv = u8(); comp[i].h = hv >> 4; comp[i].v = hv & 15; comp[i].tq = u8();
      if (comp[i].h < 1 || comp[i].h > 2 || comp[i].v < 1 || comp[i].v > 2 || comp[i].tq > 3) return fail("jpeg: sampling factors other than 1 and 2 are not read");
      if (comp[i].h > hmax) hmax = comp[i].h; if (comp[i].v > vmax) vmax = comp[i].v;
    }
    if (nComp == 1) { comp[0].h = comp[0].v = 1; hmax = vmax = 1; }                                  // a single component is never interleaved
    mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
    for (int i = 0; i < nComp; i++) {
      Comp& c = comp[i];
      c.bw = mcux * c.h; c.bh = mcuy * c.v;
      c.w = (W * c.h + hmax - 1) / hmax; c.hgt = (H * c.v + vmax - 1) / vmax;                        // the component's own size (downsampled_width / height)
      c.coef.assign((size_t)c.bw * c.bh * 64, 0);
    }
    haveFrame = true;
    return true;
  }

  // ---- one 8 x 8 block of one scan ----
  bool block(Comp& c, int16_t* b, const Huff* hd, const Huff* ha, int Ss, int Se, int Ah, int Al)
  {
    if (!progressive) {
      const int t = decode(*hd);
      c.dcPred += t ? extend(receive(t), t) : 0;
      b[0] = (int16_t)c.dcPred;
      for (int k = 1; k < 64;) {
        const int rs = decode(*ha), r = rs >> 4, s = rs & 15;
        if (s == 0) { if (r == 15) { k += 16; continue; } break; }
        k += r; if (k > 63) return fail("jpeg: coefficient index past the block");
        b[ZIGZAG[k]] = (int16_t)extend(receive(s), s); k++;
      }
      return err.empty();
    }
    if (Ss == 0) {                                                                                    // DC scans (T.81 G.1.2.1)
      if (Ah == 0) { const int t = decode(*hd); c.dcPred += t ? extend(receive(t), t) : 0; b[0] = (int16_t)(c.dcPred * (1 << Al)); }
      else if (bit()) b[0] |= (int16_t)(1 << Al);
      return err.empty();
    }
    if (Ah == 0) {                                                                                    // AC first pass (G.1.2.2)
      if (eobrun > 0) { eobrun--; return true; }
      for (int k = Ss; k <= Se; k++) {
        const int rs = decode(*ha), r = rs >> 4, s = rs & 15;
        if (s == 0) {
          if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += receive(r); break; }
          k += 15;
        } else {
          k += r; if (k > 63) return fail("jpeg: coefficient index past the block");
          b[ZIGZAG[k]] = (int16_t)(extend(receive(s), s) * (1 << Al));
        }
      }
      return err.empty();
    }
    const int p1 = 1 << Al, m1 = -(1 << Al);                                                          // AC refinement (G.1.2.3)
    int k = Ss;
    if (eobrun == 0) {
      for (; k <= Se; k++) {
        const int rs = decode(*ha); int r = rs >> 4, s = rs & 15;
        if (s) s = bit() ? p1 : m1;
        else if (r != 15) { eobrun = 1 << r; if (r) eobrun += receive(r); break; }
        do {
          int16_t& cf = b[ZIGZAG[k]];
          if (cf != 0) { if (bit() && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1)); }
          else if (--r < 0) break;
          k++;
        } while (k <= Se);
        if (s && k <= 63) b[ZIGZAG[k]] = (int16_t)s;
        if (!err.empty()) return false;
      }
    }
    if (eobrun > 0) {
      for (; k <= Se; k++) { int16_t& cf = b[ZIGZAG[k]]; if (cf != 0 && bit() && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1)); }
      eobrun--;
    }
    return err.empty();
  }
  void restartInterval()                                                                               // RSTn: byte alignment, predictors and the EOB run start over
  {
    resetBits();
    if (pos + 1 < n && d[pos] == 0xFF && d[pos + 1] >= 0xD0 && d[pos + 1] <= 0xD7) pos += 2;
    for (int i = 0; i < nComp; i++) comp[i].dcPred = 0;
    eobrun = 0;
  }
  bool readScan()
  {
    if (!haveFrame) return fail("jpeg: scan before the frame header");
    const int ns = u8();
    if (ns < 1 || ns > nComp) return fail("jpeg: bad scan header");
    int ci[3], td[3], ta[3];
    for (int i = 0; i < ns; i++) {
      const int id = u8(), t = u8(); ci[i] = -1;
      for (int j = 0; j < nComp; j++) if (comp[j].id == id) ci[i] = j;
      if (ci[i] < 0) return fail("jpeg: scan names an unknown component");
      td[i] = t >> 4; ta[i] = t & 15;
      if (td[i] > 3 || ta[i] > 3) return fail("jpeg: bad table selector");
    }
    const int Ss = u8(), Se = u8(), a = u8(), Ah = a >> 4, Al = a & 15;
    if (progressive ? (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss != 0 && ns != 1)) : false) return fail("jpeg: bad spectral selection");
    for (int i = 0; i < ns; i++) {
      const bool needDc = progressive ? (Ss == 0 && Ah == 0) : true, needAc = progressive ? Ss != 0 : true;
      if ((needDc && !dc[td[i]].set) || (needAc && !ac[ta[i]].set)) return fail("jpeg: scan uses a Huffman table that was not defined");
    }
    resetBits(); eobrun = 0;
    for (int i = 0; i < nComp; i++) comp[i].dcPred = 0;
    int count = 0;
    if (ns == 1) {                                                                                     // not interleaved: the component's own blocks, row by row
      Comp& c = comp[ci[0]];
      const int across = (c.w + 7) / 8, down = (c.hgt + 7) / 8;
      for (int by = 0; by < down; by++) for (int bx = 0; bx < across; bx++) {
        if (restart && count == restart) { restartInterval(); count = 0; }
        if (!block(c, &c.coef[((size_t)by * c.bw + bx) * 64], &dc[td[0]], &ac[ta[0]], Ss, Se, Ah, Al)) return false;
        count++;
      }
    } else {
      for (int my = 0; my < mcuy; my++) for (int mx = 0; mx < mcux; mx++) {
        if (restart && count == restart) { restartInterval(); count = 0; }
        for (int i = 0; i < ns; i++) {
          Comp& c = comp[ci[i]];
          for (int y = 0; y < c.v; y++) for (int x = 0; x < c.h; x++)
            if (!block(c, &c.coef[((size_t)(my * c.v + y) * c.bw + mx * c.h + x) * 64], &dc[td[i]], &ac[ta[i]], Ss, Se, Ah, Al)) return false;
        }
        count++;
      }
    }
    if (hitMarker) { /* pos already stands on the marker */ }
    return true;
  }

  // ---- the IJG "slow" integer inverse DCT (13-bit constants, two passes), dequantisation folded in ----
  static inline int descale(int64_t x, int nb) { return (int)((x + (int64_t(1) << (nb - 1))) >> nb); }
  static void idct(const int16_t* in, const uint16_t* q, uint8_t* out, int stride)
  {
    const int CB = 13, P1 = 2;
    const int64_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
    int ws[64];
    for (int pass = 0; pass < 2; pass++) {
      for (int i = 0; i < 8; i++) {
        int64_t v[8];
        for (int k = 0; k < 8; k++) v[k] = pass == 0 ? (int64_t)in[k * 8 + i] * q[k * 8 + i] : (int64_t)ws[i * 8 + k];
        int64_t z2 = v[2], z3 = v[6];
        int64_t z1 = (z2 + z3) * F0541;
        const int64_t t2 = z1 + z3 * (-F1847), t3 = z1 + z2 * F0765;
        const int64_t t0 = (v[0] + v[4]) * (int64_t(1) << CB), t1 = (v[0] - v[4]) * (int64_t(1) << CB);
        const int64_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        int64_t o0 = v[7], o1 = v[5], o2 = v[3], o3 = v[1];
        z1 = o0 + o3; z2 = o1 + o2; z3 = o0 + o2; int64_t z4 = o1 + o3;
        const int64_t z5 = (z3 + z4) * F1175;
        o0 *= F0298; o1 *= F2053; o2 *= F3072; o3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        o0 += z1 + z3; o1 += z2 + z4; o2 += z2 + z3; o3 += z1 + z4;
        const int64_t r[8] = { t10 + o3, t11 + o2, t12 + o1, t13 + o0, t13 - o0, t12 - o1, t11 - o2, t10 - o3 };
        if (pass == 0) for (int k = 0; k < 8; k++) ws[k * 8 + i] = descale(r[k], CB - P1);
        else for (int k = 0; k < 8; k++) { const int x = descale(r[k], CB + P1 + 3) + 128; out[i * stride + k] = (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
      }
    }
  }

  bool finish(uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba)
  {
    if (!haveFrame) return fail("jpeg: no frame");
    for (int i = 0; i < nComp; i++) {
      Comp& c = comp[i];
      if (!qtSet[c.tq]) return fail("jpeg: quantisation table missing");
      c.plane.assign((size_t)c.bw * 8 * c.bh * 8, 0);
      for (int by = 0; by < c.bh; by++) for (int bx = 0; bx < c.bw; bx++)
        idct(&c.coef[((size_t)by * c.bw + bx) * 64], qt[c.tq], &c.plane[((size_t)by * 8) * (c.bw * 8) + bx * 8], c.bw * 8);
    }
    w = (uint32_t)W; h = (uint32_t)H;
    rgba.assign((size_t)W * H * 4, 255);
    // every component at full resolution: the IJG triangle filters for a factor of two (edges replicate inside the component's own size)
    std::vector<std::vector<uint8_t>> full(nComp);
    for (int i = 0; i < nComp; i++) {
      const Comp& c = comp[i];
      const int sx = hmax / c.h, sy = vmax / c.v, cw = c.w, ch = c.hgt, stride = c.bw * 8;
      std::vector<uint8_t>& o = full[i]; o.assign((size_t)W * H, 0);
      auto at = [&](int x, int y) { x = x < 0 ? 0 : (x >= cw ? cw - 1 : x); y = y < 0 ? 0 : (y >= ch ? ch - 1 : y); return (int)c.plane[(size_t)y * stride + x]; };
      for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
        int v;
        if (sx == 1 && sy == 1) v = at(x, y);
        else if (sx == 2 && cw <= 2) v = at(x >> 1, y / sy);                                           // (the IJG code replicates components narrower than three samples)
        else if (sx == 2 && sy == 1) {                                                                 // h2v1_fancy_upsample
          const int ix = x >> 1;
          if ((x & 1) == 0) v = ix == 0 ? at(0, y) : (3 * at(ix, y) + at(ix - 1, y) + 1) >> 2;
          else v = ix == cw - 1 ? at(ix, y) : (3 * at(ix, y) + at(ix + 1, y) + 2) >> 2;
        } else if (sx == 1 && sy == 2) {                                                               // h1v2_fancy_upsample
          const int iy = y >> 1, far = (y & 1) ? iy + 1 : iy - 1;
          v = (3 * at(x, iy) + at(x, far) + ((y & 1) ? 2 : 1)) >> 2;
        } else {                                                                                       // h2v2_fancy_upsample: 3:1 vertically, then 3:1 horizontally on the column sums
          const int ix = x >> 1, iy = y >> 1, far = (y & 1) ? iy + 1 : iy - 1;
          auto colsum = [&](int xx) { return 3 * at(xx, iy) + at(xx, far); };
          const int t = colsum(ix);
          if ((x & 1) == 0) v = ix == 0 ? (t * 4 + 8) >> 4 : (t * 3 + colsum(ix - 1) + 8) >> 4;
          else v = ix == cw - 1 ? (t * 4 + 7) >> 4 : (t * 3 + colsum(ix + 1) + 7) >> 4;
        }
        o[(size_t)y * W + x] = (uint8_t)v;
      }
    }
    auto clamp8 = [](int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    for (size_t p = 0; p < (size_t)W * H; p++) {
      if (nComp == 1) { rgba[4 * p] = rgba[4 * p + 1] = rgba[4 * p + 2] = full[0][p]; continue; }
      const int y = full[0][p], cb = full[1][p] - 128, cr = full[2][p] - 128;                          // ycc_rgb_convert: 16-bit fixed point
      rgba[4 * p + 0] = clamp8(y + ((91881 * cr + 32768) >> 16));
      rgba[4 * p + 1] = clamp8(y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
      rgba[4 * p + 2] = clamp8(y + ((116130 * cb + 32768) >> 16));
    }
    return true;
  }

  bool run(uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba)
  {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail("jpeg: not a JPEG file");
    pos = 2;
    while (pos + 1 < n) {
      if (d[pos] != 0xFF) { pos++; continue; }
      const int m = d[pos + 1];
      if (m == 0xFF) { pos++; continue; }
      pos += 2;
      if (m == 0xD9) break;
      if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
      const size_t segStart = pos;
      const int len = u16();
      if (len < 2 || segStart + len > n) return fail("jpeg: truncated segment");
      if (m == 0xDB) { if (!readDQT(len - 2)) return false; }
      else if (m == 0xC4) { if (!readDHT(len - 2)) return false; }
      else if (m == 0xC0 || m == 0xC1 || m == 0xC2) { if (!readSOF(m)) return false; }
      else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8)) return fail("jpeg: lossless / hierarchical / arithmetic-coded files are not read");
      else if (m == 0xDD) restart = u16();
      else if (m == 0xDA) { if (!readScan()) return false; continue; }                                // (the scan leaves pos at the next marker)
      pos = segStart + len;
    }
    return finish(w, h, rgba);
  }
};

inline bool decode(const std::vector<uint8_t>& file, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba, std::string& err)
{
  Decoder dec(file.data(), file.size());
  if (dec.run(w, h, rgba)) return true;
  err = dec.err.empty() ? "jpeg: decode failed" : dec.err;
  return false;
}

} // namespace jpeg
} // namespace hydra_hip
