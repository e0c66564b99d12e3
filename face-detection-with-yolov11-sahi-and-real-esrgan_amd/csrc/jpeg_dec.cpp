// jpeg_dec.cpp — the sequential half of JPEG decoding: markers and Huffman decoding on the host (jdmarker.c / jdhuff.c semantics).
// The entropy-coded segment is a serial bit stream (each symbol's position depends on every previous one); what parallelises —
// dequantisation, IDCT, upsampling, colour conversion, 97 % of the arithmetic — runs on the device (jpeg.hip) on the coefficients
// this file produces, so a decoded frame is born in device memory.
#include <cstring>

#include "jpeg.hpp"

namespace ffp {

namespace {

const unsigned char kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTab {
  bool present = false;
  unsigned char look_len[512];                  // 9-bit prefix -> code length (0: longer than 9 bits)
  unsigned char look_sym[512];
  int maxcode[18];                              // largest code of each length (-1: none)
  int valptr[17], mincode[17];
  unsigned char vals[256];
  void build(const unsigned char* bits, const unsigned char* v, int n) {
    std::memcpy(vals, v, n);
    std::memset(look_len, 0, sizeof(look_len));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = code;
      FFP_CHECK(code + bits[l - 1] <= (1 << l), FFP_ERR_ARG, "jpeg: bad Huffman table (over-subscribed at length %d)", l);
      for (int i = 0; i < bits[l - 1]; ++i, ++k, ++code) {
        if (l <= 9) {
          const int lo = code << (9 - l);
          for (int f = 0; f < (1 << (9 - l)); ++f) { look_len[lo + f] = (unsigned char)l; look_sym[lo + f] = v[k]; }
        }
      }
      // a length-l code must fit in l bits and must not be all ones (jdhuff.c: JERR_BAD_HUFF_TABLE); this also keeps
      // lo + f < 512 above: an over-subscribed table is rejected before it can index past the lookup arrays
      FFP_CHECK(code < (1 << l), FFP_ERR_ARG, "jpeg: bad Huffman table (codes of length %d do not fit)", l);
      maxcode[l] = bits[l - 1] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7FFFFFFF;
    present = true;
  }
};

struct BitReader {
  const unsigned char* p;
  const unsigned char* end;
  unsigned long long acc = 0;
  int nb = 0;
  bool marker = false;                          // ran into a marker: feed zeros (jdhuff.c does the same and reports corrupt data later)
  void fill() {
    while (nb <= 56) {
      unsigned b = 0;
      if (!marker && p < end) {
        b = *p;
        if (b == 0xFF) {
          if (p + 1 < end && p[1] == 0) p += 2;
          else { marker = true; b = 0; }
        } else {
          ++p;
        }
      }
      acc = (acc << 8) | b;
      nb += 8;
    }
  }
  inline unsigned peek(int n) { if (nb < n) fill(); return (unsigned)(acc >> (nb - n)) & ((1u << n) - 1u); }
  inline void skip(int n) { nb -= n; }
  inline int receive_extend(int s) {
    if (s == 0) return 0;
    const int v = (int)peek(s);
    skip(s);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
  }
  inline int symbol(const HuffTab& t) {
    const unsigned pre = peek(16);
    const int l9 = t.look_len[pre >> 7];
    if (l9) { skip(l9); return t.look_sym[pre >> 7]; }
    for (int l = 10; l <= 16; ++l) {
      const int code = (int)(pre >> (16 - l));
      if (code <= t.maxcode[l]) { skip(l); return t.vals[t.valptr[l] + code - t.mincode[l]]; }
    }
    fail(FFP_ERR_ARG, "jpeg: bad Huffman code");
    return 0;
  }
  void restart() {                              // byte-align and step over RSTn
    nb = 0; acc = 0; marker = false;
    while (p + 1 < end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;
    if (p + 1 < end) p += 2;
  }
};

}  // namespace

void jpeg_entropy_decode(const unsigned char* d, long long n, JpegScan& out, bool header_only, JpegHead* head) {
  FFP_CHECK(d && n > 4 && d[0] == 0xFF && d[1] == 0xD8, FFP_ERR_ARG, "jpeg: not a JPEG stream");
  HuffTab dc[4], ac[4];
  bool have_qt[4] = {false, false, false, false};
  int comp_id[3] = {0, 0, 0};
  int dri = 0;
  int adobe_transform = -1;
  long long i = 2;
  bool sof = false;
  while (true) {
    FFP_CHECK(i + 4 <= n && d[i] == 0xFF, FFP_ERR_ARG, "jpeg: marker expected at byte %lld", i);
    while (i + 1 < n && d[i + 1] == 0xFF) ++i;              // fill bytes
    FFP_CHECK(i + 4 <= n, FFP_ERR_ARG, "jpeg: truncated marker at byte %lld", i);
    const int m = d[i + 1];
    const int L = (d[i + 2] << 8) | d[i + 3];
    FFP_CHECK(L >= 2 && i + 2 + L <= n, FFP_ERR_ARG, "jpeg: truncated segment");
    const unsigned char* s = d + i + 4;
    const int sl = L - 2;
    if (m == 0xDB) {
      for (int k = 0; k < sl;) {
        const int pq = s[k] >> 4, tq = s[k] & 15;
        FFP_CHECK(pq == 0 && tq < 4 && k + 65 <= sl, FFP_ERR_ARG, "jpeg: unsupported quantisation table");
        for (int z = 0; z < 64; ++z) out.qt[tq][kZigzag[z]] = s[k + 1 + z];
        have_qt[tq] = true;
        k += 65;
      }
    } else if (m == 0xC4) {
      for (int k = 0; k < sl;) {
        FFP_CHECK(k + 17 <= sl, FFP_ERR_ARG, "jpeg: truncated Huffman table");
        int cnt = 0;
        for (int b = 0; b < 16; ++b) cnt += s[k + 1 + b];
        const int tc = s[k] >> 4, th = s[k] & 15;
        FFP_CHECK(tc < 2 && th < 4 && cnt <= 256 && k + 17 + cnt <= sl, FFP_ERR_ARG, "jpeg: bad Huffman table");
        (tc ? ac : dc)[th].build(s + k + 1, s + k + 17, cnt);
        if (head) {                                          // the raw table, for the device decoder (jpeg_huff.hip)
          JpegHuffSpec& hs = (tc ? head->ac : head->dc)[th];
          hs.present = true;
          hs.n = cnt;
          std::memcpy(hs.bits, s + k + 1, 16);
          std::memcpy(hs.vals, s + k + 17, cnt);
        }
        k += 17 + cnt;
      }
    } else if (m == 0xC0 || m == 0xC1) {
      FFP_CHECK(sl >= 6 && s[0] == 8, FFP_ERR_ARG, "jpeg: only 8-bit samples");
      out.h = (s[1] << 8) | s[2]; out.w = (s[3] << 8) | s[4]; out.ncomp = s[5];
      FFP_CHECK(out.h > 0 && out.w > 0 && (out.ncomp == 1 || out.ncomp == 3) && sl >= 6 + 3 * out.ncomp, FFP_ERR_ARG, "jpeg: unsupported frame header");
      for (int c = 0; c < out.ncomp; ++c) {
        comp_id[c] = s[6 + 3 * c];
        out.comp[c].hs = s[7 + 3 * c] >> 4; out.comp[c].vs = s[7 + 3 * c] & 15; out.comp[c].tq = s[8 + 3 * c];
        FFP_CHECK(out.comp[c].tq < 4, FFP_ERR_ARG, "jpeg: bad table selector");
      }
      out.hmax = out.comp[0].hs; out.vmax = out.comp[0].vs;
      for (int c = 1; c < out.ncomp; ++c) {
        FFP_CHECK(out.comp[c].hs == 1 && out.comp[c].vs == 1, FFP_ERR_ARG, "jpeg: chroma sampling factors other than 1x1 are not supported");
      }
      FFP_CHECK((out.hmax == 1 && out.vmax == 1) || (out.hmax == 2 && out.vmax == 1) || (out.hmax == 2 && out.vmax == 2), FFP_ERR_ARG,
                "jpeg: luma sampling %dx%d is not supported (4:4:4, 4:2:2, 4:2:0 are)", out.hmax, out.vmax);
      if (out.ncomp == 1) { out.comp[0].hs = out.comp[0].vs = 1; out.hmax = out.vmax = 1; }     // a lone component is never subsampled
      sof = true;
    } else if (m == 0xC2 || m == 0xC3 || (m >= 0xC5 && m <= 0xC7) || (m >= 0xC9 && m <= 0xCB) || (m >= 0xCD && m <= 0xCF)) {
      fail(FFP_ERR_ARG, "jpeg: progressive / lossless / arithmetic-coded files are not supported (marker 0x%02X)", m);
    } else if (m == 0xEE) {
      // Adobe APP14: transform 0 on a 3-component file means the samples are RGB, not YCbCr (jdapimin.c default_decompress_parms)
      if (sl >= 12 && std::memcmp(s, "Adobe", 5) == 0) adobe_transform = s[11];
    } else if (m == 0xDD) {
      FFP_CHECK(sl >= 2, FFP_ERR_ARG, "jpeg: bad DRI");
      dri = (s[0] << 8) | s[1];
    } else if (m == 0xDA) {
      FFP_CHECK(sof, FFP_ERR_ARG, "jpeg: scan before frame header");
      FFP_CHECK(!(out.ncomp == 3 && (adobe_transform == 0 || (comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B'))), FFP_ERR_ARG,
                "jpeg: RGB-coded files (Adobe transform 0 / component ids R,G,B) are not supported");
      FFP_CHECK(sl >= 1 && s[0] == out.ncomp && sl >= 4 + 2 * out.ncomp, FFP_ERR_ARG, "jpeg: non-interleaved scans are not supported");
      for (int c = 0; c < out.ncomp; ++c) {
        int ci = -1;
        for (int q = 0; q < out.ncomp; ++q) if (comp_id[q] == s[1 + 2 * c]) ci = q;
        FFP_CHECK(ci == c, FFP_ERR_ARG, "jpeg: scan component order differs from the frame header");
        out.comp[c].td = s[2 + 2 * c] >> 4; out.comp[c].ta = s[2 + 2 * c] & 15;
        FFP_CHECK(out.comp[c].td < 4 && out.comp[c].ta < 4 && dc[out.comp[c].td].present && ac[out.comp[c].ta].present && have_qt[out.comp[c].tq], FFP_ERR_ARG,
                  "jpeg: scan refers to a missing table");
      }
      i += 2 + L;
      break;
    }
    i += 2 + L;
  }
  const int mx = (out.w + 8 * out.hmax - 1) / (8 * out.hmax), my = (out.h + 8 * out.vmax - 1) / (8 * out.vmax);
  for (int c = 0; c < out.ncomp; ++c) { out.comp[c].blocks_x = mx * out.comp[c].hs; out.comp[c].blocks_y = my * out.comp[c].vs; }
  if (head) { head->dri = dri; head->data_off = i; }
  if (header_only) return;
  for (int c = 0; c < out.ncomp; ++c) {
    FFP_CHECK(out.coef[c] != nullptr, FFP_ERR_STATE, "jpeg: no coefficient plane for component %d", c);
    std::memset(out.coef[c], 0, (size_t)out.comp[c].blocks_x * out.comp[c].blocks_y * 64 * sizeof(short));
  }
  BitReader br{d + i, d + n};
  int last[3] = {0, 0, 0};
  long long count = 0;
  for (int yy = 0; yy < my; ++yy)
    for (int xx = 0; xx < mx; ++xx) {
      if (dri && count && count % dri == 0) { br.restart(); last[0] = last[1] = last[2] = 0; }
      ++count;
      for (int c = 0; c < out.ncomp; ++c) {
        const JpegComp& cp = out.comp[c];
        const HuffTab& td = dc[cp.td];
        const HuffTab& ta = ac[cp.ta];
        for (int dy = 0; dy < cp.vs; ++dy)
          for (int dx = 0; dx < cp.hs; ++dx) {
            short* blk = out.coef[c] + ((size_t)(yy * cp.vs + dy) * cp.blocks_x + xx * cp.hs + dx) * 64;
            const int sdc = br.symbol(td);
            FFP_CHECK(sdc <= 11, FFP_ERR_ARG, "jpeg: bad DC category");
            last[c] += br.receive_extend(sdc);
            blk[0] = (short)last[c];
            for (int k = 1; k < 64;) {
              const int rs = br.symbol(ta), r = rs >> 4, sz = rs & 15;
              if (sz == 0) {
                if (r != 15) break;
                k += 16;
                continue;
              }
              k += r;
              FFP_CHECK(k < 64, FFP_ERR_ARG, "jpeg: coefficient index out of range");
              blk[kZigzag[k]] = (short)br.receive_extend(sz);
              ++k;
            }
          }
      }
    }
}

}  // namespace ffp
