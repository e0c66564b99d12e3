// asan_host_check.cpp — host-side robustness driver for the parsers of untrusted bytes (host ASan/UBSan build).
// Built by `build.py --asan` with the host compiler only (-fsanitize=address,undefined; no device code, no GPU needed) from
// common.cpp + weights.cpp + jpeg_dec.cpp + this file. It feeds jpeg_entropy_decode and WeightFile::parse
//   * every file given on the command line, unchanged (must parse),
//   * truncations of it at every length in the header region and at a stride through the body,
//   * seeded byte / bit mutations, with extra weight on the marker segments (DHT, DQT, SOF, SOS, DRI),
//   * hand-made malformed streams (over-subscribed DHT, all-ones codes, fill bytes at the end, zero-length segments, RGB-coded files),
// and expects each call to either succeed or throw ffp::Error — any out-of-bounds access, overflow or undefined shift aborts the
// process through the sanitizer runtime. Usage: asan_host_check jpeg <file>... ffpw <file>...
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <string>
#include <vector>

#include "jpeg.hpp"
#include "weights.hpp"

using namespace ffp;
typedef std::vector<unsigned char> Bytes;

static long long g_ok = 0, g_rejected = 0;

static Bytes slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
  return Bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// the same two passes jpeg_decode_to_device makes: header -> planes sized from it -> full entropy decode
static bool try_jpeg(const Bytes& b) {
  // exact-size heap copy: one byte past the stream is a redzone
  std::unique_ptr<unsigned char[]> d(new unsigned char[b.size() ? b.size() : 1]);
  if (!b.empty()) memcpy(d.get(), b.data(), b.size());
  try {
    JpegScan s;
    jpeg_entropy_decode(d.get(), (long long)b.size(), s, true);
    std::vector<short> planes[3];
    for (int c = 0; c < s.ncomp; ++c) {
      planes[c].resize((size_t)s.comp[c].blocks_x * s.comp[c].blocks_y * 64);
      s.coef[c] = planes[c].data();
    }
    jpeg_entropy_decode(d.get(), (long long)b.size(), s, false);
    ++g_ok;
    return true;
  } catch (const Error&) {
    ++g_rejected;
    return false;
  }
}

static bool try_ffpw(const Bytes& b) {
  std::unique_ptr<unsigned char[]> d(new unsigned char[b.size() ? b.size() : 1]);
  if (!b.empty()) memcpy(d.get(), b.data(), b.size());
  try {
    WeightFile wf;
    wf.parse(d.get(), b.size());
    double acc = 0;                                       // touch the first and last element of every tensor the parser accepted
    for (auto& kv : wf.t)
      if (kv.second.numel()) acc += kv.second.data[0] + kv.second.data[kv.second.numel() - 1];
    (void)acc;
    ++g_ok;
    return true;
  } catch (const Error&) {
    ++g_rejected;
    return false;
  }
}

static void put_segment(Bytes& o, int marker, const Bytes& payload) {
  o.push_back(0xFF); o.push_back((unsigned char)marker);
  const int L = (int)payload.size() + 2;
  o.push_back((unsigned char)(L >> 8)); o.push_back((unsigned char)(L & 255));
  o.insert(o.end(), payload.begin(), payload.end());
}

// end of the header region = offset just past the first SOS segment
static size_t header_end(const Bytes& b) {
  size_t i = 2;
  while (i + 4 <= b.size() && b[i] == 0xFF) {
    const int m = b[i + 1], L = (b[i + 2] << 8) | b[i + 3];
    i += 2 + L;
    if (m == 0xDA) break;
  }
  return i < b.size() ? i : b.size();
}

static void jpeg_cases(const Bytes& good, unsigned seed) {
  if (!try_jpeg(good)) { fprintf(stderr, "a valid file was rejected\n"); exit(3); }
  const size_t he = header_end(good);
  for (size_t n = 0; n <= he && n <= good.size(); ++n) try_jpeg(Bytes(good.begin(), good.begin() + n));
  for (size_t n = he; n < good.size(); n += 1 + (good.size() - he) / 97) try_jpeg(Bytes(good.begin(), good.begin() + n));
  std::mt19937 rng(seed);
  for (int it = 0; it < 3000; ++it) {
    Bytes m = good;
    const int k = 1 + (int)(rng() % 4);
    for (int q = 0; q < k; ++q) {
      const size_t pos = (rng() % 3) ? rng() % he : rng() % m.size();      // two thirds of the hits land in the header
      switch (rng() % 4) {
        case 0: m[pos] = (unsigned char)rng(); break;
        case 1: m[pos] ^= (unsigned char)(1u << (rng() % 8)); break;
        case 2: m[pos] = 0xFF; break;
        default: m[pos] = 0; break;
      }
    }
    try_jpeg(m);
  }
  // hand-made streams: the valid file's segments with one DHT replaced
  auto with_dht = [&](const Bytes& dht_payload) {
    Bytes o = {0xFF, 0xD8};
    put_segment(o, 0xC4, dht_payload);
    o.insert(o.end(), good.begin() + 2, good.end());       // the file's own tables follow and may override: also put ours last
    Bytes o2(good.begin(), good.begin() + he);
    // insert in front of SOS: find the SOS marker start
    size_t i = 2, sos = 2;
    while (i + 4 <= good.size() && good[i] == 0xFF) { const int mk = good[i + 1], L = (good[i + 2] << 8) | good[i + 3]; if (mk == 0xDA) { sos = i; break; } i += 2 + L; }
    Bytes o3(good.begin(), good.begin() + sos);
    put_segment(o3, 0xC4, dht_payload);
    o3.insert(o3.end(), good.begin() + sos, good.end());
    try_jpeg(o);
    try_jpeg(o3);
  };
  for (int tc = 0; tc < 2; ++tc) {
    Bytes p(17 + 200, 0);                                  // 200 codes of length 1
    p[0] = (unsigned char)(tc << 4); p[1] = 200;
    for (int i = 0; i < 200; ++i) p[17 + i] = (unsigned char)i;
    with_dht(p);
    Bytes q(17 + 3, 0);                                    // three codes of length 1
    q[0] = (unsigned char)(tc << 4); q[1] = 3; q[17] = 1; q[18] = 2; q[19] = 3;
    with_dht(q);
    Bytes r(17 + 2, 0);                                    // two codes of length 1: complete, but uses the all-ones code
    r[0] = (unsigned char)(tc << 4); r[1] = 2; r[17] = 0; r[18] = 1;
    with_dht(r);
    Bytes s(17 + 256, 0);                                  // 256 codes of length 8 (all-ones used), then nothing
    s[0] = (unsigned char)(tc << 4); s[8] = 255; s[9] = 1;
    for (int i = 0; i < 256; ++i) s[17 + i] = (unsigned char)i;
    with_dht(s);
    Bytes t(17 + 255, 0);                                  // over-subscribed only at length 16
    t[0] = (unsigned char)(tc << 4); t[1] = 1; t[16] = 254;
    for (int i = 0; i < 255; ++i) t[17 + i] = (unsigned char)i;
    with_dht(t);
    Bytes u(17 + 12, 0);                                   // every symbol is category 15: shifts in receive_extend
    u[0] = (unsigned char)(tc << 4); u[4] = 12;
    for (int i = 0; i < 12; ++i) u[17 + i] = 0xFF;
    with_dht(u);
  }
  {
    Bytes o(good.begin(), good.begin() + 2);                // fill bytes running into the end of the stream
    for (int i = 0; i < 9; ++i) o.push_back(0xFF);
    try_jpeg(o);
    o.push_back(0xDB);
    try_jpeg(o);
    o.push_back(0x00);
    try_jpeg(o);
    Bytes z = {0xFF, 0xD8, 0xFF, 0xDB, 0x00, 0x00};          // segment length below 2
    try_jpeg(z);
    Bytes a = {0xFF, 0xD8};
    Bytes ad = {'A', 'd', 'o', 'b', 'e', 0, 100, 0, 0, 0, 0, 0};   // APP14, transform 0: RGB-coded
    put_segment(a, 0xEE, ad);
    a.insert(a.end(), good.begin() + 2, good.end());
    const bool three = [&] { try { JpegScan s; jpeg_entropy_decode(good.data(), (long long)good.size(), s, true); return s.ncomp == 3; } catch (const Error&) { return false; } }();
    if (try_jpeg(a) && three) { fprintf(stderr, "an Adobe transform-0 file was accepted as YCbCr\n"); exit(3); }
  }
}

static void ffpw_cases(const Bytes& good, unsigned seed) {
  if (!try_ffpw(good)) { fprintf(stderr, "a valid container was rejected\n"); exit(3); }
  const size_t he = good.size() < 4096 ? good.size() : 4096;
  for (size_t n = 0; n <= he; ++n) try_ffpw(Bytes(good.begin(), good.begin() + n));
  std::mt19937 rng(seed);
  for (int it = 0; it < 4000; ++it) {
    Bytes m = good;
    const int k = 1 + (int)(rng() % 3);
    for (int q = 0; q < k; ++q) {
      const size_t pos = rng() % he;
      switch (rng() % 3) {
        case 0: m[pos] = (unsigned char)rng(); break;
        case 1: m[pos] = 0xFF; break;
        default: m[pos] ^= (unsigned char)(1u << (rng() % 8)); break;
      }
    }
    if (rng() % 4 == 0) m.resize(rng() % m.size());
    try_ffpw(m);
  }
}

int main(int argc, char** argv) {
  enum { NONE, JPEG, FFPW } mode = NONE;
  unsigned seed = 1;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "jpeg") { mode = JPEG; continue; }
    if (a == "ffpw") { mode = FFPW; continue; }
    const Bytes b = slurp(argv[i]);
    if (mode == JPEG) jpeg_cases(b, seed++);
    else if (mode == FFPW) ffpw_cases(b, seed++);
    else { fprintf(stderr, "usage: %s jpeg <file>... ffpw <file>...\n", argv[0]); return 2; }
  }
  printf("asan_host_check: %lld inputs parsed, %lld rejected, no sanitizer report\n", g_ok, g_rejected);
  return 0;
}
