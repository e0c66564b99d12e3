// common.cpp — thread-local error string, level (ragged batch) tables.
#include "common.hpp"

#include <algorithm>

namespace ffp {

static thread_local std::string g_last_error;

void set_last_error(const std::string& m) { g_last_error = m; }
const std::string& last_error() { return g_last_error; }

void fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  throw Error(code, buf);
}

Level::~Level() {
  if (stage) (void)hipHostFree(stage);
}

void Level::build(const std::vector<int>& hs, const std::vector<int>& ws, hipStream_t st, int align) {
  FFP_CHECK(hs.size() == ws.size() && !hs.empty(), FFP_ERR_ARG, "level: empty batch");
  FFP_CHECK(!capacity(), FFP_ERR_STATE, "level: build() on a capacity-mode level (use assign)");
  FFP_CHECK(align >= 1, FFP_ERR_ARG, "level: pixel alignment %d", align);
  n = (int)hs.size();
  h = hs; w = ws;
  off.resize(n);
  total_px = 0;
  real_px = 0;
  px_align = align;
  d_frag_img.release();
  std::vector<int4> tab(n);
  for (int i = 0; i < n; ++i) {
    FFP_CHECK(h[i] > 0 && w[i] > 0, FFP_ERR_ARG, "level: image %d has size %dx%d", i, w[i], h[i]);
    off[i] = total_px;
    real_px += (int64_t)h[i] * w[i];
    total_px += ((int64_t)h[i] * w[i] + align - 1) / align * align;
  }
  FFP_CHECK(total_px < (int64_t)1 << 31, FFP_ERR_ARG, "level: %lld pixels exceed the 2^31 table limit", (long long)total_px);
  for (int i = 0; i < n; ++i) tab[i] = make_int4((int)off[i], h[i], w[i], 0);
  d_tab.alloc(sizeof(int4) * n);
  FFP_HIP(hipMemcpyAsync(d_tab.p, tab.data(), sizeof(int4) * n, hipMemcpyHostToDevice, st));
  FFP_HIP(hipStreamSynchronize(st));
  tiles.clear();
}

void Level::reserve(int cap_n, int64_t cap_px, int cap_tiles16, hipStream_t st) {
  FFP_CHECK(cap_n > 0 && cap_px > 0 && cap_tiles16 > 0, FFP_ERR_ARG, "level: empty capacity");
  FFP_CHECK(cap_px < (int64_t)1 << 31, FFP_ERR_ARG, "level: capacity of %lld pixels exceeds the 2^31 table limit", (long long)cap_px);
  (void)st;
  n = cap_n; total_px = cap_px; cap_t16 = cap_tiles16;
  act_n = 0; act_px = 0;
  h.assign(n, 0); w.assign(n, 0); off.assign(n, 0);
  d_tab.alloc(sizeof(int4) * n);
  tiles.clear();
  // staging: the image table + tile tables of heights 16 and 8 (grown if another height shows up)
  stage_bytes = sizeof(int4) * ((size_t)n + (size_t)tile_cap(cap_t16, 16) + (size_t)tile_cap(cap_t16, 8) + 8);
  if (stage) { (void)hipHostFree(stage); stage = nullptr; }
  FFP_HIP(hipHostMalloc(&stage, stage_bytes, hipHostMallocDefault));
}

// 4th component of a tile entry: the image's tile grid, columns | rows << 16 (conv_trunk.hip finds a tile's neighbours with it)
static inline int tile_grid(int h, int w, int th) { return ((w + 15) / 16) | (((h + th - 1) / th) << 16); }
// key < 0: the packed form of tile_table_packed() with tile height -key
static inline int4 tile_entry(int key, int i, int y, int x, int h, int w, int64_t off) {
  if (key > 0) return make_int4(i, y, x, tile_grid(h, w, key));
  return make_int4((int)off, y | (x << 16), h | (w << 16), tile_grid(h, w, -key));
}

void Level::fill_tiles(int th, TileTab& t, hipStream_t st, size_t* stage_off) {
  // capacity mode: tile rows of the current batch -> staging -> device (only the real entries; kernels bound on d_count)
  int4* dst = reinterpret_cast<int4*>(static_cast<unsigned char*>(stage) + *stage_off);
  int k = 0;
  const int key = th;
  th = key < 0 ? -key : key;
  for (int i = 0; i < act_n; ++i)
    for (int y = 0; y < h[i]; y += th)
      for (int x = 0; x < w[i]; x += 16) {
        FFP_CHECK(k < t.cap, FFP_ERR_STATE, "level: batch needs more than %d tiles of height %d", t.cap, th);
        dst[k++] = tile_entry(key, i, y, x, h[i], w[i], off[i]);
      }
  dst[k] = make_int4(k, 0, 0, 0);          // the count travels right behind the entries
  t.n = k;
  if (k > 0) FFP_HIP(hipMemcpyAsync(t.tab.p, dst, sizeof(int4) * k, hipMemcpyHostToDevice, st));
  FFP_HIP(hipMemcpyAsync(t.d_count.p, dst + k, sizeof(int), hipMemcpyHostToDevice, st));
  *stage_off += sizeof(int4) * ((size_t)k + 1);
}

void Level::assign(const std::vector<int>& hs, const std::vector<int>& ws, hipStream_t st) {
  FFP_CHECK(capacity(), FFP_ERR_STATE, "level: assign() needs reserve()");
  FFP_CHECK(hs.size() == ws.size() && !hs.empty(), FFP_ERR_ARG, "level: empty batch");
  FFP_CHECK((int)hs.size() <= n, FFP_ERR_ARG, "level: %d images exceed the capacity of %d", (int)hs.size(), n);
  act_n = (int)hs.size();
  int64_t px = 0;
  long long t16 = 0;
  for (int i = 0; i < n; ++i) {
    if (i < act_n) {
      FFP_CHECK(hs[i] > 0 && ws[i] > 0, FFP_ERR_ARG, "level: image %d has size %dx%d", i, ws[i], hs[i]);
      h[i] = hs[i]; w[i] = ws[i]; off[i] = px;
      px += (int64_t)hs[i] * ws[i];
      t16 += (long long)((hs[i] + 15) / 16) * ((ws[i] + 15) / 16);
    } else {
      h[i] = 0; w[i] = 0;
    }
  }
  FFP_CHECK(px <= total_px && t16 <= cap_t16, FFP_ERR_ARG, "level: batch of %lld px / %lld tiles exceeds the capacity (%lld px / %d tiles)",
            (long long)px, t16, (long long)total_px, cap_t16);
  act_px = px;
  size_t need = sizeof(int4) * ((size_t)n + 8);
  for (auto& kv : tiles) need += sizeof(int4) * ((size_t)kv.second.cap + 1);
  if (need > stage_bytes) {
    FFP_HIP(hipStreamSynchronize(st));
    (void)hipHostFree(stage);
    stage = nullptr;
    stage_bytes = need;
    FFP_HIP(hipHostMalloc(&stage, stage_bytes, hipHostMallocDefault));
  }
  int4* tab = reinterpret_cast<int4*>(stage);
  for (int i = 0; i < n; ++i) {
    if (i >= act_n) off[i] = px;            // padding entries: zero-sized images behind the last real pixel
    tab[i] = make_int4((int)off[i], h[i], w[i], 0);
  }
  FFP_HIP(hipMemcpyAsync(d_tab.p, tab, sizeof(int4) * n, hipMemcpyHostToDevice, st));
  size_t so = sizeof(int4) * (size_t)n;
  for (auto& kv : tiles) fill_tiles(kv.first, kv.second, st, &so);
}

const int* Level::frag_img() {
  FFP_CHECK(!capacity() && px_align % 32 == 0, FFP_ERR_STATE, "level: frag_img() needs images aligned to 32 pixels");
  if (!d_frag_img.p) {
    FFP_CHECK(total_px < (1ll << 31), FFP_ERR_ARG, "level: frag_img() needs fewer than 2^31 pixels");
    std::vector<int> m((size_t)(total_px / 32) * 2);
    for (int i = 0; i < n; ++i) {
      const int64_t end = i + 1 < n ? off[i + 1] : total_px;
      for (int64_t f = off[i] / 32; f < end / 32; ++f) {
        m[(size_t)f * 2] = i;
        m[(size_t)f * 2 + 1] = (int)(off[i] + (int64_t)h[i] * w[i]);        // end of the image's real pixels
      }
    }
    d_frag_img.alloc(sizeof(int) * std::max<size_t>(m.size(), 2));
    FFP_HIP(hipMemcpy(d_frag_img.p, m.data(), sizeof(int) * m.size(), hipMemcpyHostToDevice));
  }
  return d_frag_img.as<int>();
}

long long Level::count_tiles(int th) const {
  long long t = 0;
  for (int i = 0; i < n; ++i) t += (long long)((h[i] + th - 1) / th) * ((w[i] + 15) / 16);
  return t;
}

const int4* Level::tile_table_packed(int th, int* n_launch, const int** d_count, hipStream_t st) {
  for (int i = 0; i < (capacity() ? act_n : n); ++i)
    FFP_CHECK(h[i] < 65536 && w[i] < 65536 && off[i] < (1ll << 31), FFP_ERR_ARG, "level: image %d (%dx%d) is too large for packed tile entries", i, w[i], h[i]);
  return tile_table(-th, n_launch, d_count, st);
}

const int4* Level::tile_table(int key, int* n_launch, const int** d_count, hipStream_t st) {
  const int th = key < 0 ? -key : key;
  auto it = tiles.find(key);
  if (it == tiles.end()) {
    TileTab t;
    if (capacity()) {
      // first use of this tile height: allocate at capacity and fill from the current batch (synchronously: rare)
      t.cap = tile_cap(cap_t16, th);
      t.tab.alloc(sizeof(int4) * (size_t)t.cap);
      t.d_count.alloc(sizeof(int));
      std::vector<int4> v;
      for (int i = 0; i < act_n; ++i)
        for (int y = 0; y < h[i]; y += th)
          for (int x = 0; x < w[i]; x += 16) v.push_back(tile_entry(key, i, y, x, h[i], w[i], off[i]));
      FFP_CHECK((int)v.size() <= t.cap, FFP_ERR_STATE, "level: batch needs more than %d tiles of height %d", t.cap, th);
      t.n = (int)v.size();
      if (t.n > 0) FFP_HIP(hipMemcpyAsync(t.tab.p, v.data(), sizeof(int4) * v.size(), hipMemcpyHostToDevice, st));
      FFP_HIP(hipMemcpyAsync(t.d_count.p, &t.n, sizeof(int), hipMemcpyHostToDevice, st));
      FFP_HIP(hipStreamSynchronize(st));
    } else {
      std::vector<int4> v;
      for (int i = 0; i < n; ++i)
        for (int y = 0; y < h[i]; y += th)
          for (int x = 0; x < w[i]; x += 16) v.push_back(tile_entry(key, i, y, x, h[i], w[i], off[i]));
      t.tab.alloc(sizeof(int4) * v.size());
      FFP_HIP(hipMemcpyAsync(t.tab.p, v.data(), sizeof(int4) * v.size(), hipMemcpyHostToDevice, st));
      FFP_HIP(hipStreamSynchronize(st));
      t.n = (int)v.size();
    }
    it = tiles.emplace(key, std::move(t)).first;
  }
  const TileTab& t = it->second;
  *n_launch = capacity() ? t.cap : t.n;
  if (d_count) *d_count = capacity() ? t.d_count.as<int>() : nullptr;
  return t.tab.as<int4>();
}

const int* Level::up2_map(const Level* src, hipStream_t st) {
  auto it = up2_maps.find(src);
  if (it == up2_maps.end()) {
    FFP_CHECK(src->n == n, FFP_ERR_ARG, "up2_map: levels hold different batches");
    std::vector<int> m((size_t)total_px);
    for (int i = 0; i < n; ++i) {
      FFP_CHECK(src->h[i] * 2 == h[i] && src->w[i] * 2 == w[i], FFP_ERR_ARG, "up2_map: image %d is %dx%d, source %dx%d", i, w[i], h[i], src->w[i], src->h[i]);
      for (int y = 0; y < h[i]; ++y)
        for (int x = 0; x < w[i]; ++x) m[(size_t)off[i] + (size_t)y * w[i] + x] = (int)(src->off[i] + (int64_t)(y >> 1) * src->w[i] + (x >> 1));
    }
    DevBuf b(sizeof(int) * m.size());
    FFP_HIP(hipMemcpyAsync(b.p, m.data(), sizeof(int) * m.size(), hipMemcpyHostToDevice, st));
    FFP_HIP(hipStreamSynchronize(st));
    it = up2_maps.emplace(src, std::move(b)).first;
  }
  return it->second.as<int>();
}

}  // namespace ffp


namespace ffp {

hipStream_t create_engine_stream(const char* env_name) {
  hipStream_t st = nullptr;
  const char* e = env_name ? getenv(env_name) : nullptr;
  int lo = 0, hi = 0;
  if (e && sscanf(e, "%d-%d", &lo, &hi) == 2 && lo >= 0 && hi > lo && hi <= 32) {
    hipDeviceProp_t prop;
    int dev = 0;
    FFP_HIP(hipGetDevice(&dev));
    FFP_HIP(hipGetDeviceProperties(&prop, dev));
    const int ncu = prop.multiProcessorCount;                 // 256 on MI355X: 8 XCDs x 32 CUs
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    // bit i of the mask = CU i in the runtime's enumeration; both plausible enumerations (XCD-major, CU-major) are covered by setting the
    // range in every group of 32 AND the matching interleaved positions is not possible at once: FFP_CU_MASK_ORDER=1 selects CU-major
    const char* ord = getenv("FFP_CU_MASK_ORDER");
    const bool cu_major = ord && ord[0] == '1';
    for (int i = 0; i < ncu; ++i) {
      const int cu_in_xcd = cu_major ? i / 8 : i % 32;
      if (cu_in_xcd >= lo && cu_in_xcd < hi) mask[i / 32] |= 1u << (i % 32);
    }
    FFP_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    return st;
  }
  FFP_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  return st;
}

}  // namespace ffp
