// common.cpp — thread-local error string, level (ragged batch) tables.
#include "common.hpp"

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

void Level::build(const std::vector<int>& hs, const std::vector<int>& ws, hipStream_t st) {
  FFP_CHECK(hs.size() == ws.size() && !hs.empty(), FFP_ERR_ARG, "level: empty batch");
  n = (int)hs.size();
  h = hs; w = ws;
  off.resize(n);
  total_px = 0;
  std::vector<int4> tab(n);
  for (int i = 0; i < n; ++i) {
    FFP_CHECK(h[i] > 0 && w[i] > 0, FFP_ERR_ARG, "level: image %d has size %dx%d", i, w[i], h[i]);
    off[i] = total_px;
    total_px += (int64_t)h[i] * w[i];
  }
  FFP_CHECK(total_px < (int64_t)1 << 31, FFP_ERR_ARG, "level: %lld pixels exceed the 2^31 table limit", (long long)total_px);
  for (int i = 0; i < n; ++i) tab[i] = make_int4((int)off[i], h[i], w[i], 0);
  d_tab.alloc(sizeof(int4) * n);
  FFP_HIP(hipMemcpyAsync(d_tab.p, tab.data(), sizeof(int4) * n, hipMemcpyHostToDevice, st));
  FFP_HIP(hipStreamSynchronize(st));
  tiles.clear();
}

long long Level::count_tiles(int th) const {
  long long t = 0;
  for (int i = 0; i < n; ++i) t += (long long)((h[i] + th - 1) / th) * ((w[i] + 15) / 16);
  return t;
}

const int4* Level::tile_table(int th, int* n_tiles, hipStream_t st) {
  auto it = tiles.find(th);
  if (it == tiles.end()) {
    std::vector<int4> t;
    for (int i = 0; i < n; ++i)
      for (int y = 0; y < h[i]; y += th)
        for (int x = 0; x < w[i]; x += 16) t.push_back(make_int4(i, y, x, 0));
    DevBuf b(sizeof(int4) * t.size());
    FFP_HIP(hipMemcpyAsync(b.p, t.data(), sizeof(int4) * t.size(), hipMemcpyHostToDevice, st));
    FFP_HIP(hipStreamSynchronize(st));
    it = tiles.emplace(th, std::make_pair(std::move(b), (int)t.size())).first;
  }
  *n_tiles = it->second.second;
  return it->second.first.as<int4>();
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
