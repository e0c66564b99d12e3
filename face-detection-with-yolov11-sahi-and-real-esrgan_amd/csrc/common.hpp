// common.hpp — error plumbing, device buffers and the tensor/level vocabulary shared by every translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ffp.h"

namespace ffp {

// An engine's stream. env `name` (FFP_DET_CU_MASK / FFP_SR_CU_MASK, experiment): "lo-hi" = only CUs lo..hi-1 of EVERY XCD run this stream's
// kernels (hipExtStreamCreateWithCUMask) — a spatial split of the card between the detector and the enhancer.
hipStream_t create_engine_stream(const char* env_name);


// ---- errors ---------------------------------------------------------------------------------------------
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
void set_last_error(const std::string& m);
[[noreturn]] void fail(int code, const char* fmt, ...);

#define FFP_HIP(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) ::ffp::fail(FFP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

#define FFP_CHECK(cond, code, ...)            \
  do {                                        \
    if (!(cond)) ::ffp::fail(code, __VA_ARGS__); \
  } while (0)

// ---- element types --------------------------------------------------------------------------------------
enum DType : int { F32 = 0, F16 = 1 };
inline int dsize(DType t) { return t == F32 ? 4 : 2; }

// ---- device memory ---------------------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  explicit DevBuf(size_t bytes) { alloc(bytes); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t bytes) {
    release();
    if (bytes == 0) bytes = 16;
    // FFP_ALLOC_FLAGS (experiment): 3 = hipDeviceMallocUncached for buffers of at least 1 MiB (activations, tables; MTYPE_UC: nothing of them is ever dirty in
    // an L2, so a kernel boundary has nothing to write back — for this stream or for the one running beside it), 1 = hipDeviceMallocFinegrained
    static const int flags = [] { const char* e = getenv("FFP_ALLOC_FLAGS"); return e ? atoi(e) : 0; }();
    if (flags && bytes >= (1u << 20)) FFP_HIP(hipExtMallocWithFlags(&p, bytes, (unsigned)flags));
    else FFP_HIP(hipMalloc(&p, bytes));
    n = bytes;
  }
  void ensure(size_t bytes) { if (bytes > n) alloc(bytes); }
  void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// pinned host memory (asynchronous uploads of small per-call tables)
struct HostPinned {
  void* p = nullptr;
  size_t n = 0;
  HostPinned() = default;
  HostPinned(const HostPinned&) = delete;
  HostPinned& operator=(const HostPinned&) = delete;
  ~HostPinned() { if (p) (void)hipHostFree(p); }
  void ensure(size_t bytes) {
    if (bytes <= n) return;
    if (p) { (void)hipHostFree(p); p = nullptr; n = 0; }
    FFP_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    n = bytes;
  }
};

// ---- a batch of images at one resolution ("level") -------------------------------------------------------
// Activations are NHWC; the images of a batch may differ in size (ragged) and are stored back to back, so a level is
// a table {pixel offset, h, w} per image. 1x1 convolutions see one flat pixel array; 3x3 ones walk a tile table.
//
// Two modes. build(): the level describes exactly one batch shape (detector: the slice grid of a resolution is fixed).
// reserve() + assign(): CAPACITY mode (super-resolution: every frame brings a new multiset of crop sizes) — device
// tables and every buffer sized from the level are allocated once for `cap_n` images / `cap_px` pixels / `cap_t16`
// 16x16 tiles, and assign() refills the tables in place for each batch that fits. `n` / `total_px` then hold the
// CAPACITY (they size buffers and flat launches: a captured launch sequence stays valid for every batch), the padding
// entries of the image table are {total actual pixels, 0, 0} so flat-pixel kernels drop the pixels past the batch, and
// tile kernels read the actual tile count from device memory (TileTab::d_count).
struct TileTab {
  DevBuf tab;        // int4 {img, y0, x0, tile columns | tile rows << 16 of the image}
  DevBuf d_count;    // int[1]: tiles of the current batch (what the kernels loop / bound on in capacity mode)
  int n = 0;         // tiles of the current batch
  int cap = 0;       // capacity mode: table capacity == launch extent
};

struct Level {
  int n = 0;
  std::vector<int> h, w;
  std::vector<int64_t> off;      // pixel offset of image i
  int64_t total_px = 0;
  DevBuf d_tab;                   // int4 {off, h, w, 0} per image
  // tile tables keyed by tile height (tile width is always 16)
  std::map<int, TileTab> tiles;
  // capacity mode
  int cap_t16 = 0;                // > 0: capacity mode
  int act_n = 0;
  int64_t act_px = 0;
  void* stage = nullptr;          // pinned host staging for the table uploads
  size_t stage_bytes = 0;

  Level() = default;
  Level(const Level&) = delete;
  Level& operator=(const Level&) = delete;
  ~Level();
  // px_align > 1: every image starts at a multiple of px_align pixels (padding pixels behind an image belong to no image: kernels that
  // walk the flat pixel array compute them like any pixel, nothing reads them; they are kept out of the max-|value| slots). With a
  // multiple of 32 no 32-pixel fragment of the flat array straddles two images: what the per-image exponent slots of the scaled
  // split need in the 1x1 kernels (frag_img()).
  void build(const std::vector<int>& hs, const std::vector<int>& ws, hipStream_t st, int px_align = 1);
  int px_align = 1;
  int64_t real_px = 0;            // pixels that belong to images (== total_px without padding)
  DevBuf d_frag_img;              // {image, end of its real pixels} per 32-pixel fragment of the flat array (px_align % 32 == 0)
  const int* frag_img();         // built on first use (synchronous copy: call it before the plan is captured)
  void reserve(int cap_n, int64_t cap_px, int cap_tiles16, hipStream_t st);
  void assign(const std::vector<int>& hs, const std::vector<int>& ws, hipStream_t st);   // asynchronous on st (pinned staging)
  bool capacity() const { return cap_t16 > 0; }
  int64_t actual_px() const { return capacity() ? act_px : real_px; }
  int actual_n() const { return capacity() ? act_n : n; }
  // *n_launch: tiles to launch (capacity mode: the table capacity); *d_count: device count of real tiles (capacity mode) or nullptr
  const int4* tile_table(int th, int* n_launch, const int** d_count, hipStream_t st);
  // the same tiles in the PACKED form of conv_trunk.hip, everything a work item needs in one 16-byte load:
  //   {first pixel of the image, y0 | x0 << 16, h | w << 16, tile columns | tile rows << 16 of the image}   (kept under key -th)
  const int4* tile_table_packed(int th, int* n_launch, const int** d_count, hipStream_t st);
  // for every pixel of this level the flat pixel index of its nearest-x2 source in `src` (same images at half size)
  std::map<const Level*, DevBuf> up2_maps;
  const int* up2_map(const Level* src, hipStream_t st);
  long long count_tiles(int th) const;     // tiles of the current batch

 private:
  void fill_tiles(int th, TileTab& t, hipStream_t st, size_t* stage_off);
  static int tile_cap(int cap_t16, int th) { return th >= 16 ? cap_t16 : cap_t16 * ((16 + th - 1) / th); }
};

// A view of `C` channels starting at `coff` inside pixel records of `cs` elements.
struct TView {
  void* ptr = nullptr;
  DType dt = F32;
  int cs = 0, coff = 0, C = 0;
  Level* lvl = nullptr;
  // device slot holding the bit pattern of max |value| over the whole BUFFER this view belongs to (every producer of a slice
  // raises it with an atomic max; reset per inference): the tensor-wide exponent of the scaled fp16 hi/lo split (FFP_PREC_F32X3)
  unsigned* amax = nullptr;
  int amax_n = 1;                 // 1: one slot for the buffer; lvl->n: one slot per image (results independent of the batch mates)
  TView slice(int c0, int c) const { TView v = *this; v.coff += c0; v.C = c; return v; }
};

// ---- activation codes ------------------------------------------------------------------------------------
enum Act : int { ACT_NONE = 0, ACT_SILU = 1, ACT_LRELU = 2 };

}  // namespace ffp
