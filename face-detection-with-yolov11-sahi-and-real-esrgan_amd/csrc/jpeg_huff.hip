// jpeg_huff.hip — Huffman decoding of a baseline JPEG scan ON THE DEVICE (what `cv2.imread` does first,
// reference call sites pipeline_v4_yolo/1_Inference.py:328-330, pipeline_v4_yolo/app_yolo_sahi.py:35). The file crosses PCIe as it is
// (1 MB for a 4K frame instead of 37 MB of coefficients) and the host only parses the markers.
//
// An entropy-coded segment is one serial bit stream, but Huffman streams SELF-SYNCHRONISE: a decoder started at a wrong bit position
// falls into step with the true symbol boundaries after a few dozen symbols. The decoder state between two symbols is
// (bit position, block of the MCU, zigzag index). The stream is cut into subsequences of SUB_BITS bits, one thread each:
//   1. unstuff: find the end of the entropy-coded data (first marker that is not RSTn), drop the 0x00 after every 0xFF and the RSTn
//      markers themselves (flags -> scan over 1 KiB tiles -> scatter); the clean offsets of the RSTn markers start the restart
//      SEGMENTS, whose first decoder state is known exactly, as is the one at the start of the scan;
//   2. sync (jh_sync_kernel<true>): every thread decodes its subsequence from a cold state, then, per workgroup, threads adopt their
//      predecessor's exit state as entry state and decode again until nothing changes any more (a thread whose entry did not change
//      keeps its result, so after the first pass only the few not-yet-synchronised threads work);
//   3. the same across workgroups (jh_sync_kernel<false>, repeated until no workgroup's last exit state changed: normally once);
//   4. an exclusive segmented scan over the subsequences of (blocks completed, sum of DC differences per component) gives every
//      thread its first block's index and the DC predictors at its entry;
//   5. write (jh_write_kernel): decode once more from the now exact entry states and store the quantised coefficients, natural order,
//      into the planes jpeg_idct_kernel (jpeg.hip) reads.
// Semantics are those of jpeg_dec.cpp (jdhuff.c); anything that decoder would treat specially — a bad code, a DC category above 11,
// a coefficient index past 63, a segment that does not hold exactly its MCUs, markers out of place — raises a flag instead, and
// jpeg_decode_to_device repeats the decode with the host decoder, which reproduces libjpeg's behaviour on damaged streams.
// Parallel Huffman decoding by self-synchronisation: Klein & Wiseman 2003; for JPEG on GPUs: Weissenberger & Schmidt 2021.
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstring>

#include "jpeg.hpp"

namespace ffp {

namespace {

constexpr int SUB_BITS = 512;                   // bits per subsequence: ~80 symbols of a quality-95 photograph
constexpr int NT = 128;                         // subsequences per workgroup
constexpr int FB = 10;                          // index bits of the fast table
constexpr int TILE = 1024;                      // bytes per unstuffing tile (256 threads x 4 bytes)
constexpr int LDS_WORDS = NT * SUB_BITS / 32 + 8;

__constant__ unsigned char kZig[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                       35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTabDev {                             // one table, as the decoding loop wants it (built on the device from the DHT segment)
  unsigned short lut[1 << FB];                  // FB-bit prefix -> length << 8 | symbol; 0: a longer code (or none)
  int maxcode[18];                              // [1..16] largest code of each length, -1: none
  int valoff[18];                               // [1..16] index of the first symbol of the length - its smallest code
  unsigned char vals[256];
};
static_assert(sizeof(HuffTabDev) % 16 == 0, "HuffTabDev is copied in 16-byte pieces");

struct HuffSpecDev { unsigned char bits[16]; unsigned char vals[256]; };

struct Scal {                                   // scalars the kernels hand to each other (and, at the end, to the host)
  int end;                                      // bytes of entropy-coded data before the terminating marker
  int total_clean;                              // bytes of the unstuffed stream
  int n_seg;                                    // restart segments
  int T;                                        // subsequences
  int changed;                                  // a workgroup's last exit state changed in the latest jh_sync_kernel<false>
  int err;                                      // bit mask, see the E_ constants
  int pad[2];
};
enum { E_SEGS = 1, E_CODE = 2, E_DCCAT = 4, E_INDEX = 8, E_COUNT = 16, E_PHASE = 32, E_CAP = 64 };

struct HuffParams {
  int ncomp, bpm;                               // components, blocks per MCU
  int blk_comp[6], blk_dx[6], blk_dy[6];        // block j of an MCU: component and position inside the MCU
  int hs[3], vs[3], blocks_x[3];
  int mcus_x, total_mcus, dri;
  int n_raw;                                    // bytes handed over (from the first entropy-coded byte to the end of the file)
  int seg_cap, t_cap;                           // capacities of the segment / subsequence arrays
  short* coef[3];
};

struct Bufs {
  const unsigned char* raw;
  unsigned char* clean;
  unsigned long long* tile_cnt;                 // kept bytes | RSTn markers << 32 per tile; after the scan: exclusive prefix
  int* seg_start;                               // [n_seg + 1] clean byte offsets
  int* first_sub;                               // [n_seg + 1]
  unsigned* sub_start;                          // [T] bit offset in the clean stream
  int* sub_seg;                                 // [T]
  unsigned long long* entry;                    // [T] packed decoder states
  unsigned long long* exit_;                    // [T]
  int4* cnt;                                    // [T] blocks completed, DC difference sums
  int4* pref;                                   // [T] exclusive prefix inside the segment
  const HuffTabDev* tabs;                       // [3 components][DC, AC]
  Scal* scal;
};

__device__ __forceinline__ bool is_rst(unsigned b) { return b >= 0xD0u && b <= 0xD7u; }

// ---- 1. unstuffing ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) jh_end_kernel(Bufs b, HuffParams P) {
  const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  for (int i = i0; i < min(i0 + 4, P.n_raw); ++i) {
    if (b.raw[i] != 0xFF) continue;
    if (i + 1 >= P.n_raw) { atomicMin(&b.scal->end, i); continue; }          // a dangling 0xFF: the host decoder stops there too
    const unsigned m = b.raw[i + 1];
    if (!(m == 0 || m == 0xFF || is_rst(m))) atomicMin(&b.scal->end, i);
  }
}

// byte i < end of the raw stream: kept in the clean stream? an RSTn code byte?
__device__ __forceinline__ void classify(const unsigned char* raw, int i, int n, bool* keep, bool* rst) {
  const unsigned v = raw[i], prev = i > 0 ? raw[i - 1] : 0u;
  if (v == 0xFF) {
    *keep = i + 1 < n && raw[i + 1] == 0;       // a data byte 0xFF is followed by a stuffed zero; otherwise it opens a marker / is a fill byte
    *rst = false;
  } else {
    *keep = prev != 0xFF;                       // the byte after an 0xFF is the stuffed zero or a marker code
    *rst = prev == 0xFF && is_rst(v);
  }
}

__global__ void __launch_bounds__(256) jh_count_kernel(Bufs b, HuffParams P) {
  __shared__ unsigned s_keep, s_rst;
  if (threadIdx.x == 0) { s_keep = 0; s_rst = 0; }
  __syncthreads();
  const int end = b.scal->end;
  const int i0 = blockIdx.x * TILE + threadIdx.x * 4;
  unsigned k = 0, r = 0;
  for (int i = i0; i < min(i0 + 4, end); ++i) {
    bool keep, rst;
    classify(b.raw, i, P.n_raw, &keep, &rst);
    k += keep; r += rst;
  }
  if (k) atomicAdd(&s_keep, k);
  if (r) atomicAdd(&s_rst, r);
  __syncthreads();
  if (threadIdx.x == 0) b.tile_cnt[blockIdx.x] = (unsigned long long)s_keep | ((unsigned long long)s_rst << 32);
}

// exclusive scan of n 64-bit values by one workgroup of 1024 threads (each thread a contiguous run); returns the total
__device__ unsigned long long block_scan_u64(unsigned long long* v, int n) {
  __shared__ unsigned long long s_part[1024];
  const int tid = threadIdx.x, per = (n + 1023) / 1024, lo = min(tid * per, n), hi = min(lo + per, n);
  unsigned long long sum = 0;
  for (int i = lo; i < hi; ++i) sum += v[i];
  s_part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const unsigned long long a = tid >= o ? s_part[tid - o] : 0ull;
    __syncthreads();
    s_part[tid] += a;
    __syncthreads();
  }
  unsigned long long run = s_part[tid] - sum;
  for (int i = lo; i < hi; ++i) { const unsigned long long x = v[i]; v[i] = run; run += x; }
  const unsigned long long total = s_part[1023];
  __syncthreads();
  return total;
}

__global__ void __launch_bounds__(1024) jh_scan_tiles_kernel(Bufs b, HuffParams P, int n_tiles, int n_seg_expected) {
  const unsigned long long total = block_scan_u64(b.tile_cnt, n_tiles);
  if (threadIdx.x == 0) {
    const int kept = (int)(unsigned)total, n_seg = (int)(total >> 32) + 1;
    b.scal->total_clean = kept;
    if (n_seg != n_seg_expected || n_seg > P.seg_cap) atomicOr(&b.scal->err, E_SEGS);
    b.scal->n_seg = min(n_seg, P.seg_cap);
    b.seg_start[min(n_seg, P.seg_cap)] = kept;
  }
}

__global__ void __launch_bounds__(256) jh_scatter_kernel(Bufs b, HuffParams P) {
  __shared__ unsigned s_k[256], s_r[256];
  const int end = b.scal->end, tid = threadIdx.x;
  const int i0 = blockIdx.x * TILE + tid * 4;
  bool keep[4], rst[4];
  unsigned k = 0, r = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    keep[j] = rst[j] = false;
    if (i0 + j < end) classify(b.raw, i0 + j, P.n_raw, &keep[j], &rst[j]);
    k += keep[j]; r += rst[j];
  }
  s_k[tid] = k; s_r[tid] = r;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned ak = tid >= o ? s_k[tid - o] : 0u, ar = tid >= o ? s_r[tid - o] : 0u;
    __syncthreads();
    s_k[tid] += ak; s_r[tid] += ar;
    __syncthreads();
  }
  const unsigned long long base = b.tile_cnt[blockIdx.x];
  unsigned ko = (unsigned)base + s_k[tid] - k, ro = (unsigned)(base >> 32) + s_r[tid] - r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (keep[j]) b.clean[ko++] = b.raw[i0 + j];
    if (rst[j]) { ++ro; if ((int)ro < P.seg_cap) b.seg_start[ro] = (int)ko; }       // segment `ro` starts where the marker stood
  }
}

// ---- subsequence layout ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) jh_segs_kernel(Bufs b, HuffParams P, unsigned long long* tmp) {
  const int n_seg = b.scal->n_seg, total_clean = b.scal->total_clean;
  for (int k = threadIdx.x; k < n_seg; k += 1024) {
    const long long bits = (long long)(b.seg_start[k + 1] - b.seg_start[k]) * 8;
    tmp[k] = (unsigned long long)((bits + SUB_BITS - 1) / SUB_BITS);
    if (bits <= 0) atomicOr(&b.scal->err, E_SEGS);          // every MCU needs bits: an empty segment is a damaged file
  }
  // the decoders look ahead of the bits they consume, and a workgroup stages a fixed window: zeros behind the stream
  for (int i = threadIdx.x; i < LDS_WORDS * 4 + 64; i += 1024) b.clean[total_clean + i] = 0;
  __syncthreads();
  const unsigned long long Tl = block_scan_u64(tmp, n_seg);
  for (int k = threadIdx.x; k < n_seg; k += 1024) b.first_sub[k] = (int)tmp[k];
  const int T = (int)min(Tl, (unsigned long long)P.t_cap);
  if (threadIdx.x == 0) {
    b.first_sub[n_seg] = (int)Tl;
    if (Tl > (unsigned long long)P.t_cap) atomicOr(&b.scal->err, E_CAP);
    b.scal->T = T;
  }
  __syncthreads();
  // every subsequence's segment and first bit
  for (int t = threadIdx.x; t < T; t += 1024) {
    int lo = 0, hi = n_seg;                      // the last segment k with first_sub[k] <= t (empty segments share their successor's value)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if ((int)tmp[mid] <= t) lo = mid; else hi = mid;
    }
    b.sub_seg[t] = lo;
    b.sub_start[t] = (unsigned)b.seg_start[lo] * 8u + (unsigned)(t - (int)tmp[lo]) * SUB_BITS;
  }
}

// ---- tables --------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) jh_tables_kernel(const HuffSpecDev* spec, HuffTabDev* tabs, Bufs b, HuffParams P) {
  __shared__ int s_min[17], s_ptr[17], s_max[17];
  if (blockIdx.x == 0 && threadIdx.x == 0) {     // the scalars of this decode
    b.scal->end = P.n_raw; b.scal->total_clean = 0; b.scal->n_seg = 1; b.scal->T = 0; b.scal->changed = 0; b.scal->err = 0;
    b.seg_start[0] = 0;
  }
  const HuffSpecDev& sp = spec[blockIdx.x];
  HuffTabDev& t = tabs[blockIdx.x];
  if (threadIdx.x == 0) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
      s_ptr[l] = k; s_min[l] = code;
      k += sp.bits[l - 1]; code += sp.bits[l - 1];
      s_max[l] = sp.bits[l - 1] ? code - 1 : -1;
      code <<= 1;
    }
    for (int l = 1; l <= 16; ++l) { t.maxcode[l] = s_max[l]; t.valoff[l] = s_ptr[l] - s_min[l]; }
    t.maxcode[0] = -1; t.valoff[0] = 0;
  }
  __syncthreads();
  t.vals[threadIdx.x] = sp.vals[threadIdx.x];
  for (int e = threadIdx.x; e < (1 << FB); e += 256) {
    unsigned short v = 0;
    for (int l = 1; l <= FB; ++l) {
      const int code = e >> (FB - l);
      if (code <= s_max[l] && code >= s_min[l]) { v = (unsigned short)((l << 8) | sp.vals[s_ptr[l] + code - s_min[l]]); break; }
    }
    t.lut[e] = v;
  }
}

// ---- the decoding loop -----------------------------------------------------------------------------------------------------------------------
struct HuffLds {
  HuffTabDev tab[6];                             // [component][DC, AC]
  unsigned words[LDS_WORDS];                     // the workgroup's part of the clean stream, big-endian words
  unsigned long long ex[NT];
};

__device__ __forceinline__ unsigned long long pack_state(unsigned p, int c, int z) { return ((unsigned long long)p << 16) | ((unsigned)c << 8) | (unsigned)z | (1ull << 63); }

// Decodes symbols from state (p, c, z) until the position reaches sub_end or the next symbol would cross seg_end. Counts completed
// blocks and sums DC differences per component; WRITE: stores the coefficients of block index `g` onwards, DC predictors in pred.
template <bool WRITE>
__device__ __forceinline__ void decode_run(const HuffLds& L, unsigned bit0, const HuffParams& P, unsigned sub_end, unsigned seg_end, unsigned& p_io, int& c_io, int& z_io,
                                           int4& cnt, long long g, long long g_end, int* err) {
  unsigned p = p_io;
  int c = c_io, z = z_io;
  int nblk = 0, dc[3] = {0, 0, 0};
  int pred[3] = {cnt.y, cnt.z, cnt.w};          // WRITE: predictors at entry come in through cnt
  int e = 0;
  // An adopted entry state can sit BELOW the workgroup's window base (the previous workgroup's last thread stopped early at the end of a
  // segment whose final subsequence is shorter than a word): p - bit0 would wrap and index far outside L.words. Such a state decodes
  // nothing here — it is handed on unchanged and flagged, so that a stream that ends up relying on it goes to the host decoder.
  if (p < bit0) {
    cnt = make_int4(0, WRITE ? cnt.y : 0, WRITE ? cnt.z : 0, WRITE ? cnt.w : 0);
    if (WRITE && err) *err = E_PHASE;
    return;
  }
  unsigned wi = (p - bit0) >> 5;
  unsigned long long acc = L.words[wi++];
  int nb = 32 - (int)((p - bit0) & 31u);
  short* blk = nullptr;
  bool have_blk = false;                        // the address of block g is worked out when its first coefficient is stored
  auto block_ptr = [&]() -> short* {
    if (g < 0 || g >= g_end) { e |= E_COUNT; return nullptr; }
    const long long m = g / P.bpm;
    const int j = (int)(g - m * P.bpm);
    if (j != c) { e |= E_PHASE; return nullptr; }
    const int comp = P.blk_comp[j], yy = (int)(m / P.mcus_x), xx = (int)(m - (long long)yy * P.mcus_x);
    return P.coef[comp] + ((size_t)(yy * P.vs[comp] + P.blk_dy[j]) * P.blocks_x[comp] + xx * P.hs[comp] + P.blk_dx[j]) * 64;
  };
  while (p < sub_end) {
    if (nb <= 32) { acc = (acc << 32) | L.words[wi++]; nb += 32; }
    const unsigned pk = (unsigned)(acc >> (nb - 16)) & 0xFFFFu;
    const int comp = P.blk_comp[c];
    const HuffTabDev& t = L.tab[comp * 2 + (z != 0)];
    const unsigned le = t.lut[pk >> (16 - FB)];
    int l = (int)(le >> 8), sym = (int)(le & 255u);
    if (l == 0) {
      for (l = FB + 1; l <= 16; ++l) {
        const int code = (int)(pk >> (16 - l));
        if (code <= t.maxcode[l]) { sym = t.vals[(t.valoff[l] + code) & 255]; break; }
      }
      if (l > 16) {                                                          // no such code: a decoder out of step, a damaged stream — or the
        if (p + 16u > seg_end) break;                                        // padding of the segment's last byte running into what follows it
        l = 16; sym = 0; e |= E_CODE;
      }
    }
    const int s = sym & 15, r = sym >> 4;
    const int total = l + s;
    if (p + (unsigned)total > seg_end) break;                                 // what is left of the segment is padding
    const unsigned extra = s ? (unsigned)(acc >> (nb - total)) & ((1u << s) - 1u) : 0u;
    const int val = s ? (extra < (1u << (s - 1)) ? (int)extra - (1 << s) + 1 : (int)extra) : 0;
    nb -= total;
    p += (unsigned)total;
    if (z == 0) {
      if (sym > 11) e |= E_DCCAT;
      dc[comp] += val;
      if (WRITE) {
        pred[comp] += val;
        if (!have_blk) { blk = block_ptr(); have_blk = true; }
        if (blk) blk[0] = (short)pred[comp];
      }
      z = 1;
    } else if (s == 0) {
      z = r == 15 ? z + 16 : 64;
    } else {
      z += r;
      if (z > 63) { e |= E_INDEX; z = 64; }
      else {
        if (WRITE) {
          if (!have_blk) { blk = block_ptr(); have_blk = true; }
          if (blk) blk[kZig[z]] = (short)val;
        }
        ++z;
      }
    }
    if (z >= 64) {
      z = 0;
      c = c + 1 == P.bpm ? 0 : c + 1;
      ++nblk;
      if (WRITE) { ++g; have_blk = false; }
    }
  }
  p_io = p; c_io = c; z_io = z;
  cnt = make_int4(nblk, dc[0], dc[1], dc[2]);
  if (WRITE && err) *err = e;
}

struct SubCtx { int t, seg; bool active, head; unsigned sub_start, sub_end, seg_end, bit0; };

// fills the workgroup's LDS (tables + its part of the stream) and the thread's subsequence
__device__ __forceinline__ SubCtx stage(const Bufs& b, HuffLds& L) {
  const int T = b.scal->T, tid = threadIdx.x;
  SubCtx x;
  x.t = blockIdx.x * NT + tid;
  x.active = x.t < T;
  const int t0 = blockIdx.x * NT;
  const unsigned start0 = b.sub_start[min(t0, T - 1)];
  x.bit0 = start0 & ~31u;
  x.seg = 0; x.head = false; x.sub_start = x.sub_end = x.seg_end = 0;
  if (x.active) {
    x.seg = b.sub_seg[x.t];
    x.head = x.t == b.first_sub[x.seg];
    x.sub_start = b.sub_start[x.t];
    x.seg_end = (unsigned)b.seg_start[x.seg + 1] * 8u;
    x.sub_end = min(x.sub_start + SUB_BITS, x.seg_end);
  }
  const uint4* src = reinterpret_cast<const uint4*>(b.tabs);
  uint4* dst = reinterpret_cast<uint4*>(L.tab);
  for (int i = tid; i < (int)(sizeof(HuffTabDev) * 6 / 16); i += NT) dst[i] = src[i];
  const unsigned* cw = reinterpret_cast<const unsigned*>(b.clean) + (x.bit0 >> 5);     // the clean buffer is padded with zeros past its end
  for (int i = tid; i < LDS_WORDS; i += NT) L.words[i] = __builtin_bswap32(cw[i]);
  __syncthreads();
  return x;
}

// 2./3. synchronisation. FIRST: cold start; otherwise continue from the stored states with the predecessor WORKGROUP's last exit state.
template <bool FIRST>
__global__ void __launch_bounds__(NT) jh_sync_kernel(Bufs b, HuffParams P) {
  __shared__ HuffLds L;
  if (blockIdx.x * NT >= b.scal->T) return;
  const SubCtx x = stage(b, L);
  const int tid = threadIdx.x;
  unsigned long long entry = 0, exit_ = 0, exit_at_load = 0;
  int4 cnt = make_int4(0, 0, 0, 0);
  auto run = [&](unsigned long long en) {
    unsigned p = (unsigned)(en >> 16);
    int c = (int)(en >> 8) & 255, z = (int)en & 255;
    cnt = make_int4(0, 0, 0, 0);
    decode_run<false>(L, x.bit0, P, x.sub_end, x.seg_end, p, c, z, cnt, 0, 0, nullptr);
    entry = en;
    exit_ = pack_state(p, c, z);
  };
  bool changed = false;
  if (x.active) {
    if (FIRST) {
      run(pack_state(x.sub_start, 0, 0));        // exact for the head of a segment, a guess for everyone else
    } else {
      entry = b.entry[x.t]; exit_ = exit_at_load = b.exit_[x.t]; cnt = b.cnt[x.t];
      if (tid == 0 && !x.head && x.t > 0) {
        const unsigned long long ne = __hip_atomic_load(&b.exit_[x.t - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ne != entry) { run(ne); changed = true; }
      }
    }
  }
  if (FIRST || __syncthreads_or(changed)) {
    for (int iter = 0; iter <= NT; ++iter) {
      L.ex[tid] = exit_;
      __syncthreads();
      changed = false;
      if (x.active && !x.head && tid > 0) {
        const unsigned long long ne = L.ex[tid - 1];
        if (ne != entry) { run(ne); changed = true; }
      }
      if (!__syncthreads_or(changed)) break;
    }
  }
  if (x.active) {
    b.entry[x.t] = entry; b.cnt[x.t] = cnt;
    __hip_atomic_store(&b.exit_[x.t], exit_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!FIRST && exit_ != exit_at_load && (tid == NT - 1 || x.t == b.scal->T - 1)) atomicOr(&b.scal->changed, 1);
  }
}

// 4. exclusive segmented scan over the subsequences: one workgroup, every thread a contiguous run
__global__ void __launch_bounds__(1024) jh_prefix_kernel(Bufs b) {
  __shared__ int4 s_sum[1024];
  __shared__ int s_flag[1024];
  const int T = b.scal->T, tid = threadIdx.x, per = (T + 1023) / 1024, lo = min(tid * per, T), hi = min(lo + per, T);
  auto add = [](int4 a, int4 c) { return make_int4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w); };
  int4 sum = make_int4(0, 0, 0, 0);
  int flag = 0;
  for (int t = lo; t < hi; ++t) {
    if (t == b.first_sub[b.sub_seg[t]]) { sum = make_int4(0, 0, 0, 0); flag = 1; }
    sum = add(sum, b.cnt[t]);
  }
  s_sum[tid] = sum; s_flag[tid] = flag;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {           // segmented inclusive scan: (f1, v1) o (f2, v2) = (f1 | f2, f2 ? v2 : v1 + v2)
    int4 a = make_int4(0, 0, 0, 0);
    int af = 0;
    if (tid >= o) { a = s_sum[tid - o]; af = s_flag[tid - o]; }
    __syncthreads();
    if (tid >= o) {
      if (!s_flag[tid]) s_sum[tid] = add(a, s_sum[tid]);
      s_flag[tid] |= af;
    }
    __syncthreads();
  }
  int4 run = tid > 0 ? s_sum[tid - 1] : make_int4(0, 0, 0, 0);
  for (int t = lo; t < hi; ++t) {
    if (t == b.first_sub[b.sub_seg[t]]) run = make_int4(0, 0, 0, 0);
    b.pref[t] = run;
    run = add(run, b.cnt[t]);
  }
}

// 5. the coefficients
__global__ void __launch_bounds__(NT) jh_write_kernel(Bufs b, HuffParams P) {
  __shared__ HuffLds L;
  if (blockIdx.x * NT >= b.scal->T) return;
  const SubCtx x = stage(b, L);
  if (!x.active) return;
  const unsigned long long en = b.entry[x.t];
  unsigned p = (unsigned)(en >> 16);
  int c = (int)(en >> 8) & 255, z = (int)en & 255;
  const int4 pre = b.pref[x.t];
  const long long seg_mcus = P.dri ? min((long long)P.dri, (long long)P.total_mcus - (long long)x.seg * P.dri) : (long long)P.total_mcus;
  const long long g0 = (long long)x.seg * P.dri * P.bpm, g_end = g0 + seg_mcus * P.bpm;
  int4 cnt = make_int4(0, pre.y, pre.z, pre.w);
  int e = 0;
  decode_run<true>(L, x.bit0, P, x.sub_end, x.seg_end, p, c, z, cnt, g0 + pre.x, g_end, &e);
  if (x.sub_end == x.seg_end) {                  // the segment's last subsequence: exactly its MCUs, and nothing but padding left
    if (g0 + pre.x + cnt.x != g_end || c != 0 || z != 0 || x.seg_end - p >= 8u) e |= E_COUNT;
  }
  if (e) atomicOr(&b.scal->err, e);
}

std::atomic<long long> g_stat_device{0}, g_stat_fallback{0}, g_stat_rounds{0};

}  // namespace

void jpeg_huff_stats(long long* device_decodes, long long* host_fallbacks, long long* extra_sync_rounds) {
  if (device_decodes) *device_decodes = g_stat_device.load();
  if (host_fallbacks) *host_fallbacks = g_stat_fallback.load();
  if (extra_sync_rounds) *extra_sync_rounds = g_stat_rounds.load();
}
void jpeg_huff_note_fallback() { g_stat_fallback.fetch_add(1); }

struct JpegHuffWs {
  HostPinned stage;                              // table specs + raw bytes on their way up, the scalars on their way back
  DevBuf raw, clean, tiles, tmp, seg_start, first_sub, sub_start, sub_seg, entry, exit_, cnt, pref, tabs, scal;
  Bufs b;                                        // the decode in flight
  HuffParams P;
  int sub_wgs = 0;
  size_t scal_off = 0;
};
void jpeg_huff_ws_delete(JpegHuffWs* w) { delete w; }

namespace {
void queue_scan_and_write(JpegHuffWs& H, const JpegScan& s, JpegDecodeWs& ws, hipStream_t st) {
  hipLaunchKernelGGL(jh_prefix_kernel, dim3(1), dim3(1024), 0, st, H.b);
  hipLaunchKernelGGL(jh_write_kernel, dim3(H.sub_wgs), dim3(NT), 0, st, H.b, H.P);
  FFP_HIP(hipGetLastError());
  FFP_HIP(hipMemcpyAsync(static_cast<unsigned char*>(H.stage.p) + H.scal_off, H.scal.p, sizeof(Scal), hipMemcpyDeviceToHost, st));
  (void)s; (void)ws;
}
void zero_planes(const JpegScan& s, JpegDecodeWs& ws, hipStream_t st) {
  (void)s;
  FFP_HIP(hipMemsetAsync(ws.coef_all.p, 0, ws.coef_bytes, st));        // the three planes are one allocation
}
}  // namespace

// Queues the whole entropy decode of the scan on `st` (coefficient planes: ws.dev[c]) under the assumption that ONE round of
// synchronisation across workgroups is enough — it practically always is; jpeg_huff_finish() checks after the caller's
// synchronisation. Nothing is synchronised here.
bool jpeg_huff_decode_async(const unsigned char* data, long long n, const JpegScan& s, const JpegHead& head, JpegDecodeWs& ws, hipStream_t st) {
  // limits of THIS decoder are not errors of the file: 32-bit bit offsets (a scan + trailer of 256 MiB or more) and six blocks per MCU.
  // The caller then takes the host decoder, as round 2 did for every file.
  const long long n_raw = n - head.data_off;
  if (n_raw <= 0 || n_raw >= (1ll << 28)) return false;
  {
    int bpm = 0;
    for (int c = 0; c < s.ncomp; ++c) bpm += s.comp[c].hs * s.comp[c].vs;
    if (bpm > 6) return false;
  }
  if (!ws.huff) ws.huff = new JpegHuffWs();
  JpegHuffWs& H = *ws.huff;
  HuffParams P;
  std::memset(&P, 0, sizeof(P));
  P.ncomp = s.ncomp;
  for (int c = 0; c < s.ncomp; ++c) {
    P.hs[c] = s.comp[c].hs; P.vs[c] = s.comp[c].vs; P.blocks_x[c] = s.comp[c].blocks_x;
    for (int dy = 0; dy < s.comp[c].vs; ++dy)
      for (int dx = 0; dx < s.comp[c].hs; ++dx) {
        FFP_CHECK(P.bpm < 6, FFP_ERR_ARG, "jpeg: more than 6 blocks per MCU");
        P.blk_comp[P.bpm] = c; P.blk_dx[P.bpm] = dx; P.blk_dy[P.bpm] = dy; ++P.bpm;
      }
    P.coef[c] = ws.dev[c];
  }
  P.mcus_x = s.comp[0].blocks_x / s.comp[0].hs;
  P.total_mcus = P.mcus_x * (s.comp[0].blocks_y / s.comp[0].vs);
  P.dri = head.dri;
  P.n_raw = (int)n_raw;
  const int n_seg_expected = head.dri ? (P.total_mcus + head.dri - 1) / head.dri : 1;
  P.seg_cap = n_seg_expected + 1;
  P.t_cap = (int)((n_raw * 8 + SUB_BITS - 1) / SUB_BITS) + n_seg_expected + 1;
  const int n_tiles = (int)((n_raw + TILE - 1) / TILE);
  const size_t clean_bytes = (size_t)n_raw + LDS_WORDS * 4 + 64;

  auto grow = [](DevBuf& d, size_t bytes) { if (bytes > d.n) d.alloc(bytes + (bytes >> 2)); };
  constexpr size_t SPEC_BYTES = (sizeof(HuffSpecDev) * 6 + 255) & ~(size_t)255;       // [table specs][entropy-coded bytes]: one upload
  grow(H.raw, SPEC_BYTES + (size_t)n_raw + 16); grow(H.clean, clean_bytes);
  grow(H.tiles, sizeof(unsigned long long) * (size_t)(n_tiles + 1));
  grow(H.tmp, sizeof(unsigned long long) * (size_t)(P.seg_cap + 1));
  grow(H.seg_start, sizeof(int) * (size_t)(P.seg_cap + 2)); grow(H.first_sub, sizeof(int) * (size_t)(P.seg_cap + 2));
  grow(H.sub_start, sizeof(unsigned) * (size_t)P.t_cap); grow(H.sub_seg, sizeof(int) * (size_t)P.t_cap);
  grow(H.entry, sizeof(unsigned long long) * (size_t)P.t_cap); grow(H.exit_, sizeof(unsigned long long) * (size_t)P.t_cap);
  grow(H.cnt, sizeof(int4) * (size_t)P.t_cap); grow(H.pref, sizeof(int4) * (size_t)P.t_cap);
  grow(H.tabs, sizeof(HuffTabDev) * 6); grow(H.scal, sizeof(Scal));
  const size_t scal_off = (SPEC_BYTES + (size_t)n_raw + 63) & ~(size_t)63;
  if (scal_off + sizeof(Scal) > H.stage.n) H.stage.ensure((scal_off + sizeof(Scal)) * 5 / 4);
  unsigned char* hp = static_cast<unsigned char*>(H.stage.p);
  HuffSpecDev* specs = reinterpret_cast<HuffSpecDev*>(hp);
  std::memset(specs, 0, SPEC_BYTES);
  for (int c = 0; c < s.ncomp; ++c)
    for (int k = 0; k < 2; ++k) {
      const JpegHuffSpec& hs = k ? head.ac[s.comp[c].ta] : head.dc[s.comp[c].td];
      std::memcpy(specs[c * 2 + k].bits, hs.bits, 16);
      std::memcpy(specs[c * 2 + k].vals, hs.vals, (size_t)hs.n);
    }
  std::memcpy(hp + SPEC_BYTES, data + head.data_off, (size_t)n_raw);
  FFP_HIP(hipMemcpyAsync(H.raw.p, hp, SPEC_BYTES + (size_t)n_raw, hipMemcpyHostToDevice, st));
  zero_planes(s, ws, st);

  Bufs b;
  b.raw = H.raw.as<unsigned char>() + SPEC_BYTES; b.clean = H.clean.as<unsigned char>(); b.tile_cnt = H.tiles.as<unsigned long long>();
  b.seg_start = H.seg_start.as<int>(); b.first_sub = H.first_sub.as<int>(); b.sub_start = H.sub_start.as<unsigned>(); b.sub_seg = H.sub_seg.as<int>();
  b.entry = H.entry.as<unsigned long long>(); b.exit_ = H.exit_.as<unsigned long long>(); b.cnt = H.cnt.as<int4>(); b.pref = H.pref.as<int4>();
  b.tabs = H.tabs.as<HuffTabDev>(); b.scal = H.scal.as<Scal>();
  H.b = b; H.P = P; H.sub_wgs = (P.t_cap + NT - 1) / NT; H.scal_off = scal_off;
  hipLaunchKernelGGL(jh_tables_kernel, dim3(6), dim3(256), 0, st, H.raw.as<HuffSpecDev>(), H.tabs.as<HuffTabDev>(), b, P);
  hipLaunchKernelGGL(jh_end_kernel, dim3(n_tiles), dim3(256), 0, st, b, P);
  hipLaunchKernelGGL(jh_count_kernel, dim3(n_tiles), dim3(256), 0, st, b, P);
  hipLaunchKernelGGL(jh_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, b, P, n_tiles, n_seg_expected);
  hipLaunchKernelGGL(jh_scatter_kernel, dim3(n_tiles), dim3(256), 0, st, b, P);
  hipLaunchKernelGGL(jh_segs_kernel, dim3(1), dim3(1024), 0, st, b, P, H.tmp.as<unsigned long long>());
  hipLaunchKernelGGL(jh_sync_kernel<true>, dim3(H.sub_wgs), dim3(NT), 0, st, b, P);
  hipLaunchKernelGGL(jh_sync_kernel<false>, dim3(H.sub_wgs), dim3(NT), 0, st, b, P);
  queue_scan_and_write(H, s, ws, st);
  return true;
}

// After the stream has been synchronised: 1 = the coefficient planes are right; 2 = they are right NOW, after more rounds of
// synchronisation (the reconstruction kernels have to run again); 0 = the stream needs the host decoder.
int jpeg_huff_finish(const JpegScan& s, JpegDecodeWs& ws, hipStream_t st) {
  JpegHuffWs& H = *ws.huff;
  const Scal* sc = reinterpret_cast<const Scal*>(static_cast<unsigned char*>(H.stage.p) + H.scal_off);
  int rc = 1;
  if (sc->changed && !(sc->err & (E_SEGS | E_CAP))) {
    // a workgroup's last exit state moved in the round that was queued blindly: keep going until a round changes none, then redo the rest
    rc = 2;
    int rounds = 0;
    const int limit = H.sub_wgs + 2;             // information travels at least one workgroup per round
    do {
      FFP_HIP(hipMemsetAsync(reinterpret_cast<unsigned char*>(H.scal.p) + offsetof(Scal, changed), 0, sizeof(int), st));
      hipLaunchKernelGGL(jh_sync_kernel<false>, dim3(H.sub_wgs), dim3(NT), 0, st, H.b, H.P);
      FFP_HIP(hipMemcpyAsync(static_cast<unsigned char*>(H.stage.p) + H.scal_off, H.scal.p, sizeof(Scal), hipMemcpyDeviceToHost, st));
      FFP_HIP(hipStreamSynchronize(st));
      g_stat_rounds.fetch_add(1);
    } while (sc->changed && ++rounds < limit);
    if (sc->changed) return 0;
    FFP_HIP(hipMemsetAsync(reinterpret_cast<unsigned char*>(H.scal.p) + offsetof(Scal, err), 0, sizeof(int), st));     // flags of the premature write pass
    zero_planes(s, ws, st);
    queue_scan_and_write(H, s, ws, st);
    FFP_HIP(hipStreamSynchronize(st));
  }
  if (sc->err) return 0;
  g_stat_device.fetch_add(1);
  return rc;
}

}  // namespace ffp
