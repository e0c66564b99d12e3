// jpeg.hip — baseline JPEG encoding of a device-resident image (the reference's cv2.imwrite boundaries,
// /root/reference/utils/visualization.py:218-221, utils/enhancer.py:273-278), bit-compatible with libjpeg(-turbo)'s default path:
// YCbCr 4:2:0, integer "islow" DCT, Annex K tables, quality scaling, dummy edge blocks — the file equals what cv2 / Pillow write.
//   mcu     one wave per 16x16 MCU: colour conversion + edge replication + 2x2 chroma box, six 8x8 forward DCTs (row pass, column
//           pass: 48 lanes x one 1-D transform), quantisation, dummy-block rule -> int16 coefficients in zigzag order
//   bits    one thread per block: length of its entropy-coded segment (DC difference against the previous block of the component)
//   scan    exclusive prefix sum of the lengths = bit offset of every block in the scan (two levels)
//   emit    one thread per block: Huffman codes OR-ed into the zeroed bit stream at its offset (big-endian words, atomics only
//           matter for the words two blocks share); the last block pads the final byte with ones
//   stuff   0xFF -> 0xFF 0x00: per-256-byte counts, prefix sum, scatter
// Headers are written by the host (a few hundred bytes).
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include <cstdlib>

#include "jpeg.hpp"

namespace ffp {

namespace {

__constant__ unsigned char c_zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                           35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// one image of a batch: where it is, where its MCUs / blocks / raw scan bytes start in the batch-wide arrays
struct JpegImg {
  const unsigned char* ptr;
  long long stride;
  int h, w, mcus_x;
  int mcu_base;                 // first MCU (blocks: * 6)
  long long raw_base;           // first byte of its unstuffed scan in the batch stream (multiple of 256), set after the length scan
  long long bit_base;           // bit offset of its first block in the length scan, set after the scan
  long long raw_before;         // unstuffed scan bytes of all earlier images (without the alignment gaps)
};

__device__ __forceinline__ int image_of_mcu(const JpegImg* imgs, int n_img, int mcu) {
  int lo = 0, hi = n_img - 1;                          // last image with mcu_base <= mcu
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (imgs[mid].mcu_base <= mcu) lo = mid; else hi = mid - 1;
  }
  return lo;
}

struct HuffDev {              // code << 8 | length, per symbol
  unsigned dc[2][12];
  unsigned ac[2][256];
};

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jfdctint.c, one 8-point pass; first: rows (results scaled up by 4), second: columns (scaled back, overall x8)
__device__ __forceinline__ void fdct8(int (&d)[8], bool first) {
  const int t0 = d[0] + d[7], t7 = d[0] - d[7], t1 = d[1] + d[6], t6 = d[1] - d[6], t2 = d[2] + d[5], t5 = d[2] - d[5], t3 = d[3] + d[4], t4 = d[3] - d[4];
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  const int so = first ? 11 : 15;
  d[0] = first ? (t10 + t11) << 2 : descale(t10 + t11, 2);
  d[4] = first ? (t10 - t11) << 2 : descale(t10 - t11, 2);
  int z1 = (t12 + t13) * 4433;
  d[2] = descale(z1 + t13 * 6270, so);
  d[6] = descale(z1 - t12 * 15137, so);
  z1 = t4 + t7;
  int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  const int z5 = (z3 + z4) * 9633;
  const int a4 = t4 * 2446, a5 = t5 * 16819, a6 = t6 * 25172, a7 = t7 * 12299;
  z1 = -z1 * 7373; z2 = -z2 * 20995; z3 = -z3 * 16069 + z5; z4 = -z4 * 3196 + z5;
  d[7] = descale(a4 + z1 + z3, so);
  d[5] = descale(a5 + z2 + z4, so);
  d[3] = descale(a6 + z2 + z3, so);
  d[1] = descale(a7 + z1 + z4, so);
}

__device__ __forceinline__ void ycc(const unsigned char* px, int bgr, int& y, int& cb, int& cr) {
  // jccolor.c rgb_ycc_convert (16-bit fixed point)
  const int r = px[bgr ? 2 : 0], g = px[1], b = px[bgr ? 0 : 2];
  y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16;
  cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16;
  cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16;
}

__global__ void __launch_bounds__(64) jpeg_mcu_kernel(const JpegImg* __restrict__ imgs, int n_img, int bgr,
                                                      const unsigned short* __restrict__ qdiv /*[2][64] natural order, q*8*/, short* __restrict__ coef) {
  __shared__ int blk[6][64];
  const int lane = threadIdx.x;
  const JpegImg im = imgs[image_of_mcu(imgs, n_img, blockIdx.x)];
  const unsigned char* img = im.ptr;
  const int h = im.h, w = im.w, mcus_x = im.mcus_x;
  const long long stride = im.stride;
  const int local = (int)blockIdx.x - im.mcu_base;
  const int mx = local % mcus_x, my = local / mcus_x;
  // luma: 256 samples, edge pixels replicated
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int s = lane + 64 * k, yy = s >> 4, xx = s & 15;
    const int gy = min(my * 16 + yy, h - 1), gx = min(mx * 16 + xx, w - 1);
    int y, cb, cr;
    ycc(img + gy * stride + gx * 3, bgr, y, cb, cr);
    blk[(yy >> 3) * 2 + (xx >> 3)][(yy & 7) * 8 + (xx & 7)] = y - 128;
  }
  // chroma: one 2x2 box per lane. Columns replicate at full resolution, rows only up to an even height; beyond that the last
  // DOWNSAMPLED row repeats (jcsample.c / jcprepct.c order of edge expansion)
  {
    const int cy = lane >> 3, cx = lane & 7;
    const int r = min(my * 8 + cy, ((h + 1) >> 1) - 1);
    const int y0 = 2 * r, y1 = min(2 * r + 1, h - 1);
    const int x0 = min((mx * 8 + cx) * 2, w - 1), x1 = min((mx * 8 + cx) * 2 + 1, w - 1);
    int sb = 0, sr = 0, y, cb, cr;
    ycc(img + y0 * stride + x0 * 3, bgr, y, cb, cr); sb += cb; sr += cr;
    ycc(img + y0 * stride + x1 * 3, bgr, y, cb, cr); sb += cb; sr += cr;
    ycc(img + y1 * stride + x0 * 3, bgr, y, cb, cr); sb += cb; sr += cr;
    ycc(img + y1 * stride + x1 * 3, bgr, y, cb, cr); sb += cb; sr += cr;
    const int bias = (cx & 1) ? 2 : 1;
    blk[4][lane] = ((sb + bias) >> 2) - 128;
    blk[5][lane] = ((sr + bias) >> 2) - 128;
  }
  __syncthreads();
  if (lane < 48) {                                     // rows
    const int b = lane >> 3, r = lane & 7;
    int d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = blk[b][r * 8 + i];
    fdct8(d, true);
#pragma unroll
    for (int i = 0; i < 8; ++i) blk[b][r * 8 + i] = d[i];
  }
  __syncthreads();
  if (lane < 48) {                                     // columns
    const int b = lane >> 3, c = lane & 7;
    int d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = blk[b][i * 8 + c];
    fdct8(d, false);
#pragma unroll
    for (int i = 0; i < 8; ++i) blk[b][i * 8 + c] = d[i];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 6; ++k) {                        // jcdctmgr.c: symmetric round-half-up division by 8 * q
    const int v = blk[k][lane], q = qdiv[(k >= 4 ? 64 : 0) + lane];
    const int a = (abs(v) + (q >> 1)) / q;
    blk[k][lane] = v < 0 ? -a : a;
  }
  __syncthreads();
  // jccoefct.c compress_data: luma blocks outside the image's own block grid are dummies (AC 0, DC of the previous block of the MCU)
  const int hb = (h + 7) >> 3, wb = (w + 7) >> 3;
  if (lane == 0) {
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int by = my * 2 + dy, bx = mx * 2 + dx;
        if (by < hb && bx < wb) continue;
        blk[dy * 2 + dx][0] = by < hb ? blk[dy * 2 + dx - 1][0] : blk[(dy - 1) * 2 + 1][0];
      }
  }
  __syncthreads();
  short* out = coef + (size_t)blockIdx.x * 384;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int nat = c_zigzag[lane];
    int v = blk[k][nat];
    if (k < 4 && lane > 0) {
      const int by = my * 2 + (k >> 1), bx = mx * 2 + (k & 1);
      if (by >= hb || bx >= wb) v = 0;
    }
    out[k * 64 + lane] = (short)v;
  }
}

__device__ __forceinline__ int prev_dc(const short* coef, int blk_id, int first_mcu) {
  // previous block of the same component in scan order (MCU = Y Y Y Y Cb Cr); predictions start at 0 in every image
  const int m = blk_id / 6, j = blk_id % 6;
  if (j >= 1 && j <= 3) return coef[(size_t)(blk_id - 1) * 64];
  if (m == first_mcu) return 0;
  return coef[((size_t)(m - 1) * 6 + (j == 0 ? 3 : j)) * 64];
}

__device__ __forceinline__ int nbits_of(int v) { return v ? 32 - __clz(v) : 0; }

// jchuff.c encode_one_block, as a visitor over (code, length) pairs
template <typename F> __device__ __forceinline__ void encode_block(const short* c, int last_dc, const HuffDev* hd, int comp, F&& put) {
  const int tab = comp ? 1 : 0;
  int diff = c[0] - last_dc;
  int t = diff < 0 ? -diff : diff, t2 = diff < 0 ? diff - 1 : diff;
  int nb = nbits_of(t);
  unsigned e = hd->dc[tab][nb];
  put(e >> 8, e & 0xFF);
  if (nb) put((unsigned)t2 & ((1u << nb) - 1u), nb);
  int r = 0;
  for (int k = 1; k < 64; ++k) {
    const int v = c[k];
    if (v == 0) { ++r; continue; }
    while (r > 15) { e = hd->ac[tab][0xF0]; put(e >> 8, e & 0xFF); r -= 16; }
    t = v < 0 ? -v : v; t2 = v < 0 ? v - 1 : v;
    nb = nbits_of(t);
    e = hd->ac[tab][(r << 4) + nb];
    put(e >> 8, e & 0xFF);
    put((unsigned)t2 & ((1u << nb) - 1u), nb);
    r = 0;
  }
  if (r > 0) { e = hd->ac[tab][0]; put(e >> 8, e & 0xFF); }
}

__global__ void jpeg_bits_kernel(const short* __restrict__ coef, int n_blocks, const JpegImg* __restrict__ imgs, int n_img, const HuffDev* __restrict__ hd,
                                 unsigned* __restrict__ len) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  const int first = imgs[image_of_mcu(imgs, n_img, b / 6)].mcu_base;
  unsigned n = 0;
  encode_block(coef + (size_t)b * 64, prev_dc(coef, b, first), hd, b % 6 >= 4, [&](unsigned, int l) { n += l; });
  len[b] = n;
}

// bit offsets of the images' first blocks (and the grand total) out of the block scan: what the host needs to lay the streams out
__global__ void jpeg_image_bits_kernel(const JpegImg* __restrict__ imgs, int n_img, const unsigned long long* __restrict__ off,
                                       const unsigned long long* __restrict__ total, unsigned long long* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_img) out[i] = off[(size_t)imgs[i].mcu_base * 6];
  if (i == n_img) out[i] = *total;
}

// two-level exclusive scan of 32-bit counts into 64-bit offsets: 1024 items per workgroup
__global__ void __launch_bounds__(256) scan_local_kernel(const unsigned* __restrict__ in, int n, unsigned long long* __restrict__ out, unsigned long long* __restrict__ sums) {
  __shared__ unsigned long long part[256];
  const int base = blockIdx.x * 1024 + threadIdx.x * 4;
  unsigned v[4];
  unsigned long long s = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = base + i < n ? in[base + i] : 0u; s += v[i]; }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned long long a = threadIdx.x >= o ? part[threadIdx.x - o] : 0ull;
    __syncthreads();
    part[threadIdx.x] += a;
    __syncthreads();
  }
  unsigned long long run = part[threadIdx.x] - s;
#pragma unroll
  for (int i = 0; i < 4; ++i) { if (base + i < n) out[base + i] = run; run += v[i]; }
  if (threadIdx.x == 255) sums[blockIdx.x] = part[255];
}
__global__ void __launch_bounds__(1024) scan_sums_kernel(unsigned long long* sums, int n, unsigned long long* total) {
  __shared__ unsigned long long part[1024];
  const unsigned long long v = (int)threadIdx.x < n ? sums[threadIdx.x] : 0ull;
  part[threadIdx.x] = v;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const unsigned long long a = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0ull;
    __syncthreads();
    part[threadIdx.x] += a;
    __syncthreads();
  }
  if ((int)threadIdx.x < n) sums[threadIdx.x] = part[threadIdx.x] - v;
  if (threadIdx.x == 1023) *total = part[1023];
}
__global__ void scan_add_kernel(unsigned long long* out, int n, const unsigned long long* sums) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] += sums[i >> 10];
}

__global__ void jpeg_emit_kernel(const short* __restrict__ coef, int n_blocks, const JpegImg* __restrict__ imgs, int n_img, const HuffDev* __restrict__ hd,
                                 const unsigned long long* __restrict__ off, unsigned* __restrict__ words) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  const int ii = image_of_mcu(imgs, n_img, b / 6);
  const JpegImg im = imgs[ii];
  const int last_block = (ii + 1 < n_img ? imgs[ii + 1].mcu_base : n_blocks / 6) * 6 - 1;
  unsigned long long pos = (unsigned long long)im.raw_base * 8ull + (off[b] - (unsigned long long)im.bit_base);
  auto put = [&](unsigned code, int l) {
    const unsigned long long v = (unsigned long long)code << (64 - (int)(pos & 31) - l);
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    atomicOr(words + (pos >> 5), __builtin_bswap32(hi));
    if (lo) atomicOr(words + (pos >> 5) + 1, __builtin_bswap32(lo));
    pos += l;
  };
  encode_block(coef + (size_t)b * 64, prev_dc(coef, b, im.mcu_base), hd, b % 6 >= 4, put);
  if (b == last_block) {                               // jchuff.c flush_bits: fill the image's last byte with ones
    const int rem = (int)(pos & 7);
    if (rem) put(0x7Fu >> (rem - 1), 8 - rem);
  }
}

// unstuffed streams: image i occupies bytes [raw_base, raw_base + raw_len) of `raw`, raw_base a multiple of 256, so that every
// 256-byte chunk belongs to one image; chunk_img[c] = its image, the tail of an image's last chunk is not data
__device__ __forceinline__ bool stuff_valid(const JpegImg* imgs, const int* chunk_img, const long long* raw_len, int chunk, long long i) {
  const int ii = chunk_img[chunk];
  return i - imgs[ii].raw_base < raw_len[ii];
}

__global__ void __launch_bounds__(256) stuff_count_kernel(const unsigned char* __restrict__ raw, const JpegImg* __restrict__ imgs, const int* __restrict__ chunk_img,
                                                          const long long* __restrict__ raw_len, unsigned* __restrict__ cnt) {
  __shared__ unsigned s[256];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  s[threadIdx.x] = (stuff_valid(imgs, chunk_img, raw_len, blockIdx.x, i) && raw[i] == 0xFF) ? 1u : 0u;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) cnt[blockIdx.x] = s[0];
}
__global__ void __launch_bounds__(256) stuff_scatter_kernel(const unsigned char* __restrict__ raw, const JpegImg* __restrict__ imgs, const int* __restrict__ chunk_img,
                                                            const long long* __restrict__ raw_len, const unsigned long long* __restrict__ before,
                                                            unsigned char* __restrict__ out) {
  __shared__ unsigned s[256];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool ok = stuff_valid(imgs, chunk_img, raw_len, blockIdx.x, i);
  const unsigned char v = ok ? raw[i] : 0;
  const unsigned f = (ok && v == 0xFF) ? 1u : 0u;
  s[threadIdx.x] = f;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned a = (int)threadIdx.x >= o ? s[threadIdx.x - o] : 0u;
    __syncthreads();
    s[threadIdx.x] += a;
    __syncthreads();
  }
  if (ok) {                                            // compact output: earlier images' bytes + this image's bytes so far + every 0x00 inserted so far
    const JpegImg im = imgs[chunk_img[blockIdx.x]];
    const long long dst = im.raw_before + (i - im.raw_base) + (long long)before[blockIdx.x] + (s[threadIdx.x] - f);
    out[dst] = v;
    if (f) out[dst + 1] = 0;
  }
}

// number of 0x00 bytes inserted before each image's first chunk (and in total): with raw_before, where every stuffed stream starts
__global__ void jpeg_image_ff_kernel(const int* __restrict__ first_chunk, int n_img, const unsigned long long* __restrict__ before, const unsigned long long* __restrict__ total,
                                     unsigned long long* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_img) out[i] = before[first_chunk[i]];
  if (i == n_img) out[i] = *total;
}

void derive(const unsigned char* bits, const unsigned char* vals, int nvals, unsigned* out) {
  unsigned code = 0;
  int k = 0;
  for (int l = 1; l <= 16; ++l) {
    for (int i = 0; i < bits[l - 1]; ++i, ++k) out[vals[k]] = (code++ << 8) | (unsigned)l;
    code <<= 1;
  }
  (void)nvals;
}

const unsigned char kStdLuma[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const unsigned char kStdChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                      99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const unsigned char kZig[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const unsigned char kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const unsigned char kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const unsigned char kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d}, kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const unsigned char kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
    0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
    0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const unsigned char kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
    0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

void quality_tables(int quality, unsigned char (&ql)[64], unsigned char (&qc)[64]) {
  // jcparam.c jpeg_quality_scaling + jpeg_add_quant_table with force_baseline
  const int q = std::max(1, std::min(100, quality));
  const int scale = q < 50 ? 5000 / q : 200 - 2 * q;
  for (int i = 0; i < 64; ++i) {
    ql[i] = (unsigned char)std::max(1, std::min(255, (kStdLuma[i] * scale + 50) / 100));
    qc[i] = (unsigned char)std::max(1, std::min(255, (kStdChroma[i] * scale + 50) / 100));
  }
}

void put_seg(std::vector<unsigned char>& o, int marker, const std::vector<unsigned char>& payload) {
  o.push_back(0xFF); o.push_back((unsigned char)marker);
  const int L = (int)payload.size() + 2;
  o.push_back((unsigned char)(L >> 8)); o.push_back((unsigned char)L);
  o.insert(o.end(), payload.begin(), payload.end());
}

struct Workspace {
  DevBuf coef, len, off, sums, total, words, cnt, cnt_off, cnt_sums, cnt_total, out, hd, qdiv, imgs, img_vals, img_len, first_chunk, chunk_img;
  size_t cap_blocks = 0, cap_words = 0, cap_imgs = 0;
  bool tables = false;
};
std::mutex g_mu;
std::map<int, Workspace> g_ws;

void exclusive_scan(const unsigned* d_in, int n, unsigned long long* d_out, unsigned long long* d_sums, unsigned long long* d_total, hipStream_t st) {
  const int groups = (n + 1023) / 1024;
  FFP_CHECK(groups <= 1024, FFP_ERR_ARG, "jpeg: image too large for the two-level scan (%d items)", n);
  hipLaunchKernelGGL(scan_local_kernel, dim3(groups), dim3(256), 0, st, d_in, n, d_out, d_sums);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, d_sums, groups, d_total);
  hipLaunchKernelGGL(scan_add_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_out, n, d_sums);
}

}  // namespace

std::vector<unsigned char> jpeg_header(int h, int w, int quality) {
  // jcmarker.c: SOI, JFIF APP0 (1.01, aspect 1:1), DQT x2, SOF0 (2x2 / 1x1 / 1x1), DHT x4, SOS
  unsigned char ql[64], qc[64];
  quality_tables(quality, ql, qc);
  std::vector<unsigned char> o = {0xFF, 0xD8};
  put_seg(o, 0xE0, {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0});
  for (int t = 0; t < 2; ++t) {
    std::vector<unsigned char> p = {(unsigned char)t};
    for (int i = 0; i < 64; ++i) p.push_back((t ? qc : ql)[kZig[i]]);
    put_seg(o, 0xDB, p);
  }
  put_seg(o, 0xC0, {8, (unsigned char)(h >> 8), (unsigned char)h, (unsigned char)(w >> 8), (unsigned char)w, 3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1});
  const struct { int id; const unsigned char* bits; const unsigned char* vals; int n; } tabs[4] = {
      {0x00, kDcLumaBits, kDcVals, 12}, {0x10, kAcLumaBits, kAcLumaVals, 162}, {0x01, kDcChromaBits, kDcVals, 12}, {0x11, kAcChromaBits, kAcChromaVals, 162}};
  for (const auto& t : tabs) {
    std::vector<unsigned char> p = {(unsigned char)t.id};
    p.insert(p.end(), t.bits, t.bits + 16);
    p.insert(p.end(), t.vals, t.vals + t.n);
    put_seg(o, 0xC4, p);
  }
  put_seg(o, 0xDA, {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0});
  return o;
}

void jpeg_encode_batch_device(const JpegSrc* src, int n_img, int bgr, int quality, std::vector<std::vector<unsigned char>>& files, hipStream_t st) {
  files.assign(n_img, {});
  if (n_img == 0) return;
  int dev = 0;
  FFP_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_mu);
  Workspace& ws = g_ws[dev];
  if (!ws.tables) {
    HuffDev hd;
    std::memset(&hd, 0, sizeof(hd));
    derive(kDcLumaBits, kDcVals, 12, hd.dc[0]); derive(kDcChromaBits, kDcVals, 12, hd.dc[1]);
    derive(kAcLumaBits, kAcLumaVals, 162, hd.ac[0]); derive(kAcChromaBits, kAcChromaVals, 162, hd.ac[1]);
    ws.hd = DevBuf(sizeof(hd));
    FFP_HIP(hipMemcpy(ws.hd.p, &hd, sizeof(hd), hipMemcpyHostToDevice));
    ws.qdiv = DevBuf(sizeof(unsigned short) * 128);
    ws.total = DevBuf(16); ws.cnt_total = DevBuf(16);
    ws.sums = DevBuf(sizeof(unsigned long long) * 1024); ws.cnt_sums = DevBuf(sizeof(unsigned long long) * 1024);
    ws.tables = true;
  }
  std::vector<JpegImg> imgs(n_img);
  long long n_mcu = 0;
  for (int i = 0; i < n_img; ++i) {
    const JpegSrc& q = src[i];
    FFP_CHECK(q.d_img && q.h > 0 && q.w > 0 && q.h < 65536 && q.w < 65536 && q.stride >= (long long)q.w * 3, FFP_ERR_ARG, "jpeg: bad geometry of image %d (%dx%d)", i, q.w, q.h);
    imgs[i] = JpegImg{q.d_img, q.stride, q.h, q.w, (q.w + 15) / 16, (int)n_mcu, 0, 0, 0};
    n_mcu += (long long)((q.w + 15) / 16) * ((q.h + 15) / 16);
    FFP_CHECK(n_mcu * 6 <= 1024 * 1024, FFP_ERR_ARG, "jpeg: batch too large for the two-level scan (%lld blocks)", n_mcu * 6);
  }
  const int n_blocks = (int)n_mcu * 6;
  if ((size_t)n_blocks > ws.cap_blocks) {
    ws.cap_blocks = (size_t)n_blocks * 5 / 4;
    ws.coef = DevBuf(ws.cap_blocks * 64 * sizeof(short));
    ws.len = DevBuf(ws.cap_blocks * sizeof(unsigned));
    ws.off = DevBuf(ws.cap_blocks * sizeof(unsigned long long));
  }
  if ((size_t)n_img + 1 > ws.cap_imgs) {
    ws.cap_imgs = (size_t)n_img * 2 + 8;
    ws.imgs = DevBuf(ws.cap_imgs * sizeof(JpegImg));
    ws.img_vals = DevBuf(ws.cap_imgs * sizeof(unsigned long long));
    ws.img_len = DevBuf(ws.cap_imgs * sizeof(long long));
    ws.first_chunk = DevBuf(ws.cap_imgs * sizeof(int));
  }
  unsigned char ql[64], qc[64];
  quality_tables(quality, ql, qc);
  unsigned short qd[128];
  for (int i = 0; i < 64; ++i) { qd[i] = (unsigned short)(ql[i] * 8); qd[64 + i] = (unsigned short)(qc[i] * 8); }
  FFP_HIP(hipMemcpyAsync(ws.qdiv.p, qd, sizeof(qd), hipMemcpyHostToDevice, st));
  FFP_HIP(hipMemcpyAsync(ws.imgs.p, imgs.data(), sizeof(JpegImg) * n_img, hipMemcpyHostToDevice, st));

  hipLaunchKernelGGL(jpeg_mcu_kernel, dim3((unsigned)n_mcu), dim3(64), 0, st, ws.imgs.as<JpegImg>(), n_img, bgr, ws.qdiv.as<unsigned short>(), ws.coef.as<short>());
  hipLaunchKernelGGL(jpeg_bits_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, st, ws.coef.as<short>(), n_blocks, ws.imgs.as<JpegImg>(), n_img, ws.hd.as<HuffDev>(),
                     ws.len.as<unsigned>());
  exclusive_scan(ws.len.as<unsigned>(), n_blocks, ws.off.as<unsigned long long>(), ws.sums.as<unsigned long long>(), ws.total.as<unsigned long long>(), st);
  hipLaunchKernelGGL(jpeg_image_bits_kernel, dim3((n_img + 256) / 256), dim3(256), 0, st, ws.imgs.as<JpegImg>(), n_img, ws.off.as<unsigned long long>(),
                     ws.total.as<unsigned long long>(), ws.img_vals.as<unsigned long long>());
  std::vector<unsigned long long> bits(n_img + 1);
  FFP_HIP(hipMemcpyAsync(bits.data(), ws.img_vals.p, sizeof(unsigned long long) * (n_img + 1), hipMemcpyDeviceToHost, st));
  FFP_HIP(hipStreamSynchronize(st));                   // (also: qd and imgs live on this stack frame)
  // lay the unstuffed streams out: each image starts on a 256-byte boundary of the raw buffer
  std::vector<long long> raw_len(n_img);
  std::vector<int> first_chunk(n_img + 1);
  long long raw_pos = 0, raw_sum = 0;
  for (int i = 0; i < n_img; ++i) {
    raw_len[i] = (long long)((bits[i + 1] - bits[i] + 7) / 8);
    imgs[i].bit_base = (long long)bits[i];
    imgs[i].raw_base = raw_pos;
    imgs[i].raw_before = raw_sum;
    first_chunk[i] = (int)(raw_pos / 256);
    raw_pos += (raw_len[i] + 255) / 256 * 256;
    raw_sum += raw_len[i];
  }
  const int n_chunks = (int)(raw_pos / 256);
  first_chunk[n_img] = n_chunks;
  FFP_CHECK(n_chunks <= 1024 * 1024, FFP_ERR_ARG, "jpeg: batch too large for the two-level scan (%d stream chunks)", n_chunks);
  const size_t need_words = (size_t)raw_pos / 4 + 64;
  if (need_words > ws.cap_words) {
    ws.cap_words = need_words * 5 / 4;
    ws.words = DevBuf(ws.cap_words * 4);
    ws.cnt = DevBuf((ws.cap_words / 64 + 2) * sizeof(unsigned));
    ws.cnt_off = DevBuf((ws.cap_words / 64 + 2) * sizeof(unsigned long long));
    ws.chunk_img = DevBuf((ws.cap_words / 64 + 2) * sizeof(int));
    ws.out = DevBuf(ws.cap_words * 8 + 16);
  }
  std::vector<int> chunk_img((size_t)std::max(n_chunks, 1));
  for (int i = 0; i < n_img; ++i) std::fill(chunk_img.begin() + first_chunk[i], chunk_img.begin() + first_chunk[i + 1], i);
  FFP_HIP(hipMemcpyAsync(ws.imgs.p, imgs.data(), sizeof(JpegImg) * n_img, hipMemcpyHostToDevice, st));
  FFP_HIP(hipMemcpyAsync(ws.img_len.p, raw_len.data(), sizeof(long long) * n_img, hipMemcpyHostToDevice, st));
  FFP_HIP(hipMemcpyAsync(ws.first_chunk.p, first_chunk.data(), sizeof(int) * (n_img + 1), hipMemcpyHostToDevice, st));
  FFP_HIP(hipMemcpyAsync(ws.chunk_img.p, chunk_img.data(), sizeof(int) * (size_t)n_chunks, hipMemcpyHostToDevice, st));
  FFP_HIP(hipMemsetAsync(ws.words.p, 0, (size_t)raw_pos + 64, st));
  hipLaunchKernelGGL(jpeg_emit_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, st, ws.coef.as<short>(), n_blocks, ws.imgs.as<JpegImg>(), n_img, ws.hd.as<HuffDev>(),
                     ws.off.as<unsigned long long>(), ws.words.as<unsigned>());
  std::vector<unsigned long long> ff(n_img + 1, 0ull);
  if (n_chunks > 0) {
    hipLaunchKernelGGL(stuff_count_kernel, dim3(n_chunks), dim3(256), 0, st, ws.words.as<unsigned char>(), ws.imgs.as<JpegImg>(), ws.chunk_img.as<int>(), ws.img_len.as<long long>(),
                       ws.cnt.as<unsigned>());
    exclusive_scan(ws.cnt.as<unsigned>(), n_chunks, ws.cnt_off.as<unsigned long long>(), ws.cnt_sums.as<unsigned long long>(), ws.cnt_total.as<unsigned long long>(), st);
    hipLaunchKernelGGL(stuff_scatter_kernel, dim3(n_chunks), dim3(256), 0, st, ws.words.as<unsigned char>(), ws.imgs.as<JpegImg>(), ws.chunk_img.as<int>(),
                       ws.img_len.as<long long>(), ws.cnt_off.as<unsigned long long>(), ws.out.as<unsigned char>());
    hipLaunchKernelGGL(jpeg_image_ff_kernel, dim3((n_img + 256) / 256), dim3(256), 0, st, ws.first_chunk.as<int>(), n_img, ws.cnt_off.as<unsigned long long>(),
                       ws.cnt_total.as<unsigned long long>(), ws.img_vals.as<unsigned long long>());
    FFP_HIP(hipMemcpyAsync(ff.data(), ws.img_vals.p, sizeof(unsigned long long) * (n_img + 1), hipMemcpyDeviceToHost, st));
  }
  FFP_HIP(hipStreamSynchronize(st));                   // (the host vectors above are done with, too)
  FFP_HIP(hipGetLastError());
  const long long stuffed_total = raw_sum + (long long)ff[n_img];
  std::vector<unsigned char> scans((size_t)std::max<long long>(stuffed_total, 1));
  if (stuffed_total > 0) FFP_HIP(hipMemcpy(scans.data(), ws.out.p, (size_t)stuffed_total, hipMemcpyDeviceToHost));
  for (int i = 0; i < n_img; ++i) {
    const long long a = imgs[i].raw_before + (long long)ff[i], b = (i + 1 < n_img ? imgs[i + 1].raw_before : raw_sum) + (long long)ff[i + 1];
    std::vector<unsigned char>& f = files[i];
    f = jpeg_header(src[i].h, src[i].w, quality);
    f.insert(f.end(), scans.begin() + a, scans.begin() + b);
    f.push_back(0xFF); f.push_back(0xD9);
  }
}

long long jpeg_encode_device(const unsigned char* d_img, int h, int w, long long stride, int bgr, int quality, unsigned char* out, long long cap, hipStream_t st) {
  JpegSrc one{d_img, h, w, stride};
  std::vector<std::vector<unsigned char>> files;
  jpeg_encode_batch_device(&one, 1, bgr, quality, files, st);
  const long long total = (long long)files[0].size();
  if (out == nullptr || cap < total) return -total;    // caller learns the size it needs
  std::memcpy(out, files[0].data(), (size_t)total);
  return total;
}

}  // namespace ffp

namespace ffp {

// ---- decoding: everything after the entropy decoder ----------------------------------------------------------------------------------
namespace {

// jidctint.c, one 8-point pass on dequantised values; first: columns (keeps 2 fractional bits), second: rows (final scaling)
__device__ __forceinline__ void idct8(int (&v)[8], bool first) {
  int z2 = v[2], z3 = v[6];
  int z1 = (z2 + z3) * 4433;
  const int e2 = z1 - z3 * 15137, e3 = z1 + z2 * 6270;
  const int e0 = (v[0] + v[4]) << 13, e1 = (v[0] - v[4]) << 13;
  const int t10 = e0 + e3, t13 = e0 - e3, t11 = e1 + e2, t12 = e1 - e2;
  int t0 = v[7], t1 = v[5], t2 = v[3], t3 = v[1];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
  int z4 = t1 + t3;
  const int z5 = (z3 + z4) * 9633;
  t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299;
  z1 = -z1 * 7373; z2 = -z2 * 20995; z3 = -z3 * 16069 + z5; z4 = -z4 * 3196 + z5;
  t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
  const int n = first ? 11 : 18;
  v[0] = descale(t10 + t3, n); v[7] = descale(t10 - t3, n);
  v[1] = descale(t11 + t2, n); v[6] = descale(t11 - t2, n);
  v[2] = descale(t12 + t1, n); v[5] = descale(t12 - t1, n);
  v[3] = descale(t13 + t0, n); v[4] = descale(t13 - t0, n);
}

// 8 blocks per workgroup: lane = (block, column) in the first pass, (block, row) in the second; plane row pitch = blocks_x * 8.
// One launch covers the components of the scan: workgroups [wg0[c], wg0[c + 1]) work on component c.
struct IdctComp { const short* coef; unsigned char* plane; const unsigned short* qt; int n_blocks, blocks_x, wg0; };
struct IdctArgs { IdctComp c[3]; int ncomp; };
__global__ void __launch_bounds__(64) jpeg_idct_kernel(const IdctArgs A) {
  __shared__ int ws[8][64];
  int ci = 0;
  if (A.ncomp > 1 && (int)blockIdx.x >= A.c[1].wg0) ci = 1;
  if (A.ncomp > 2 && (int)blockIdx.x >= A.c[2].wg0) ci = 2;
  const short* __restrict__ coef = ci == 0 ? A.c[0].coef : ci == 1 ? A.c[1].coef : A.c[2].coef;
  unsigned char* __restrict__ plane = ci == 0 ? A.c[0].plane : ci == 1 ? A.c[1].plane : A.c[2].plane;
  const unsigned short* __restrict__ qt = ci == 0 ? A.c[0].qt : ci == 1 ? A.c[1].qt : A.c[2].qt;
  const int n_blocks = ci == 0 ? A.c[0].n_blocks : ci == 1 ? A.c[1].n_blocks : A.c[2].n_blocks;
  const int blocks_x = ci == 0 ? A.c[0].blocks_x : ci == 1 ? A.c[1].blocks_x : A.c[2].blocks_x;
  const int wg0 = ci == 0 ? A.c[0].wg0 : ci == 1 ? A.c[1].wg0 : A.c[2].wg0;
  const int lb = threadIdx.x >> 3, idx = threadIdx.x & 7;
  const int b = ((int)blockIdx.x - wg0) * 8 + lb;
  int v[8];
  if (b < n_blocks) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = (int)coef[(size_t)b * 64 + r * 8 + idx] * (int)qt[r * 8 + idx];
    idct8(v, true);
#pragma unroll
    for (int r = 0; r < 8; ++r) ws[lb][r * 8 + idx] = v[r];
  }
  __syncthreads();
  if (b < n_blocks) {
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = ws[lb][idx * 8 + c];
    idct8(v, false);
    const int by = b / blocks_x, bx = b % blocks_x;
    unsigned char* dst = plane + ((size_t)(by * 8 + idx) * blocks_x + bx) * 8;
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      lo |= (unsigned)min(max(v[c] + 128, 0), 255) << (8 * c);
      hi |= (unsigned)min(max(v[4 + c] + 128, 0), 255) << (8 * c);
    }
    *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
  }
}

// one thread per output pixel: chroma through libjpeg's "fancy" triangle upsampling (jdsample.c; replication when the component is
// at most 2 samples wide), then jdcolor.c YCbCr -> RGB
__global__ void jpeg_colour_kernel(const unsigned char* __restrict__ py, const unsigned char* __restrict__ pcb, const unsigned char* __restrict__ pcr, int h, int w,
                                   int pitch_y, int pitch_c, int ch, int cw, int hsub, int vsub, int ncomp, int bgr, long long stride, unsigned char* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const int Y = py[(size_t)y * pitch_y + x];
  int r, g, b;
  if (ncomp == 1) {
    r = g = b = Y;
  } else {
    int cbv, crv;
    auto up = [&](const unsigned char* p) {
      if (hsub == 1) return (int)p[(size_t)y * pitch_c + x];
      const int c = x >> 1;
      if (cw <= 2) return (int)p[(size_t)(vsub == 2 ? y >> 1 : y) * pitch_c + c];
      if (vsub == 1) {                                   // h2v1
        const int v0 = p[(size_t)y * pitch_c + c];
        if (x == 0 || x == 2 * cw - 1) return v0;
        return (x & 1) ? (3 * v0 + p[(size_t)y * pitch_c + c + 1] + 2) >> 2 : (3 * v0 + p[(size_t)y * pitch_c + c - 1] + 1) >> 2;
      }
      const int rr = y >> 1, far = (y & 1) ? min(rr + 1, ch - 1) : max(rr - 1, 0);
      auto colsum = [&](int cc) { return 3 * (int)p[(size_t)rr * pitch_c + cc] + (int)p[(size_t)far * pitch_c + cc]; };
      const int s0 = colsum(c);
      if (x == 0) return (s0 * 4 + 8) >> 4;
      if (x == 2 * cw - 1) return (s0 * 4 + 7) >> 4;
      return (x & 1) ? (3 * s0 + colsum(c + 1) + 7) >> 4 : (3 * s0 + colsum(c - 1) + 8) >> 4;
    };
    cbv = up(pcb) - 128;
    crv = up(pcr) - 128;
    r = Y + ((91881 * crv + 32768) >> 16);
    g = Y + ((-22554 * cbv + 32768 - 46802 * crv) >> 16);
    b = Y + ((116130 * cbv + 32768) >> 16);
    r = min(max(r, 0), 255); g = min(max(g, 0), 255); b = min(max(b, 0), 255);
  }
  unsigned char* o = out + (size_t)y * stride + (size_t)x * 3;
  o[0] = (unsigned char)(bgr ? b : r); o[1] = (unsigned char)g; o[2] = (unsigned char)(bgr ? r : b);
}

}  // namespace

void JpegDecodeWs::ensure(const JpegScan& s) {
  if (!qt.p) qt = DevBuf(sizeof(unsigned short) * 64 * 4);
  size_t off[4] = {0, 0, 0, 0};
  for (int c = 0; c < s.ncomp; ++c) {
    const size_t nb = (size_t)s.comp[c].blocks_x * s.comp[c].blocks_y;
    if (nb > cap[c]) {
      cap[c] = nb * 5 / 4 + 64;
      plane[c] = DevBuf(cap[c] * 64 + 16);
    }
    off[c + 1] = off[c] + ((nb * 64 * sizeof(short) + 255) & ~(size_t)255);
  }
  coef_bytes = off[s.ncomp];
  if (coef_bytes > coef_all.n) coef_all.alloc(coef_bytes * 5 / 4);
  for (int c = 0; c < s.ncomp; ++c) dev[c] = reinterpret_cast<short*>(static_cast<unsigned char*>(coef_all.p) + off[c]);
}

void JpegDecodeWs::ensure_host(const JpegScan& s) {
  ensure(s);
  for (int c = 0; c < s.ncomp; ++c) {
    host[c].ensure(cap[c] * 64 * sizeof(short));
    const_cast<JpegScan&>(s).coef[c] = static_cast<short*>(host[c].p);
  }
}

namespace {
std::mutex g_dec_mu;
std::vector<JpegDecodeWs*> g_dec_free;
}  // namespace

JpegDecodeWs* jpeg_ws_acquire() {
  int dev = 0;
  FFP_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_dec_mu);
  for (size_t i = 0; i < g_dec_free.size(); ++i)
    if (g_dec_free[i]->device == dev) {               // a workspace's buffers live on the device that was current when it was made
      JpegDecodeWs* w = g_dec_free[i];
      g_dec_free.erase(g_dec_free.begin() + i);
      return w;
    }
  JpegDecodeWs* w = new JpegDecodeWs();
  w->device = dev;
  return w;
}

void jpeg_ws_release(JpegDecodeWs* ws) {
  constexpr size_t kPoolCap = 32;                     // a handful of frames' worth of staging per device; beyond that, free
  std::unique_lock<std::mutex> lock(g_dec_mu);
  if (g_dec_free.size() < kPoolCap) { g_dec_free.push_back(ws); return; }
  lock.unlock();
  int cur = 0;
  const int owner = ws->device;
  (void)hipGetDevice(&cur);
  if (cur != owner) (void)hipSetDevice(owner);
  delete ws;
  if (cur != owner) (void)hipSetDevice(cur);
}

void jpeg_reconstruct_device(const JpegScan& s, JpegDecodeWs& ws, unsigned char* d_out, long long stride, int bgr, hipStream_t st, bool upload) {
  FFP_HIP(hipMemcpyAsync(ws.qt.p, s.qt, sizeof(unsigned short) * 64 * 4, hipMemcpyHostToDevice, st));
  IdctArgs A;
  A.ncomp = s.ncomp;
  int wg = 0;
  for (int c = 0; c < 3; ++c) A.c[c] = IdctComp{nullptr, nullptr, nullptr, 0, 1, 0};
  for (int c = 0; c < s.ncomp; ++c) {
    const JpegComp& cp = s.comp[c];
    const int nb = cp.blocks_x * cp.blocks_y;
    if (upload) FFP_HIP(hipMemcpyAsync(ws.dev[c], s.coef[c], (size_t)nb * 64 * sizeof(short), hipMemcpyHostToDevice, st));
    A.c[c] = IdctComp{ws.dev[c], ws.plane[c].as<unsigned char>(), ws.qt.as<unsigned short>() + 64 * cp.tq, nb, cp.blocks_x, wg};
    wg += (nb + 7) / 8;
  }
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3(wg), dim3(64), 0, st, A);
  const int hsub = s.ncomp == 3 ? s.hmax : 1, vsub = s.ncomp == 3 ? s.vmax : 1;
  const int ch = (s.h + vsub - 1) / vsub, cw = (s.w + hsub - 1) / hsub;
  hipLaunchKernelGGL(jpeg_colour_kernel, dim3((s.w + 255) / 256, s.h), dim3(256), 0, st, ws.plane[0].as<unsigned char>(),
                     s.ncomp == 3 ? ws.plane[1].as<unsigned char>() : nullptr, s.ncomp == 3 ? ws.plane[2].as<unsigned char>() : nullptr, s.h, s.w,
                     s.comp[0].blocks_x * 8, s.ncomp == 3 ? s.comp[1].blocks_x * 8 : 0, ch, cw, hsub, vsub, s.ncomp, bgr, stride, d_out);
  FFP_HIP(hipGetLastError());
}

// FFP_JPEG_HOST_HUFFMAN=1: entropy decoding on the host for every file (round-2 behaviour; A/B and the fallback's own tests)
static bool host_huffman_forced() {
  static const bool v = [] { const char* e = std::getenv("FFP_JPEG_HOST_HUFFMAN"); return e && e[0] == '1'; }();
  return v;
}

void jpeg_decode_to_device(const unsigned char* data, long long n, unsigned char* d_out, long long stride, long long cap, int bgr, hipStream_t st, int* out_h, int* out_w) {
  JpegScan s;
  JpegHead head;
  jpeg_entropy_decode(data, n, s, true, &head);
  if (out_h) *out_h = s.h;
  if (out_w) *out_w = s.w;
  if (stride == 0) stride = (long long)s.w * 3;
  FFP_CHECK(stride >= (long long)s.w * 3 && cap >= stride * (s.h - 1) + (long long)s.w * 3, FFP_ERR_ARG, "jpeg_decode: output buffer too small for %dx%d", s.w, s.h);
  JpegDecodeWs* ws = jpeg_ws_acquire();
  try {
    ws->ensure(s);
    bool done = false;
    if (!host_huffman_forced()) {
      // the whole decode is queued in one go — entropy decoding (jpeg_huff.hip) and reconstruction — and checked after ONE synchronisation
      if (jpeg_huff_decode_async(data, n, s, head, *ws, st)) {
        jpeg_reconstruct_device(s, *ws, d_out, stride, bgr, st, false);
        FFP_HIP(hipStreamSynchronize(st));
        const int rc = jpeg_huff_finish(s, *ws, st);
        if (rc == 2) {
          jpeg_reconstruct_device(s, *ws, d_out, stride, bgr, st, false);
          FFP_HIP(hipStreamSynchronize(st));
        }
        done = rc != 0;
      }
      if (!done) jpeg_huff_note_fallback();
    }
    if (!done) {                                 // the host decoder: libjpeg's behaviour on damaged streams, and its error reports
      ws->ensure_host(s);
      jpeg_entropy_decode(data, n, s, false);
      jpeg_reconstruct_device(s, *ws, d_out, stride, bgr, st, true);
      FFP_HIP(hipStreamSynchronize(st));         // the staging planes go back to the pool after this
    }
  } catch (...) {
    jpeg_ws_release(ws);
    throw;
  }
  jpeg_ws_release(ws);
}

}  // namespace ffp
