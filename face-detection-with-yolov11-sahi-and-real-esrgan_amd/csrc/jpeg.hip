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

#include "jpeg.hpp"

namespace ffp {

namespace {

__constant__ unsigned char c_zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                           35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

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

__global__ void __launch_bounds__(64) jpeg_mcu_kernel(const unsigned char* __restrict__ img, int h, int w, long long stride, int bgr, int mcus_x,
                                                      const unsigned short* __restrict__ qdiv /*[2][64] natural order, q*8*/, short* __restrict__ coef) {
  __shared__ int blk[6][64];
  const int lane = threadIdx.x;
  const int mx = blockIdx.x % mcus_x, my = blockIdx.x / mcus_x;
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

__device__ __forceinline__ int prev_dc(const short* coef, int blk_id) {
  // previous block of the same component in scan order (MCU = Y Y Y Y Cb Cr)
  const int m = blk_id / 6, j = blk_id % 6;
  if (j >= 1 && j <= 3) return coef[(size_t)(blk_id - 1) * 64];
  if (m == 0) return 0;
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

__global__ void jpeg_bits_kernel(const short* __restrict__ coef, int n_blocks, const HuffDev* __restrict__ hd, unsigned* __restrict__ len) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  unsigned n = 0;
  encode_block(coef + (size_t)b * 64, prev_dc(coef, b), hd, b % 6 >= 4, [&](unsigned, int l) { n += l; });
  len[b] = n;
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

__global__ void jpeg_emit_kernel(const short* __restrict__ coef, int n_blocks, const HuffDev* __restrict__ hd, const unsigned long long* __restrict__ off,
                                 const unsigned long long* __restrict__ total_bits, unsigned* __restrict__ words) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_blocks) return;
  unsigned long long pos = off[b];
  auto put = [&](unsigned code, int l) {
    const unsigned long long v = (unsigned long long)code << (64 - (int)(pos & 31) - l);
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    atomicOr(words + (pos >> 5), __builtin_bswap32(hi));
    if (lo) atomicOr(words + (pos >> 5) + 1, __builtin_bswap32(lo));
    pos += l;
  };
  encode_block(coef + (size_t)b * 64, prev_dc(coef, b), hd, b % 6 >= 4, put);
  if (b == n_blocks - 1) {                             // jchuff.c flush_bits: fill the last byte with ones
    const int rem = (int)(*total_bits & 7);
    if (rem) put(0x7Fu >> (rem - 1), 8 - rem);
  }
}

__global__ void __launch_bounds__(256) stuff_count_kernel(const unsigned char* __restrict__ raw, long long n, unsigned* __restrict__ cnt) {
  __shared__ unsigned s[256];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  s[threadIdx.x] = (i < n && raw[i] == 0xFF) ? 1u : 0u;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) cnt[blockIdx.x] = s[0];
}
__global__ void __launch_bounds__(256) stuff_scatter_kernel(const unsigned char* __restrict__ raw, long long n, const unsigned long long* __restrict__ before,
                                                            unsigned char* __restrict__ out) {
  __shared__ unsigned s[256];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const unsigned char v = i < n ? raw[i] : 0;
  const unsigned f = (i < n && v == 0xFF) ? 1u : 0u;
  s[threadIdx.x] = f;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned a = (int)threadIdx.x >= o ? s[threadIdx.x - o] : 0u;
    __syncthreads();
    s[threadIdx.x] += a;
    __syncthreads();
  }
  if (i < n) {
    const long long dst = i + (long long)before[blockIdx.x] + (s[threadIdx.x] - f);
    out[dst] = v;
    if (f) out[dst + 1] = 0;
  }
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
  DevBuf coef, len, off, sums, total, words, cnt, cnt_off, cnt_sums, cnt_total, out, hd, qdiv;
  size_t cap_blocks = 0, cap_words = 0;
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

long long jpeg_encode_device(const unsigned char* d_img, int h, int w, long long stride, int bgr, int quality, unsigned char* out, long long cap, hipStream_t st) {
  FFP_CHECK(d_img && h > 0 && w > 0 && h < 65536 && w < 65536 && stride >= (long long)w * 3, FFP_ERR_ARG, "jpeg: bad image geometry %dx%d", w, h);
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
  const int mcus_x = (w + 15) / 16, mcus_y = (h + 15) / 16, n_mcu = mcus_x * mcus_y, n_blocks = n_mcu * 6;
  if ((size_t)n_blocks > ws.cap_blocks) {
    ws.cap_blocks = (size_t)n_blocks * 5 / 4;
    ws.coef = DevBuf(ws.cap_blocks * 64 * sizeof(short));
    ws.len = DevBuf(ws.cap_blocks * sizeof(unsigned));
    ws.off = DevBuf(ws.cap_blocks * sizeof(unsigned long long));
  }
  // worst case 16 + 11 bits for the DC and 63 x (16 + 10) for the ACs of a block
  const size_t max_words = ((size_t)n_blocks * 1665 + 31) / 32 + 2;
  if (max_words > ws.cap_words) {
    ws.cap_words = max_words * 5 / 4;
    ws.words = DevBuf(ws.cap_words * 4);
    const size_t groups = (ws.cap_words * 4 + 255) / 256;
    ws.cnt = DevBuf(groups * sizeof(unsigned));
    ws.cnt_off = DevBuf(groups * sizeof(unsigned long long));
    ws.out = DevBuf(ws.cap_words * 8 + 16);
  }
  unsigned char ql[64], qc[64];
  quality_tables(quality, ql, qc);
  unsigned short qd[128];
  for (int i = 0; i < 64; ++i) { qd[i] = (unsigned short)(ql[i] * 8); qd[64 + i] = (unsigned short)(qc[i] * 8); }
  FFP_HIP(hipMemcpyAsync(ws.qdiv.p, qd, sizeof(qd), hipMemcpyHostToDevice, st));
  FFP_HIP(hipStreamSynchronize(st));                   // qd lives on this stack frame

  hipLaunchKernelGGL(jpeg_mcu_kernel, dim3(n_mcu), dim3(64), 0, st, d_img, h, w, stride, bgr, mcus_x, ws.qdiv.as<unsigned short>(), ws.coef.as<short>());
  hipLaunchKernelGGL(jpeg_bits_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, st, ws.coef.as<short>(), n_blocks, ws.hd.as<HuffDev>(), ws.len.as<unsigned>());
  exclusive_scan(ws.len.as<unsigned>(), n_blocks, ws.off.as<unsigned long long>(), ws.sums.as<unsigned long long>(), ws.total.as<unsigned long long>(), st);
  unsigned long long total_bits = 0;
  FFP_HIP(hipMemcpyAsync(&total_bits, ws.total.p, 8, hipMemcpyDeviceToHost, st));
  FFP_HIP(hipStreamSynchronize(st));
  const long long raw_bytes = (long long)((total_bits + 7) / 8);
  FFP_HIP(hipMemsetAsync(ws.words.p, 0, (size_t)((raw_bytes + 3) / 4 + 1) * 4, st));
  hipLaunchKernelGGL(jpeg_emit_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, st, ws.coef.as<short>(), n_blocks, ws.hd.as<HuffDev>(),
                     ws.off.as<unsigned long long>(), ws.total.as<unsigned long long>(), ws.words.as<unsigned>());
  const int groups = (int)((raw_bytes + 255) / 256);
  hipLaunchKernelGGL(stuff_count_kernel, dim3(groups), dim3(256), 0, st, ws.words.as<unsigned char>(), raw_bytes, ws.cnt.as<unsigned>());
  exclusive_scan(ws.cnt.as<unsigned>(), groups, ws.cnt_off.as<unsigned long long>(), ws.cnt_sums.as<unsigned long long>(), ws.cnt_total.as<unsigned long long>(), st);
  hipLaunchKernelGGL(stuff_scatter_kernel, dim3(groups), dim3(256), 0, st, ws.words.as<unsigned char>(), raw_bytes, ws.cnt_off.as<unsigned long long>(),
                     ws.out.as<unsigned char>());
  unsigned long long n_ff = 0;
  FFP_HIP(hipMemcpyAsync(&n_ff, ws.cnt_total.p, 8, hipMemcpyDeviceToHost, st));
  FFP_HIP(hipStreamSynchronize(st));
  FFP_HIP(hipGetLastError());
  const std::vector<unsigned char> head = jpeg_header(h, w, quality);
  const long long scan_bytes = raw_bytes + (long long)n_ff, total = (long long)head.size() + scan_bytes + 2;
  if (out == nullptr || cap < total) return -total;    // caller learns the size it needs
  std::memcpy(out, head.data(), head.size());
  FFP_HIP(hipMemcpy(out + head.size(), ws.out.p, (size_t)scan_bytes, hipMemcpyDeviceToHost));
  out[total - 2] = 0xFF; out[total - 1] = 0xD9;
  return total;
}

}  // namespace ffp
