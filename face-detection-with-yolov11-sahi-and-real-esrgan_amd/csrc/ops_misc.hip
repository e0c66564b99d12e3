// ops_misc.hip — the HBM-bound operators around the MFMA convolutions: depthwise 3x3, SPPF pooling, nearest x2,
// C2PSA attention, LetterBox preprocessing. All work on ragged NHWC batches described by level tables.
#include "ops.hpp"
#include "letterbox.hpp"

namespace ffp {

namespace {

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { *p = (T)v; }

__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == ACT_SILU) return v / (1.0f + expf(-v));
  if (act == ACT_LRELU) return v >= 0.f ? v : v * 0.2f;
  return v;
}

// locate the image containing flat pixel `gp` (tables are tiny: linear scan from a per-thread guess is fine, binary here)
__device__ __forceinline__ int find_img(const int4* tab, int n, long long gp) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((long long)tab[mid].x <= gp) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// 4 consecutive channels as floats (16-byte fp32 / 8-byte fp16 accesses)
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<_Float16>(const _Float16* p) {
  union { uint2 u; _Float16 h[4]; } v;
  v.u = *reinterpret_cast<const uint2*>(p);
  return make_float4((float)v.h[0], (float)v.h[1], (float)v.h[2], (float)v.h[3]);
}
template <typename T> __device__ __forceinline__ void st4(T* p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void st4<_Float16>(_Float16* p, float4 v) {
  union { uint2 u; _Float16 h[4]; } o;
  o.h[0] = (_Float16)v.x; o.h[1] = (_Float16)v.y; o.h[2] = (_Float16)v.z; o.h[3] = (_Float16)v.w;
  *reinterpret_cast<uint2*>(p) = o.u;
}

// ---- depthwise 3x3 s1 p1: one thread = one pixel x 4 channels ------------------------------------------------------
template <typename T>
__global__ void dwconv3x3_kernel(const T* __restrict__ in, int in_cs, int in_coff, int grp, int grp_stride, int grp_off,
                                 T* __restrict__ out, int out_cs, int out_coff, const float* __restrict__ w /*[9][C]*/,
                                 const float* __restrict__ bias, const T* __restrict__ res, int r_cs, int r_coff, int C,
                                 int act, const int4* __restrict__ tab, int n_img, long long total_px) {
  const int C4 = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total_px * C4) return;
  const int c = (int)(idx % C4) * 4;
  const long long gp = idx / C4;
  const int im = find_img(tab, n_img, gp);
  const int4 t = tab[im];
  const int lp = (int)(gp - t.x), y = lp / t.z, x = lp - y * t.z;
  const int cin = in_coff + (c / grp) * grp_stride + grp_off + (c % grp);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = y + ky - 1;
    if ((unsigned)yy >= (unsigned)t.y) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xx = x + kx - 1;
      if ((unsigned)xx >= (unsigned)t.z) continue;
      const float4 v = ld4<T>(in + ((size_t)t.x + (size_t)yy * t.z + xx) * in_cs + cin);
      const float4 k = *reinterpret_cast<const float4*>(w + (ky * 3 + kx) * C + c);
      acc.x = fmaf(v.x, k.x, acc.x); acc.y = fmaf(v.y, k.y, acc.y); acc.z = fmaf(v.z, k.z, acc.z); acc.w = fmaf(v.w, k.w, acc.w);
    }
  }
  const float4 b = *reinterpret_cast<const float4*>(bias + c);
  float4 v = make_float4(act_fn(acc.x + b.x, act), act_fn(acc.y + b.y, act), act_fn(acc.z + b.z, act), act_fn(acc.w + b.w, act));
  if (res) {
    const float4 r = ld4<T>(res + (size_t)gp * r_cs + r_coff + c);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  st4<T>(out + (size_t)gp * out_cs + out_coff + c, v);
}

// Same op, one thread = FOUR horizontally adjacent pixels x 4 channels: 18 input vectors instead of 36 and one image lookup
// instead of four (the per-pixel form runs at ~40 % of its HBM roofline on the stride-8 cls-tower layers). Taps are
// accumulated in the same (ky, kx) order with zeros for out-of-image taps: bit-identical to dwconv3x3_kernel.
template <typename T>
__device__ __forceinline__ void dwconv_strip_body(const T* __restrict__ in, int in_cs, int in_coff, int grp, int grp_stride, int grp_off,
                                                  T* __restrict__ out, int out_cs, int out_coff, const float* __restrict__ w,
                                                  const float* __restrict__ bias, const T* __restrict__ res, int r_cs, int r_coff, int C,
                                                  int act, const int4* __restrict__ tab, int n_img, long long total_px, float& mx);

template <typename T>
__global__ void dwconv3x3_strip_kernel(const T* __restrict__ in, int in_cs, int in_coff, int grp, int grp_stride, int grp_off,
                                       T* __restrict__ out, int out_cs, int out_coff, const float* __restrict__ w /*[9][C]*/,
                                       const float* __restrict__ bias, const T* __restrict__ res, int r_cs, int r_coff, int C,
                                       int act, const int4* __restrict__ tab, int n_img, long long total_px, unsigned* __restrict__ amax) {
  float mx = 0.f;                        // largest |value| this thread stores (see TView::amax)
  dwconv_strip_body<T>(in, in_cs, in_coff, grp, grp_stride, grp_off, out, out_cs, out_coff, w, bias, res, r_cs, r_coff, C, act, tab, n_img, total_px, mx);
  if (amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const unsigned b = __float_as_uint(mx);
    if ((threadIdx.x & 63) == 0 && b > __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax, b);
  }
}

template <typename T>
__device__ __forceinline__ void dwconv_strip_body(const T* __restrict__ in, int in_cs, int in_coff, int grp, int grp_stride, int grp_off,
                                       T* __restrict__ out, int out_cs, int out_coff, const float* __restrict__ w /*[9][C]*/,
                                       const float* __restrict__ bias, const T* __restrict__ res, int r_cs, int r_coff, int C,
                                       int act, const int4* __restrict__ tab, int n_img, long long total_px, float& mx) {
  const int C4 = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long groups = (total_px + 3) >> 2;
  if (idx >= groups * C4) return;
  const int c = (int)(idx % C4) * 4;
  const long long gp0 = (idx / C4) << 2;
  const int cin = in_coff + (c / grp) * grp_stride + grp_off + (c % grp);
  float4 k[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) k[i] = *reinterpret_cast<const float4*>(w + i * C + c);
  const float4 b = *reinterpret_cast<const float4*>(bias + c);
  const int im0 = find_img(tab, n_img, gp0);
  const int4 t0 = tab[im0];
  const int lp0 = (int)(gp0 - t0.x), y0 = lp0 / t0.z, x0 = lp0 - y0 * t0.z;
  auto finish = [&](float4 acc, long long gp) {
    float4 v = make_float4(act_fn(acc.x + b.x, act), act_fn(acc.y + b.y, act), act_fn(acc.z + b.z, act), act_fn(acc.w + b.w, act));
    if (res) {
      const float4 r = ld4<T>(res + (size_t)gp * r_cs + r_coff + c);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    st4<T>(out + (size_t)gp * out_cs + out_coff + c, v);
  };
  if (x0 + 3 < t0.z) {          // the four pixels sit in one row of one image
    float4 v[3][6];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y0 + ky - 1;
      const bool rok = (unsigned)yy < (unsigned)t0.y;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int xx = x0 + j - 1;
        v[ky][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rok && (unsigned)xx < (unsigned)t0.z) v[ky][j] = ld4<T>(in + ((size_t)t0.x + (size_t)yy * t0.z + xx) * in_cs + cin);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float4 a = v[ky][i + kx], q = k[ky * 3 + kx];
          acc.x = fmaf(a.x, q.x, acc.x); acc.y = fmaf(a.y, q.y, acc.y); acc.z = fmaf(a.z, q.z, acc.z); acc.w = fmaf(a.w, q.w, acc.w);
        }
      finish(acc, gp0 + i);
    }
    return;
  }
  for (int i = 0; i < 4; ++i) {   // row end / image boundary / tail: pixel by pixel
    const long long gp = gp0 + i;
    if (gp >= total_px) return;
    const int im = find_img(tab, n_img, gp);
    const int4 t = tab[im];
    const int lp = (int)(gp - t.x), y = lp / t.z, x = lp - y * t.z;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      if ((unsigned)yy >= (unsigned)t.y) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = x + kx - 1;
        if ((unsigned)xx >= (unsigned)t.z) continue;
        const float4 a = ld4<T>(in + ((size_t)t.x + (size_t)yy * t.z + xx) * in_cs + cin), q = k[ky * 3 + kx];
        acc.x = fmaf(a.x, q.x, acc.x); acc.y = fmaf(a.y, q.y, acc.y); acc.z = fmaf(a.z, q.z, acc.z); acc.w = fmaf(a.w, q.w, acc.w);
      }
    }
    finish(acc, gp);
  }
}

// ---- SPPF pooling: 5x5, 9x9, 13x13 windows (== 3 chained 5x5 pools with -inf padding), pixel x 4 channels per thread
template <typename T>
__global__ void sppf_pool_kernel(const T* __restrict__ in, int in_cs, int in_coff, T* __restrict__ y1, T* __restrict__ y2,
                                 T* __restrict__ y3, int o_cs, int o1, int o2, int o3, int C, const int4* __restrict__ tab,
                                 int n_img, long long total_px) {
  const int C4 = C >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total_px * C4) return;
  const int c = (int)(idx % C4) * 4;
  const long long gp = idx / C4;
  const int im = find_img(tab, n_img, gp);
  const int4 t = tab[im];
  const int lp = (int)(gp - t.x), y = lp / t.z, x = lp - y * t.z;
  const float ninf = -INFINITY;
  float4 m5 = make_float4(ninf, ninf, ninf, ninf), m9 = m5, m13 = m5;
  for (int dy = -6; dy <= 6; ++dy) {
    const int yy = y + dy;
    if ((unsigned)yy >= (unsigned)t.y) continue;
    for (int dx = -6; dx <= 6; ++dx) {
      const int xx = x + dx;
      if ((unsigned)xx >= (unsigned)t.z) continue;
      const float4 v = ld4<T>(in + ((size_t)t.x + (size_t)yy * t.z + xx) * in_cs + in_coff + c);
      const int r = max(abs(dy), abs(dx));
      m13.x = fmaxf(m13.x, v.x); m13.y = fmaxf(m13.y, v.y); m13.z = fmaxf(m13.z, v.z); m13.w = fmaxf(m13.w, v.w);
      if (r <= 4) { m9.x = fmaxf(m9.x, v.x); m9.y = fmaxf(m9.y, v.y); m9.z = fmaxf(m9.z, v.z); m9.w = fmaxf(m9.w, v.w); }
      if (r <= 2) { m5.x = fmaxf(m5.x, v.x); m5.y = fmaxf(m5.y, v.y); m5.z = fmaxf(m5.z, v.z); m5.w = fmaxf(m5.w, v.w); }
    }
  }
  st4<T>(y1 + (size_t)gp * o_cs + o1 + c, m5);
  st4<T>(y2 + (size_t)gp * o_cs + o2 + c, m9);
  st4<T>(y3 + (size_t)gp * o_cs + o3 + c, m13);
}

// Small maps (the stride-32 level: 16x16 at 512, 32x32 at 1024): one workgroup = one image x 4 channels, the map lives in
// LDS and the square windows are separated into a row pass and a column pass (26 LDS reads per pixel instead of 169
// global reads; max is exact, so the result is identical to the direct kernel above).
template <typename T>
__global__ void __launch_bounds__(256) sppf_pool_lds_kernel(const T* __restrict__ in, int in_cs, int in_coff, T* __restrict__ y1, T* __restrict__ y2,
                                                            T* __restrict__ y3, int o_cs, int o1, int o2, int o3, const int4* __restrict__ tab) {
  constexpr int MAXPX = 1024;
  __shared__ float4 sv[MAXPX], r5[MAXPX], r9[MAXPX], r13[MAXPX];
  const int4 t = tab[blockIdx.y];
  const int H = t.y, W = t.z, N = H * W, c = blockIdx.x * 4;
  const float ninf = -INFINITY;
  for (int i = threadIdx.x; i < N; i += 256) sv[i] = ld4<T>(in + ((size_t)t.x + i) * in_cs + in_coff + c);
  __syncthreads();
  auto mx = [](float4& a, const float4& b) { a.x = fmaxf(a.x, b.x); a.y = fmaxf(a.y, b.y); a.z = fmaxf(a.z, b.z); a.w = fmaxf(a.w, b.w); };
  for (int i = threadIdx.x; i < N; i += 256) {
    const int y = i / W, x = i - y * W;
    float4 m5 = make_float4(ninf, ninf, ninf, ninf), m9 = m5, m13 = m5;
    for (int dx = -6; dx <= 6; ++dx) {
      const int xx = x + dx;
      if ((unsigned)xx >= (unsigned)W) continue;
      const float4 v = sv[y * W + xx];
      mx(m13, v);
      if (abs(dx) <= 4) mx(m9, v);
      if (abs(dx) <= 2) mx(m5, v);
    }
    r5[i] = m5; r9[i] = m9; r13[i] = m13;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += 256) {
    const int y = i / W, x = i - y * W;
    float4 m5 = make_float4(ninf, ninf, ninf, ninf), m9 = m5, m13 = m5;
    for (int dy = -6; dy <= 6; ++dy) {
      const int yy = y + dy;
      if ((unsigned)yy >= (unsigned)H) continue;
      const int j = yy * W + x;
      mx(m13, r13[j]);
      if (abs(dy) <= 4) mx(m9, r9[j]);
      if (abs(dy) <= 2) mx(m5, r5[j]);
    }
    const size_t gp = (size_t)t.x + i;
    st4<T>(y1 + gp * o_cs + o1 + c, m5);
    st4<T>(y2 + gp * o_cs + o2 + c, m9);
    st4<T>(y3 + gp * o_cs + o3 + c, m13);
  }
}

// ---- nearest x2: one thread copies 16 bytes ---------------------------------------------------------------------------
__global__ void upsample2x_kernel(const unsigned char* __restrict__ in, int in_cs_b, int in_coff_b, const int4* __restrict__ in_tab,
                                  unsigned char* __restrict__ out, int out_cs_b, int out_coff_b, const int4* __restrict__ out_tab,
                                  int n_img, int nvec, long long total_out_px) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total_out_px * nvec) return;
  const int v = (int)(idx % nvec);
  const long long gp = idx / nvec;
  const int im = find_img(out_tab, n_img, gp);
  const int4 to = out_tab[im], ti = in_tab[im];
  const int lp = (int)(gp - to.x), y = lp / to.z, x = lp - y * to.z;
  const uint4 d = *reinterpret_cast<const uint4*>(in + ((size_t)ti.x + (size_t)(y >> 1) * ti.z + (x >> 1)) * in_cs_b + in_coff_b + v * 16);
  *reinterpret_cast<uint4*>(out + (size_t)gp * out_cs_b + out_coff_b + v * 16) = d;
}

// ---- direct 3x3 conv for image inputs (3 real channels, NHWC4 fp32 / NHWC8 fp16): one thread = one output pixel x COUT ----
// HBM-bound (27 inputs in, COUT out per pixel); weights [tap][3][COUT] broadcast from LDS.
template <typename T, int COUT>
__global__ void __launch_bounds__(256) conv3x3_c3_direct_kernel(const T* __restrict__ in, int in_cs, T* __restrict__ out, int out_cs, int out_coff,
                                                                const float* __restrict__ w, const float* __restrict__ bias, int stride, int act,
                                                                const int4* __restrict__ in_tab, const int4* __restrict__ out_tab, int n_img,
                                                                long long total_out_px) {
  __shared__ __attribute__((aligned(16))) float ws[27 * COUT];
  for (int i = threadIdx.x; i < 27 * COUT; i += 256) ws[i] = w[i];
  __syncthreads();
  const long long gp = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gp >= total_out_px) return;
  const int im = find_img(out_tab, n_img, gp);
  const int4 to = out_tab[im], ti = in_tab[im];
  const int lp = (int)(gp - to.x);
  if (lp >= to.y * to.z) return;              // capacity-mode level: pixel past the current batch
  const int oy = lp / to.z, ox = lp - oy * to.z;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {        // taps stay a loop: unrolling all 27 x COUT/4 LDS reads spills
    const int ky = tap / 3, kx = tap - ky * 3;
    const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
    float px[3] = {0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)ti.y && (unsigned)ix < (unsigned)ti.z) {
      const float4 v = ld4<T>(in + ((size_t)ti.x + (size_t)iy * ti.z + ix) * in_cs);
      px[0] = v.x; px[1] = v.y; px[2] = v.z;
    }
    const float* wt = ws + tap * 3 * COUT;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int c = 0; c < COUT; c += 4) {
        const float4 k = *reinterpret_cast<const float4*>(wt + ci * COUT + c);
        acc[c] = fmaf(px[ci], k.x, acc[c]); acc[c + 1] = fmaf(px[ci], k.y, acc[c + 1]);
        acc[c + 2] = fmaf(px[ci], k.z, acc[c + 2]); acc[c + 3] = fmaf(px[ci], k.w, acc[c + 3]);
      }
  }
  T* op = out + (size_t)gp * out_cs + out_coff;
#pragma unroll
  for (int c = 0; c < COUT; c += 4) {
    const float4 b = *reinterpret_cast<const float4*>(bias + c);
    st4<T>(op + c, make_float4(act_fn(acc[c] + b.x, act), act_fn(acc[c + 1] + b.y, act), act_fn(acc[c + 2] + b.z, act), act_fn(acc[c + 3] + b.w, act)));
  }
}

// ---- C2PSA attention ----------------------------------------------------------------------------------------------
// grid (ceil(Nmax/256), nh, n_img), 256 threads: one query per thread, keys/values streamed through LDS 64 at a time,
// fp32 online softmax. qkv pixel record per head: [q(kd) | k(kd) | v(hd)].
template <typename T, int KD, int HD>
__global__ void __launch_bounds__(256) psa_attention_kernel(const T* __restrict__ qkv, int q_cs, int q_coff,
                                                            T* __restrict__ out, int o_cs, int o_coff,
                                                            const int4* __restrict__ tab, float scale) {
  constexpr int KB = 64;
  __shared__ float ks[KB][KD];
  __shared__ float vs[KB][HD];
  const int4 t = tab[blockIdx.z];
  const int N = t.y * t.z;
  const int head = blockIdx.y;
  const int qi = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x * 256 >= N) return;
  const int hoff = q_coff + head * (2 * KD + HD);
  float q[KD];
  const bool active = qi < N;
  {
    const T* qp = qkv + ((size_t)t.x + (active ? qi : 0)) * q_cs + hoff;
#pragma unroll
    for (int d = 0; d < KD; ++d) q[d] = ldf(qp + d);
  }
  float m = -INFINITY, l = 0.f;
  float o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  for (int j0 = 0; j0 < N; j0 += KB) {
    __syncthreads();
    for (int i = threadIdx.x; i < KB * (KD + HD); i += 256) {
      const int j = i / (KD + HD), d = i - j * (KD + HD);
      float v = 0.f;
      if (j0 + j < N) v = ldf(qkv + ((size_t)t.x + j0 + j) * q_cs + hoff + KD + d);
      if (d < KD) ks[j][d] = v; else vs[j][d - KD] = v;
    }
    __syncthreads();
    const int jn = min(KB, N - j0);
    for (int j = 0; j < jn; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < KD; ++d) s = fmaf(q[d], ks[j][d], s);
      s *= scale;
      if (s > m) {
        const float f = expf(m - s);
        l *= f;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= f;
        m = s;
      }
      const float pj = expf(s - m);
      l += pj;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] = fmaf(pj, vs[j][d], o[d]);
    }
  }
  if (!active) return;
  const float inv = 1.0f / l;
  T* op = out + ((size_t)t.x + qi) * o_cs + o_coff + head * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) stf(op + d, o[d] * inv);
}

// The same attention on the matrix cores (exact-fp32 MFMA 32x32x2): a wave owns 32 queries (one per lane column), a
// workgroup 128. Per block of 32 keys: S[key][query] = K Q^T (16 MFMAs, A = K rows from LDS, B = the lane's own query,
// pre-scaled), online softmax per query (a lane holds 16 of its query's 32 scores, the other 16 sit in lane ^ 32), then
// O[dim][query] += V^T P (2 x 16 MFMAs): the k-order of that product is chosen so that the B operand of step t is exactly
// the lane's score register t — no data movement between the two products.
template <typename T>
__global__ void __launch_bounds__(256) psa_attention_mfma_kernel(const T* __restrict__ qkv, int q_cs, int q_coff, T* __restrict__ out, int o_cs,
                                                                 int o_coff, const int4* __restrict__ tab, float scale) {
  constexpr int KD = 32, HD = 64, KB = 64;
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  __shared__ float ks[KB][KD + 1];          // +1: the 32 lanes of an A-operand read walk 32 different keys at one d
  __shared__ float vs[KB][HD + 4];
  const int4 t = tab[blockIdx.z];
  const int N = t.y * t.z;
  if (blockIdx.x * 128 >= N) return;
  const int head = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, hh = lane >> 5;
  const int hoff = q_coff + head * (2 * KD + HD);
  const int qi = blockIdx.x * 128 + wave * 32 + n;
  const bool active = qi < N;
  float qreg[16];
  {
    const T* qp = qkv + ((size_t)t.x + (active ? qi : N - 1)) * q_cs + hoff;
#pragma unroll
    for (int s = 0; s < 16; ++s) qreg[s] = ldf(qp + 2 * s + hh) * scale;
  }
  float m = -INFINITY, l = 0.f;
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  for (int j0 = 0; j0 < N; j0 += KB) {
    __syncthreads();
    for (int i = threadIdx.x; i < KB * ((KD + HD) / 4); i += 256) {       // 24 float4 per key: [k(32) | v(64)]
      const int j = i / ((KD + HD) / 4), d = (i - j * ((KD + HD) / 4)) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j0 + j < N) v = ld4<T>(qkv + ((size_t)t.x + j0 + j) * q_cs + hoff + KD + d);
      float* dst = d < KD ? &ks[j][d] : &vs[j][d - KD];
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();
#pragma unroll 1
    for (int kb = 0; kb < KB && j0 + kb < N; kb += 32) {
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 16; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[kb + n][2 * s + hh], qreg[s], sc, 0, 0, 0);
      // register r = 4g + j holds key kb + 8g + 4hh + j of this lane's query
      float bm = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = j0 + kb + 8 * (r >> 2) + 4 * hh + (r & 3);
        if (key >= N) sc[r] = -INFINITY;
        bm = fmaxf(bm, sc[r]);
      }
      bm = fmaxf(bm, __shfl_xor(bm, 32));
      const float mn = fmaxf(m, bm);
      const float f = expf(m - mn);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = expf(sc[r] - mn); ps += sc[r]; }
      ps += __shfl_xor(ps, 32);
      l = l * f + ps;
      m = mn;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= f; o1[r] *= f; }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int key = kb + 8 * (s >> 2) + 4 * hh + (s & 3);
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[key][n], sc[s], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[key][32 + n], sc[s], o1, 0, 0, 0);
      }
    }
  }
  if (!active) return;
  const float inv = 1.0f / l;
  T* op = out + ((size_t)t.x + qi) * o_cs + o_coff + head * HD;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    st4<T>(op + 8 * g + 4 * hh, make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
    st4<T>(op + 32 + 8 * g + 4 * hh, make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
  }
}

// ---- LetterBox + normalise ------------------------------------------------------------------------------------------
// cv2.resize INTER_LINEAR for uint8 restated in fixed point (11-bit coefficients, two-pass rounding), pad value 114,
// /255, optional channel flip, NHWC with CPAD channels (>= 3, rest zero).
template <typename T, int CPAD>
__global__ void letterbox_kernel(const uint8_t* __restrict__ frame, int W, int flip, const LetterboxImg* __restrict__ imgs,
                                 const int4* __restrict__ tab, int n_img, T* __restrict__ out, long long total_px) {
  const long long gp = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gp >= total_px) return;
  const int im = find_img(tab, n_img, gp);
  const int4 t = tab[im];
  const LetterboxImg L = imgs[im];
  const int lp = (int)(gp - t.x), y = lp / t.z, x = lp - y * t.z;
  int px[3];
  letterbox_sample(frame, W, L, y, x, px);
  T* op = out + (size_t)gp * CPAD;
  const float v0 = (float)px[flip ? 2 : 0] / 255.0f, v1 = (float)px[1] / 255.0f, v2 = (float)px[flip ? 0 : 2] / 255.0f;
  stf(op + 0, v0); stf(op + 1, v1); stf(op + 2, v2);
#pragma unroll
  for (int c = 3; c < CPAD; ++c) stf(op + c, 0.f);
}

// ---- stem fused with the letterbox: the image-input 3x3 conv reads the u8 frame through letterbox_sample() instead of
// a letterboxed fp32 copy (61 x 512^2 x 16 B = 256 MB written and read back per 4K frame). Same arithmetic, same order
// as letterbox_kernel + conv3x3_c3_direct_kernel: results are bit-identical to the two-kernel path.
template <typename T, int COUT>
__global__ void __launch_bounds__(256) stem_from_frame_kernel(const uint8_t* __restrict__ frame, int W, int flip, const LetterboxImg* __restrict__ imgs,
                                                              T* __restrict__ out, int out_cs, int out_coff, const float* __restrict__ w,
                                                              const float* __restrict__ bias, int stride, int act, const int4* __restrict__ in_tab,
                                                              const int4* __restrict__ out_tab, int n_img, long long total_out_px) {
  // The 27 x COUT weights are read at compile-time offsets from the (wave-uniform) kernel argument: hipcc turns that into scalar
  // loads and the multiply-adds take the weight as their SGPR operand — no LDS copy of the weights, no LDS read per four FMAs
  // (the LDS-broadcast form spent half its issue slots on ds_read_b128).
  // q / 255 for the 256 possible sample values, rounded exactly as the division the two-kernel path performs (through T)
  __shared__ float norm[256];
  norm[threadIdx.x] = (float)(T)((float)threadIdx.x / 255.0f);
  __syncthreads();
  const long long gp = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gp >= total_out_px) return;
  const int im = find_img(out_tab, n_img, gp);
  const int4 to = out_tab[im], ti = in_tab[im];
  const LetterboxImg L = imgs[im];
  const int lp = (int)(gp - to.x), oy = lp / to.z, ox = lp - oy * to.z;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {                  // one tap = 3 x COUT weights live in scalar registers at a time
    const int ky = tap / 3, kx = tap - ky * 3;
    const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
    float px[3] = {0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)ti.y && (unsigned)ix < (unsigned)ti.z) {
      int q[3];
      letterbox_sample(frame, W, L, iy, ix, q);
      px[0] = norm[q[flip ? 2 : 0] & 255]; px[1] = norm[q[1] & 255]; px[2] = norm[q[flip ? 0 : 2] & 255];
    }
    const float* wt = w + tap * 3 * COUT;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc[c] = fmaf(px[ci], wt[ci * COUT + c], acc[c]);
  }
  T* op = out + (size_t)gp * out_cs + out_coff;
#pragma unroll
  for (int c = 0; c < COUT; c += 4) {
    const float4 b = *reinterpret_cast<const float4*>(bias + c);
    st4<T>(op + c, make_float4(act_fn(acc[c] + b.x, act), act_fn(acc[c + 1] + b.y, act), act_fn(acc[c + 2] + b.z, act), act_fn(acc[c + 3] + b.w, act)));
  }
}

// Depthwise 3x3 as a walk down the image: a thread owns 4 columns x 4 channels of an RB-row x 16-column tile and slides a three-row
// window over it — six 16-byte loads per output row of four pixels (1.5 loads per output) instead of the strip kernel's eighteen per
// four outputs; everything else (weights, bias, activation, residual, max-|value|) as in dwconv_strip_body.
template <typename T, int RB>
__global__ void __launch_bounds__(256) dwconv3x3_col_kernel(const T* __restrict__ in, int in_cs, int in_coff, int grp, int grp_stride, int grp_off,
                                                            T* __restrict__ out, int out_cs, int out_coff, const float* __restrict__ w /*[9][C]*/,
                                                            const float* __restrict__ bias, const T* __restrict__ res, int r_cs, int r_coff, int C, int act,
                                                            const int4* __restrict__ tab, const int4* __restrict__ tiles, int n_tiles, unsigned* __restrict__ amax,
                                                            int amax_img /* 1: a slot per image (TView::amax_n > 1) */) {
  const int C4 = C >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  float mx = 0.f;
  int img = 0;
  if (idx < (long long)n_tiles * 4 * C4) {
    const int c = (int)(idx % C4) * 4;
    const int strip = (int)((idx / C4) & 3);
    const int4 tl = tiles[idx / (4 * C4)];
    img = amax_img ? tl.x : 0;
    const int4 t = tab[tl.x];
    const int Hh = t.y, Ww = t.z, x0 = tl.z + 4 * strip, y0 = tl.y;
    if (x0 < Ww) {
      const int cin = in_coff + (c / grp) * grp_stride + grp_off + (c % grp);
      float4 k[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) k[i] = *reinterpret_cast<const float4*>(w + i * C + c);
      const float4 b = *reinterpret_cast<const float4*>(bias + c);
      const T* ibase = in + (size_t)t.x * in_cs + cin;
      auto load_row = [&](int yy, float4 (&r)[6]) {
        const bool rok = (unsigned)yy < (unsigned)Hh;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int xx = x0 + j - 1;
          r[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (rok && (unsigned)xx < (unsigned)Ww) r[j] = ld4<T>(ibase + ((size_t)yy * Ww + xx) * in_cs);
        }
      };
      float4 v[3][6];
      load_row(y0 - 1, v[0]);
      load_row(y0, v[1]);
      const int rows = min(RB, Hh - y0);
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        if (r < rows) {
          load_row(y0 + r + 1, v[(r + 2) % 3]);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (x0 + i < Ww) {
              float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
              for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                  const float4 a = v[(r + ky) % 3][i + kx], q = k[ky * 3 + kx];
                  acc.x = fmaf(a.x, q.x, acc.x); acc.y = fmaf(a.y, q.y, acc.y); acc.z = fmaf(a.z, q.z, acc.z); acc.w = fmaf(a.w, q.w, acc.w);
                }
              const long long gp = (long long)t.x + (long long)(y0 + r) * Ww + x0 + i;
              float4 o = make_float4(act_fn(acc.x + b.x, act), act_fn(acc.y + b.y, act), act_fn(acc.z + b.z, act), act_fn(acc.w + b.w, act));
              if (res) {
                const float4 rr = ld4<T>(res + (size_t)gp * r_cs + r_coff + c);
                o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
              }
              mx = fmaxf(mx, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
              st4<T>(out + (size_t)gp * out_cs + out_coff + c, o);
            }
          }
        }
      }
    }
  }
  if (amax) {
    const int i0 = __builtin_amdgcn_readfirstlane(img);
    if (__all(img == i0)) {                              // the wave's threads share an image (the common case): one atomic per wave
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      const unsigned bb = __float_as_uint(mx);
      if ((threadIdx.x & 63) == 0 && bb > __hip_atomic_load(amax + i0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax + i0, bb);
    } else {
      const unsigned bb = __float_as_uint(mx);
      if (bb > __hip_atomic_load(amax + img, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(amax + img, bb);
    }
  }
}

inline unsigned blocks_for(long long n, int bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

bool conv_direct_eligible(const ConvOp& op) {
  const PackedConv& pc = *op.pc;
  return pc.w_direct.p != nullptr && pc.k == 3 && !op.up && !op.has_res1 && !op.has_res2 && op.out.dt == pc.dt && op.in.coff == 0 &&
         (pc.cout == 16 || pc.cout == 32 || pc.cout == 64) && op.out.cs % 4 == 0 && op.out.coff % 4 == 0 && op.in.cs == (pc.dt == F16 ? 8 : 4);
}

template <typename T, int COUT>
static void launch_direct_t(const ConvOp& op, hipStream_t st) {
  const PackedConv& pc = *op.pc;
  const long long total = op.out.lvl->total_px;
  hipLaunchKernelGGL((conv3x3_c3_direct_kernel<T, COUT>), dim3(blocks_for(total, 256)), dim3(256), 0, st, (const T*)op.in.ptr, op.in.cs,
                     (T*)op.out.ptr, op.out.cs, op.out.coff, pc.w_direct.as<float>(), pc.bias.as<float>(), op.stride, op.act,
                     op.in.lvl->d_tab.as<int4>(), op.out.lvl->d_tab.as<int4>(), op.out.lvl->n, total);
}

void launch_conv_direct(const ConvOp& op, hipStream_t st) {
  const PackedConv& pc = *op.pc;
  FFP_CHECK(conv_direct_eligible(op), FFP_ERR_ARG, "conv %s: not eligible for the direct kernel", pc.name.c_str());
  if (pc.dt == F32) {
    if (pc.cout == 16) launch_direct_t<float, 16>(op, st); else if (pc.cout == 32) launch_direct_t<float, 32>(op, st); else launch_direct_t<float, 64>(op, st);
  } else {
    if (pc.cout == 16) launch_direct_t<_Float16, 16>(op, st); else if (pc.cout == 32) launch_direct_t<_Float16, 32>(op, st); else launch_direct_t<_Float16, 64>(op, st);
  }
  FFP_HIP(hipGetLastError());
}

template <typename T, int COUT>
static void launch_stem_t(const uint8_t* d_frame, int W, int flip, const DevBuf& d_imgs, const ConvOp& op, hipStream_t st) {
  const PackedConv& pc = *op.pc;
  const long long total = op.out.lvl->total_px;
  hipLaunchKernelGGL((stem_from_frame_kernel<T, COUT>), dim3(blocks_for(total, 256)), dim3(256), 0, st, d_frame, W, flip, d_imgs.as<LetterboxImg>(),
                     (T*)op.out.ptr, op.out.cs, op.out.coff, pc.w_direct.as<float>(), pc.bias.as<float>(), op.stride, op.act,
                     op.in.lvl->d_tab.as<int4>(), op.out.lvl->d_tab.as<int4>(), op.out.lvl->n, total);
}

void launch_stem_from_frame(const uint8_t* d_frame, int H, int W, int flip, const DevBuf& d_imgs, const ConvOp& op, hipStream_t st) {
  const PackedConv& pc = *op.pc;
  (void)H;
  FFP_CHECK(conv_direct_eligible(op), FFP_ERR_ARG, "conv %s: not eligible for the fused stem", pc.name.c_str());
  if (pc.dt == F32) {
    if (pc.cout == 16) launch_stem_t<float, 16>(d_frame, W, flip, d_imgs, op, st); else if (pc.cout == 32) launch_stem_t<float, 32>(d_frame, W, flip, d_imgs, op, st); else launch_stem_t<float, 64>(d_frame, W, flip, d_imgs, op, st);
  } else {
    if (pc.cout == 16) launch_stem_t<_Float16, 16>(d_frame, W, flip, d_imgs, op, st); else if (pc.cout == 32) launch_stem_t<_Float16, 32>(d_frame, W, flip, d_imgs, op, st); else launch_stem_t<_Float16, 64>(d_frame, W, flip, d_imgs, op, st);
  }
  FFP_HIP(hipGetLastError());
}

void launch_dwconv(const DwConvOp& op, hipStream_t st) {
  const PackedConv& pc = *op.pc;
  FFP_CHECK(pc.depthwise() && pc.k == 3, FFP_ERR_ARG, "dwconv %s: expects depthwise 3x3", pc.name.c_str());
  const int C = pc.cout;
  FFP_CHECK(op.out.C == C && op.in.lvl == op.out.lvl, FFP_ERR_ARG, "dwconv %s: view mismatch", pc.name.c_str());
  const int grp = op.grp > 0 ? op.grp : C;
  const int gstride = op.grp > 0 ? op.grp_stride : C;
  FFP_CHECK(C % 4 == 0 && grp % 4 == 0 && op.in.cs % 4 == 0 && (op.in.coff + op.grp_off) % 4 == 0 && gstride % 4 == 0 && op.out.cs % 4 == 0 &&
                op.out.coff % 4 == 0 && (!op.has_res || (op.res.cs % 4 == 0 && op.res.coff % 4 == 0)),
            FFP_ERR_ARG, "dwconv %s: channel counts/offsets must be multiples of 4", pc.name.c_str());
  const int4* tab = op.out.lvl->d_tab.as<int4>();
  static const bool col_walk = [] { const char* e = getenv("FFP_DW_STRIP"); return !(e && e[0] == '1'); }();      // FFP_DW_STRIP=1: the strip kernel (A/B aid)
  const int amax_img = op.out.amax && op.out.amax_n > 1 ? 1 : 0;
  if ((col_walk || amax_img) && !op.out.lvl->capacity()) {
    constexpr int RB = 8;
    int n_tiles = 0;
    const int* d_count = nullptr;
    const int4* tiles = op.out.lvl->tile_table(RB, &n_tiles, &d_count, st);
    const unsigned nbc = blocks_for((long long)n_tiles * 4 * (C / 4), 256);
    if (nbc == 0) return;
    if (op.in.dt == F32)
      hipLaunchKernelGGL((dwconv3x3_col_kernel<float, RB>), dim3(nbc), dim3(256), 0, st, (const float*)op.in.ptr, op.in.cs, op.in.coff, grp, gstride, op.grp_off,
                         (float*)op.out.ptr, op.out.cs, op.out.coff, pc.w.as<float>(), pc.bias.as<float>(), op.has_res ? (const float*)op.res.ptr : nullptr,
                         op.res.cs, op.res.coff, C, op.act, tab, tiles, n_tiles, op.out.amax, amax_img);
    else
      hipLaunchKernelGGL((dwconv3x3_col_kernel<_Float16, RB>), dim3(nbc), dim3(256), 0, st, (const _Float16*)op.in.ptr, op.in.cs, op.in.coff, grp, gstride,
                         op.grp_off, (_Float16*)op.out.ptr, op.out.cs, op.out.coff, pc.w.as<float>(), pc.bias.as<float>(),
                         op.has_res ? (const _Float16*)op.res.ptr : nullptr, op.res.cs, op.res.coff, C, op.act, tab, tiles, n_tiles, op.out.amax, amax_img);
    FFP_HIP(hipGetLastError());
    return;
  }
  FFP_CHECK(!amax_img, FFP_ERR_STATE, "dwconv: per-image exponent slots need the tile-walking kernel (exact-mode level)");
  const unsigned nb = blocks_for(((op.out.lvl->total_px + 3) / 4) * (C / 4), 256);
  if (op.in.dt == F32)
    hipLaunchKernelGGL(dwconv3x3_strip_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)op.in.ptr, op.in.cs, op.in.coff, grp,
                       gstride, op.grp_off, (float*)op.out.ptr, op.out.cs, op.out.coff, pc.w.as<float>(), pc.bias.as<float>(),
                       op.has_res ? (const float*)op.res.ptr : nullptr, op.res.cs, op.res.coff, C, op.act, tab, op.out.lvl->n,
                       op.out.lvl->total_px, op.out.amax);
  else
    hipLaunchKernelGGL(dwconv3x3_strip_kernel<_Float16>, dim3(nb), dim3(256), 0, st, (const _Float16*)op.in.ptr, op.in.cs, op.in.coff,
                       grp, gstride, op.grp_off, (_Float16*)op.out.ptr, op.out.cs, op.out.coff, pc.w.as<float>(),
                       pc.bias.as<float>(), op.has_res ? (const _Float16*)op.res.ptr : nullptr, op.res.cs, op.res.coff, C, op.act,
                       tab, op.out.lvl->n, op.out.lvl->total_px, op.out.amax);
  FFP_HIP(hipGetLastError());
}

namespace {
__global__ void amax_init_kernel(unsigned* __restrict__ slots, const unsigned* __restrict__ init, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) slots[i] = init[i];
}
__global__ void amax_max_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = max(dst[i], src[i]);
}
}  // namespace

void launch_amax_init(unsigned* slots, const unsigned* init, int n, hipStream_t st) {
  hipLaunchKernelGGL(amax_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, slots, init, n);
  FFP_HIP(hipGetLastError());
}

void launch_amax_max(unsigned* dst, const unsigned* src, hipStream_t st, int n) {
  hipLaunchKernelGGL(amax_max_kernel, dim3((n + 63) / 64), dim3(64), 0, st, dst, src, n);
  FFP_HIP(hipGetLastError());
}

void launch_sppf_pool(const TView& in, const TView& y1, const TView& y2, const TView& y3, hipStream_t st) {
  FFP_CHECK(y1.ptr == y2.ptr && y2.ptr == y3.ptr && y1.cs == y2.cs && y2.cs == y3.cs, FFP_ERR_ARG, "sppf: outputs must be slices of one buffer");
  const int C = in.C;
  FFP_CHECK(C % 4 == 0 && in.cs % 4 == 0 && in.coff % 4 == 0 && y1.cs % 4 == 0 && y1.coff % 4 == 0 && y2.coff % 4 == 0 && y3.coff % 4 == 0,
            FFP_ERR_ARG, "sppf: channel counts/offsets must be multiples of 4");
  const long long total = in.lvl->total_px * (C / 4);
  const unsigned nb = blocks_for(total, 256);
  const int4* tab = in.lvl->d_tab.as<int4>();
  int max_px = 0;
  for (int i = 0; i < in.lvl->n; ++i) max_px = std::max(max_px, in.lvl->h[i] * in.lvl->w[i]);
  if (max_px <= 1024) {
    const dim3 grid(C / 4, in.lvl->n);
    if (in.dt == F32)
      hipLaunchKernelGGL(sppf_pool_lds_kernel<float>, grid, dim3(256), 0, st, (const float*)in.ptr, in.cs, in.coff, (float*)y1.ptr, (float*)y2.ptr,
                         (float*)y3.ptr, y1.cs, y1.coff, y2.coff, y3.coff, tab);
    else
      hipLaunchKernelGGL(sppf_pool_lds_kernel<_Float16>, grid, dim3(256), 0, st, (const _Float16*)in.ptr, in.cs, in.coff, (_Float16*)y1.ptr,
                         (_Float16*)y2.ptr, (_Float16*)y3.ptr, y1.cs, y1.coff, y2.coff, y3.coff, tab);
    FFP_HIP(hipGetLastError());
    return;
  }
  if (in.dt == F32)
    hipLaunchKernelGGL(sppf_pool_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)in.ptr, in.cs, in.coff, (float*)y1.ptr,
                       (float*)y2.ptr, (float*)y3.ptr, y1.cs, y1.coff, y2.coff, y3.coff, C, tab, in.lvl->n, in.lvl->total_px);
  else
    hipLaunchKernelGGL(sppf_pool_kernel<_Float16>, dim3(nb), dim3(256), 0, st, (const _Float16*)in.ptr, in.cs, in.coff,
                       (_Float16*)y1.ptr, (_Float16*)y2.ptr, (_Float16*)y3.ptr, y1.cs, y1.coff, y2.coff, y3.coff, C, tab, in.lvl->n,
                       in.lvl->total_px);
  FFP_HIP(hipGetLastError());
}

void launch_upsample2x(const TView& in, const TView& out, hipStream_t st) {
  FFP_CHECK(in.C == out.C && in.dt == out.dt && in.lvl->n == out.lvl->n, FFP_ERR_ARG, "upsample: view mismatch");
  const int es = dsize(in.dt);
  FFP_CHECK((in.C * es) % 16 == 0 && (in.cs * es) % 16 == 0 && (in.coff * es) % 16 == 0 && (out.cs * es) % 16 == 0 && (out.coff * es) % 16 == 0,
            FFP_ERR_ARG, "upsample: views must be 16-byte aligned");
  const int nvec = in.C * es / 16;
  const long long total = out.lvl->total_px * nvec;
  const unsigned nb = blocks_for(total, 256);
  hipLaunchKernelGGL(upsample2x_kernel, dim3(nb), dim3(256), 0, st, (const unsigned char*)in.ptr, in.cs * es, in.coff * es,
                     in.lvl->d_tab.as<int4>(), (unsigned char*)out.ptr, out.cs * es, out.coff * es, out.lvl->d_tab.as<int4>(), out.lvl->n, nvec,
                     out.lvl->total_px);
  FFP_HIP(hipGetLastError());
}

void launch_psa_attention(const TView& qkv, const TView& out, int nh, int kd, int hd, hipStream_t st) {
  FFP_CHECK(kd == 32 && hd == 64, FFP_ERR_ARG, "psa attention: only key_dim 32 / head_dim 64 (YOLO11 n/s) is instantiated");
  FFP_CHECK(qkv.C == nh * (2 * kd + hd) && out.C == nh * hd && qkv.lvl == out.lvl, FFP_ERR_ARG, "psa attention: view mismatch");
  int nmax = 0;
  for (int i = 0; i < qkv.lvl->n; ++i) nmax = std::max(nmax, qkv.lvl->h[i] * qkv.lvl->w[i]);
  const float scale = 1.0f / sqrtf((float)kd);
  static const bool valu = [] { const char* e = getenv("FFP_ATTN_VALU"); return e && e[0] == '1'; }();
  if (!valu && qkv.cs % 4 == 0 && qkv.coff % 4 == 0 && out.cs % 4 == 0 && out.coff % 4 == 0) {
    dim3 g2((nmax + 127) / 128, nh, qkv.lvl->n);
    if (qkv.dt == F32)
      hipLaunchKernelGGL((psa_attention_mfma_kernel<float>), g2, dim3(256), 0, st, (const float*)qkv.ptr, qkv.cs, qkv.coff, (float*)out.ptr, out.cs,
                         out.coff, qkv.lvl->d_tab.as<int4>(), scale);
    else
      hipLaunchKernelGGL((psa_attention_mfma_kernel<_Float16>), g2, dim3(256), 0, st, (const _Float16*)qkv.ptr, qkv.cs, qkv.coff, (_Float16*)out.ptr,
                         out.cs, out.coff, qkv.lvl->d_tab.as<int4>(), scale);
    FFP_HIP(hipGetLastError());
    return;
  }
  dim3 grid((nmax + 255) / 256, nh, qkv.lvl->n);
  if (qkv.dt == F32)
    hipLaunchKernelGGL((psa_attention_kernel<float, 32, 64>), grid, dim3(256), 0, st, (const float*)qkv.ptr, qkv.cs, qkv.coff,
                       (float*)out.ptr, out.cs, out.coff, qkv.lvl->d_tab.as<int4>(), scale);
  else
    hipLaunchKernelGGL((psa_attention_kernel<_Float16, 32, 64>), grid, dim3(256), 0, st, (const _Float16*)qkv.ptr, qkv.cs,
                       qkv.coff, (_Float16*)out.ptr, out.cs, out.coff, qkv.lvl->d_tab.as<int4>(), scale);
  FFP_HIP(hipGetLastError());
}

void launch_letterbox(const uint8_t* d_frame, int H, int W, int flip, const DevBuf& d_imgs, const TView& out, hipStream_t st) {
  (void)H;
  const long long total = out.lvl->total_px;
  const unsigned nb = blocks_for(total, 256);
  FFP_CHECK(out.coff == 0 && out.cs == out.C, FFP_ERR_ARG, "letterbox: output must be a whole buffer");
  if (out.dt == F32) {
    FFP_CHECK(out.C == 4, FFP_ERR_ARG, "letterbox: fp32 output has 4 channels");
    hipLaunchKernelGGL((letterbox_kernel<float, 4>), dim3(nb), dim3(256), 0, st, d_frame, W, flip, d_imgs.as<LetterboxImg>(),
                       out.lvl->d_tab.as<int4>(), out.lvl->n, (float*)out.ptr, total);
  } else {
    FFP_CHECK(out.C == 8, FFP_ERR_ARG, "letterbox: fp16 output has 8 channels");
    hipLaunchKernelGGL((letterbox_kernel<_Float16, 8>), dim3(nb), dim3(256), 0, st, d_frame, W, flip, d_imgs.as<LetterboxImg>(),
                       out.lvl->d_tab.as<int4>(), out.lvl->n, (_Float16*)out.ptr, total);
  }
  FFP_HIP(hipGetLastError());
}

}  // namespace ffp
