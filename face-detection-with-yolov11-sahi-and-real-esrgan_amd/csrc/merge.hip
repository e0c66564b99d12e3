// merge.hip — SAHI post-processing (NMSPostprocess / GreedyNMMPostprocess) on the device.
//
// Reference: constructed at docs sahi/predict.py:254-259, applied at :317-320; arithmetic from sahi 0.11.34
// `postprocess/combine.py` + `postprocess/utils.py` (not vendored).
// Greedy matching is order dependent, so the parallel formulation keeps the exact greedy order:
//   1. one workgroup sorts the boxes by (category when batched NMM, score desc, original index asc)      [bitonic]
//   2. a grid computes the upper-triangular match matrix M[i][j] = !(metric(i,j) < thr) as 64-bit words  [fp32, torch order]
//   3. ONE wave sweeps the sorted list: a live box i becomes a keeper, absorbs  M[i] & alive  and clears it; for NMM the
//      absorbed boxes are folded in ascending order with sahi's float64 has_match() against the GROWING union box.
//   4. a grid writes the surviving rows.
#include "det_post.hpp"

namespace ffp {

namespace {

constexpr int SORT_THREADS = 1024;
constexpr int SORT_LDS = 4096;

__global__ void compact_prefix_kernel(const int* __restrict__ counts, int n_slices, int max_det, int* __restrict__ prefix,
                                      int* __restrict__ d_n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int t = 0;
    for (int i = 0; i < n_slices; ++i) { prefix[i] = t; t += min(counts[i], max_det); }
    prefix[n_slices] = t;
    *d_n = t;
  }
}

__global__ void compact_rows_kernel(const float* __restrict__ dets, const int* __restrict__ counts, const int* __restrict__ prefix,
                                    int n_slices, int max_det, int stride, float* __restrict__ rows) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_slices * max_det) return;
  const int s = g / max_det, k = g - s * max_det;
  if (k >= min(counts[s], max_det)) return;
  const float* src = dets + (size_t)g * stride;
  float* dst = rows + (size_t)(prefix[s] + k) * stride;
  for (int c = 0; c < stride; ++c) dst[c] = src[c];
}

// ---- 1. sort ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(SORT_THREADS) merge_sort_kernel(const float* __restrict__ rows, const int* __restrict__ d_n,
                                                                   int stride, int cat_major, unsigned long long* __restrict__ gkeys,
                                                                   int* __restrict__ order, float4* __restrict__ sbox,
                                                                   float* __restrict__ sscore, int* __restrict__ scat) {
  __shared__ unsigned long long lkeys[SORT_LDS];
  const int n = *d_n;
  if (n <= 0) return;
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  unsigned long long* keys = np2 <= SORT_LDS ? lkeys : gkeys;
  for (int i = threadIdx.x; i < np2; i += SORT_THREADS) {
    unsigned long long k = ~0ull;
    if (i < n) {
      const float s = rows[(size_t)i * stride + 4];
      const unsigned sb = ~__float_as_uint(s);                    // descending score (scores are positive floats)
      const unsigned cat = cat_major ? (unsigned)min(max((int)rows[(size_t)i * stride + 5], 0), 255) : 0u;
      k = ((unsigned long long)cat << 56) | ((unsigned long long)sb << 24) | (unsigned long long)(i & 0xFFFFFF);
    }
    keys[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < np2; i += SORT_THREADS) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = keys[i], b = keys[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[l] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int p = threadIdx.x; p < n; p += SORT_THREADS) {
    const int idx = (int)(keys[p] & 0xFFFFFFull);
    order[p] = idx;
    const float* r = rows + (size_t)idx * stride;
    sbox[p] = make_float4(r[0], r[1], r[2], r[3]);
    sscore[p] = r[4];
    scat[p] = (int)r[5];
  }
}

// ---- 2. match matrix ----------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) merge_mask_kernel(const float4* __restrict__ sbox, const int* __restrict__ scat,
                                                        const int* __restrict__ d_n, int nws, int metric, float thr,
                                                        int class_agnostic, unsigned long long* __restrict__ mask) {
  const int n = *d_n;
  const int rt = blockIdx.y, cw = blockIdx.x;
  if (cw < rt || rt * 64 >= n || cw * 64 >= n) return;
  __shared__ float4 cb[64];
  __shared__ int cc[64];
  const int t = threadIdx.x;
  const int j0 = cw * 64;
  if (j0 + t < n) { cb[t] = sbox[j0 + t]; cc[t] = scat[j0 + t]; }
  __syncthreads();
  const int i = rt * 64 + t;
  if (i >= n) return;
  const float4 bi = sbox[i];
  const int ci = scat[i];
  const float area_i = (bi.z - bi.x) * (bi.w - bi.y);
  unsigned long long word = 0ull;
  const int jn = min(64, n - j0);
  for (int q = 0; q < jn; ++q) {
    const int j = j0 + q;
    if (j <= i) continue;
    if (!class_agnostic && cc[q] != ci) continue;
    const float4 bj = cb[q];
    const float xx1 = fmaxf(bj.x, bi.x), yy1 = fmaxf(bj.y, bi.y), xx2 = fminf(bj.z, bi.z), yy2 = fminf(bj.w, bi.w);
    const float w = fmaxf(xx2 - xx1, 0.f), h = fmaxf(yy2 - yy1, 0.f);
    const float inter = w * h;
    const float area_j = (bj.z - bj.x) * (bj.w - bj.y);
    float v;
    if (metric == FFP_METRIC_IOU) v = inter / ((area_j - inter) + area_i);
    else v = inter / fminf(area_j, area_i);
    if (!(v < thr)) word |= 1ull << q;
  }
  mask[(size_t)i * nws + cw] = word;
}

// ---- 3. sweep -----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool has_match_f64(const double* c, const float4& o, int metric, double thr) {
  const double a1 = (c[2] - c[0]) * (c[3] - c[1]);
  const double a2 = ((double)o.z - (double)o.x) * ((double)o.w - (double)o.y);
  const double w = fmin(c[2], (double)o.z) - fmax(c[0], (double)o.x);
  const double h = fmin(c[3], (double)o.w) - fmax(c[1], (double)o.y);
  const double it = fmax(w, 0.0) * fmax(h, 0.0);
  const double v = metric == FFP_METRIC_IOU ? it / (a1 + a2 - it) : it / fmin(a1, a2);
  return v > thr;
}

__global__ void __launch_bounds__(64) merge_sweep_kernel(const float4* __restrict__ sbox, const float* __restrict__ sscore,
                                                         const int* __restrict__ d_n, int nws, int type, int metric, double thr,
                                                         const unsigned long long* __restrict__ mask, float4* __restrict__ kbox,
                                                         float* __restrict__ kscore, int* __restrict__ ksrc,
                                                         int* __restrict__ d_k) {
  extern __shared__ unsigned long long alive[];
  const int n = *d_n;
  const int lane = threadIdx.x;
  const int nw = (n + 63) >> 6;
  for (int w = lane; w < nw; w += 64) {
    const int rem = n - w * 64;
    alive[w] = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
  }
  __syncthreads();
  int K = 0;
  for (int i = 0; i < n; ++i) {
    const unsigned long long aw = alive[i >> 6];
    if (!((aw >> (i & 63)) & 1ull)) continue;
    double cur[4];
    const float4 bi = sbox[i];
    cur[0] = bi.x; cur[1] = bi.y; cur[2] = bi.z; cur[3] = bi.w;
    float cscore = sscore[i];
    int csrc = i;
    const int w0 = i >> 6;
    for (int base = w0; base < nw; base += 64) {
      const int w = base + lane;
      unsigned long long m = 0ull;
      if (w < nw) {
        const unsigned long long row = (w >= w0) ? mask[(size_t)i * nws + w] : 0ull;
        const unsigned long long al = alive[w];
        m = row & al;
        unsigned long long nal = al & ~row;
        if (w == w0) nal &= ~(1ull << (i & 63));
        alive[w] = nal;
      }
      if (type == FFP_PP_GREEDYNMM) {
        unsigned long long lanes = __ballot(m != 0ull);
        while (lanes) {
          const int l = __ffsll((long long)lanes) - 1;
          lanes &= lanes - 1ull;
          unsigned long long mw = __shfl(m, l);
          while (mw) {
            const int b = __ffsll((long long)mw) - 1;
            mw &= mw - 1ull;
            const int j = (base + l) * 64 + b;
            const float4 bj = sbox[j];
            if (has_match_f64(cur, bj, metric, thr)) {
              cur[0] = fmin(cur[0], (double)bj.x); cur[1] = fmin(cur[1], (double)bj.y);
              cur[2] = fmax(cur[2], (double)bj.z); cur[3] = fmax(cur[3], (double)bj.w);
              const float sj = sscore[j];
              if (!(cscore > sj)) csrc = j;
              cscore = fmaxf(cscore, sj);
            }
          }
        }
      }
    }
    __syncthreads();
    if (lane == 0) {
      kbox[K] = make_float4((float)cur[0], (float)cur[1], (float)cur[2], (float)cur[3]);
      kscore[K] = cscore;
      ksrc[K] = csrc;
    }
    ++K;
  }
  if (lane == 0) *d_k = K;
}

// ---- 4. emit ---------------------------------------------------------------------------------------------------------------
__global__ void merge_emit_kernel(const float* __restrict__ rows, int stride, const int* __restrict__ order,
                                  const float4* __restrict__ kbox, const float* __restrict__ kscore, const int* __restrict__ ksrc,
                                  const int* __restrict__ d_k, int cap, float* __restrict__ out, int* __restrict__ out_src,
                                  int* __restrict__ out_n) {
  // *out_n is the UNTRUNCATED count: a caller whose buffer is smaller sees n > cap and knows rows were dropped
  const int K = min(*d_k, cap);
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0) *out_n = *d_k;
  if (k >= K) return;
  const int src = order[ksrc[k]];
  const float* r = rows + (size_t)src * stride;
  float* o = out + (size_t)k * stride;
  const float4 b = kbox[k];
  o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = kscore[k];
  for (int c = 5; c < stride; ++c) o[c] = r[c];
  if (out_src) out_src[k] = src;
}

__global__ void merge_passthrough_kernel(const float* __restrict__ rows, int stride, const int* __restrict__ d_n, int cap,
                                         float* __restrict__ out, int* __restrict__ out_src, int* __restrict__ out_n) {
  // fewer than two boxes: SAHI skips the post-process (docs sahi/predict.py:317)
  const int n = min(*d_n, cap);
  if (threadIdx.x == 0) *out_n = *d_n;
  for (int k = 0; k < n; ++k) {
    for (int c = threadIdx.x; c < stride; c += blockDim.x) out[(size_t)k * stride + c] = rows[(size_t)k * stride + c];
    if (threadIdx.x == 0 && out_src) out_src[k] = k;
  }
}

}  // namespace

void launch_compact_rows(const float* d_dets, const int* d_counts, int n_slices, int max_det, int stride, float* d_rows,
                         int* d_n, int* d_prefix, hipStream_t st) {
  hipLaunchKernelGGL(compact_prefix_kernel, dim3(1), dim3(64), 0, st, d_counts, n_slices, max_det, d_prefix, d_n);
  const int total = n_slices * max_det;
  hipLaunchKernelGGL(compact_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, st, d_dets, d_counts, d_prefix, n_slices,
                     max_det, stride, d_rows);
  FFP_HIP(hipGetLastError());
}

void run_merge(MergeWork& w, const float* d_rows, const int* d_n, int n_max, int stride, int type, int metric, double thr,
               int class_agnostic, float* d_out, int* d_out_src, int* d_out_n, int cap, hipStream_t st) {
  FFP_CHECK(type == FFP_PP_NMS || type == FFP_PP_GREEDYNMM, FFP_ERR_ARG, "merge: postprocess type %d not implemented", type);
  FFP_CHECK(metric == FFP_METRIC_IOU || metric == FFP_METRIC_IOS, FFP_ERR_ARG, "merge: metric %d", metric);
  FFP_CHECK(n_max < (1 << 24), FFP_ERR_ARG, "merge: %d boxes exceed 2^24", n_max);
  if (n_max <= 0) { FFP_HIP(hipMemsetAsync(d_out_n, 0, sizeof(int), st)); return; }
  int np2 = 1;
  while (np2 < n_max) np2 <<= 1;
  const int nws = (n_max + 63) / 64;
  w.keys.ensure(sizeof(unsigned long long) * (size_t)np2);
  w.order.ensure(sizeof(int) * (size_t)n_max);
  w.sbox.ensure(sizeof(float4) * (size_t)n_max);
  w.sscore.ensure(sizeof(float) * (size_t)n_max);
  w.scat.ensure(sizeof(int) * (size_t)n_max);
  w.mask.ensure(sizeof(unsigned long long) * (size_t)n_max * nws);
  w.kbox.ensure(sizeof(float4) * (size_t)n_max);
  w.kscore.ensure(sizeof(float) * (size_t)n_max);
  w.ksrc.ensure(sizeof(int) * (size_t)n_max);
  w.dk.ensure(sizeof(int));
  const int cat_major = (type == FFP_PP_GREEDYNMM && !class_agnostic) ? 1 : 0;
  hipLaunchKernelGGL(merge_sort_kernel, dim3(1), dim3(SORT_THREADS), 0, st, d_rows, d_n, stride, cat_major,
                     w.keys.as<unsigned long long>(), w.order.as<int>(), w.sbox.as<float4>(), w.sscore.as<float>(), w.scat.as<int>());
  hipLaunchKernelGGL(merge_mask_kernel, dim3(nws, nws), dim3(64), 0, st, w.sbox.as<float4>(), w.scat.as<int>(), d_n, nws, metric,
                     (float)thr, class_agnostic, w.mask.as<unsigned long long>());
  hipLaunchKernelGGL(merge_sweep_kernel, dim3(1), dim3(64), sizeof(unsigned long long) * (size_t)nws, st, w.sbox.as<float4>(),
                     w.sscore.as<float>(), d_n, nws, type, metric, thr, w.mask.as<unsigned long long>(), w.kbox.as<float4>(),
                     w.kscore.as<float>(), w.ksrc.as<int>(), w.dk.as<int>());
  hipLaunchKernelGGL(merge_emit_kernel, dim3((n_max + 255) / 256), dim3(256), 0, st, d_rows, stride, w.order.as<int>(),
                     w.kbox.as<float4>(), w.kscore.as<float>(), w.ksrc.as<int>(), w.dk.as<int>(), cap, d_out, d_out_src, d_out_n);
  FFP_HIP(hipGetLastError());
}

void run_merge_passthrough(const float* d_rows, const int* d_n, int stride, float* d_out, int* d_out_src, int* d_out_n, int cap,
                           hipStream_t st) {
  hipLaunchKernelGGL(merge_passthrough_kernel, dim3(1), dim3(64), 0, st, d_rows, stride, d_n, cap, d_out, d_out_src, d_out_n);
  FFP_HIP(hipGetLastError());
}

}  // namespace ffp
