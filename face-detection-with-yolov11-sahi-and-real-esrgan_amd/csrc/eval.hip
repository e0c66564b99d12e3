// eval.hip — WIDER FACE evaluation on the device: IoU rows + greedy matching + PR accumulation. One wave per image: the match
// loop over an image's predictions is sequential by definition (each decision depends on which faces are already taken), the
// faces of a row are spread over the lanes, and images are independent. Everything is float64 like the reference's numpy /
// Cython code, compiled without fp contraction, so that IoUs compare against the threshold exactly as they do there.
// Integer results only (counts, flags): the host turns them into precision / recall / AP with the reference's own expressions.
#include <algorithm>

#include "eval.hpp"

namespace ffp {

namespace {

// wave-wide (max value, smallest index with that value): numpy argmax semantics
__device__ __forceinline__ void wave_argmax(double& v, int& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(v, o);
    const int oi = __shfl_xor(i, o);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

__device__ __forceinline__ int ld_state(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_state(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// eval_official_widerface.py:302-375 for one image per wave
__global__ void __launch_bounds__(64) wider_pr_kernel(const double* __restrict__ pred, const long long* __restrict__ pred_off, const double* __restrict__ gt,
                                                      const long long* __restrict__ gt_off, const unsigned char* __restrict__ evalf, double iou_thr, int T,
                                                      int* state, long long total_gt, long long total_pred, unsigned long long* counts) {
  const int img = blockIdx.x, lane = threadIdx.x;
  const long long p0 = pred_off[img], g0 = gt_off[img];
  const int N = (int)(pred_off[img + 1] - p0), G = (int)(gt_off[img + 1] - g0);
  if (N == 0 || G == 0) return;                          // :430-431 `continue`
  const double* P = pred + p0 * 5;
  const double* Q = gt + g0 * 4;
  int* recall = state + g0;                              // 0 free, 1 matched, -1 ignored-and-hit   (zeroed by the launcher)
  int* cum_valid = state + total_gt + p0;                // # proposals == 1 among predictions [0, h]
  int* cum_rec = state + total_gt + total_pred + p0;     // pred_recall[h]
  int matched = 0, valid = 0;
  bool sorted = true;
  double prev = 0.0;
  for (int h = 0; h < N; ++h) {
    const double bx1 = P[h * 5], by1 = P[h * 5 + 1], bx2 = P[h * 5 + 2] + bx1, by2 = P[h * 5 + 3] + by1;     // :321-324 xywh -> xyxy
    const double barea = (bx2 - bx1 + 1) * (by2 - by1 + 1);
    double best = -1.0;
    int bi = 0x7FFFFFFF;
    for (int g = lane; g < G; g += 64) {
      const double qx1 = Q[g * 4], qy1 = Q[g * 4 + 1], qx2 = Q[g * 4 + 2] + qx1, qy2 = Q[g * 4 + 3] + qy1;
      double ov = 0.0;
      const double iw = fmin(bx2, qx2) - fmax(bx1, qx1) + 1;
      if (iw > 0) {
        const double ih = fmin(by2, qy2) - fmax(by1, qy1) + 1;
        if (ih > 0) {
          const double qarea = (qx2 - qx1 + 1) * (qy2 - qy1 + 1);
          const double ua = barea + qarea - iw * ih;
          ov = iw * ih / ua;
        }
      }
      if (ov > best) { best = ov; bi = g; }              // first maximum inside the lane (g ascending)
    }
    wave_argmax(best, bi);
    int prop = 1;
    if (best >= iou_thr) {
      const int s = ld_state(recall + bi);
      if (evalf[g0 + bi] == 0) {
        if (lane == 0) st_state(recall + bi, -1);        // (never a face that was counted: the flag is per face)
        prop = -1;
      } else if (s == 0) {
        if (lane == 0) st_state(recall + bi, 1);
        ++matched;
      }
    }
    if (prop == 1) ++valid;
    if (lane == 0) { st_state(cum_valid + h, valid); st_state(cum_rec + h, matched); }
    const double sc = P[h * 5 + 4];
    if (h > 0 && sc > prev) sorted = false;
    prev = sc;
  }
  // :362-373 per threshold: r = LAST prediction with score >= 1 - (t + 1) / T
  for (int t = lane; t < T; t += 64) {
    const double thresh = 1 - (double)(t + 1) / (double)T;
    int r = -1;
    if (sorted) {                                        // scores descending: the predicate is monotone, bisect
      int lo = 0, hi = N;                                // first index with score < thresh
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (P[mid * 5 + 4] >= thresh) lo = mid + 1; else hi = mid;
      }
      r = lo - 1;
    } else {
      for (int h = N - 1; h >= 0; --h)
        if (P[h * 5 + 4] >= thresh) { r = h; break; }
    }
    if (r >= 0) {
      atomicAdd(counts + 2 * t, (unsigned long long)ld_state(cum_valid + r));
      atomicAdd(counts + 2 * t + 1, (unsigned long long)ld_state(cum_rec + r));
    }
  }
}

__device__ __forceinline__ double dual_iou(double x1, double y1, double w1, double h1, double x2, double y2, double w2, double h2) {
  // eval_dual.py:272-291
  const double ix1 = fmax(x1, x2), iy1 = fmax(y1, y2);
  const double ix2 = fmin(x1 + w1, x2 + w2), iy2 = fmin(y1 + h1, y2 + h2);
  if (ix2 < ix1 || iy2 < iy1) return 0.0;
  const double inter = (ix2 - ix1) * (iy2 - iy1);
  const double uni = (w1 * h1) + (w2 * h2) - inter;
  return uni > 0 ? inter / uni : 0.0;
}

// eval_dual.py:369-399 for one image per wave
__global__ void __launch_bounds__(64) dual_match_kernel(const double* __restrict__ pred, const long long* __restrict__ pred_off, const double* __restrict__ faces,
                                                        const long long* __restrict__ face_off, const unsigned char* __restrict__ validf, double iou_thr,
                                                        int* state, int* flags) {
  const int img = blockIdx.x, lane = threadIdx.x;
  const long long p0 = pred_off[img], f0 = face_off[img];
  const int N = (int)(pred_off[img + 1] - p0), F = (int)(face_off[img + 1] - f0);
  if (N == 0) return;
  const double* P = pred + p0 * 5;
  const double* Q = faces + f0 * 4;
  int* taken = state + f0;                               // zeroed by the launcher
  int nv = 0;
  for (int g = lane; g < F; g += 64) nv += validf[f0 + g] ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nv += __shfl_xor(nv, o);
  if (nv == 0) {                                         // :354-355: image skipped, its predictions are not counted at all
    for (int h = lane; h < N; h += 64) flags[p0 + h] = 2;
    return;
  }
  for (int h = 0; h < N; ++h) {
    const double x = P[h * 5], y = P[h * 5 + 1], w = P[h * 5 + 2], hh = P[h * 5 + 3];
    double best = 0.0;                                   // :370-379 `if iou > best_iou` from 0: a face with IoU 0 never becomes the best
    int bi = 0x7FFFFFFF;
    bool ign = false;
    for (int g = lane; g < F; g += 64) {
      const double iou = dual_iou(x, y, w, hh, Q[g * 4], Q[g * 4 + 1], Q[g * 4 + 2], Q[g * 4 + 3]);
      if (validf[f0 + g]) {
        if (iou > best) { best = iou; bi = g; }
      } else if (iou >= iou_thr) {
        ign = true;
      }
    }
    wave_argmax(best, bi);
    int flag = 0;
    if (best >= iou_thr && bi != 0x7FFFFFFF && ld_state(taken + bi) == 0) {
      if (lane == 0) st_state(taken + bi, 1);
      flag = 1;
    } else if (__any(ign)) {
      flag = 2;
    }
    if (lane == 0) flags[p0 + h] = flag;
  }
}

}  // namespace

void launch_wider_pr(const double* d_pred, const long long* d_pred_off, const double* d_gt, const long long* d_gt_off, const unsigned char* d_eval,
                     int n_img, double iou_thr, int thresh_num, int* d_state, long long total_pred, long long total_gt,
                     unsigned long long* d_counts, hipStream_t st) {
  FFP_HIP(hipMemsetAsync(d_counts, 0, sizeof(unsigned long long) * 2 * thresh_num, st));
  if (n_img == 0) return;
  FFP_HIP(hipMemsetAsync(d_state, 0, sizeof(int) * (size_t)std::max<long long>(total_gt, 1), st));
  hipLaunchKernelGGL(wider_pr_kernel, dim3(n_img), dim3(64), 0, st, d_pred, d_pred_off, d_gt, d_gt_off, d_eval, iou_thr, thresh_num, d_state,
                     total_gt, total_pred, d_counts);
  FFP_HIP(hipGetLastError());
}

void launch_dual_match(const double* d_pred, const long long* d_pred_off, const double* d_faces, const long long* d_face_off,
                       const unsigned char* d_valid, int n_img, double iou_thr, int* d_state, long long total_faces, int* d_flags,
                       hipStream_t st) {
  if (n_img == 0) return;
  FFP_HIP(hipMemsetAsync(d_state, 0, sizeof(int) * (size_t)std::max<long long>(total_faces, 1), st));
  hipLaunchKernelGGL(dual_match_kernel, dim3(n_img), dim3(64), 0, st, d_pred, d_pred_off, d_faces, d_face_off, d_valid, iou_thr, d_state, d_flags);
  FFP_HIP(hipGetLastError());
}

}  // namespace ffp
