// det_post.hip — Pose-head decode, per-image NMS and un-letterboxing, batched over all crops of a frame.
//
// Restates (on the device) what Ultralytics does after the forward pass of `model.predict`
// (reference call site utils/yolo_wrapper.py:74-80; semantics SURVEY.md Appendix B steps 4-6):
//   Detect._inference + Pose.kpts_decode  -> decode_kernel
//   non_max_suppression (conf > thr, xywh2xyxy, class offset, torchvision nms IoU > thr, max_det) -> nms_kernel
//   scale_boxes / scale_coords (pad, gain, clip)                                                 -> nms_kernel tail
// Greedy NMS is done as repeated arg-max over the live candidates (one wave-reduced max per kept box), which visits
// boxes in exactly the (score desc, anchor index asc) order of a stable sort, without sorting.
#include "det_post.hpp"

namespace ffp {

namespace {

__device__ __forceinline__ int upper_img(const int* off, int n, int g) {   // largest i with off[i] <= g
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= g) lo = mid; else hi = mid - 1;
  }
  return lo;
}

struct AnchorPos { int lvl, y, x, w; long long px; };

__device__ __forceinline__ AnchorPos locate(const DecodeArgs& a, int img, int la) {
  AnchorPos r;
  int l = 0;
  int4 t = a.tab[0][img];
  int n = t.y * t.z;
  if (la >= n) { la -= n; l = 1; t = a.tab[1][img]; n = t.y * t.z; if (la >= n) { la -= n; l = 2; t = a.tab[2][img]; } }
  r.lvl = l; r.w = t.z; r.y = la / t.z; r.x = la - r.y * t.z; r.px = (long long)t.x + la;
  return r;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float dfl_expect(const float* p) {
  float m = p[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) m = fmaxf(m, p[i]);
  float e[16], s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { e[i] = expf(p[i] - m); s += e[i]; }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc = fmaf(e[i] / s, (float)i, acc);
  return acc;
}

// one thread per anchor: box (xyxy, net-input pixels), best class score/index; score = -1 when not > conf
__global__ void decode_kernel(const DecodeArgs a, float conf, float4* __restrict__ boxes, float* __restrict__ scores,
                              int* __restrict__ classes) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= a.total_anchors) return;
  const int img = upper_img(a.anchor_off, a.n_img, g);
  const AnchorPos ap = locate(a, img, g - a.anchor_off[img]);
  const float* r = a.head[ap.lvl] + (size_t)ap.px * a.head_cs;
  const float stride = (float)(8 << ap.lvl);
  float best = -INFINITY; int bj = 0;
  for (int c = 0; c < a.nc; ++c) { const float v = r[64 + c]; if (v > best) { best = v; bj = c; } }
  const float sc = sigmoidf_(best);
  float out_s = -1.f;
  float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
  if (sc > conf) {
    float d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = dfl_expect(r + 16 * k);
    const float ax = (float)ap.x + 0.5f, ay = (float)ap.y + 0.5f;
    const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
    const float cx = ((x1 + x2) / 2.0f) * stride, cy = ((y1 + y2) / 2.0f) * stride;
    const float w = (x2 - x1) * stride, h = (y2 - y1) * stride;
    bx = make_float4(cx - w / 2.0f, cy - h / 2.0f, cx + w / 2.0f, cy + h / 2.0f);
    out_s = sc;
  }
  boxes[g] = bx; scores[g] = out_s; classes[g] = bj;
}

// raw inference-mode head output (4+nc+3*nkpt, A) per image, for parity tests
__global__ void decode_raw_kernel(const DecodeArgs a, float* __restrict__ out, const long long* __restrict__ out_off) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= a.total_anchors) return;
  const int img = upper_img(a.anchor_off, a.n_img, g);
  const int la = g - a.anchor_off[img];
  const int A = a.anchor_off[img + 1] - a.anchor_off[img];
  const AnchorPos ap = locate(a, img, la);
  const float* r = a.head[ap.lvl] + (size_t)ap.px * a.head_cs;
  const float stride = (float)(8 << ap.lvl);
  float* o = out + out_off[img] + la;
  float d[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) d[k] = dfl_expect(r + 16 * k);
  const float ax = (float)ap.x + 0.5f, ay = (float)ap.y + 0.5f;
  const float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
  o[0 * (size_t)A] = ((x1 + x2) / 2.0f) * stride;
  o[1 * (size_t)A] = ((y1 + y2) / 2.0f) * stride;
  o[2 * (size_t)A] = (x2 - x1) * stride;
  o[3 * (size_t)A] = (y2 - y1) * stride;
  for (int c = 0; c < a.nc; ++c) o[(size_t)(4 + c) * A] = sigmoidf_(r[64 + c]);
  const float* kp = r + 64 + a.nc;
  for (int k = 0; k < a.nkpt; ++k) {
    o[(size_t)(4 + a.nc + 3 * k + 0) * A] = (kp[3 * k + 0] * 2.0f + (ax - 0.5f)) * stride;
    o[(size_t)(4 + a.nc + 3 * k + 1) * A] = (kp[3 * k + 1] * 2.0f + (ay - 0.5f)) * stride;
    o[(size_t)(4 + a.nc + 3 * k + 2) * A] = sigmoidf_(kp[3 * k + 2]);
  }
}

constexpr int NMS_THREADS = 512;
constexpr int NMS_LCAP = 2048;   // candidates kept in LDS; more spill to the global scratch

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_xor(v, o);
    v = t > v ? t : v;
  }
  return v;
}

// one workgroup per image
__global__ void __launch_bounds__(NMS_THREADS) nms_kernel(const DecodeArgs a, const float4* __restrict__ boxes,
                                                          const float* __restrict__ scores, const int* __restrict__ classes,
                                                          int* __restrict__ cand /*[total_anchors]*/,
                                                          float* __restrict__ cscore /*[total_anchors]*/,
                                                          const DetImg* __restrict__ imgs, float iou_thr, int max_det,
                                                          int round_boxes, int det_stride, float* __restrict__ out_dets,
                                                          int* __restrict__ out_counts) {
  __shared__ float4 lbox[NMS_LCAP];
  __shared__ float lsc[NMS_LCAP];
  __shared__ int wsum[NMS_THREADS / 64];
  __shared__ unsigned long long wmax[NMS_THREADS / 64];
  __shared__ int s_base, s_n;
  __shared__ int keep[1024];
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int a0 = a.anchor_off[img], A = a.anchor_off[img + 1] - a0;
  if (max_det > 1024) max_det = 1024;

  // ---- 1. ordered compaction of candidates (score > conf) -------------------------------------------------------
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < A; c0 += NMS_THREADS) {
    const int i = c0 + tid;
    const bool f = i < A && scores[a0 + i] >= 0.f;
    const unsigned long long m = __ballot(f);
    const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int before = s_base;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (f) {
      const int pos = before + in_wave;
      cand[a0 + pos] = i;
      const float s = scores[a0 + i];
      cscore[a0 + pos] = s;
      if (pos < NMS_LCAP) {
        float4 b = boxes[a0 + i];
        const float c = (float)classes[a0 + i] * 7680.0f;    // class offset (no-op for a single class)
        b.x += c; b.y += c; b.z += c; b.w += c;
        lbox[pos] = b; lsc[pos] = s;
      }
    }
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < NMS_THREADS / 64; ++w) t += wsum[w]; s_base += t; }
    __syncthreads();
  }
  const int n = s_base;

  // ---- 2. greedy NMS by repeated arg-max ----------------------------------------------------------------------------
  int nk = 0;
  for (; nk < max_det; ++nk) {
    unsigned long long best = 0ull;
    for (int j = tid; j < n; j += NMS_THREADS) {
      const float s = j < NMS_LCAP ? lsc[j] : cscore[a0 + j];
      if (s >= 0.f) {
        const unsigned long long k = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)j);
        best = k > best ? k : best;
      }
    }
    best = wave_max_u64(best);
    if (lane == 0) wmax[wave] = best;
    __syncthreads();
    best = wmax[0];
#pragma unroll
    for (int w = 1; w < NMS_THREADS / 64; ++w) best = wmax[w] > best ? wmax[w] : best;
    if (best == 0ull) { __syncthreads(); break; }
    const int js = (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
    float4 bs;
    if (js < NMS_LCAP) bs = lbox[js];
    else {
      bs = boxes[a0 + cand[a0 + js]];
      const float c = (float)classes[a0 + cand[a0 + js]] * 7680.0f;
      bs.x += c; bs.y += c; bs.z += c; bs.w += c;
    }
    const float area_s = (bs.z - bs.x) * (bs.w - bs.y);
    if (tid == 0) keep[nk] = js;
    __syncthreads();   // everyone has read lsc[js] / wmax before they are modified
    for (int j = tid; j < n; j += NMS_THREADS) {
      const bool inl = j < NMS_LCAP;
      const float s = inl ? lsc[j] : cscore[a0 + j];
      if (s < 0.f) continue;
      bool kill = (j == js);
      if (!kill) {
        float4 b;
        if (inl) b = lbox[j];
        else {
          b = boxes[a0 + cand[a0 + j]];
          const float c = (float)classes[a0 + cand[a0 + j]] * 7680.0f;
          b.x += c; b.y += c; b.z += c; b.w += c;
        }
        const float xx1 = fmaxf(bs.x, b.x), yy1 = fmaxf(bs.y, b.y), xx2 = fminf(bs.z, b.z), yy2 = fminf(bs.w, b.w);
        const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
        const float inter = w * h;
        const float area_j = (b.z - b.x) * (b.w - b.y);
        const float ovr = inter / (area_s + area_j - inter);
        kill = ovr > iou_thr;
      }
      if (kill) { if (inl) lsc[j] = -1.f; else cscore[a0 + j] = -1.f; }
    }
    __syncthreads();
  }
  if (tid == 0) out_counts[img] = nk;
  __syncthreads();

  // ---- 3. scale_boxes / scale_coords and write rows (kept order = score descending) ------------------------------------
  const DetImg im = imgs[img];
  for (int k = tid; k < nk; k += NMS_THREADS) {
    const int la = cand[a0 + keep[k]];
    const float4 b = boxes[a0 + la];
    float* o = out_dets + ((size_t)img * max_det + k) * det_stride;
    float bx[4] = {b.x, b.y, b.z, b.w};
    bx[0] -= im.pad_x; bx[2] -= im.pad_x; bx[1] -= im.pad_y; bx[3] -= im.pad_y;
#pragma unroll
    for (int q = 0; q < 4; ++q) bx[q] = bx[q] / im.gain;
    bx[0] = fminf(fmaxf(bx[0], 0.f), (float)im.sw); bx[2] = fminf(fmaxf(bx[2], 0.f), (float)im.sw);
    bx[1] = fminf(fmaxf(bx[1], 0.f), (float)im.sh); bx[3] = fminf(fmaxf(bx[3], 0.f), (float)im.sh);
    if (round_boxes) {
#pragma unroll
      for (int q = 0; q < 4; ++q) bx[q] = rintf(bx[q]);
    }
    o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2]; o[3] = bx[3];
    o[4] = scores[a0 + la];
    o[5] = (float)classes[a0 + la];
    const AnchorPos ap = locate(a, img, la);
    const float* kp = a.head[ap.lvl] + (size_t)ap.px * a.head_cs + 64 + a.nc;
    const float stride = (float)(8 << ap.lvl);
    const float axm = ((float)ap.x + 0.5f) - 0.5f, aym = ((float)ap.y + 0.5f) - 0.5f;
    for (int q = 0; q < a.nkpt; ++q) {
      float kx = (kp[3 * q + 0] * 2.0f + axm) * stride;
      float ky = (kp[3 * q + 1] * 2.0f + aym) * stride;
      kx = (kx - im.pad_x) / im.gain; ky = (ky - im.pad_y) / im.gain;
      kx = fminf(fmaxf(kx, 0.f), (float)im.sw); ky = fminf(fmaxf(ky, 0.f), (float)im.sh);
      o[6 + 3 * q + 0] = kx; o[6 + 3 * q + 1] = ky; o[6 + 3 * q + 2] = sigmoidf_(kp[3 * q + 2]);
    }
  }
}

// utils/yolo_wrapper.py:137-162 + docs sahi/prediction.py:94-120: int-truncate the box, clip like ObjectAnnotation
// (x2 <= full_w, y2 <= full_h applied to crop-local coords), add the crop origin to box and keypoints.
__global__ void truncate_shift_kernel(float* __restrict__ dets, const int* __restrict__ counts, const DetImg* __restrict__ imgs,
                                      int max_det, int det_stride, int nkpt, int full_h, int full_w, int n_img) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_img * max_det) return;
  const int img = g / max_det, k = g - img * max_det;
  if (k >= counts[img]) return;
  const DetImg im = imgs[img];
  float* o = dets + (size_t)g * det_stride;
  int x1 = (int)o[0], y1 = (int)o[1], x2 = (int)o[2], y2 = (int)o[3];
  x1 = max(x1, 0); y1 = max(y1, 0); x2 = min(x2, full_w); y2 = min(y2, full_h);
  o[0] = (float)(x1 + im.x0); o[1] = (float)(y1 + im.y0); o[2] = (float)(x2 + im.x0); o[3] = (float)(y2 + im.y0);
  for (int q = 0; q < nkpt; ++q) { o[6 + 3 * q] += (float)im.x0; o[6 + 3 * q + 1] += (float)im.y0; }
}

}  // namespace

void launch_decode(const DecodeArgs& a, float conf, float4* boxes, float* scores, int* classes, hipStream_t st) {
  if (a.total_anchors == 0) return;
  hipLaunchKernelGGL(decode_kernel, dim3((a.total_anchors + 255) / 256), dim3(256), 0, st, a, conf, boxes, scores, classes);
  FFP_HIP(hipGetLastError());
}

void launch_decode_raw(const DecodeArgs& a, float* out, const long long* out_off, hipStream_t st) {
  if (a.total_anchors == 0) return;
  hipLaunchKernelGGL(decode_raw_kernel, dim3((a.total_anchors + 255) / 256), dim3(256), 0, st, a, out, out_off);
  FFP_HIP(hipGetLastError());
}

void launch_nms(const DecodeArgs& a, const float4* boxes, const float* scores, const int* classes, int* cand, float* cscore,
                const DetImg* imgs, float iou_thr, int max_det, int round_boxes, int det_stride, float* out_dets,
                int* out_counts, hipStream_t st) {
  FFP_CHECK(max_det >= 1 && max_det <= 1024, FFP_ERR_ARG, "max_det %d outside [1,1024]", max_det);
  hipLaunchKernelGGL(nms_kernel, dim3(a.n_img), dim3(NMS_THREADS), 0, st, a, boxes, scores, classes, cand, cscore, imgs, iou_thr,
                     max_det, round_boxes, det_stride, out_dets, out_counts);
  FFP_HIP(hipGetLastError());
}

void launch_truncate_shift(float* dets, const int* counts, const DetImg* imgs, int n_img, int max_det, int det_stride, int nkpt,
                           int full_h, int full_w, hipStream_t st) {
  const int total = n_img * max_det;
  if (total == 0) return;
  hipLaunchKernelGGL(truncate_shift_kernel, dim3((total + 255) / 256), dim3(256), 0, st, dets, counts, imgs, max_det, det_stride,
                     nkpt, full_h, full_w, n_img);
  FFP_HIP(hipGetLastError());
}

}  // namespace ffp
