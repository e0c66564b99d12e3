// sr_ops.hip — Real-ESRGAN pre/post-processing around RRDBNet, batched over ragged tiles.
//
// Restates RealESRGANer.enhance's array handling (reference call utils/enhancer.py:214; realesrgan 0.3.0, SURVEY.md
// Appendix D.2): uint8 BGR -> /255 -> RGB -> (reflect pre_pad / mod pad) -> [pixel_unshuffle for x2] -> net ->
// clamp(0,1) -> BGR -> (x*255).round() uint8, with the tile loop's "copy the core of each padded tile" done by
// the store kernel.
#include "sr_ops.hpp"

namespace ffp {

namespace {

__device__ __forceinline__ int find_img(const int4* tab, int n, long long gp) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((long long)tab[mid].x <= gp) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ int reflect2(int i, int n_pre, int n) {   // F.pad(..., 'reflect') twice on the high side:
  if (i >= n_pre) i = 2 * (n_pre - 1) - i;                           // mod pad reflects the pre-padded image,
  if (i >= n) i = 2 * (n - 1) - i;                                   // pre_pad reflects the source
  return i;
}

// SHUF = 1: one thread per input pixel, 3 channels. SHUF = 2: pixel_unshuffle(2): 12 channels c*4 + dy*2 + dx.
template <typename T, int SHUF, int CPAD>
__global__ void sr_pre_kernel(const uint8_t* __restrict__ base, const SrSrc* __restrict__ srcs, const int4* __restrict__ tab,
                              int n_img, T* __restrict__ out, long long total_px) {
  const long long gp = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gp >= total_px) return;
  const int im = find_img(tab, n_img, gp);
  const int4 t = tab[im];
  const SrSrc s = srcs[im];
  const int lp = (int)(gp - t.x);
  if (lp >= t.y * t.z) return;                // capacity-mode level: pixel past the current batch
  const int y = lp / t.z, x = lp - y * t.z;
  T* op = out + (size_t)gp * CPAD;
  const uint8_t* img = base + s.src_off;
#pragma unroll
  for (int dy = 0; dy < SHUF; ++dy)
#pragma unroll
    for (int dx = 0; dx < SHUF; ++dx) {
      const int sy = reflect2(s.y0 + y * SHUF + dy, s.pre_h, s.src_h), sx = reflect2(s.x0 + x * SHUF + dx, s.pre_w, s.src_w);
      const uint8_t* p = img + (size_t)sy * s.src_stride + (size_t)sx * 3;
      // source is BGR; network channel order is RGB
      const float r = (float)p[2] / 255.0f, g = (float)p[1] / 255.0f, b = (float)p[0] / 255.0f;
      if (SHUF == 1) { op[0] = (T)r; op[1] = (T)g; op[2] = (T)b; }
      else {
        const int k = dy * SHUF + dx;
        op[0 * SHUF * SHUF + k] = (T)r; op[1 * SHUF * SHUF + k] = (T)g; op[2 * SHUF * SHUF + k] = (T)b;
      }
    }
#pragma unroll
  for (int c = 3 * SHUF * SHUF; c < CPAD; ++c) op[c] = (T)0.f;
}

// one thread per pixel of each tile's core region in the output canvas
__global__ void sr_post_kernel(const float* __restrict__ net_out, int cs, const int4* __restrict__ tab, const SrDst* __restrict__ dsts,
                               const long long* __restrict__ core_off, int n_img, long long total_core, uint8_t* __restrict__ base) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= total_core) return;
  int lo = 0, hi = n_img - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (core_off[mid] <= g) lo = mid; else hi = mid - 1; }
  const int im = lo;
  const SrDst d = dsts[im];
  const int4 t = tab[im];
  const int lp = (int)(g - core_off[im]);
  if (lp >= d.cw * d.ch) return;              // padding entry (capacity-mode batch) or empty core
  const int y = lp / d.cw, x = lp - y * d.cw;
  const float* p = net_out + ((size_t)t.x + (size_t)(d.ty + y) * t.z + (d.tx + x)) * cs;
  uint8_t* o = base + d.dst_off + (size_t)(d.oy + y) * d.dst_stride + (size_t)(d.ox + x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = fminf(fmaxf(p[2 - c], 0.f), 1.f);       // RGB -> BGR
    o[c] = (uint8_t)rintf(v * 255.0f);
  }
}

// crop gather: utils/visualization.py:185-223 semantics are applied on the host (int box, clamp); this copies rows.
__global__ void crop_gather_kernel(const uint8_t* __restrict__ frame, int W, const int4* __restrict__ boxes /*x0,y0,w,h*/,
                                   const long long* __restrict__ offs, int n, uint8_t* __restrict__ out) {
  const int i = blockIdx.y;
  if (i >= n) return;
  const int4 b = boxes[i];
  const long long total = (long long)b.z * b.w * 3;
  for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(k / (b.z * 3)), col = (int)(k - (long long)row * b.z * 3);
    out[offs[i] + k] = frame[((size_t)(b.y + row) * W + b.x) * 3 + col];
  }
}

}  // namespace

void launch_sr_pre(const uint8_t* d_base, const SrSrc* d_srcs, const TView& out, int shuf, hipStream_t st) {
  const long long total = out.lvl->total_px;
  const unsigned nb = (unsigned)((total + 255) / 256);
  const int4* tab = out.lvl->d_tab.as<int4>();
  FFP_CHECK(out.coff == 0 && out.cs == out.C, FFP_ERR_ARG, "sr_pre: output must be a whole buffer");
  if (out.dt == F32 && shuf == 1 && out.C == 4)
    hipLaunchKernelGGL((sr_pre_kernel<float, 1, 4>), dim3(nb), dim3(256), 0, st, d_base, d_srcs, tab, out.lvl->n, (float*)out.ptr, total);
  else if (out.dt == F32 && shuf == 2 && out.C == 12)
    hipLaunchKernelGGL((sr_pre_kernel<float, 2, 12>), dim3(nb), dim3(256), 0, st, d_base, d_srcs, tab, out.lvl->n, (float*)out.ptr, total);
  else if (out.dt == F16 && shuf == 1 && out.C == 8)
    hipLaunchKernelGGL((sr_pre_kernel<_Float16, 1, 8>), dim3(nb), dim3(256), 0, st, d_base, d_srcs, tab, out.lvl->n, (_Float16*)out.ptr, total);
  else if (out.dt == F16 && shuf == 2 && out.C == 16)
    hipLaunchKernelGGL((sr_pre_kernel<_Float16, 2, 16>), dim3(nb), dim3(256), 0, st, d_base, d_srcs, tab, out.lvl->n, (_Float16*)out.ptr, total);
  else
    fail(FFP_ERR_ARG, "sr_pre: no kernel for dtype %d shuffle %d channels %d", (int)out.dt, shuf, out.C);
  FFP_HIP(hipGetLastError());
}

void launch_sr_post(const TView& net_out, const SrDst* d_dsts, const long long* d_core_off, long long total_core, uint8_t* d_base,
                    hipStream_t st) {
  FFP_CHECK(net_out.dt == F32, FFP_ERR_ARG, "sr_post: network output must be fp32");
  if (total_core == 0) return;
  const unsigned nb = (unsigned)((total_core + 255) / 256);
  hipLaunchKernelGGL(sr_post_kernel, dim3(nb), dim3(256), 0, st, (const float*)net_out.ptr + net_out.coff, net_out.cs,
                     net_out.lvl->d_tab.as<int4>(), d_dsts, d_core_off, net_out.lvl->n, total_core, d_base);
  FFP_HIP(hipGetLastError());
}

void launch_crop_gather(const uint8_t* d_frame, int W, const int4* d_boxes, const long long* d_offs, int n, uint8_t* d_out,
                        hipStream_t st) {
  if (n == 0) return;
  hipLaunchKernelGGL(crop_gather_kernel, dim3(16, n), dim3(256), 0, st, d_frame, W, d_boxes, d_offs, n, d_out);
  FFP_HIP(hipGetLastError());
}

}  // namespace ffp
