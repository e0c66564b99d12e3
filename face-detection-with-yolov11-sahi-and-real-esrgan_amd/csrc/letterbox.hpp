// letterbox.hpp — device-side LetterBox sampling shared by the image-input kernels (ops_misc.hip) and the stem-fused loader of
// the first MFMA conv (conv_mfma.hip).
#pragma once
#include "ops.hpp"

namespace ffp {

// cv2.resize INTER_LINEAR for uint8 restated in fixed point (11-bit coefficients, two-pass rounding)
__device__ __forceinline__ void lin_coef(int d, double scale, int n_src, bool zero_at_border, int& s, int& c0, int& c1) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int si = (int)floorf(f);
  f -= (float)si;
  if (zero_at_border) {
    if (si < 0) { f = 0.f; si = 0; }
    if (si >= n_src - 1) { f = 0.f; si = n_src - 1; }
  }
  s = si;
  c0 = (int)rintf((1.0f - f) * 2048.0f);
  c1 = (int)rintf(f * 2048.0f);
}

// One pixel of the letterboxed network image (u8 per channel, source channel order): pad 114 outside the resized picture,
// a plain copy when the crop is not resized, else the fixed-point bilinear sample.
__device__ __forceinline__ void letterbox_sample(const uint8_t* __restrict__ frame, int W, const LetterboxImg& L, int y, int x, int (&px)[3]) {
  px[0] = px[1] = px[2] = 114;
  const int ry = y - L.top, rx = x - L.left;
  if (ry >= 0 && ry < L.new_h && rx >= 0 && rx < L.new_w) {
    const uint8_t* src = frame + ((size_t)L.y0 * W + L.x0) * 3;
    const size_t rs = (size_t)W * 3;
    if (L.new_w == L.sw && L.new_h == L.sh) {
      const uint8_t* p = src + (size_t)ry * rs + (size_t)rx * 3;
      px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
    } else {
      int sx, ax0, ax1, sy, by0, by1;
      lin_coef(rx, (double)L.sw / (double)L.new_w, L.sw, true, sx, ax0, ax1);
      lin_coef(ry, (double)L.sh / (double)L.new_h, L.sh, false, sy, by0, by1);
      const int sx1 = min(sx + 1, L.sw - 1);
      const int y0 = min(max(sy, 0), L.sh - 1), y1 = min(max(sy + 1, 0), L.sh - 1);
      const uint8_t* r0 = src + (size_t)y0 * rs;
      const uint8_t* r1 = src + (size_t)y1 * rs;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int S0 = r0[sx * 3 + c] * ax0 + r0[sx1 * 3 + c] * ax1;
        const int S1 = r1[sx * 3 + c] * ax0 + r1[sx1 * 3 + c] * ax1;
        const int v = (((by0 * (S0 >> 4)) >> 16) + ((by1 * (S1 >> 4)) >> 16) + 2) >> 2;
        px[c] = min(max(v, 0), 255);
      }
    }
  }
}

}  // namespace ffp
