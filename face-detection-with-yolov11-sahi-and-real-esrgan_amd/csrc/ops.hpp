// ops.hpp — host-side descriptors and launchers of the device operators (one launch covers a whole ragged batch).
#pragma once
#include "common.hpp"
#include "weights.hpp"

namespace ffp {

// Dense convolution k in {1,3}, stride in {1,2}, pad k/2, as an implicit GEMM on MFMA (conv_mfma.hip).
//   out = epilogue( conv(in) + bias ),   epilogue: act, then optional  v = v*s1 + res1,  v = v*s2 + res2
// `up` = 1 reads the input through a nearest x2 upsample (ESRGAN conv_up1/2). Depthwise convs go to DwConvOp.
struct ConvOp {
  const PackedConv* pc = nullptr;
  TView in, out;
  int stride = 1;
  int act = ACT_NONE;
  int up = 0;
  bool has_res1 = false, has_res2 = false;
  TView res1, res2;
  float s1 = 1.f, s2 = 1.f;
  // 1x1 convs over a virtual concat [nearest_x2(up2) | rest]: input channels [0, up2_c) are read from `up2` (a view one
  // level coarser) through the level's x2 pixel map instead of from `in` (YOLO neck: Upsample + Concat + C3k2.cv1 without
  // materialising the upsampled tensor)
  bool has_up2 = false;
  TView up2;
  int up2_c = 0;
  const int* up2_map = nullptr;   // device: Level::up2_map of in.lvl over up2.lvl
  double flops = 0;   // 2*MAC, algorithmic (unpadded)
  int dbg = 0;        // tuning only: phase-skip mask (see ConvArgs::dbg)
  int force_shape = -1;   // tuning only: 0 wide .. 5 narrow1H
};
void launch_conv(const ConvOp& op, hipStream_t st);
// image-input 3x3 convs (3 real channels: YOLO stem, ESRGAN conv_first) as a direct VALU kernel (ops_misc.hip)
bool conv_direct_eligible(const ConvOp& op);
void launch_conv_direct(const ConvOp& op, hipStream_t st);
void conv_kernels_init();   // raise dynamic-LDS limits once per process
// times the valid workgroup shapes of a generic-kernel conv on its own buffers; returns the fastest shape id, or -1 when
// the op has a single candidate / goes to a specialised kernel (direct stem, row-reuse)
int conv_tune(const ConvOp& op, hipStream_t st);
// The image-input conv computed inside the loader of the conv that consumes it (fp32-split plans, YOLO11s: model.0 -> model.1): the
// stem's output (1 GB per 2-frame group of 4K slices) is neither written nor read back. stem_conv_pack() lays the stem weights out
// as MFMA fragments for one channel order; launch_stem_conv() needs the frame, so it runs outside the plan's captured steps.
bool stem_conv_eligible(const ConvOp& stem, const ConvOp& conv);       // env FFP_NO_STEM_FUSE=1: never
void stem_conv_pack(const ConvOp& stem, int flip, DevBuf& out, hipStream_t st);
void launch_stem_conv(const uint8_t* d_frame, int H, int W, const DevBuf& d_imgs, const DevBuf& stem_w, const ConvOp& stem, const ConvOp& conv, hipStream_t st);

// Depthwise 3x3, stride 1 (YOLO11 cls-tower DWConv and the PSA positional conv), fp32 math.
// Input channel c is read from  in.coff + (c / grp) * grp_stride + grp_off + c % grp  (grp = C: identity) so that the
// attention kernel's per-head [q|k|v] layout can be read in place.
struct DwConvOp {
  const PackedConv* pc = nullptr;
  TView in, out;
  int act = ACT_NONE;
  int grp = 0, grp_stride = 0, grp_off = 0;
  bool has_res = false;
  TView res;
};
void launch_dwconv(const DwConvOp& op, hipStream_t st);      // raises op.out.amax when set
// slots[i] = init[i] (bit patterns of max-|value| bounds known without looking at data; 0 elsewhere): first step of a plan
void launch_amax_init(unsigned* slots, const unsigned* init, int n, hipStream_t st);
// *dst = max(*dst, *src): an op that only moves values (nearest upsample into a concat slice) hands its input's bound on
void launch_amax_max(unsigned* dst, const unsigned* src, hipStream_t st, int n = 1);       // n slots (per-image slots: TView::amax_n)

// SPPF: y1,y2,y3 = 5x5/9x9/13x13 stride-1 max pools of `in` (== three chained MaxPool2d(5,1,2)), written to three slices.
void launch_sppf_pool(const TView& in, const TView& y1, const TView& y2, const TView& y3, hipStream_t st);

// nearest x2 upsample of `in` (level L) into `out` (level with doubled dims)
void launch_upsample2x(const TView& in, const TView& out, hipStream_t st);

// Multi-head attention of C2PSA: qkv [px][nh*(2kd+hd)] -> out [px][nh*hd] = softmax(q^T k * scale) applied to v.
void launch_psa_attention(const TView& qkv, const TView& out, int nh, int kd, int hd, hipStream_t st);

// ---- detector pre/post ------------------------------------------------------------------------------------
struct LetterboxImg {     // one network input image cut from the frame
  int x0, y0, sw, sh;     // source rect in the frame
  int new_w, new_h;       // resized size
  int top, left;          // padding
  int net_h, net_w;       // = new + pads
};
// frame HxWx3 u8 (device) -> NHWC T with CPAD channels (3 used), /255, pad 114/255, optional channel flip.
void launch_letterbox(const uint8_t* d_frame, int H, int W, int flip, const DevBuf& d_imgs /*LetterboxImg[n]*/,
                      const TView& out, hipStream_t st);

// the image-input conv (YOLO stem) reading the u8 frame through the letterbox arithmetic: letterbox + stem in one launch,
// bit-identical to launch_letterbox followed by launch_conv_direct (op.in only carries the level and the channel padding)
void launch_stem_from_frame(const uint8_t* d_frame, int H, int W, int flip, const DevBuf& d_imgs, const ConvOp& op, hipStream_t st);

}  // namespace ffp
