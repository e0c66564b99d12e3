// sr_ops.hpp — Real-ESRGAN pre/post kernels (sr_ops.hip) and the SR engine (rrdb.cpp).
#pragma once
#include "engine.hpp"

namespace ffp {

struct SrSrc {          // where one network tile reads its pixels
  long long src_off;    // byte offset of the source image inside the base allocation
  int src_stride;       // bytes per source row
  int src_h, src_w;     // source size
  int pre_h, pre_w;     // size after pre_pad; rows beyond reflect about it first (mod pad), then about the source
  int x0, y0;           // tile origin in padded-source coordinates
};
struct SrDst {          // where the core of one tile's output lands
  long long dst_off;    // byte offset of the output canvas inside the base allocation
  int dst_stride;       // bytes per canvas row
  int ox, oy;           // core origin in the canvas
  int cw, ch;           // core size (output pixels)
  int tx, ty;           // core origin inside the tile's network output
};

void launch_sr_pre(const uint8_t* d_base, const SrSrc* d_srcs, const TView& out, int shuf, hipStream_t st);
void launch_sr_post(const TView& net_out, const SrDst* d_dsts, const long long* d_core_off, long long total_core, uint8_t* d_base,
                    hipStream_t st);
void launch_crop_gather(const uint8_t* d_frame, int W, const int4* d_boxes, const long long* d_offs, int n, uint8_t* d_out,
                        hipStream_t st);

// A plan is laid out for a CAPACITY (16x16 tiles at the body level), not for a list of image sizes: its levels are in
// capacity mode (Level::reserve / assign), so one allocated, tuned and graph-captured plan serves every batch that fits.
struct SrPlan : Plan {
  Level* Lb = nullptr;     // body level (input res, or half res for x2)
  Level* L1 = nullptr;     // x2 of body
  Level* L2 = nullptr;     // x4 of body
  TView input, out;
  DevBuf d_srcs, d_dsts, d_core_off;
  int cap_t16 = 0;
  void* stage = nullptr;   // pinned staging of the per-batch tile descriptors
  size_t stage_bytes = 0;
  unsigned long long last_use = 0;
  ~SrPlan();
};

struct SrImage {           // one image of a batch: offsets are relative to the in/out base allocations
  long long in_off; int in_stride; int h, w;
  long long out_off; int out_stride;
};

class SrEngine {
 public:
  SrEngine(const void* weights, size_t nbytes, int scale, int num_block, int device, int half);
  ~SrEngine();
  int scale() const { return scale_; }
  int device() const { return device_; }
  hipStream_t stream() const { return st_; }
  // all images live in device memory: inputs at d_in + in_off (BGR u8, in_stride bytes/row), outputs at d_out + out_off
  // wait = false: return as soon as the work is enqueued on the engine's stream (wait_done() completes it), so that the
  // caller can overlap the next frame's detection (own stream) with this frame's super-resolution
  void enhance_dev(const uint8_t* d_in, uint8_t* d_out, const std::vector<SrImage>& imgs, int tile, int tile_pad, int pre_pad, bool wait = true);
  void wait_done();
  // the body (every conv of the residual dense blocks) as ONE persistent launch (conv_trunk.hip) or one launch per layer; bit-identical
  // results either way. Default: fused in fp16 unless FFP_TRUNK=0. Changing it drops the resident plans.
  void set_fused_body(bool on);
  bool fused_body() const { return fused_body_; }
  void drop_plans();            // release every resident plan (waits for the batch in flight and the stream first)
  size_t plan_bytes() const;    // device memory held by the resident plans (activations, tables), without the packed weights
  size_t weight_bytes() const { return weight_bytes_; }
  int plans_resident() const { return (int)plans_.size(); }

  DevBuf scratch_in, scratch_out, scratch_boxes, scratch_offs;
  HostPinned crop_stage;
  ConvProfile prof;
  float last_ms = 0.f;
  double last_conv_flops = 0;
  int last_conv_launches = 0;
  bool last_graph = false;      // the last call replayed a captured hipGraph (false: eager launches)
  int plans_built = 0;          // plans laid out so far (a steady stream of varying crop sizes must not grow this)

 private:
  struct TileDesc { SrSrc src; SrDst dst; int h, w; };
  void build_plan(SrPlan& P, int cap_t16);
  SrPlan& plan_for(long long t16);
  unsigned long long use_clock_ = 0;
  const PackedConv* conv(const std::string& name) const;
  std::map<std::string, PackedConv> convs_;
  std::map<int, std::unique_ptr<SrPlan>> plans_;      // keyed by capacity
  int scale_ = 4, num_block_ = 23, device_ = 0;
  DType dt_ = F16;
  hipStream_t st_ = nullptr;
  hipEvent_t ev_[2];
  bool pending_ = false;
  bool fused_body_ = false;
  size_t weight_bytes_ = 0;
};

}  // namespace ffp
