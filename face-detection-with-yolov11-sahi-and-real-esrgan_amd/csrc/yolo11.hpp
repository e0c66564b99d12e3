// yolo11.hpp — the detector engine: YOLO11{n,s}-pose laid out as a plan of HIP launches per ragged batch shape.
#pragma once
#include "engine.hpp"

namespace ffp {

struct DetPlan : Plan {
  unsigned long long last_use = 0;
  Level* L[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // strides 1,2,4,8,16,32
  TView input;
  TView head[3];
  std::vector<int> anchor_off;
  DevBuf d_anchor_off, d_boxes, d_scores, d_classes, d_cand, d_cscore, d_lb, d_imgs;
  int total_anchors = 0;
  ConvOp stem;              // model.0 when it is fused with the letterbox (runs ahead of the captured launch sequence)
  bool fused_stem = false;
  // fp32-split plans of YOLO11s: model.0 is computed inside model.1's loader (launch_stem_conv) and its output never stored
  ConvOp stemconv;
  bool fused_stem_conv = false;
  std::string stemconv_variant;
  DevBuf stem_w[2];         // the stem's MFMA fragments per channel order (RGB, BGR frames)
};

struct TileGeom {       // host-side geometry of one crop
  LetterboxImg lb;
  DetImg di;
};

class DetEngine {
 public:
  DetEngine(const void* weights, size_t nbytes, int arch, int nc, int nkpt, int device, int precision);
  ~DetEngine();

  int det_stride() const { return 6 + 3 * nkpt_; }
  // Plan::lanes of every plan built from now on (cached plans are dropped): parallel graph branches for the head towers and C3k
  void set_lanes(int mode) { if (mode != lanes_) { lanes_ = mode; plans_.clear(); } }
  int device() const { return device_; }
  hipStream_t stream() const { return st_; }
  int nc() const { return nc_; }
  int nkpt() const { return nkpt_; }
  // device memory: resident plans (activations + tables; one plan per batch shape, least recently used ones are dropped beyond 8 plans or
  // beyond the byte budget — env FFP_DET_PLAN_GIB, default 64) and the packed weights
  size_t plan_bytes() const;
  void drop_plans();            // release every resident plan (the next call lays its plan out again); waits for the stream first
  size_t weight_bytes() const { return weight_bytes_; }
  int plans_resident() const { return (int)plans_.size(); }

  // crops of a device-resident frame -> per-crop detections (device), crop-local float coords
  void infer_tiles_dev(const uint8_t* d_frame, int H, int W, int chan_order, const int32_t* tiles, int n_tiles, int imgsz,
                       float conf, float iou, int max_det, int round_boxes, float* d_out_dets, int32_t* d_out_counts);
  // raw head output for parity tests, host output
  void forward_raw(const uint8_t* d_frame, int H, int W, int chan_order, const int32_t* tiles, int n_tiles, int imgsz,
                   float* out_raw, size_t out_cap, int32_t* out_anchor_counts);
  // + int-truncate, clip, shift to frame coordinates (wrapper + SAHI shift semantics)
  void truncate_shift_dev(float* d_dets, const int32_t* d_counts, int n_tiles, int max_det, int H, int W);

  void merge_dev(const float* d_dets, const int32_t* d_counts, int n_slices, int max_det, int type, int metric, double thr,
                 int class_agnostic, float* d_out, int32_t* d_out_src, int cap, int32_t* d_out_n);

  // scratch the API layer may use (grown on demand)
  DevBuf scratch_frame, scratch_dets, scratch_counts, scratch_rows, scratch_n, scratch_prefix, scratch_out, scratch_outn,
      scratch_src;
  MergeWork merge_work;
  ConvProfile prof;
  float last_ms[5] = {0, 0, 0, 0, 0};
  double last_conv_flops = 0;
  int last_conv_launches = 0;
  int last_graph_state = 0;     // Plan::graph_state() of the plan the last call ran
  hipEvent_t ev_[6];

 private:
  DetPlan* plan_for(const std::vector<TileGeom>& g);
  void build_plan(DetPlan& p, const std::vector<int>& hs, const std::vector<int>& ws);
  std::vector<TileGeom> geometry(int H, int W, const int32_t* tiles, int n_tiles, int imgsz) const;
  DetPlan* prepare(const uint8_t* d_frame, int H, int W, int chan_order, const int32_t* tiles, int n_tiles, int imgsz);
  const PackedConv* conv(const std::string& name) const;

  std::map<std::string, PackedConv> convs_;
  std::map<std::vector<int>, std::unique_ptr<DetPlan>> plans_;
  std::vector<TileGeom> geom_last_;
  int lanes_ = 0;
  unsigned long long use_clock_ = 0;
  size_t weight_bytes_ = 0;
  int device_ = 0, nc_ = 1, nkpt_ = 5;
  char scale_ = 's';
  DType dt_ = F32;
  bool split_ = false;
  hipStream_t st_ = nullptr;
};

void letterbox_geometry(int h, int w, int imgsz, int32_t* out6);

}  // namespace ffp
