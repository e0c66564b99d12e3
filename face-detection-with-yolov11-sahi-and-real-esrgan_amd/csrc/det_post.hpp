// det_post.hpp — launchers of the detector post-processing kernels (det_post.hip) and the SAHI merge (merge.hip).
#pragma once
#include "common.hpp"

namespace ffp {

struct DecodeArgs {
  const float* head[3];      // per level [px][head_cs] fp32: 64 DFL logits | nc class logits | 3*nkpt keypoint raws
  const int4* tab[3];        // level tables {off,h,w,0}
  int head_cs;               // floats per pixel record
  int nc, nkpt;
  int n_img;
  const int* anchor_off;     // [n_img+1] prefix of anchors per image (device)
  int total_anchors;
};

// geometry of one crop: how to map net-input pixels back to crop pixels, and where the crop sits in the frame
struct DetImg {
  float gain, pad_x, pad_y;
  int sw, sh;      // crop size (clip bounds)
  int x0, y0;      // crop origin in the frame (SAHI shift_amount)
};

void launch_decode(const DecodeArgs& a, float conf, float4* boxes, float* scores, int* classes, hipStream_t st);
void launch_decode_raw(const DecodeArgs& a, float* out, const long long* out_off, hipStream_t st);
void launch_nms(const DecodeArgs& a, const float4* boxes, const float* scores, const int* classes, int* cand, float* cscore,
                const DetImg* imgs, float iou_thr, int max_det, int round_boxes, int det_stride, float* out_dets,
                int* out_counts, hipStream_t st);
void launch_truncate_shift(float* dets, const int* counts, const DetImg* imgs, int n_img, int max_det, int det_stride, int nkpt,
                           int full_h, int full_w, hipStream_t st);

// ---- SAHI merge (merge.hip) ------------------------------------------------------------------------------------
struct MergeWork {            // device scratch, grown on demand
  DevBuf keys, order, sbox, sscore, scat, mask, kbox, kscore, ksrc, dk;
};
// d_rows: device [*d_n][stride] (first six x1,y1,x2,y2,score,class), *d_n <= n_max (grid bound).
// d_out: device [cap][stride]; d_out_src [cap] (may be null): index of the source row of each output; d_out_n[1].
void run_merge(MergeWork& w, const float* d_rows, const int* d_n, int n_max, int stride, int type, int metric, double thr,
               int class_agnostic, float* d_out, int* d_out_src, int* d_out_n, int cap, hipStream_t st);
void run_merge_passthrough(const float* d_rows, const int* d_n, int stride, float* d_out, int* d_out_src, int* d_out_n, int cap,
                           hipStream_t st);
// gather fixed-cap per-slice detections [n_slices][max_det][stride] into a compact row list (slice order, then kept
// order); d_n[1] receives the row count, d_prefix[n_slices+1] the per-slice offsets.
void launch_compact_rows(const float* d_dets, const int* d_counts, int n_slices, int max_det, int stride, float* d_rows,
                         int* d_n, int* d_prefix, hipStream_t st);

}  // namespace ffp
