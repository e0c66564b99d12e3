/*
 * ffp.h — C-ABI of the MI355X-native sliced-inference face pipeline (libffp.so).
 *
 * Drop-in boundary for the hot path  SAHI slice -> YOLO11-pose detect -> per-slice NMS -> SAHI merge ->
 * Real-ESRGAN x4 on crops.  The reference has no native code and no FFI: it reaches this arithmetic through four
 * pip packages called from Python.  Each entry point below names the reference interface it replaces (paths are
 * relative to /root/reference).  Binding shown in INTEGRATION.md (ctypes).
 *
 * Conventions: return 0 (FFP_OK) on success, a positive FFP_ERR_* code otherwise; ffp_last_error() returns a
 * thread-local message.  Callers own every input and output buffer; outputs are caller-allocated.  A handle is
 * bound to one HIP device and one internal stream and is NOT thread-safe (the reference wrapper is not re-entrant
 * either: utils/yolo_wrapper.py keeps per-call state in self._original_predictions).
 * No function here falls back to a CPU implementation: without a gfx950 device every compute call fails.
 */
#ifndef FFP_H
#define FFP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFP_OK 0
#define FFP_ERR_ARG 1      /* bad argument */
#define FFP_ERR_HIP 2      /* HIP runtime / no device */
#define FFP_ERR_WEIGHTS 3  /* malformed FFPW container or missing tensor */
#define FFP_ERR_NOMEM 4
#define FFP_ERR_STATE 5

/* arithmetic type of the convolution path */
#define FFP_PREC_F32 0 /* fp32 activations, v_mfma_f32_32x32x2_f32 (exact-f32 parity mode) */
#define FFP_PREC_F16 1 /* fp16 activations, v_mfma_f32_32x32x16_f16, fp32 accumulate */
#define FFP_PREC_F32X3 2 /* fp32 activations; each product computed as fp16 hi/lo split, 3 fp16 MFMAs (~2^-21 relative),
                            fp32 accumulate: fp32-grade results at ~5x the exact-fp32 MFMA rate */

/* how a caller's HxWx3 uint8 array maps to network channels.
 * AS_BGR reproduces Ultralytics' ndarray path (network channel 0 = array[...,2]) which is what the reference gets,
 * also when SAHI hands it an RGB array (docs sahi/predict.py:103-106 + utils/yolo_wrapper.py:74-80). */
#define FFP_CHAN_AS_BGR 0
#define FFP_CHAN_AS_RGB 1

/* SAHI post-process selectors (docs sahi/predict.py:44-49,150-153) */
#define FFP_PP_NMS 0
#define FFP_PP_GREEDYNMM 1
#define FFP_METRIC_IOU 0
#define FFP_METRIC_IOS 1

/* floats per detection row: x1,y1,x2,y2,score,class, then nkpt*(x,y,conf) */
#define FFP_DET_STRIDE(nkpt) (6 + 3 * (nkpt))

typedef struct ffp_det ffp_det;
typedef struct ffp_sr ffp_sr;

const char* ffp_last_error(void);
int ffp_version(void);
/* number of visible HIP devices (0 and FFP_OK when none) */
int ffp_device_count(int* out_n);

/* ---------------------------------------------------------------------------------------------------------
 * host logic
 * ------------------------------------------------------------------------------------------------------- */

/* sahi.slicing.get_slice_bboxes (called through slice_image at docs sahi/predict.py:229-238).
 * Writes up to cap [xmin,ymin,xmax,ymax] rows, row-major over the grid; *out_n = number of slices (may exceed cap,
 * in which case only cap rows were written). */
int ffp_slice_bboxes(int image_h, int image_w, int slice_h, int slice_w, float overlap_h_ratio, float overlap_w_ratio,
                     int32_t* out_xyxy, int cap, int32_t* out_n);

/* Ultralytics LetterBox(new_shape=imgsz, auto=True, stride=32) geometry for one (h,w) source:
 * out[6] = {new_w, new_h, top, bottom, left, right}.  (inside YOLO.predict, utils/yolo_wrapper.py:74-80) */
int ffp_letterbox_geometry(int h, int w, int imgsz, int32_t* out6);

/* ---------------------------------------------------------------------------------------------------------
 * detector — replaces ultralytics.YOLO(model_path) + .predict(...) behind
 * YOLOv11PoseDetectionModel.load_model / perform_inference (utils/yolo_wrapper.py:47-56, 63-82)
 * ------------------------------------------------------------------------------------------------------- */

/* weights: FFPW container (see weights_io.py) holding the fused conv tensors of YOLO11{n,s}-pose.
 * arch: 'n' or 's'.  nc classes, nkpt keypoints of 3 values.  precision: FFP_PREC_*. */
int ffp_det_create(const void* weights, size_t nbytes, int arch, int nc, int nkpt, int device, int precision,
                   ffp_det** out);
void ffp_det_destroy(ffp_det* d);

/* One batched `predict` over n_tiles crops of one frame (each crop = one `model.predict(source=slice)` call of the
 * reference; the crop [0,0,W,H] is the "standard" full-frame prediction of docs sahi/predict.py:301-314).
 * frame_hwc: HxWx3 uint8 (host).  tiles_xyxy: n_tiles x [x0,y0,x1,y1] in frame pixels.
 * Each crop is letterboxed to imgsz (imgsz <= 0: native — the longer crop side rounded up to a multiple of 32),
 * run through the network, decoded, conf-filtered (score > conf), NMS'ed (IoU > iou suppressed, at most max_det
 * kept, score descending) and mapped back to crop pixels (scale_boxes / scale_coords, clipped to the crop).
 * out_dets: [n_tiles][max_det][FFP_DET_STRIDE(nkpt)] float32, crop-local coordinates, NOT truncated.
 * out_counts: [n_tiles].  round_boxes != 0 applies the `.round()` of older Ultralytics PosePredictor. */
int ffp_det_infer_tiles(ffp_det* d, const uint8_t* frame_hwc, int H, int W, int chan_order,
                        const int32_t* tiles_xyxy, int n_tiles, int imgsz, float conf, float iou, int max_det,
                        int round_boxes, float* out_dets, int32_t* out_counts);

/* Same, with the frame already resident in device memory and outputs written to device memory
 * (all three pointers are device pointers on the handle's device).  Work is enqueued on the handle's stream and
 * the call returns after the stream has drained. */
int ffp_det_infer_tiles_dev(ffp_det* d, const uint8_t* d_frame_hwc, int H, int W, int chan_order,
                            const int32_t* tiles_xyxy_host, int n_tiles, int imgsz, float conf, float iou,
                            int max_det, int round_boxes, float* d_out_dets, int32_t* d_out_counts);

/* The wrapper's conversion on the device, for the tiles of the handle's LAST infer call: boxes int-truncated and
 * clipped like sahi's ObjectAnnotation, then boxes and keypoints shifted by the tile origin
 * (utils/yolo_wrapper.py:137-162 + docs sahi/prediction.py:94-120). d_dets/d_counts as written by
 * ffp_det_infer_tiles_dev. */
int ffp_det_truncate_shift_dev(ffp_det* d, float* d_dets, const int32_t* d_counts, int n_tiles, int max_det, int full_h,
                               int full_w);

/* Raw network output for parity tests: for each tile the inference-mode Pose head output (4+nc+3*nkpt, A_t)
 * float32 = [cx,cy,w,h, class sigmoid.., kpt x,y,sigmoid(v)..] in net-input pixels, written back to back.
 * out_anchor_counts[n_tiles] receives A_t.  out_cap = capacity of out_raw in floats. */
int ffp_det_forward_raw(ffp_det* d, const uint8_t* frame_hwc, int H, int W, int chan_order,
                        const int32_t* tiles_xyxy, int n_tiles, int imgsz, float* out_raw, size_t out_cap,
                        int32_t* out_anchor_counts);

/* ---------------------------------------------------------------------------------------------------------
 * SAHI merge — replaces sahi.postprocess.combine.{NMSPostprocess,GreedyNMMPostprocess}.__call__
 * (constructed at docs sahi/predict.py:254-259, invoked :297,319)
 * ------------------------------------------------------------------------------------------------------- */

/* dets: n rows of `stride` floats (host), first six = x1,y1,x2,y2,score,class (full-frame coords, as SAHI holds
 * them: ints stored as floats).  out: up to n rows of `stride` floats, in SAHI's output order; for GREEDYNMM the box
 * is the union, the score the max, and the remaining columns come from the higher-scored source row.
 * out_src_index[k] = index of that source row.  Runs on `device`. */
int ffp_merge(int device, const float* dets, int n, int stride, int type, int metric, double thr, int class_agnostic,
              float* out, int32_t* out_src_index, int32_t* out_n);

/* Fused get_sliced_prediction (docs sahi/predict.py:142-345) for a host frame: slice grid -> batched predict ->
 * int-truncate + shift (utils/yolo_wrapper.py:137-162, docs sahi/prediction.py:94-120) -> optional full-frame
 * prediction when more than one slice -> merge when more than one box.  out rows as in ffp_merge. */
int ffp_sliced_predict(ffp_det* d, const uint8_t* frame_hwc, int H, int W, int chan_order, int slice_h, int slice_w,
                       float overlap_h_ratio, float overlap_w_ratio, int perform_standard_pred, int imgsz,
                       float conf, float iou, int max_det, int round_boxes, int pp_type, int pp_metric, double pp_thr,
                       int class_agnostic, float* out, int cap, int32_t* out_n);

/* Device-resident variant used by bench.py / multi-GPU: frame and outputs are device pointers. rank/world shard the
 * slice list contiguously (rank r takes slices [r*ceil(n/world), ...)); with world > 1 the call stops after the
 * per-rank detections are written to d_local_dets/d_local_counts (global slice order, fixed cap max_det) so that the
 * caller can all-gather them (RCCL) and finish with ffp_merge_dev. */
int ffp_det_stage_dev(ffp_det* d, const uint8_t* d_frame_hwc, int H, int W, int chan_order, int slice_h, int slice_w,
                      float overlap_h_ratio, float overlap_w_ratio, int perform_standard_pred, int imgsz, float conf,
                      float iou, int max_det, int round_boxes, int rank, int world, float* d_local_dets,
                      int32_t* d_local_counts, int32_t* out_n_local, int32_t* out_n_total);

/* Merge of fixed-cap per-slice detections already on the device: d_dets [n_slices][max_det][stride] (full-frame,
 * truncated+shifted), d_counts[n_slices].  d_out [cap][stride], d_out_n[1] device pointers.
 * *d_out_n receives the UNTRUNCATED number of merged detections; when it exceeds cap only the first cap rows were
 * written and the caller must treat the result as overflowed (ffp_sliced_predict returns FFP_ERR_ARG in that case). */
int ffp_merge_dev(ffp_det* d, const float* d_dets, const int32_t* d_counts, int n_slices, int max_det, int type,
                  int metric, double thr, int class_agnostic, float* d_out, int cap, int32_t* d_out_n);

/* ---------------------------------------------------------------------------------------------------------
 * super-resolution — replaces basicsr RRDBNet + realesrgan.RealESRGANer behind FaceEnhancer
 * (utils/enhancer.py:99-156 construction, :214 `self.upsampler.enhance(image, outscale=self.scale)`)
 * ------------------------------------------------------------------------------------------------------- */

/* weights: FFPW container with RRDBNet tensors.  scale 4 (x4plus) or 2 (x2plus, pixel-unshuffle front).
 * half != 0 -> FFP_PREC_F16 (the reference's GPU default, utils/enhancer.py:22), else fp32. */
int ffp_sr_create(const void* weights, size_t nbytes, int scale, int num_block, int device, int half, ffp_sr** out);
void ffp_sr_destroy(ffp_sr* s);

/* RealESRGANer.enhance for one 3-channel uint8 BGR image (host). out_bgr: (scale*h) x (scale*w) x 3 uint8.
 * tile <= 0: whole image in one pass; else tiles of `tile` px padded by tile_pad (clipped). */
int ffp_sr_enhance(ffp_sr* s, const uint8_t* bgr_hwc, int h, int w, int tile, int tile_pad, int pre_pad,
                   uint8_t* out_bgr);

/* The same for an image resident in device memory, output written to device memory ((scale*h) x (scale*w) x 3, tightly
 * packed): the enhance-first ordering (pipeline_v4_yolo/app_yolo_full.py:87-123 — FaceEnhancer.enhance_image on the
 * whole picture, utils/enhancer.py:189-235, then get_sliced_prediction on the result) without a host round trip. */
int ffp_sr_enhance_dev(ffp_sr* s, const uint8_t* d_bgr_hwc, int h, int w, int tile, int tile_pad, int pre_pad, uint8_t* d_out_bgr);

/* n independent images in one ragged batch (one launch per layer for all of them). */
int ffp_sr_enhance_batch(ffp_sr* s, int n, const uint8_t* const* imgs, const int32_t* hs, const int32_t* ws, int tile,
                         int tile_pad, int pre_pad, uint8_t* const* outs);

/* Crops gathered on the device from a resident frame (utils/visualization.py:185-223 crop semantics: int box,
 * clamped to the frame, an empty crop is skipped), each enhanced like FaceEnhancer.enhance_image does
 * (utils/enhancer.py:189-235 -> RealESRGANer.enhance with the enhancer's tile setting: tile 400 / tile_pad 10 at
 * utils/enhancer.py:21,135-142; tile <= 0: one pass per crop), outputs packed back to back in d_out (device), byte
 * offsets in out_offsets (host, n+1 entries; a skipped crop has out_offsets[i+1] == out_offsets[i]).
 * frame is BGR. boxes_xyxy: host int32 [n][4]. */
int ffp_sr_enhance_crops_dev(ffp_sr* s, const uint8_t* d_frame_bgr, int H, int W, const int32_t* boxes_xyxy, int n, int tile,
                             int tile_pad, uint8_t* d_out, size_t out_cap, int64_t* out_offsets);

/* Same, but returns once the work is enqueued on the enhancer's own HIP stream; ffp_sr_wait() blocks until d_out is
 * complete. Lets a caller overlap frame i's super-resolution with frame i+1's detection (the detector handle has its
 * own stream). d_frame and d_out must stay untouched until ffp_sr_wait (or the next call on the handle) returns. */
int ffp_sr_enhance_crops_dev_async(ffp_sr* s, const uint8_t* d_frame_bgr, int H, int W, const int32_t* boxes_xyxy, int n, int tile,
                                   int tile_pad, uint8_t* d_out, size_t out_cap, int64_t* out_offsets);
int ffp_sr_wait(ffp_sr* s);
/* Crops of SEVERAL resident frames (all H x W) as one ragged batch: frame_of_box[i] selects d_frames[...] for box i
 * (boxes of one frame must be contiguous). More pixels per launch = better use of the chip when a frame yields few crops. */
int ffp_sr_enhance_crops_multi_dev_async(ffp_sr* s, int n_frames, const uint8_t* const* d_frames, const int32_t* frame_of_box,
                                         int H, int W, const int32_t* boxes_xyxy, int n, int tile, int tile_pad, uint8_t* d_out,
                                         size_t out_cap, int64_t* out_offsets);

/* Engine state for tests and monitoring. The enhancer lays its network out for a CAPACITY (pixels / tiles), not for a list
 * of crop sizes: out_plans_built counts the layouts built so far (a stream of frames with ever-changing crop sizes must
 * not grow it), out_last_graph tells whether the last call replayed a captured hipGraph (1) or launched eagerly (0). */
int ffp_sr_plan_state(ffp_sr* s, int32_t* out_plans_built, int32_t* out_last_graph);
/* The body of RRDBNet (the 345 convs of the residual dense blocks constructed at utils/enhancer.py:121-128) as ONE persistent launch
 * with per-tile dependency counters (1; fp16 only) or as one launch per layer (0, the default: the faster of the two on MI355X as
 * measured in round 4, profiles/r04_sr_batch_sweep.txt; FFP_TRUNK=1 in the environment makes 1 the default). Results are bit-identical
 * either way; each form is the other's parity oracle and A/B partner. Drops the resident plans. */
int ffp_sr_set_fused_body(ffp_sr* s, int on);
/* Device memory a handle holds: the packed weights and the resident plans (activations + tables; the detector keeps at most 8 plans
 * and FFP_DET_PLAN_GIB (default 64) GiB of them, the enhancer at most 4 capacity buckets, least recently used first out). The
 * reference holds one torch module per model (utils/yolo_wrapper.py:47-61, utils/enhancer.py:156) and lets torch's caching
 * allocator grow; here a caller that switches image_size or batch shapes can see — and bound — what stays resident. */
int ffp_sr_mem_bytes(ffp_sr* s, uint64_t* out_weight_bytes, uint64_t* out_plan_bytes, int32_t* out_plans_resident);
/* Release every resident plan of a handle now (activations, tables, captured graphs; the packed weights stay). The next call lays its plan out
 * again (tens of ms: layout, tuning, graph capture). The counterpart of `del model; torch.cuda.empty_cache()` around the reference's modules. */
int ffp_det_drop_plans(ffp_det* d);
int ffp_sr_drop_plans(ffp_sr* s);
int ffp_det_mem_bytes(ffp_det* d, uint64_t* out_weight_bytes, uint64_t* out_plan_bytes, int32_t* out_plans_resident);
/* hipGraph state of the plan the detector's last call ran: 1 replayed a captured graph, 0 not captured yet (first runs of a
 * plan are eager), -1 capture failed and the plan keeps launching eagerly (also reported once on stderr). */
int ffp_det_graph_status(ffp_det* d, int32_t* out_state);
/* Parallel branches in the detector's captured launch graph (no counterpart in the reference, which runs one layer at a time:
 * /root/reference/utils/yolo_wrapper.py:74-80 hands the whole forward pass to one call). mode 0 (default): one stream;
 * 1: the head's nine towers and C3k's side convs run as parallel graph branches (3-5 % faster when the detector has the device
 * to itself, slower when an enhancer stream runs beside it); 2 / 3: coarser variants for A/B. Drops the cached plans. */
int ffp_det_set_lanes(ffp_det* d, int mode);
/* Stream interop for the multi-GPU exchange (no counterpart in the reference, which is single-device:
 * /root/reference/pipeline_v4_yolo/app_yolo_sahi.py:136): work enqueued on the handle's stream AFTER this call waits for `hip_event`
 * (a hipEvent_t of the same process, e.g. torch.cuda.Event.cuda_event recorded behind the RCCL all-gather of the boxes) — the
 * merge that follows is ordered behind the collective without the host waiting for either. */
int ffp_det_stream_wait_event(ffp_det* d, void* hip_event);

/* ---------------------------------------------------------------------------------------------------------
 * single-operator entry points (layer-wise parity tests; host NHWC fp32 in/out, computed on `device` in `precision`)
 * ------------------------------------------------------------------------------------------------------- */

/* y = act(conv2d(x, w) + b) [* s1 + res]; x: [n][h][w][cin] fp32 NHWC; w: OIHW fp32 (groups: 1 or cin==cout
 * depthwise); k in {1,3}; stride in {1,2}; pad = k/2; act: 0 none, 1 SiLU, 2 LeakyReLU(0.2); up: nearest x2 of x
 * before the conv (0/1); res: optional residual [n][ho][wo][cout] (NULL: none). y: [n][ho][wo][cout] fp32. */
int ffp_op_conv2d(int device, int precision, const float* x, int n, int h, int w, int cin, const float* wt,
                  const float* bias, int cout, int k, int stride, int groups, int act, int up, const float* res,
                  float res_scale, float* y);

/* ---------------------------------------------------------------------------------------------------------
 * WIDER FACE evaluation (host arrays in, integer results out; float64 throughout like the reference's numpy code)
 * ------------------------------------------------------------------------------------------------------- */

/* "Official" protocol: /root/reference/eval/eval_official_widerface.py:302-375 (`_image_eval` + `_img_pr_info`, with the
 * WiderFace-Evaluation `bbox_overlaps` it imports at :24-33) for all images of one difficulty setting, summed as in
 * `_evaluate_setting` :397-443. preds [pred_off[n_img]][5] = x, y, w, h, score in the order the caller evaluates them; gts
 * [gt_off[n_img]][4] = x, y, w, h; evaluate[g] = 1: the setting evaluates this face (the reference's `ignore[keep_index-1] = 1`),
 * 0: a proposal matching it is dropped. Images with no prediction or no face contribute nothing (:430-431). out_counts
 * [thresh_num][2] = {valid proposals, matched faces} at the last prediction with score >= 1 - (t+1)/thresh_num — the
 * reference's pr_curve before `_dataset_pr_info` (counts are integers; precision / recall / AP are the caller's three lines). */
int ffp_eval_wider_pr(int device, const double* preds, const int64_t* pred_off, const double* gts, const int64_t* gt_off, const uint8_t* evaluate,
                      int n_img, double iou_thr, int thresh_num, int64_t* out_counts);

/* "Dual" protocol matching: /root/reference/eval/eval_dual.py:272-291 (`calculate_iou`) and :369-399 of `evaluate_single_set`.
 * faces [face_off[n_img]][4] in annotation order, valid[f] = 1: face belongs to the evaluated category set, 0: ignored face.
 * out_flags [pred_off[n_img]]: 1 true positive, 0 false positive, 2 not counted (overlaps an ignored face, or the image has no
 * valid face and is skipped :354-355). */
int ffp_eval_dual_match(int device, const double* preds, const int64_t* pred_off, const double* faces, const int64_t* face_off, const uint8_t* valid,
                        int n_img, double iou_thr, int32_t* out_flags);

/* ---------------------------------------------------------------------------------------------------------
 * JPEG at the file boundaries
 * ------------------------------------------------------------------------------------------------------- */

/* Replaces `cv2.imwrite(path, img)` / `cv2.imwrite(path, img, [cv2.IMWRITE_JPEG_QUALITY, q])` for .jpg paths
 * (/root/reference/utils/visualization.py:218-221 crops, utils/enhancer.py:273-278 enhanced crops, quality 95): baseline JFIF,
 * YCbCr 4:2:0, integer DCT, Annex K Huffman tables — byte-for-byte the file libjpeg-turbo (OpenCV's and Pillow's codec) writes.
 * img: h x w x 3 uint8, bgr = 1 for OpenCV's channel order (0: RGB). out: caller's buffer of `cap` bytes; *out_size = file size
 * (also set when cap is too small: FFP_ERR_ARG, call again). _dev: the image is already in device memory (row pitch in bytes),
 * e.g. a crop inside the buffer ffp_sr_enhance_crops_dev filled — only the compressed bytes cross PCIe. */
int ffp_jpeg_encode(int device, const uint8_t* img, int h, int w, int bgr, int quality, uint8_t* out, int64_t cap, int64_t* out_size);
int ffp_jpeg_encode_dev(int device, const uint8_t* d_img, int h, int w, int64_t row_stride, int bgr, int quality, uint8_t* out, int64_t cap,
                        int64_t* out_size);
/* The per-crop loop of enhance_face_crops_batch (/root/reference/utils/enhancer.py:344-391: one cv2.imwrite per enhanced crop) as ONE
 * pass over n device-resident images: image i is hs[i] x ws[i] x 3 at d_base + offsets[i] (row pitch strides[i], or ws[i]*3 when
 * strides is NULL) — e.g. the out_offsets layout of ffp_sr_enhance_crops_dev with hs/ws = 4 x the crop sizes. File i is
 * out[out_offsets[i] .. out_offsets[i+1]) (out_offsets has n+1 entries and is filled even when cap is too small: FFP_ERR_ARG). */
int ffp_jpeg_encode_batch_dev(int device, const uint8_t* d_base, int n, const int64_t* offsets, const int32_t* hs, const int32_t* ws, const int64_t* strides,
                              int bgr, int quality, uint8_t* out, int64_t cap, int64_t* out_offsets);

/* Replaces `cv2.imread(path)` for .jpg files (/root/reference/utils/enhancer.py:254, utils/visualization.py:200, and the image
 * load inside sahi's get_sliced_prediction): baseline / extended-sequential 8-bit JFIF, 4:4:4 / 4:2:2 / 4:2:0 or grayscale, restart
 * intervals; libjpeg's default reconstruction (integer IDCT, triangle chroma upsampling, fixed-point YCbCr -> RGB), pixel-identical
 * to what libjpeg-turbo returns. Huffman decoding (see ffp_jpeg_decode_stats below), IDCT, upsampling and colour conversion all run
 * on the device; the host parses the markers. Progressive / arithmetic-coded files: FFP_ERR_ARG. ffp_jpeg_info reads the header only. out: h*w*3 bytes, BGR
 * when bgr = 1 (cv2's order; grayscale files give three equal channels like IMREAD_COLOR). _dev writes into device memory (row
 * pitch in bytes): the frame is ready for ffp_det_infer_tiles_dev; what crosses PCIe is the file itself (through pinned staging kept in a
 * pool), neither coefficients nor pixels are ever materialised on the host. */
int ffp_jpeg_info(const uint8_t* data, int64_t n, int32_t* out_h, int32_t* out_w, int32_t* out_ncomp);
int ffp_jpeg_decode(int device, const uint8_t* data, int64_t n, int bgr, uint8_t* out, int64_t cap);
int ffp_jpeg_decode_dev(int device, const uint8_t* data, int64_t n, int bgr, uint8_t* d_out, int64_t row_stride, int64_t cap);
/* Since round 3 the Huffman decoding of ffp_jpeg_decode* runs on the device as well (csrc/jpeg_huff.hip: self-synchronising parallel
 * decoding; the file crosses PCIe as it is). Streams the device decoder flags — damaged data, anything libjpeg treats specially — are
 * decoded again by the host decoder (FFP_JPEG_HOST_HUFFMAN=1 forces it for every file). Counters since the library was loaded:
 * files decoded on the device, files that fell back to the host decoder, extra synchronisation rounds (beyond the one queued blindly). */
int ffp_jpeg_decode_stats(int64_t* device_decodes, int64_t* host_fallbacks, int64_t* extra_sync_rounds);

/* 1x1 conv over the virtual concat [nearest_x2(coarse) | fine] (the YOLO neck's Upsample + Concat + C3k2.cv1 without
 * materialising the upsampled tensor: /root/reference's model graph via ultralytics' yolo11-pose.yaml layers 11-13 and 14-16).
 * coarse: [n][h/2][w/2][c_up] (c_up a multiple of 64), fine: [n][h][w][c_fine]; wt: [cout][c_up + c_fine] (input channel
 * order = concat order); y = act(conv + bias): [n][h][w][cout]. fp32 and fp32-split precisions. */
int ffp_op_conv1x1_up2(int device, int precision, const float* coarse, const float* fine, int n, int h, int w, int c_up, int c_fine,
                       const float* wt, const float* bias, int cout, int act, float* y);

/* Tuning / test hook: pin the workgroup shape ffp_op_conv2d uses from now on (process-wide; -1 = automatic, the default).
 * 0..5 generic shapes, 9 conv_rows16, 10..16 the pointwise kernels of conv_pw.hip. A shape that cannot run the op is an error
 * of the following ffp_op_conv2d call. */
int ffp_op_conv2d_shape(int force_shape);

/* Tuning hook: mean device time (HIP events, microseconds) of `iters` back-to-back launches of one dense convolution on
 * synthetic data. dbg_mask skips kernel phases (1 stores, 2 MFMAs, 4 chunk refetch, 8 LDS stash) to attribute time —
 * results are then wrong by construction; force_shape pins the workgroup shape (-1: automatic). */
int ffp_op_conv2d_time(int device, int precision, int n, int h, int w, int cin, int cout, int k, int stride, int up, int iters,
                       int dbg_mask, int force_shape, float* out_us);

/* timing hooks for bench.py: HIP-event milliseconds of the last call on the handle, by stage
 * stage: 0 total, 1 preprocess, 2 network, 3 decode+nms, 4 merge */
int ffp_det_last_ms(ffp_det* d, int stage, float* out_ms);
int ffp_sr_last_ms(ffp_sr* s, float* out_ms);

/* algorithmic conv FLOPs (2*MAC) of the last call, its number of convolution launches, and (profiling enabled) the
 * HIP-event time spent in convolution kernels */
int ffp_det_last_conv_stats(ffp_det* d, double* out_flops, float* out_ms, int32_t* out_launches);
int ffp_sr_last_conv_stats(ffp_sr* s, double* out_flops, float* out_ms, int32_t* out_launches);

/* Per-launch HIP-event profiling of the convolution kernels on the handle's stream. While enabled every conv launch
 * of a call is bracketed by two events; afterwards the table lists, per kernel variant ("f16_k3s1_narrow1", ...),
 * the algorithmic FLOPs, the summed event time and the launch count of the LAST call. */
int ffp_det_set_profile(ffp_det* d, int enable);
int ffp_det_profile_count(ffp_det* d, int32_t* out_n);
int ffp_det_profile_get(ffp_det* d, int i, char* name, int name_cap, double* out_flops, float* out_ms, int32_t* out_launches);
int ffp_sr_set_profile(ffp_sr* s, int enable);
int ffp_sr_profile_count(ffp_sr* s, int32_t* out_n);
int ffp_sr_profile_get(ffp_sr* s, int i, char* name, int name_cap, double* out_flops, float* out_ms, int32_t* out_launches);
/* algorithmic bytes of entry i of the same table: every input, residual and output element of the launches once (real channels, the tensors'
 * element size) + weights and bias once — the denominator for the PMC traffic of the same launches (bench.py: roofline.algorithmic_bytes) */
int ffp_det_profile_bytes(ffp_det* d, int i, double* out_bytes);
int ffp_sr_profile_bytes(ffp_sr* s, int i, double* out_bytes);
/* Process-wide lifetime totals per kernel variant over EVERY plan execution since enable(1) (eager, profiled or hipGraph replay; all handles):
 * launches, algorithmic FLOPs, algorithmic bytes. A rocprofv3 --pmc pass over a command sums its counters over all launches of a kernel; with
 * these totals printed by the same command (bench.py --conv-totals), traffic per launch and algorithmic bytes per launch describe one population.
 * Off by default. Diagnostics only: the reference has no counterpart (its torch modules are opaque to it). */
int ffp_conv_totals_enable(int on);
int ffp_conv_totals_count(int32_t* out_n);
int ffp_conv_totals_get(int i, char* name, int name_cap, double* out_flops, double* out_bytes, int64_t* out_launches);
/* per-launch entries of the last profiled call, in launch order: "<variant> <layer name>"; returns FFP_ERR_ARG past the end */
int ffp_det_profile_detail(ffp_det* d, int i, char* name, int name_cap, double* out_flops, float* out_ms);
int ffp_sr_profile_detail(ffp_sr* s, int i, char* name, int name_cap, double* out_flops, float* out_ms);

#ifdef __cplusplus
}
#endif
#endif /* FFP_H */
