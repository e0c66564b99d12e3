// eval.hpp — launchers of the WIDER FACE evaluation kernels (eval.hip): SURVEY.md §8 row f4.
#pragma once
#include "common.hpp"

namespace ffp {

// "official" protocol (/root/reference/eval/eval_official_widerface.py:302-375): per image the greedy match of predictions
// (in the given order) against faces with the inclusive-pixel IoU, then per score threshold the counts of valid proposals and
// matched faces at the last prediction scoring >= the threshold, summed over images into d_counts [thresh_num][2] (zeroed here).
// d_pred [sum N][5] = x, y, w, h, score; d_gt [sum G][4] = x, y, w, h; d_eval [sum G]: 1 = face is evaluated, 0 = a match
// with it drops the proposal. d_state: scratch, >= sum G + 2 * sum N ints.
void launch_wider_pr(const double* d_pred, const long long* d_pred_off, const double* d_gt, const long long* d_gt_off, const unsigned char* d_eval,
                     int n_img, double iou_thr, int thresh_num, int* d_state, long long total_pred, long long total_gt,
                     unsigned long long* d_counts, hipStream_t st);

// "dual" protocol (/root/reference/eval/eval_dual.py:272-291, 369-399): per image, per prediction in the given order,
// flag 1 = true positive (first valid face with the strictly largest continuous-coordinate IoU, >= thr, not matched yet),
// 2 = dropped (not a TP and IoU >= thr with an ignored face; every prediction of an image without a valid face), 0 = false positive.
// d_valid [sum F]: 1 = face belongs to the evaluated set, 0 = ignored face. d_state: scratch, >= sum F ints.
void launch_dual_match(const double* d_pred, const long long* d_pred_off, const double* d_faces, const long long* d_face_off,
                       const unsigned char* d_valid, int n_img, double iou_thr, int* d_state, long long total_faces, int* d_flags,
                       hipStream_t st);

}  // namespace ffp
