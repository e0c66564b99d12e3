"""Placeholder for the reference's RetinaFace adapter (utils/insightface_wrapper.py:7-113, InsightFace / ONNX Runtime): a different
detector that SURVEY.md §2 marks out of scope. The module exists so that the import block of
pipeline_v1_detection_first/app_v1.py:10-14 resolves; constructing the model says what to use instead."""


class InsightFaceDetectionModel:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("the RetinaFace / InsightFace detector is not part of this build (hot path = YOLO11-pose): "
                                  "use utils.yolo_wrapper.YOLOv11PoseDetectionModel with the same get_sliced_prediction call")
