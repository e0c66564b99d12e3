"""CPU oracle for the hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package, and
only as the checker / the timed CPU baseline. The product path (the package
`face-detection-with-yolov11-sahi-and-real-esrgan_amd`) never imports it and fails loudly without its HIP library.

What it restates (each function cites the reference file:line it follows):
    yolo11_ref.py   YOLO11{n,s}-pose forward (Ultralytics graph behind utils/yolo_wrapper.py:74-80)      torch CPU fp32
    ultra_post.py   LetterBox, DFL/box/keypoint decode, non_max_suppression, scale_boxes/scale_coords      numpy / torch fp32
    sahi_ref.py     get_slice_bboxes, nms, greedy_nmm, merge, get_prediction / get_sliced_prediction        numpy + python
    rrdbnet_ref.py  RRDBNet forward + RealESRGANer.enhance (utils/enhancer.py:121-156,214)                  torch CPU fp32
    wrapper_ref.py  YOLOv11PoseDetectionModel result conversion + keypoint cache / attach (utils/yolo_wrapper.py:84-217)  python

PARITY UNPINNED against the reference's own outputs: the arithmetic of this path lives in third-party packages
that are neither vendored under /root/reference nor installed here (ultralytics — unpinned/unlisted;
sahi==0.11.34; basicsr==1.4.2; realesrgan==0.3.0; opencv-python==4.11.0.86; torchvision==0.14.1 —
/root/reference/requirements.txt:12,98,134,140,170), no model weights exist offline, and the reference holds no
tests, golden vectors or fixtures for the path. The restatement follows the published
algorithms of those packages and the reference's own call sites; what pins it is structural: exact parameter
counts (2,662,416 / 9,715,744), anchor counts, the slice-grid values and letterbox geometry listed in SURVEY.md
§8c / Appendix C, plus hand-computed micro-cases (tests/test_oracle_*.py).
"""
