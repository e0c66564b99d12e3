"""Where the host time of the drop-in API goes: cProfile over get_sliced_prediction (4K frame, 512 / 0.2) and over per-crop enhance_image."""
import cProfile, io, os, pstats, sys, time, contextlib
sys.path.insert(0, os.getcwd())
import numpy as np
import ffp_amd  # noqa: F401
from ffp_amd import synth, pipeline
sys.path.insert(0, os.path.join(os.getcwd(), "face-detection-with-yolov11-sahi-and-real-esrgan_amd", "compat"))
from sahi.predict import get_sliced_prediction
from utils.enhancer import FaceEnhancer
from utils.yolo_wrapper import YOLOv11PoseDetectionModel
q = io.StringIO()
with contextlib.redirect_stdout(q):
    model = YOLOv11PoseDetectionModel(model_path=synth.yolo11_pose_weights("s"), confidence_threshold=0.5, device="cuda:0", image_size=512)
    enh = FaceEnhancer("RealESRGAN_x4plus", model_path=synth.rrdbnet_weights(4, 23), scale=4, tile=400, half=True)
frames = [synth.synthetic_frame(2160, 3840, seed=i) for i in range(2)]
def det(k):
    with contextlib.redirect_stdout(q):
        return get_sliced_prediction(frames[k % 2], model, slice_height=512, slice_width=512, overlap_height_ratio=0.2, overlap_width_ratio=0.2, verbose=0)
def sr(k):
    boxes = pipeline.crop_boxes_for_sr(np.zeros((0, 5), np.float32), 2160, 3840, 32, pipeline.sr_crop_sizes(32, seed=1000 + k), seed=k)
    f = frames[k % 2]
    for x0, y0, x1, y1 in boxes:
        with contextlib.redirect_stdout(q):
            enh.enhance_image(np.ascontiguousarray(f[y0:y1, x0:x1, ::-1]))
for name, fn in (("get_sliced_prediction", det), ("enhance_image x 32", sr)):
    for k in range(3):
        fn(k)
    t0 = time.perf_counter()
    for k in range(10):
        fn(k)
    print(f"== {name}: {(time.perf_counter() - t0) * 100:.2f} ms per frame", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for k in range(10):
        fn(k)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000], flush=True)
