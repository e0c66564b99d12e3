"""Per-layer time, algorithmic HBM bytes and achieved GB/s of the detector's convolutions for a 2-frame group (122 items).
Usage on the GPU box: python tools/det_layers.py [--f32] [--all]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ffp_amd  # noqa
from ffp_amd import _lib, pipeline, synth
import torch

H, W, NF = 2160, 3840, 2
cfg = pipeline.PipeConfig(sr_crops=0)
Wd = synth.yolo11_pose_weights("s")
pipe = pipeline.FramePipeline(Wd, None, cfg, arch="s", det_precision=_lib.PREC_F32 if "--f32" in sys.argv else _lib.PREC_F32X3)
frame = torch.from_numpy(np.concatenate([synth.synthetic_frame(H, W, seed=i) for i in range(NF)], 0)).cuda()
for it in range(4):
    pipe.det.set_profile(it == 3)
    pipe.detect(frame, H, W, NF)
det = pipe.det.profile_detail()
S2 = {"model.1.conv", "model.3.conv", "model.5.conv", "model.7.conv", "model.17.conv", "model.20.conv", "model.0.conv"}
tot_ms = tot_b = 0
rows = []
for x in det:
    variant, name = x["name"].split(" ", 1)
    if "+" in name:         # model.0 computed in model.1's loader: reads the u8 network image, writes model.1's output; nothing in between
        w0, w1 = Wd["model.0.conv.weight"], Wd["model.1.conv.weight"]
        px1 = x["flops"] / (2.0 * (w0.shape[1] * 9 * w0.shape[0] * 4 + w1.shape[1] * 9 * w1.shape[0]))
        by = px1 * 16 * 3 + px1 * w1.shape[0] * 4
        rows.append((x["ms"], name, variant, 3, w1.shape[0], 3, by))
        tot_ms += x["ms"]; tot_b += by
        continue
    w = Wd[name + ".weight"]
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    px_out = x["flops"] / (2.0 * cin * k * k * cout)
    px_in = px_out * (4 if name in S2 else 1)
    es = 1 if name == "model.0.conv" else 4
    by = px_in * (3 if name == "model.0.conv" else cin) * es + px_out * cout * 4
    rows.append((x["ms"], name, variant, cin, cout, k, by))
    tot_ms += x["ms"]; tot_b += by
print(f"total conv {tot_ms:.3f} ms for {NF} frames, algorithmic bytes {tot_b/1e9:.2f} GB -> {tot_b/tot_ms/1e9:.2f} TB/s average; graph-mode stage ms:", pipe.det.last_ms())
flops = {x["name"].split(" ", 1)[1]: x["flops"] for x in det}
split = 1 if "--f32" in sys.argv else 3              # MFMA products per algorithmic MAC in the split arithmetic
# which roof binds a layer: its algorithmic bytes at 8 TB/s (what streams at best is 6.0-6.2, MI355X_MICROARCH.md) or its MFMA work (x3 in the split
# arithmetic) at 2.5 PF; `of roof` = that floor / the measured time
PEAK_BW, PEAK_MFMA = 8.0e12, 2.5e15
sum_floor = 0.0
shown = sorted(rows, reverse=True)
for ms, name, variant, cin, cout, k, by in shown:
    sum_floor += max(by / PEAK_BW, flops[name] * split / PEAK_MFMA) * 1e3
print(f"layer-wise roofline floor (sum over layers of max(bytes / 8 TB/s, MFMA flops / 2.5 PF)): {sum_floor:.3f} ms -> the convs run at {sum_floor / tot_ms:.2f} of it")
for ms, name, variant, cin, cout, k, by in shown[:(200 if "--all" in sys.argv else 40)]:
    t_bw, t_mf = by / PEAK_BW * 1e3, flops[name] * split / PEAK_MFMA * 1e3
    bound = "hbm " if t_bw >= t_mf else "mfma"
    print(f"{name:34s} {variant:22s} {cin:4d}->{cout:4d} k{k} {ms*1e3:8.1f} us {by/1e6:8.1f} MB {by/ms/1e9:6.2f} TB/s  {flops[name]*split/ms/2.5e12*100:5.1f} % of 2.5 PF   bound {bound} {max(t_bw, t_mf) / ms:5.2f} of roof")
