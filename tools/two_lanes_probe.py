"""Pipeline lanes in ONE process: L detector + enhancer handle pairs driven by L host threads, each on its share of the frames —
does the card finish K frames sooner than with one lane? (2 ranks sharing one GPU do: profiles/r03_bench_2ranks_one_gpu_gloo_rehearsal.json.)
usage: two_lanes_probe.py [bench.py flags]; env LANES="1,2,3,4" KS="20,60" """
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
import ffp_amd  # noqa
from ffp_amd import synth

sys.argv = ["bench.py", "--no-cpu-baseline", "--no-secondary"] + sys.argv[1:]
args = bench.parse()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ctx = {"rank": 0, "world": 1, "local_rank": 0, "dev": dev, "backend": "nccl", "det_w": synth.yolo11_pose_weights(args.arch), "sr_w": synth.rrdbnet_weights(4, 23),
       "host_frames": [synth.synthetic_frame(args.height, args.width, seed=i) for i in range(2)]}
LANES = [int(x) for x in os.environ.get("LANES", "1,2,3,4").split(",")]
KS = [int(x) for x in os.environ.get("KS", "20,60").split(",")]
lanes = [bench.Runner(args, ctx) for _ in range(max(LANES))]
for K in KS:
    row = []
    for L in LANES:
        share = [K // L + (1 if i < K % L else 0) for i in range(L)]
        for r, n in zip(lanes[:L], share):
            r.setup(args.warmup, n)
            r.loop(args.warmup)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(2):
            th = [threading.Thread(target=r.loop, args=(n,)) for r, n in zip(lanes[:L], share)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        row.append(f"{L} lane(s) {K / best:6.1f}")
    print(f"K={K} DB={args.det_batch_frames} SB={args.sr_batch_frames}: " + " | ".join(row) + "  frames/s", flush=True)
