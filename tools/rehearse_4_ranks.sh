#!/bin/bash
# Four ranks on ONE card over gloo (no RCCL peers on a one-GPU box): the N > 1 code path of bench.py with every secondary row, and the memory
# bound of VERDICT r3 item 7 — each handle keeps at most FFP_DET_PLAN_GIB of detector plans (least recently used first out).
set -o pipefail
export FFP_BENCH_BACKEND=gloo FFP_BENCH_ONE_DEVICE=1 FFP_DET_PLAN_GIB=${FFP_DET_PLAN_GIB:-12} HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 4 --steps 10 --warmup 5 --lanes 1 --secondary-steps 10 > gpurun_out/r4_rehearse_4_ranks.json 2> gpurun_out/r4_rehearse_4_ranks.err
echo rc=$?
