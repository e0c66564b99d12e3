#!/bin/bash
# The round's rocprofv3 evidence for bench.py's default command (run on the GPU box): kernel stats, HBM traffic (two PMC passes),
# SQ utilisation (two PMC passes). Summaries are written by tools/pmc_traffic.py / tools/pmc_util.py into profiles/<tag>_*.
# usage: tools/profile_round.sh <tag>      (results under gpurun_out/<tag>_prof/, copy the summaries from profiles/ back into the repo)
set -o pipefail
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${TAG}_prof
mkdir -p $O
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- $B > $O/bench_under_rocprof.json 2> $O/stats.log && echo stats ok
S="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --conv-totals"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- $S > $O/fetch_bench.json 2> $O/fetch.log && echo fetch ok
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- $S > /dev/null 2> $O/write.log && echo write ok
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/sq1 -o p --output-format csv -- $S > /dev/null 2> $O/sq1.log && echo sq1 ok
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -d $O/sq2 -o p --output-format csv -- $S > /dev/null 2> $O/sq2.log && echo sq2 ok
python3 tools/pmc_traffic.py $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv $TAG $O/fetch_bench.json > $O/traffic.txt 2>&1; tail -5 $O/traffic.txt
python3 tools/pmc_util.py $TAG $O/sq1/p_counter_collection.csv $O/sq2/p_counter_collection.csv > $O/util.txt 2>&1; tail -14 $O/util.txt
cp profiles/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_util.json $O/ 2>/dev/null
cp $O/stats/s_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
rm -rf $O/fetch $O/write $O/sq1 $O/sq2 $O/stats/s_kernel_trace.csv           # raw traces are large; the summaries stay
ls -la $O
