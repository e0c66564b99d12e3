#!/bin/bash
# SQ counters of the fused body launch (conv_trunk_kernel, opt-in) on the headline's SR batch alone, beside conv_rows16's on the same batch: two PMC passes each
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_trunk_prof; mkdir -p $O
for mode in 1 0; do
  export FFP_TRUNK=$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/sq1_$mode -o p --output-format csv -- python3 tools/sr_only_trace.py > $O/events_$mode.json 2> $O/sq1_$mode.log || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -d $O/sq2_$mode -o p --output-format csv -- python3 tools/sr_only_trace.py > /dev/null 2> $O/sq2_$mode.log || exit 1
  python3 tools/pmc_util.py r04_sr_alone_trunk$mode $O/sq1_$mode/p_counter_collection.csv $O/sq2_$mode/p_counter_collection.csv > $O/util_$mode.txt 2>&1
  tail -4 $O/util_$mode.txt
  cp profiles/r04_sr_alone_trunk${mode}_pmc_util.json $O/ 2>/dev/null
  rm -rf $O/sq1_$mode $O/sq2_$mode
done
