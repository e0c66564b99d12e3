# spatial split of the card between the enhancer's and the detector's streams (hipExtStreamCreateWithCUMask): CUs lo-hi of every XCD
run() { echo "$1"; env $1 timeout -k 10 200 python tools/overlap_probe.py 2>&1 | grep -v amdgpu; }
run "FFP_NONE=1"
run "FFP_SR_CU_MASK=0-12 FFP_DET_CU_MASK=12-32"
run "FFP_SR_CU_MASK=0-12 FFP_DET_CU_MASK=12-32 FFP_CU_MASK_ORDER=1"
run "FFP_SR_CU_MASK=0-16 FFP_DET_CU_MASK=16-32"
run "FFP_SR_CU_MASK=0-16"
run "FFP_SR_CU_MASK=0-32 FFP_DET_CU_MASK=0-32"
