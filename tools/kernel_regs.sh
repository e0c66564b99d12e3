#!/bin/bash
# usage: tools/kernel_regs.sh <file.hip> : per-kernel VGPR / SGPR / spill / scratch summary from hipcc's resource-usage remarks
cd "$(dirname "$0")/../face-detection-with-yolov11-sahi-and-real-esrgan_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off $FFP_EXTRA_FLAGS -c "$1" -o /tmp/_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        cur={'name':subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()[:110]};rows.append(cur);continue
    for k in ('TotalSGPRs','VGPRs','AGPRs','ScratchSize [bytes/lane]','Occupancy [waves/SIMD]','SGPRs Spill','VGPRs Spill'):
        m=re.search(re.escape(k)+r': (\d+)',l)
        if m and cur is not None and k not in cur: cur[k]=m.group(1)
for r in rows:
    print(r['name'],'| vgpr',r.get('VGPRs'),'agpr',r.get('AGPRs'),'sgpr',r.get('TotalSGPRs'),'scratch',r.get('ScratchSize [bytes/lane]'),'occ',r.get('Occupancy [waves/SIMD]'),'spill s/v',r.get('SGPRs Spill'),r.get('VGPRs Spill'))
"
