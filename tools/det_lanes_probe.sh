# branches of the detector's launch graph as parallel lanes (FFP_LANES: 0 off, 1 every branch, 2 one per head level, 3 head level 0 only): frame-by-frame order and the headline
for m in 0 1 2 3; do
  FFP_LANES=$m timeout -k 10 300 python bench.py --no-cpu-baseline --secondary-only frame_by_frame --steps 40 > gpurun_out/dl_$m.json 2>/dev/null || { echo "FFP_LANES $m failed"; continue; }
  python3 -c "
import json; d=json.load(open('gpurun_out/dl_$m.json')); f=d['secondary']['frame_by_frame']; print('FFP_LANES', $m, 'headline', d['value'], 'frame_by_frame', f['value'], f['latency_ms_rank0'], f['host_stage_ms_per_step_rank0'])"
done
