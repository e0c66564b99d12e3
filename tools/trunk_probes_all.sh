set -e
B=face-detection-with-yolov11-sahi-and-real-esrgan_amd/csrc/build
FFP_TRUNK_DUMP=1 FFP_LIB=$B/libffp_trunk_dbg.so timeout -k 10 120 python tools/trunk_stamp_probe.py > gpurun_out/r4_trunk_stamps.txt 2>&1
for v in "" _trunk_skip2 _trunk_skip4 _trunk_skip5 _trunk_skip23; do
  if [ -z "$v" ]; then L=face-detection-with-yolov11-sahi-and-real-esrgan_amd/libffp.so; else L=$B/libffp$v.so; fi
  echo "== $v" >> gpurun_out/r4_trunk_phase.txt
  FFP_LIB=$L timeout -k 10 120 python tools/trunk_phase_probe.py >> gpurun_out/r4_trunk_phase.txt 2>&1
done
