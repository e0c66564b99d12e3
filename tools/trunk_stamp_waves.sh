B=face-detection-with-yolov11-sahi-and-real-esrgan_amd/csrc/build
for w in w5 w11; do
  echo "== loader wave $w" >> gpurun_out/r4_trunk_stamps_loaders.txt
  FFP_TRUNK_DUMP=1 FFP_LIB=$B/libffp_trunk_dbg_$w.so timeout -k 10 120 python tools/trunk_stamp_probe.py >> gpurun_out/r4_trunk_stamps_loaders.txt 2>&1 || exit 1
done
