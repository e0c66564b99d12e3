B=face-detection-with-yolov11-sahi-and-real-esrgan_amd/csrc/build
for v in "" r16_st16 r16_st17 r16_st2; do
  if [ -z "$v" ]; then L=face-detection-with-yolov11-sahi-and-real-esrgan_amd/libffp.so; else L=$B/libffp_$v.so; fi
  echo "== stores: ${v:-default}"
  FFP_LIB=$L timeout -k 10 200 python tools/sr_batch_sweep.py 2>&1 | grep -v amdgpu | grep "crops=  32\|crops= 320" | sed "s/fused.*//"
done
