SPECS=("fwd 64 150 64 128 3 1 SSD_CONV_TILE 0" "fwd 64 150 128 128 3 1 SSD_CONV_TILE 0" "fwd 64 75 128 256 3 1 SSD_CONV_TILE 0" "dgrad 64 150 128 128 3 1 SSD_CONV_TILE 0" "dgrad 64 150 128 64 3 1 SSD_CONV_TILE 0" "head 64 38 512 340 3 1 SSD_CONV_TILE 0")
for rep in 1 2; do
for lib in libssd_hip_base.so libssd_hip_new.so; do
  echo "== $lib"
  AB_LIB=$lib python tools_dev/ab_conv_multi.py "${SPECS[@]}"
done
done
python tools_dev/time_loss.py; python tools_dev/time_detect.py 2>&1 | grep -v "stop after"
timeout -k 10 300 python -m pytest tests/test_loss_gpu.py tests/test_detect_gpu.py tests/test_conv_gpu.py -x -q 2>&1 | tail -3
