for rep in 1 2; do
echo "m0 one launch  $(python tools/profile_workload.py c5m0b32 30 2>&1 | grep c5m0b32:)"
echo "m0 two lanes   $(SJPEG_HIP_BATCH_LANES_M0=1 python tools/profile_workload.py c5m0b32 30 2>&1 | grep c5m0b32:)"
done
