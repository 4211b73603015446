#!/bin/bash
# round 6: one iteration on the histogram kind -- its tests, its pass time (16 x 4K, coefficients kept), its stamps, the
# default-parameter batch call.   gpurun -- 'bash tools/r6_hist_iter.sh TAG'
TAG=${1:-x}
O=gpurun_out/r6_$TAG.txt
{
python -m pytest tests -m gpu -x -q -k "histogram or adaptive or methods or default_parameters or c5" 2>&1 | tail -3
SJPEG_HIP_FORCE_COEF_KEEP=1 python tools/histogram_pass_time.py
python tools/histogram_pass_time.py
python tools/hist_stamps.py
python tools/profile_workload.py m4 20
python tools/profile_workload.py c5m4x32 20
} > $O 2>&1
cat $O
