for rep in 1 2; do
for v in "tools/lib_pf0.bin" "-" "tools/lib_pf2.bin"; do
if [ "$v" = "-" ]; then unset SJPEG_AMD_LIB; else export SJPEG_AMD_LIB=$(readlink -f $v); fi
echo "== $v"
SJPEG_HIP_FORCE_COEF_KEEP=1 python tools/histogram_pass_time.py 2>&1 | grep histogram
python tools/histogram_pass_time.py 2>&1 | grep histogram
python tools/profile_workload.py m4 20 2>&1 | grep m4
done; done
