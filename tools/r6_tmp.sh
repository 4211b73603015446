for sl in 384 512 640 768 1024; do for n in 16 32; do
echo "slots $sl n $n  $(SJPEG_HIP_HISTO_SLOTS=$sl python tools/profile_workload.py m4n$n 30 2>&1 | grep "m4n$n:")"
done; done
