python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
for rep in 1 2; do
echo "fused big: $(python tools/profile_workload.py c3x1 30 2>&1 | grep c3x1:)"
echo "no fused big: $(SJPEG_HIP_NO_FUSED_BIG=1 python tools/profile_workload.py c3x1 30 2>&1 | grep c3x1:)"
done
