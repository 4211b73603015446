python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
bash tools/lib_multi_ab.sh 3 tools/lib_k3old.bin -
