export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r5w
python tools/adjoint_sizes.py 2>&1 | tee gpurun_out/r5w/adjoint_sizes.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k adjoint 2>&1 | grep -v amdgpu.ids | tail -4
