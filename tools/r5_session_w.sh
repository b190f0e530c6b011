export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r5w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8
RMX_W2_INTEG=bdf1 timeout 600 python tools/w2_check.py 64 256 512 2>&1 | tee $O/w2_check11.txt
