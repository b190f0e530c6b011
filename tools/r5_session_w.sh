export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r5w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 | tee $O/tests_full.txt
