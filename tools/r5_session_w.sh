export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r5w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q -k two_wave 2>&1 | tail -15 | tee $O/tests_full.txt
