mkdir -p gpurun_out/r04w; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_big_trees.py tests/test_gpu_bench_contract.py -m gpu -q > gpurun_out/r04w/pytest.log 2>&1; tail -5 gpurun_out/r04w/pytest.log
timeout 600 python tools/big_tree_bench.py > gpurun_out/r04w/big_tree_bench.txt 2>&1; cat gpurun_out/r04w/big_tree_bench.txt
PMC_WORKLOADS=chain128 bash tools/gpu_session.sh r04w pmc
