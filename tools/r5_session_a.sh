export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5a; mkdir -p $O
timeout 300 python tools/coop_check.py 1024 100 24 > $O/coop_check.txt 2>&1
timeout 300 python tools/quick_bench.py > $O/quick_bench.txt 2>&1
timeout 600 python bench.py --workload ground > $O/bench_ground.json 2> $O/bench_ground.err
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 1500 python -m pytest tests -m gpu -q --durations=60 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log; cat $O/coop_check.txt $O/quick_bench.txt
