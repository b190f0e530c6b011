export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -5
rm -rf gpurun_out/r5u; PMC_WORKLOADS=tree64 bash tools/gpu_session.sh r5u pmc
O=$PWD/gpurun_out/r5u
timeout 600 python tools/w2c_check.py 128 256 512 > $O/w2c_check.txt 2>&1
timeout 600 python bench.py --batch 512 --no-cpu-baseline --no-side-legs > $O/bench_chain_b512.json 2> $O/bench_chain_b512.err
RMX_W2_MAX=0 timeout 600 python bench.py --batch 512 --no-cpu-baseline --no-side-legs > $O/bench_chain_b512_one_wave.json 2>> $O/bench_chain_b512.err
timeout 600 python bench.py --workload tree64 > $O/bench_tree64.json 2> $O/bench_tree64.err
