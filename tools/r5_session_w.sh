export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
PMC_WORKLOADS=tree64 bash tools/gpu_session.sh r5y pmc
O=$PWD/gpurun_out/r5y
timeout 600 python bench.py --workload tree64 > $O/bench_tree64.json 2> $O/bench_tree64.err
timeout 600 python bench.py --workload tree64 --batch 256 --no-cpu-baseline --no-side-legs > $O/bench_tree64_b256.json 2>> $O/bench_tree64.err
timeout 600 python bench.py --workload tree64 --batch 64 --no-cpu-baseline --no-side-legs > $O/bench_tree64_b64.json 2>> $O/bench_tree64.err
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tree64 -- python $GRAFT_REPO_ROOT/bench.py --workload tree64 --no-cpu-baseline --no-side-legs --repeats 0 > /dev/null 2>&1
find $O -name "*.db" -delete
