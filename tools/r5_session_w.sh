export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rm -rf gpurun_out/r5v; PMC_WORKLOADS=tree64 bash tools/gpu_session.sh r5v pmc
O=$PWD/gpurun_out/r5v
timeout 600 python tools/w2_check.py 64 256 512 > $O/w2_check.txt 2>&1
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tree64 -- python $GRAFT_REPO_ROOT/bench.py --workload tree64 --no-cpu-baseline --no-side-legs --repeats 0 > /dev/null 2>&1
find $O -name "*.db" -delete
