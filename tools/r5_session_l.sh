export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$PWD/gpurun_out/r5l; mkdir -p $O
cd /tmp
for mode in 0 1 2; do
  RMX_GROUND_FUSED=$mode timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f$mode -- python $GRAFT_REPO_ROOT/tools/ground_once.py > $O/run_f$mode.txt 2>&1
  find $O/trace_f$mode -name "*.db" -delete
  f=$(find $O/trace_f$mode -name "*kernel_stats.csv" | head -1)
  echo "== fused $mode"; cat $O/run_f$mode.txt | tail -3; head -8 $f
done
