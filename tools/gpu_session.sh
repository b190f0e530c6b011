#!/bin/bash
# One GPU-box session: tests, bench lines, rocprofv3 kernel trace and PMC passes.  Usage: tools/gpu_session.sh <tag> [what...]
# what: tests bench bench2 prof pmc extra   (default: all).  Everything lands in gpurun_out/<tag>/.
TAG=${1:-r02a}; shift
WHAT=${*:-tests bench bench2 prof pmc extra}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has tests; then
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
if has bench; then
  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
  timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2>> $OUT/bench.err
  head -c 600 $OUT/bench.json; echo
fi
if has bench2; then
  timeout 900 python bench.py --gpus 2 --no-side-legs > $OUT/bench_gpus2.json 2> $OUT/bench_gpus2.err; echo "bench --gpus 2 rc=$?"
  tail -3 $OUT/bench_gpus2.err
fi
BENCH="python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-side-legs --repeats 0"
if has prof; then
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- $BENCH > $OUT/prof_bench.json 2> $OUT/prof.err; echo "prof rc=$?"
  cd $ROOT
fi
if has pmc; then
  cd /tmp
  rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*\|SQ_[A-Z_]*F64[A-Z_]*" | sort -u > $OUT/counters_valu.txt
  P() { name=$1; shift; timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- $BENCH $EXTRA > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err; echo "pmc $name rc=$?"; }
  EXTRA=""
  P f64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  P sq SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_F64 SQ_WAVES
  P fetch FETCH_SIZE
  P write WRITE_SIZE
  EXTRA="--tol 1e-3"       # another iterations-per-step mix for the two-parameter calibration
  P f64_tol3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  EXTRA=""
  BENCH="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs --repeats 0"
  P fetch_k20 FETCH_SIZE
  P write_k20 WRITE_SIZE
  cd $ROOT
fi
if has extra; then
  timeout 600 python bench.py --workload adjoint > $OUT/bench_adjoint.json 2> $OUT/bench_adjoint.err; echo "adjoint rc=$?"
  timeout 600 python bench.py --workload tree64 > $OUT/bench_tree64.json 2> $OUT/bench_tree64.err; echo "tree64 rc=$?"
  timeout 600 python bench.py --workload ground > $OUT/bench_ground.json 2> $OUT/bench_ground.err; echo "ground rc=$?"
  timeout 300 python tools/phase_profile.py > $OUT/phase_profile.txt 2>&1
  timeout 300 python tools/quick_bench.py > $OUT/quick_bench.txt 2>&1; cat $OUT/quick_bench.txt
fi
# keep the merge-back small: drop rocprof's big raw files
find $OUT -name "*.db" -size +8M -delete 2>/dev/null
du -sh $OUT
