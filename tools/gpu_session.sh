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
  find $OUT/prof -name "*.db" -delete 2>/dev/null
  cd $ROOT
fi
if has pmc; then
  # counter passes for every bench workload (tools/roofline_from_pmc.py reads them): separate rocprofv3 runs, --kernel-trace only
  cd /tmp
  P() { wl=$1; name=$2; shift 2; timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_${wl}_$name -- $BENCH $EXTRA > $OUT/pmc_${wl}_$name.json 2> $OUT/pmc_${wl}_$name.err; echo "pmc $wl $name rc=$?"; find $OUT/pmc_${wl}_$name -name "*.db" -delete 2>/dev/null; }
  F64="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  SQ="SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES"
  COMMON="--no-cpu-baseline --no-side-legs --repeats 0"
  for wl in ${PMC_WORKLOADS:-chain tree64 tree64x ground adjoint chain128}; do
    case $wl in
      chain)    BENCH="python $ROOT/bench.py --steps 100 --warmup 10 $COMMON";;
      tree64)   BENCH="python $ROOT/bench.py --workload tree64 $COMMON";;
      tree64x)  BENCH="python $ROOT/bench.py --workload tree64 --batch 1024 $COMMON";;
      ground)   BENCH="python $ROOT/bench.py --workload ground $COMMON";;
      adjoint)  BENCH="python $ROOT/bench.py --workload adjoint $COMMON";;
      chain128) BENCH="python $ROOT/bench.py --workload chain --links 128 --batch 256 --steps 10 --warmup 2 --burn-in 0 $COMMON";;
    esac
    EXTRA=""
    P $wl f64 $F64
    P $wl sq $SQ
    P $wl fetch FETCH_SIZE
    P $wl write WRITE_SIZE
    case $wl in chain|tree64|tree64x) EXTRA="--tol 1e-3"; P $wl f64_tol3 $F64; EXTRA="";; esac   # another iterations-per-step mix: the two-parameter model
  done
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
