export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5t; mkdir -p $O
timeout 300 python tools/pair_bench.py > $O/pair_bench.txt 2>&1; cat $O/pair_bench.txt
timeout 300 python tools/coop_stress.py 30
timeout 600 python bench.py --workload ground > $O/bench_ground.json 2> $O/bench_ground.err; python -c "
import json;d=json.loads(open('$O/bench_ground.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['config'].get('not_converged_trajectories'))"
timeout 900 python -m pytest -q tests/test_gpu_contact.py tests/test_gpu_straggler_policy.py tests/test_gpu_full_size.py tests/test_gpu_multi_device.py tests/test_gpu_fuzz.py tests/test_gpu_soak.py -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
