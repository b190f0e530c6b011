export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5o; mkdir -p $O
timeout 600 python tools/pair_bench.py > $O/pair_bench.txt 2>&1; cat $O/pair_bench.txt
timeout 900 python -m pytest -q -x tests/test_gpu_contact.py tests/test_gpu_straggler_policy.py tests/test_gpu_full_size.py tests/test_gpu_multi_device.py tests/test_mex_gateway.py tests/test_gpu_reference_tol.py::test_chain32_newton_counts_vs_literal_oracle_at_reference_tol tests/test_gpu_bench_contract.py::test_rccl_group_of_two_ranks_on_one_gpu_is_refused_as_expected tests/test_gpu_big_trees.py::test_compute_values_on_a_big_tree tests/test_gpu_parity.py::test_single_step_matches_oracle -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
grep -n "^tol\|rollout .*iterations gpu\|passed\|failed\|rc=" $O/pytest.log | tail -20
