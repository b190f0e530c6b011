export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r5d; mkdir -p $O
timeout 300 python tools/coop_check.py 1024 100 24 > $O/coop_check.txt 2>&1
cat $O/coop_check.txt
timeout 900 python -m pytest -q -x tests/test_gpu_contact.py tests/test_gpu_straggler_policy.py -s > $O/pytest_contact.log 2>&1; echo "rc=$?" >> $O/pytest_contact.log
tail -25 $O/pytest_contact.log
