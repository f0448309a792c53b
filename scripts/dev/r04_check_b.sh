# K16 join: A/B at 1B and 300M rows, then the join tests
python scripts/ab_env.py 1000000000 GOLAP_JOIN_K16 0 1 > gpurun_out/r04_ab_k16.log 2>&1 || exit 1
python scripts/ab_env.py 300000000 GOLAP_JOIN_K16 0 1 >> gpurun_out/r04_ab_k16.log 2>&1 || exit 1
grep -v amdgpu.ids gpurun_out/r04_ab_k16.log
timeout -k 10 900 python -m pytest tests/test_join_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/r04_t6.log 2>&1; rc=$?
tail -5 gpurun_out/r04_t6.log
exit $rc
