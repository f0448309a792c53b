# packed-row join: A/B over sizes (0 = 32-bit table, 1 = packed rows forced), then the join tests
: > gpurun_out/r04_ab_k16c.log
for n in 100000000 300000000 600000000 1000000000 32000000; do
  python scripts/ab_env.py $n GOLAP_JOIN_K16 0 1 >> gpurun_out/r04_ab_k16c.log 2>&1 || exit 1
done
grep -v amdgpu.ids gpurun_out/r04_ab_k16c.log
timeout -k 10 900 python -m pytest tests/test_join_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/r04_t7.log 2>&1; rc=$?
tail -5 gpurun_out/r04_t7.log
exit $rc
