# host layer: join strategy by build-side size, no pair sort under a GROUP BY -- operator tests, OP line + breakdown, operator fuzzer
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_operators_gpu.py tests/test_c_abi_demo.py -x -q -m gpu > gpurun_out/r04_k_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r04_k_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python scripts/dev/op_breakdown.py > gpurun_out/r04_k_op_breakdown.log 2>&1 || { tail -5 gpurun_out/r04_k_op_breakdown.log; exit 1; }
cat gpurun_out/r04_k_op_breakdown.log | tail -16
for i in 1 2; do
  python bench.py --config OP --steps 4 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OP', d['ms_per_step'], 'h2d', d['h2d_ms'], d['h2d_GBps'], 'rest', d['device_and_host_ms'], 'kwait', d['kernel_wait_ms'])" | tee -a gpurun_out/r04_k_op.log || exit 1
done
python bench.py --config OP --steps 4 --warmup 2 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r04_k_op_line.json
timeout -k 10 200 python scripts/fuzz_operators.py 90 71 > gpurun_out/r04_k_fuzz_ops.log 2>&1; rc=$?
tail -2 gpurun_out/r04_k_fuzz_ops.log
exit $rc
