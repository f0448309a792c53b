# pipelined GpuFilter (host layer only): the operator tests, the OP line with and without pipelining (interleaved), the operator fuzzer
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_operators_gpu.py tests/test_ingest_gpu.py -x -q -m gpu > gpurun_out/r04_j_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r04_j_tests.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  for f in "" "--no-pipelining"; do
    python bench.py --config OP --steps 4 --warmup 2 --cpu-sample 0 $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('OP pipelined' if d['pipelined_filter'] else 'OP one-phase', d['ms_per_step'], 'h2d', d['h2d_ms'], d['h2d_GBps'], 'rest', d['device_and_host_ms'], 'kwait', d['kernel_wait_ms'], 'd2h', d['d2h_bytes'])" | tee -a gpurun_out/r04_j_op_ab.log || exit 1
  done
done
python bench.py --config OP --steps 4 --warmup 2 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r04_j_op_line.json
timeout -k 10 200 python scripts/fuzz_operators.py 60 61 > gpurun_out/r04_j_fuzz_ops.log 2>&1; rc=$?
tail -2 gpurun_out/r04_j_fuzz_ops.log
exit $rc
