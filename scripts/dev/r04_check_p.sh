# the round's last check: the whole GPU suite (472 tests), then the join fuzzers at the final sources
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r04_p_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r04_p_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python scripts/fuzz_join_big.py 90 91 > gpurun_out/r04_p_fuzz_big.log 2>&1; rc=$?
tail -2 gpurun_out/r04_p_fuzz_big.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python scripts/fuzz_join.py 90 92 > gpurun_out/r04_p_fuzz.log 2>&1; rc=$?
tail -2 gpurun_out/r04_p_fuzz.log
exit $rc
