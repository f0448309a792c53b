# after k_part2_wide: the whole GPU suite, then the join fuzzers (wide inputs included)
set -o pipefail
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r04_h_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r04_h_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/fuzz_join_big.py 120 41 > gpurun_out/r04_h_fuzz_big.log 2>&1; rc=$?
tail -3 gpurun_out/r04_h_fuzz_big.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/fuzz_join.py 120 42 > gpurun_out/r04_h_fuzz.log 2>&1; rc=$?
tail -3 gpurun_out/r04_h_fuzz.log
exit $rc
