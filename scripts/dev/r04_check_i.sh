# final sources (sha a6c08efd): the join over the whole size range (scripts/size_curve.py), then the fuzzers whose paths the last
# full-suite run does not randomise (aggregate / filter, operator layer, sharded join at world 2)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 420 python scripts/size_curve.py > gpurun_out/r04_size_curve.jsonl 2> gpurun_out/r04_size_curve.err; rc=$?
cat gpurun_out/r04_size_curve.jsonl | cut -c1-230
[ $rc -ne 0 ] && { tail -5 gpurun_out/r04_size_curve.err; exit $rc; }
timeout -k 10 200 python scripts/size_curve.py --keys random 100000000 400000000 1000000000 > gpurun_out/r04_size_curve_wide.jsonl 2>> gpurun_out/r04_size_curve.err; rc=$?
cat gpurun_out/r04_size_curve_wide.jsonl | cut -c1-230
[ $rc -ne 0 ] && { tail -5 gpurun_out/r04_size_curve.err; exit $rc; }
timeout -k 10 200 python scripts/fuzz_agg.py 75 51 > gpurun_out/r04_i_fuzz_agg.log 2>&1; rc=$?
tail -2 gpurun_out/r04_i_fuzz_agg.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python scripts/fuzz_operators.py 75 52 > gpurun_out/r04_i_fuzz_ops.log 2>&1; rc=$?
tail -2 gpurun_out/r04_i_fuzz_ops.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 240 python scripts/fuzz_sharded.py 75 53 2 > gpurun_out/r04_i_fuzz_sharded.log 2>&1; rc=$?
tail -2 gpurun_out/r04_i_fuzz_sharded.log
exit $rc
