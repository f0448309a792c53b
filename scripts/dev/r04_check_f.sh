# k_part2_wide: forms test, the join tests, A/B on the 1B random-upper-halves join
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_join_gpu.py -x -q -m gpu -k "wide_last_pass or last_pass_and_its_overflow or unaudited" > gpurun_out/r04_f_t1.log 2>&1; rc=$?
tail -5 gpurun_out/r04_f_t1.log
[ $rc -ne 0 ] && exit $rc
for v in "0 1" "1 1" "1 2" "1 4" "2 1" "1 1" "0 1"; do
  set -- $v
  GOLAP_DEBUG_ENV=1 GOLAP_P2W=$1 GOLAP_P2W_SPLIT=$2 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-others --no-wide --keys random 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P2W=$1 SPLIT=$2 random keys', d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if v>0.05})" || exit 1
done
