# k_part2_wide: chunk split A/B on the 1B random-upper-halves join (whole tiles per part)
set -o pipefail
for v in "1 1" "1 2" "1 4" "1 8" "0 1" "1 2" "1 4"; do
  set -- $v
  GOLAP_DEBUG_ENV=1 GOLAP_P2W=$1 GOLAP_P2W_SPLIT=$2 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-others --no-wide --keys random 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P2W=$1 SPLIT=$2 random keys', d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if v>0.05})" || exit 1
done
