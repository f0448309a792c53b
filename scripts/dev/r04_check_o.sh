# the OP profile at the final executor (join strategy by build-side size): trace + FETCH / WRITE + SQ passes, NOT merged into pmc_traffic.json
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash $ROOT/scripts/prof_bench.sh r04o_ops_OP --config OP --steps 2 --warmup 1 --cpu-sample 0 > $ROOT/gpurun_out/prof_r04o_ops_OP.log 2>&1 || { tail -5 $ROOT/gpurun_out/prof_r04o_ops_OP.log; exit 1; }
out=$ROOT/gpurun_out/prof_r04o_ops_OP
KEEP=$ROOT/gpurun_out/profiles_r04o/ops_OP
mkdir -p $KEEP
cp $out/summary.txt $out/summary.json $out/traffic.json $KEEP/ 2>/dev/null
cp $(find $out/trace -name '*kernel_stats.csv' | head -1) $KEEP/kernel_stats.csv 2>/dev/null
grep '^{' $out/trace.log | tail -1 > $KEEP/bench_line_under_trace.json
(cd $ROOT && python3 bench.py --config OP --steps 4 --warmup 2 --cpu-sample 0 2>/dev/null | grep '^{' | tail -1 > $KEEP/bench_line_unprofiled.json)
head -12 $out/summary.txt | cut -c1-180
python3 -c "import json; d=json.load(open('$KEEP/bench_line_unprofiled.json')); print('OP', d['ms_per_step'], d['h2d_ms'], d['h2d_GBps'], d['device_and_host_ms'])"
