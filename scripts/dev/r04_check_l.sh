# final executor: operator tests, then the driver's default bench line (wall clock of the whole run beside it)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_operators_gpu.py -x -q -m gpu > gpurun_out/r04_l_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r04_l_tests.log
[ $rc -ne 0 ] && exit $rc
t0=$(date +%s)
timeout -k 10 600 python bench.py > gpurun_out/r04_l_bench.out 2> gpurun_out/r04_l_bench.err; rc=$?
t1=$(date +%s)
echo "default bench: rc=$rc wall=$((t1-t0)) s"
grep '^{' gpurun_out/r04_l_bench.out | tail -1 > gpurun_out/r04_l_default_line.json
python -c "
import json; d=json.load(open('gpurun_out/r04_l_default_line.json'))
print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['workspace_draws']['ms_per_join'])
print({k:(v['ms_per_step'], v.get('alg_roofline_frac')) for k,v in d['other_configs'].items()})
print({k:v['ms_per_step'] for k,v in d['wide_keys'].items() if isinstance(v,dict) and 'ms_per_step' in v}, d['wide_keys']['ms_per_step'])
print(d['cpu_baseline'])
"
exit $rc
