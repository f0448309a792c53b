# the bench line with workspace draws, four processes on one box; then the default line once
for i in 1 2 3 4; do
  python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-others --no-wide 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run $i', d['ms_per_step'], d['workspace_draws']['ms_per_join'], d['workspace_draws']['kept'], {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if v>0.05})" || exit 1
done
python bench.py > gpurun_out/r04_default_line2.json 2>gpurun_out/r04_default_err2.log || exit 1
python -c "
import json
d=json.loads(open('gpurun_out/r04_default_line2.json').read().strip().splitlines()[-1])
print('default', d['ms_per_step'], d['roofline']['frac'], d['workspace_draws']['ms_per_join'], d['workspace_draws']['kept'])
print('C3', d['other_configs']['C3']['ms_per_step'], d['other_configs']['C3']['alg_roofline_frac'], d['other_configs']['C3'].get('workspace_draws'))
print('wide', d['wide_keys']['random_upper_halves']['ms_per_step'], d['wide_keys']['random_upper_halves']['roofline'])
"
