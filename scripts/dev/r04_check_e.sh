# light profiling: bench lines (C3, default) and the join tests (full timing, profiling = 1, unchanged)
python bench.py --config C3 --steps 30 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3', d['ms_per_step'], d['alg_roofline_frac'], {k:v for k,v in d['kernels_ms_per_step'].items() if v>0.01})"
python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-others 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], {k:v for k,v in d['kernels_ms_per_step'].items() if v>0.01}); w=d['wide_keys']; print('wide', w['ms_per_step'], w['two_upper_halves']['ms_per_step'], w['random_upper_halves']['ms_per_step'], w['random_upper_halves']['roofline'], w['random_upper_halves']['kernels_ms_per_step'])"
python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-others --no-wide --keys random 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 random keys', d['ms_per_step'], d['roofline'], d['kernels_ms_per_step'])"
timeout -k 10 900 python -m pytest tests/test_join_gpu.py -x -q -m gpu > gpurun_out/r04_t9.log 2>&1; rc=$?
tail -4 gpurun_out/r04_t9.log
exit $rc
