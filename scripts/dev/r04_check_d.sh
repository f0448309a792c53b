# key sample size: C3 / C5 end to end with 8 vs 2 samples per thread and side, then the tests that depend on what the sample sees
export GOLAP_DEBUG_ENV=1
for v in 8 2 8 2; do
  GOLAP_DICT_SAMPLES=$v python bench.py --config C3 --steps 30 --warmup 5 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 samples $v:', d['ms_per_step'])"
done
for v in 8 2; do
  GOLAP_DICT_SAMPLES=$v python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-others 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C5 samples $v:', d['ms_per_step'], {k: (v['ms_per_step'] if isinstance(v, dict) and 'ms_per_step' in v else None) for k, v in d['wide_keys'].items() if isinstance(v, dict)}, d['wide_keys']['ms_per_step'])"
done
unset GOLAP_DEBUG_ENV
timeout -k 10 900 python -m pytest tests/test_join_gpu.py -x -q -m gpu -k "narrow or wide or dict or hidden or window or upper" > gpurun_out/r04_t8.log 2>&1; rc=$?
tail -5 gpurun_out/r04_t8.log
exit $rc
