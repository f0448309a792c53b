# join tests + C3 / C5 bench lines (round-4 launch trimming check)
timeout -k 10 900 python -m pytest tests/test_join_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/r04_t3.log 2>&1; rc=$?
tail -5 gpurun_out/r04_t3.log
[ $rc -eq 0 ] || exit $rc
python bench.py --config C3 --steps 20 --warmup 3 --cpu-sample 0 > gpurun_out/r04_c3_a.json 2>/dev/null || exit 1
python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-others > gpurun_out/r04_c5_a.json 2>/dev/null || exit 1
python - <<PY
import json
for f in ("r04_c3_a","r04_c5_a"):
    d=json.load(open("gpurun_out/%s.json"%f)); print(f, d["ms_per_step"], {k:v for k,v in d["kernels_ms_per_step"].items() if v>0.02}, d.get("wide_keys",{}).get("random_upper_halves",{}).get("ms_per_step"))
PY
