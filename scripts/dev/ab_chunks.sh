set -e
export GOLAP_DEBUG_ENV=1
for r in "4:8" "1:8" "1:8:200" "1:2" "2:4"; do
  echo "== C3 rounds $r"
  GOLAP_CHUNK_ROUNDS=$r python bench.py --config C3 --steps 20 --warmup 3 --cpu-sample 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if v>0.05}, d['config']['plan'].get('chunks1'))"
done
for r in "4:8" "1:8" "2:4"; do
  echo "== C5 rounds $r"
  GOLAP_CHUNK_ROUNDS=$r python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-wide --no-others | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if v>0.05}, d['config']['plan'].get('chunks1'))"
done
for n in 16000000 32000000 60000000 250000000; do
 for r in "4:8" "1:8"; do
  echo "== rows $n rounds $r"
  GOLAP_CHUNK_ROUNDS=$r python bench.py --config C3 --rows $n --steps 20 --warmup 3 --cpu-sample 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v for k,v in d['kernels_ms_per_step'].items() if v>0.05})"
 done
done
