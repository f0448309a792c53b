# is the 3 % spread of the first pass a property of the box or of the process (where its buffers landed)?  the same bench line six times on one box
for i in 1 2 3 4 5 6; do
  python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-others --no-wide 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('run $i', d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms_per_step'].items() if v>0.05})" || exit 1
done
