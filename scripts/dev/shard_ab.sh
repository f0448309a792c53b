set -o pipefail
for K in 1 4 2 1 4; do
  GOLAP_DEBUG_ENV=1 GOLAP_SHARD_CHUNKS=$K python bench.py --sharded --steps 10 --warmup 2 --no-wide --no-others --cpu-sample 0 --no-verify 2>/dev/null | grep '^{' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); sj=d['sharded_join']; print('K=$K', d['ms_per_step'], {k:round(v[1]-v[0],3) for k,v in sj['stage_ms'].items()}, d.get('kernels_ms_per_step'))"
done
python bench.py --steps 10 --warmup 2 --no-wide --no-others --cpu-sample 0 --no-verify 2>/dev/null | grep '^{' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'], d.get('kernels_ms_per_step'))"
