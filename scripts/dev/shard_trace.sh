# host timeline of golap_sharded_join calls at world size 1 (1B rows and a rank-sized 125M-row problem)
for R in 1000000000 125000000; do
  GOLAP_DEBUG_ENV=1 GOLAP_SHARD_TRACE=1 python bench.py --rows $R --sharded --steps 6 --warmup 2 --no-wide --no-others --cpu-sample 0 --no-verify 2>&1 | grep -E "host us|^\{" | tail -4 | cut -c1-400
done
