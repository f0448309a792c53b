#!/bin/bash
# rocprofv3 evidence for every bench configuration at the current sources (run on the GPU box from the repo root):
#   kernel trace + FETCH_SIZE + WRITE_SIZE + SQ passes per configuration (scripts/prof_bench.sh), the unprofiled bench line,
#   profiles/pmc_traffic.json rebuilt for the sources' hash.  Small files only are kept: gpurun_out/profiles_<round>/<tag>/.
# usage: bash scripts/prof_all.sh r02   (then copy gpurun_out/profiles_r02/* to profiles/ and profiles/pmc_traffic.json comes back via gpurun_out)
set -o pipefail
R=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
KEEP=$ROOT/gpurun_out/profiles_$R
mkdir -p $KEEP
rm -f $ROOT/profiles/pmc_traffic.json
run() {   # tag, merge size, bench args...
  local tag=$1 size=$2; shift 2
  MERGE="--merge $size" bash $ROOT/scripts/prof_bench.sh ${R}_$tag "$@" --cpu-sample 0 > $ROOT/gpurun_out/prof_${R}_$tag.log 2>&1 || { echo "profile $tag failed"; tail -5 $ROOT/gpurun_out/prof_${R}_$tag.log; return 1; }
  local out=$ROOT/gpurun_out/prof_${R}_$tag
  mkdir -p $KEEP/$tag
  cp $out/summary.txt $out/summary.json $out/traffic.json $KEEP/$tag/ 2>/dev/null
  cp $(find $out/trace -name '*kernel_stats.csv' | head -1) $KEEP/$tag/kernel_stats.csv 2>/dev/null
  grep '^{' $out/trace.log | tail -1 > $KEEP/$tag/bench_line_under_trace.json
  (cd $ROOT && python3 bench.py "$@" --cpu-sample 0 2>/dev/null | grep '^{' | tail -1 > $KEEP/$tag/bench_line_unprofiled.json)
  echo "== $tag"; tail -12 $out/summary.txt
}
run bench_1b 1000000000x1 --steps 3 --warmup 1 --no-wide --no-others || exit 1
run ops_C4 1000000000x1 --config C4 --steps 3 --warmup 1
run ops_C4L 1000000000x1 --config C4L --steps 3 --warmup 1
run ops_C1 1000000x1 --config C1 --steps 20 --warmup 2
run bench_C3 100000000x1 --config C3 --steps 3 --warmup 1 --no-wide
run bench_1b_wide 1000000000x1 --steps 3 --warmup 1 --no-wide --no-others --keys random
run ops_OP 100000000x1 --config OP --steps 2 --warmup 1
# the partition-exchange path at world size 1 (stage timestamps in the line): unprofiled line only
(cd $ROOT && python3 bench.py --sharded --gpus 1 --steps 5 --cpu-sample 0 2>/dev/null | grep '^{' | tail -1 > $KEEP/sharded_1gpu_bench_line.json)
cp $ROOT/profiles/pmc_traffic.json $KEEP/pmc_traffic.json
echo done
