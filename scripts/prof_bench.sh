#!/bin/bash
# rocprofv3 evidence for bench.py: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in their own passes.
# usage (on the GPU box, from the repo root): [MERGE="--merge ROWSxGPUS [--reset]"] bash scripts/prof_bench.sh <tag> [bench args...]
# (MERGE: also write the traffic into profiles/pmc_traffic.json, stamped with the kernel sources' hash)
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py "$@" > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py "$@" > $OUT/pmc_fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py "$@" > $OUT/pmc_write.log 2>&1 || { echo "write pass failed"; tail -5 $OUT/pmc_write.log; exit 1; }
# SQ pass (own run, 8 SQ counters fit one pass): where the waves' cycles go and how busy the LDS is
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/bench.py "$@" > $OUT/pmc_sq.log 2>&1 || { echo "sq pass failed"; tail -5 $OUT/pmc_sq.log; }
cd $ROOT
python3 scripts/prof_summarize.py $OUT $MERGE > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
