#!/bin/bash
# Counter passes of the balls2 1080p frame (the frame the first profiles of the traversal kernel were taken on), for a
# like-for-like before/after of that kernel.  Run from the repo root through gpurun.
set -e -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/prof_${TAG}_balls2
REPO=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 $REPO/tools/gpu_breakdown.py balls2"
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/kt -o kt -- $CMD > $REPO/$OUT/kt.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $REPO/$OUT/pmc_sq -o pmc -- $CMD > $REPO/$OUT/pmc_sq.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/$OUT/pmc_fetch -o pmc -- $CMD > $REPO/$OUT/pmc_fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $REPO/$OUT/pmc_write -o pmc -- $CMD > $REPO/$OUT/pmc_write.log 2>&1
cd $REPO
