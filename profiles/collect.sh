#!/bin/bash
# Collects the measurements behind bench.py's roofline object on the GPU box (run from the repo root
# through gpurun).  Every GPU step has its own time limit and the chain stops at the first failure.
#   1. bench.py as the driver runs it (JSON line)
#   2. rocprofv3 --kernel-trace --stats of the same command (per-kernel durations)
#   3. separate --pmc passes: FETCH_SIZE, WRITE_SIZE (HBM-side bytes), SQ busy/wait counters, L2 requests / hits / misses with the
#      vector L1's read requests to L2, the vector L1's tag requests
# Raw output lands in gpurun_out/prof_<tag>/; profiles/summarize.py condenses it into profiles/<tag>_*.
set -e -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
REPO=$(pwd)
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 20 --warmup 6"
timeout -k 10 300 $BENCH > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/kt -o kt -- $BENCH --no-cpu-baseline --no-secondary > $REPO/$OUT/kt.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $REPO/$OUT/pmc_fetch -o pmc -- $BENCH --no-cpu-baseline --no-secondary > $REPO/$OUT/pmc_fetch.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $REPO/$OUT/pmc_write -o pmc -- $BENCH --no-cpu-baseline --no-secondary > $REPO/$OUT/pmc_write.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $REPO/$OUT/pmc_sq -o pmc -- $BENCH --no-cpu-baseline --no-secondary > $REPO/$OUT/pmc_sq.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $REPO/$OUT/pmc_l2 -o pmc -- $BENCH --no-cpu-baseline --no-secondary > $REPO/$OUT/pmc_l2.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum --output-format csv -d $REPO/$OUT/pmc_l1 -o pmc -- $BENCH --no-cpu-baseline --no-secondary > $REPO/$OUT/pmc_l1.log 2>&1
cd $REPO
python3 profiles/summarize.py $TAG
