#!/bin/bash
# The counters of EVERY kernel of the pipelined headline step (front end + FEC), one rocprofv3 --pmc pass over 2 steps + 1 warm-up of the bench
# (counter collection runs the dispatches one after the other: the durations are each kernel's time ALONE, the counts are what they are beside the
# decoder too).  usage (GPU box): gpurun -- bash tools/pmc_frontend.sh r04 [bench.py arguments]   -> gpurun_out/pmc_fe_<tag>.csv
TAG=${1:-r04}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
O=$R/gpurun_out/pmc_fe_$TAG
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $O.log 2>&1
cd $R
DB=$(find $O -name "*.db" | head -1)
python tools/pmc_aggregate.py $DB > gpurun_out/pmc_fe_$TAG.csv 2>&1
rm -rf $O
cut -c1-200 gpurun_out/pmc_fe_$TAG.csv | head -30
