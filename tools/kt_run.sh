#!/bin/bash
# Development aid, on the GPU box: kernel trace of a short bench run -> gpurun_out/kt_<tag>.csv (per-kernel stats) and
# gpurun_out/kt_<tag>_timeline.txt (dispatches >= 2 ms).   usage: gpurun -- bash tools/kt_run.sh <tag> [bench.py arguments]
set -e
TAG=${1:-x}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
O=$R/gpurun_out/kt_$TAG
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary "$@" > $O.log 2>&1 || true
cd $R
DB=$(find $O -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/kt_$TAG.csv 2>&1
python tools/timeline.py $DB 2.0 > gpurun_out/kt_${TAG}_timeline.txt 2>&1 || true
rm -rf $O
head -14 gpurun_out/kt_$TAG.csv | cut -c1-150
