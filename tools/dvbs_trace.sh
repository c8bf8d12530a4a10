#!/bin/bash
# Development aid (GPU box): per-kernel times of the DVB-S receiver bank (tools/dvbs_bank_bench.py) for the carrier counts given.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/dvbs_trace; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for S in "$@"; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/s$S -- python3 $R/tools/dvbs_bank_bench.py $S > $O/s$S.log 2>&1
  python3 $R/tools/rocpd_summary.py $(find $O/s$S -name "*.db" | head -1) > $O/s$S.csv 2>&1
  tail -1 $O/s$S.log; head -14 $O/s$S.csv | cut -c1-150
done
find $O -name "*.db" -delete
