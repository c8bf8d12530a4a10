#!/bin/bash
# Development aid (GPU box): kernel trace + dispatch timeline of small synchronous S2 calls (tools/s2_single_stream.py: 1, 8, 64, 512 streams x 4 frames)
# -- shows the stage pipeline: RRC / walk / frame loops of a slice on the auxiliary stream beside the timing loop of the next slice.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s2_single; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -- python3 $R/tools/s2_single_stream.py > $O/run.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.csv 2>&1
python3 $R/tools/timeline.py $(find $O/kt -name "*.db" | head -1) ${MINMS:-1.0} > $O/timeline.txt 2>&1
find $O -name "*.db" -delete
grep "stream(s)" $O/run.log; head -12 $O/kernel_stats.csv | cut -c1-150
