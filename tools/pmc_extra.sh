#!/bin/bash
# Development aid (GPU box): extra SQ counter passes over one forced LDPC launch (tools/pmc_ldpc.py) -- where do the wave cycles go
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmcx; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp FRAMES=4096 ITERS=50
timeout 300 rocprofv3 --kernel-trace --stats --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU -d $O/a -- python3 $R/tools/pmc_ldpc.py 6 > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --pmc SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES -d $O/b -- python3 $R/tools/pmc_ldpc.py 6 > $O/b.log 2>&1
cd $R
for d in a b; do python tools/rocpd_summary.py $(find $O/$d -name "*.db" | head -1) > $O/$d.csv 2>&1; done
find $O -name "*.db" -delete
grep -h ldpc $O/a.csv $O/b.csv
