#!/bin/bash
# Development aid, on the GPU box: SQ counter passes over ONE forced LDPC launch (rate 3/4 normal, FRAMES x ITERS) -> gpurun_out/pmc_split_<tag>.txt
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_split_$TAG
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export FRAMES=${FRAMES:-4096} ITERS=${ITERS:-50}
timeout 300 rocprofv3 --kernel-trace --stats --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/p3 -- python3 $R/tools/pmc_ldpc.py 6 > $O/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU -d $O/p4 -- python3 $R/tools/pmc_ldpc.py 6 > $O/p4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU -d $O/p5 -- python3 $R/tools/pmc_ldpc.py 6 > $O/p5.log 2>&1
cd $R
for d in p3 p4 p5; do python tools/rocpd_summary.py $(find $O/$d -name "*.db" | head -1) 2>&1 | grep -i "ldpc" ; done > $R/gpurun_out/pmc_split_$TAG.txt
find $O -name "*.db" -delete
cat $R/gpurun_out/pmc_split_$TAG.txt
