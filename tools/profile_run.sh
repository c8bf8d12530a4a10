#!/bin/bash
# The round's evidence run on the GPU box (gpurun -- bash tools/profile_run.sh r02): bench line, kernel trace of the pipelined bench,
# four separate PMC passes over ONE forced LDPC launch (the bench's dominant kernel).  Outputs under gpurun_out/fin_<tag>/;
# tools/collect_profiles.py <tag> turns them into the committed summaries under profiles/.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/fin_$TAG
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/kt.log 2>&1
export FRAMES=4096 ITERS=50
timeout 300 rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $O/p1 -- python3 $R/tools/pmc_ldpc.py 6 > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $O/p2 -- python3 $R/tools/pmc_ldpc.py 6 > $O/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/p3 -- python3 $R/tools/pmc_ldpc.py 6 > $O/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU -d $O/p4 -- python3 $R/tools/pmc_ldpc.py 6 > $O/p4.log 2>&1
# the same issue counters on DECODABLE frames (the speculative layers' fast case; the passes above run on noise, their slow one)
SNR=8 timeout 300 rocprofv3 --kernel-trace --stats --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/p5 -- python3 $R/tools/pmc_ldpc.py 6 > $O/p5.log 2>&1
# the wave-per-frame decoder on the config 5 stand-in's code (rate 8/9 short, 16384 frames x 50 iterations): issue counters + traffic
export FRAMES=16384
timeout 300 rocprofv3 --kernel-trace --stats --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/w1 -- python3 $R/tools/pmc_ldpc.py 9 1 > $O/w1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $O/w2 -- python3 $R/tools/pmc_ldpc.py 9 1 > $O/w2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $O/w3 -- python3 $R/tools/pmc_ldpc.py 9 1 > $O/w3.log 2>&1
cd $R
for d in kt p1 p2 p3 p4 p5 w1 w2 w3; do python tools/rocpd_summary.py $(find $O/$d -name "*.db" | head -1) > $O/$d.csv 2>&1; done
python tools/timeline.py $(find $O/kt -name "*.db" | head -1) 2.0 > $O/timeline.txt 2>&1
find $O -name "*.db" -delete
tail -c 400 $O/bench.json; head -5 $O/kt.csv; grep ldpc $O/p1.csv $O/p2.csv | head
