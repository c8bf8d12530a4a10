cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
DVBS_DEMOD_BENCH=1 DVBS_BANK_STREAMS=4096 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dvbsp -- python3 $R/tools/dvbs_bench.py > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py $(find gpurun_out/dvbsp -name "*.db" | head -1) | head -12 | cut -c1-150
