R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/dvbs_ts_tl; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp DVBS2GPU_BENCH_DVBS_BANKS=1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/s1 -- python3 $R/tools/dvbs_ts_bench.py > $O/s1.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $O/s1 -name "*.db" | head -1) > $O/stats.csv 2>&1
python3 $R/tools/timeline.py $(find $O/s1 -name "*.db" | head -1) 0 > $O/tl.txt
find $O -name "*.db" -delete
grep bank_1_msym $O/s1.log; head -30 $O/stats.csv | cut -c1-140
