R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/dvbs_tl; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/s1 -- python3 $R/tools/dvbs_bank_bench.py 1 > $O/s1.log 2>&1
python3 $R/tools/timeline.py $(find $O/s1 -name "*.db" | head -1) 0 > $O/tl.txt
find $O -name "*.db" -delete
tail -2 $O/s1.log; wc -l $O/tl.txt
