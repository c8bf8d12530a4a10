set -x
mkdir -p gpurun_out/fin
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/fin/gpu_tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/fin/smoke.txt
timeout 400 python bench.py 2>&1 | tail -1 > gpurun_out/fin/bench.json
timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-aux 2>&1 | tail -1 > gpurun_out/fin/bench10.json
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/fin/kt -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-aux > $R/gpurun_out/fin/kt.log 2>&1
export FRAMES=4096 ITERS=50
timeout 200 rocprofv3 --kernel-trace --stats --pmc FETCH_SIZE -d $R/gpurun_out/fin/p1 -- python3 $R/tools/pmc_ldpc.py 6 > $R/gpurun_out/fin/p1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --pmc WRITE_SIZE -d $R/gpurun_out/fin/p2 -- python3 $R/tools/pmc_ldpc.py 6 > $R/gpurun_out/fin/p2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/fin/p3 -- python3 $R/tools/pmc_ldpc.py 6 > $R/gpurun_out/fin/p3.log 2>&1
cd $R
for d in kt p1 p2 p3; do python tools/rocpd_summary.py $(find gpurun_out/fin/$d -name "*.db" | head -1) > gpurun_out/fin/$d.csv 2>&1; done
find gpurun_out/fin -name "*.db" -size +20M -delete
cat gpurun_out/fin/gpu_tests.txt gpurun_out/fin/smoke.txt gpurun_out/fin/bench.json gpurun_out/fin/bench10.json
