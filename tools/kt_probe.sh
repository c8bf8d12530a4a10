cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
FRAMES=4096 ITERS=50 timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/probe1 -- python3 $R/tools/pmc_ldpc.py 6 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/probe2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aux --no-pipeline > $R/gpurun_out/probe2.log 2>&1
cd $R
python - <<'PY'
import sqlite3,glob
for d in ('probe1','probe2'):
    db=sqlite3.connect(glob.glob('gpurun_out/%s/**/*.db'%d, recursive=True)[0])
    rows=db.execute("select start,end,stream_id from kernels where name like '%ldpc_decode%' order by start").fetchall()
    print(d, [round((e-s)/1e6,2) for s,e,_ in rows][-8:])
PY
grep -o '"kernel_ms": [0-9.]*' gpurun_out/probe2.log
