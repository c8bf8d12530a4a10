cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
STEPS=3 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/mb -- python3 $R/tools/mixed_bench.py > $R/gpurun_out/mb.log 2>&1
cd $R
python - <<'PY'
import sqlite3,glob
db=sqlite3.connect(glob.glob('gpurun_out/mb/**/*.db', recursive=True)[0])
rows=db.execute("select name,start,end,stream_id from kernels order by start").fetchall()
# last ~1 step window: find last 3 LDPC groups
t_end=rows[-1][2]
sel=[r for r in rows if r[1] > t_end-260e6 and 's2::' in r[0] and (r[2]-r[1])>200e3]
t0=sel[0][1]
for r in sel[:90]:
    print('%-26s start %8.2f dur %7.2f ms stream %s'%(r[0].split('(')[0].replace('void ','')[4:30],(r[1]-t0)/1e6,(r[2]-r[1])/1e6,r[3]))
PY
