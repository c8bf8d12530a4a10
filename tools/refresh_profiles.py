#!/usr/bin/env python3
"""Development aid: copy the summaries of tools/final_run.sh (gpurun_out/fin/) into profiles/ (kernel stats of the pipelined bench, PMC
passes and traffic of one LDPC launch, per-launch list, bench line)."""
import glob, json, os, re, sqlite3
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, P = os.path.join(ROOT, 'gpurun_out', 'fin'), os.path.join(ROOT, 'profiles')
ks = os.path.join(P, 'r01_fullchain_v9_pipelined_kernel_stats.csv')
hdr = open(ks).readline()
open(ks, 'w').write(hdr + open(os.path.join(F, 'kt.csv')).read())
pm = os.path.join(P, 'r01_ldpc_pmc.txt')
head = [l for l in open(pm).read().split('\n') if l.startswith('#')]
rows = [l.rstrip() for p in ('p1', 'p2', 'p3') for l in open(os.path.join(F, p + '.csv')) if 'ldpc_decode_kernel' in l]
open(pm, 'w').write('\n'.join(head + rows) + '\n')
get = lambda c: [float(l.split(',')[-1]) for l in rows if '"' + c + '"' in l][0]
tj = os.path.join(P, 'r01_ldpc_traffic.json')
t = json.load(open(tj))
t['fetch_size_kb'], t['write_size_kb'] = get('FETCH_SIZE'), get('WRITE_SIZE')
t['traffic_bytes_per_launch'] = (2 * t['fetch_size_kb'] + t['write_size_kb']) * 1024
json.dump(t, open(tj, 'w'), indent=2)
open(os.path.join(P, 'r01_bench_final.json'), 'w').write(open(os.path.join(F, 'bench.json')).read())
kms = re.search(r'"kernel_ms": ([0-9.]+)', open(os.path.join(F, 'kt.log')).read()).group(1)
db = sqlite3.connect(sorted(glob.glob(os.path.join(F, 'kt', '*', '*.db')), key=os.path.getmtime)[-1])
cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
ix = {c: i for i, c in enumerate(cols)}
rws = list(db.execute("select * from kernels where name like '%ldpc_decode%' order by start"))
ll = os.path.join(P, 'r01_ldpc_launches_in_bench.csv')
lines = [l for l in open(ll) if l.startswith('#')][:2]
lines += ['#        alone = the launches bench.py times for roofline.kernel_ms / kernel_ms_normal_mode_same_iterations (4096 frames, nothing else running; the first four are the forced ones): bench line of this very run: kernel_ms = %s\n' % kms,
          'launch,start_ms,duration_ms,hip_stream,grid_x_threads,phase\n']
t0, alone = rws[0][ix['start']], []
for k, r in enumerate(rws):
    dur = (r[ix['end']] - r[ix['start']]) / 1e6
    ph = 'preroll' if r[ix['grid_x']] < 196608 else ('alone' if r[ix['stream_id']] == 0 else 'step')
    if ph == 'alone':
        alone.append(dur)
    lines.append('%d,%.3f,%.3f,%d,%d,%s\n' % (k, (r[ix['start']] - t0) / 1e6, dur, r[ix['stream_id']], r[ix['grid_x']], ph))
lines.append('# mean of the 3 timed forced stand-alone launches (2nd to 4th): %.3f ms (HIP events in bench.py, same launches: %s ms)\n' % (sum(alone[1:4]) / 3, kms))
open(ll, 'w').writelines(lines)
print(t['traffic_bytes_per_launch'], kms, alone)
