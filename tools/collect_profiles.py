#!/usr/bin/env python3
"""Development aid: turn the outputs of tools/profile_run.sh (gpurun_out/fin_<tag>/) into the committed summaries under profiles/:
  <tag>_bench.json                 the bench line of that run
  <tag>_bench_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of the pipelined bench (3 steps)
  <tag>_bench_timeline.txt         every dispatch >= 2 ms with start / end (what overlaps what)
  <tag>_ldpc_pmc.txt               the PMC passes over one forced LDPC launch (4096 frames x 50 iterations)
  <tag>_ldpc_traffic.json          fabric traffic and VALU issue fraction derived from them (read by bench.py)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
sys.path.insert(0, ROOT)
from bench import ldpc_source_hash
F, P = os.path.join(ROOT, 'gpurun_out', 'fin_' + tag), os.path.join(ROOT, 'profiles')
rd = lambda n: open(os.path.join(F, n)).read()
open(os.path.join(P, tag + '_bench.json'), 'w').write(rd('bench.json'))
open(os.path.join(P, tag + '_bench_kernel_stats.csv'), 'w').write('# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary (tools/profile_run.sh)\n' + rd('kt.csv'))
open(os.path.join(P, tag + '_bench_timeline.txt'), 'w').write('# start_ms end_ms duration_ms queue kernel -- dispatches >= 2 ms of the same run (tools/timeline.py); q=3: FEC stream, q=2: front-end stream\n' + rd('timeline.txt'))
rows = [l.rstrip() for p in ('p1', 'p2', 'p3', 'p4') if os.path.exists(os.path.join(F, p + '.csv')) for l in open(os.path.join(F, p + '.csv')) if 'ldpc_decode_kernel' in l or 'ldpc_split_kernel' in l]
get = lambda c: ([float(l.split(',')[-1]) for l in rows if '"' + c + '"' in l] or [float('nan')])[0]
dur_ns = [float(l.split(',')[-4]) for l in rows if 'FETCH' not in l and 'SQ_' not in l and 'WRITE' not in l]   # (name,calls,total,AVERAGE,min,max,percentage)
ms = sum(dur_ns) / len(dur_ns) / 1e6
frames, iters = 4096, 50
head = ['# %s: PMC passes over ONE forced LDPC launch = the bench\'s dominant kernel (rate 3/4 normal, %d frames, %d iterations): tools/pmc_ldpc.py 6, FRAMES=%d ITERS=%d' % (tag, frames, iters, frames, iters),
        '# separate passes: --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_*  (rocprofv3, ROCm 7.2; values summed over XCDs; FETCH/WRITE_SIZE unit = KB)']
rows5 = [l.rstrip() for l in (open(os.path.join(F, 'p5.csv')) if os.path.exists(os.path.join(F, 'p5.csv')) else []) if 'ldpc_split_kernel' in l]
tail5 = (['# the issue counters again on DECODABLE frames (SNR=8 tools/pmc_ldpc.py 6: 32 noisy codewords repeated) -- the passes above decode uniform noise, which never converges:',
          '# the layers with shared bits are speculative (round 6), their instruction count and time depend on the data'] + rows5) if rows5 else []
open(os.path.join(P, tag + '_ldpc_pmc.txt'), 'w').write('\n'.join(head + rows + tail5) + '\n')
fetch, write, valu = get('FETCH_SIZE'), get('WRITE_SIZE'), get('SQ_INSTS_VALU')
traffic = (2 * fetch + write) * 1024
cus, simds, clk = 256, 4, 2.4e9
t = {
    'kernel': 'ldpc_split_kernel<12>' if any('ldpc_split_kernel' in l for l in rows) else 'ldpc_decode_kernel<12,4,false>',
    'kernel_source_sha16': ldpc_source_hash(),     # bench.py flags these figures as stale when the decoder sources have changed since
    'launch': {'rate': '3/4 normal', 'frames': frames, 'iterations': iters, 'forced': True, 'kernel_ms': round(ms, 3)},
    'source': 'profiles/%s_ldpc_pmc.txt (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE and --pmc SQ_* in separate passes, tools/pmc_ldpc.py); gfx950: FETCH_SIZE tallies '
              '128-B read requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is; both count fabric-side requests, Infinity-Cache hits '
              'included; KB = 1024 B' % tag,
    'fetch_size_kb': fetch, 'write_size_kb': write,
    'traffic_bytes_per_launch': traffic, 'traffic_bytes_per_frame': traffic / frames,
    'sq_insts_valu': valu,
    'input': 'uniform noise (never converges: the slow case of the speculative layers)',
    'on_decodable_frames': ({'kernel_ms': round([float(l.split(',')[-4]) for l in rows5 if 'SQ_' not in l][0] / 1e6, 3), 'sq_insts_valu': [float(l.split(',')[-1]) for l in rows5 if '"SQ_INSTS_VALU"' in l][0]} if rows5 else None),
    'valu_per_simd_cycle': round(valu / (cus * simds * ms * 1e-3 * clk), 4),
    'valu_per_simd_cycle_formula': 'SQ_INSTS_VALU / (256 CUs x 4 SIMDs x kernel time x 2.4 GHz); tools/ubench/valu_cu.hip: a wave issues one VALU instruction per 4.6 (4-byte '
                                   'encoding) / 5.6 (8-byte) cycles whatever else runs on its SIMD, and a SIMD sustains >= 0.88 per cycle with four such waves '
                                   '(profiles/r04_ubench_notes.txt) -- rounds 1-3 multiplied this figure by 4 and called it a utilisation; it is not one',
    'wait_fraction': round(get('SQ_WAIT_ANY') / get('SQ_WAVE_CYCLES'), 4),
    # where a wave's resident cycles go (SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES, both in quad-cycles; fourth PMC pass)
    'wave_cycles_fraction': {k: round(get(c) / get('SQ_WAVE_CYCLES'), 4) for k, c in (('valu', 'SQ_ACTIVE_INST_VALU'), ('scalar', 'SQ_ACTIVE_INST_SCA'),
                             ('lds', 'SQ_ACTIVE_INST_LDS'), ('misc', 'SQ_ACTIVE_INST_MISC'), ('waiting_for_lds', 'SQ_WAIT_INST_LDS'))},
    'reading': 'valu_per_simd_cycle x 4.3 = the share of the launch the vector ALUs are busy: the decoder\'s instructions are packed 16-bit / DPP / byte-permute forms, which '
               'a SIMD issues every 4.3 cycles (plain 32-bit VOP2: 2.3; profiles/r05_valu_rates.txt) -- rounds 1-4 read the same counter against a full-rate issue model and '
               'called the kernel latency-bound.  The rest is the serial sections of the layers with shared bits (profiles/r05_ldpc_split_layers.txt; DESIGN.md section 5)',
}
json.dump(t, open(os.path.join(P, tag + '_ldpc_traffic.json'), 'w'), indent=2)
wrows = [l.rstrip() for p in ('w1', 'w2', 'w3') if os.path.exists(os.path.join(F, p + '.csv')) for l in open(os.path.join(F, p + '.csv')) if 'ldpc_wave_kernel' in l]
if wrows:
    open(os.path.join(P, tag + '_ldpc_wave_pmc.txt'), 'w').write('\n'.join(['# %s: the wave-per-frame decoder (ldpc_wave_kernel<4,3>) on rate 8/9 short, 16384 frames x 50 forced iterations: tools/pmc_ldpc.py 9 1, FRAMES=16384 ITERS=50' % tag,
        '# separate passes: --pmc SQ_* | --pmc FETCH_SIZE | --pmc WRITE_SIZE (values summed over XCDs; FETCH/WRITE_SIZE unit = KB; gfx950: FETCH_SIZE counts 128-B requests at 64 B)'] + wrows) + '\n')
print(json.dumps(t, indent=1))
