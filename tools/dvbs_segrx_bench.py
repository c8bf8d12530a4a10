#!/usr/bin/env python3
"""ONE fast DVB-S carrier through the DVB-S segment receiver (dvbs2gpu_dvbs_segrx_*): Msymbols/s of that single stream (BASELINE config D
is 2 Msym/s), and the returned bit stream compared with the transmitted information bits (continuity: one anchor, then no slip)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import __graft_entry__ as g
import orc_dvbs as od
pkg = g.load_package(); eng = pkg.Engine(0)
RATE = int(os.environ.get('RATE', '0'))
NSEG, OWN, WARM = int(os.environ.get('NSEG', '64')), int(os.environ.get('OWN', '65536')), int(os.environ.get('WARM', '32768'))
CALLS = int(os.environ.get('CALLS', '3'))
nsym = CALLS * NSEG * OWN
iq, bits = od.dvbs_iq(RATE, nsym, seed=7, esn0_db=float(os.environ.get('ESN0', '12')), cfo=1e-4, timing=0.3, phase0=0.6)
ref = np.asarray(bits, np.uint8)
rx = pkg.DvbsSegmentReceiver(eng, NSEG, OWN, WARM)
d_iq = torch.from_numpy(iq).cuda()
out = torch.zeros(2 * NSEG * OWN + 8 * 65536, dtype=torch.uint8, device='cuda')
got, times, unmatched = [], [], 0
for c in range(CALLS):
    a = c * rx.chunk_samples
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nb = rx.process(d_iq[a:a + rx.chunk_samples], out)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    got.append(out[:nb].cpu().numpy().copy())
    st = rx.stats(); unmatched += st['unmatched']
    print('call', c, '%.1f ms' % (times[-1] * 1e3), st)
got = np.concatenate(got)
skip = 70000
k = ref.tobytes().find(got[skip:skip + 256].tobytes()); inv = 0
if k < 0:
    k = ref.tobytes().find((got[skip:skip + 256] ^ 1).tobytes()); inv = 1
n = min(got.size - skip, ref.size - k) if k >= 0 else 0
errs = int(np.count_nonzero((got[skip:skip + n] ^ inv) != ref[k:k + n])) if n else -1
dt = min(times[1:]) if len(times) > 1 else times[0]
print(json.dumps({'workload': 'one continuous DVB-S carrier, rate index %d, %d segments x (%d own + %d warm-up) symbols per call' % (RATE, NSEG, OWN, WARM),
                  'ms_per_call': round(dt * 1e3, 1), 'Msymbols_per_s_single_stream': round(NSEG * OWN / dt / 1e6, 2), 'realtime_factor_at_2_Msym_s': round(NSEG * OWN / dt / 2e6, 1),
                  'bits_returned': int(got.size), 'compared_bits': n, 'bit_errors_after_anchor': errs, 'segments_without_match': unmatched}))
