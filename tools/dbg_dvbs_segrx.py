#!/usr/bin/env python3
"""Development aid: the DVB-S segment receiver call by call (matches, discontinuities, error profile against the transmitted bits)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import __graft_entry__ as g
import orc_dvbs as od
pkg = g.load_package(); eng = pkg.Engine(0)
rate = int(os.environ.get('RATE', '0'))
nseg, own, warm = int(os.environ.get('NSEG', '4')), int(os.environ.get('OWN', '49152')), int(os.environ.get('WARM', '32768'))
nsym = 5 * nseg * own // 2 + 20000
iq, bits = od.dvbs_iq(rate, nsym, seed=31 + rate, esn0_db=float(os.environ.get('ESN0', 9.0 + 3 * rate)), cfo=float(os.environ.get('CFO', '5e-4')), timing=0.3, phase0=0.6)
ref = np.asarray(bits, np.uint8)
rx = pkg.DvbsSegmentReceiver(eng, nseg, own, warm)
d_iq = torch.from_numpy(iq).cuda()
out = torch.zeros(2 * nseg * own * 2 + 4 * 65536, dtype=torch.uint8, device='cuda')
got, a, k = [], 0, 0
sizes = [rx.chunk_samples, rx.chunk_samples // 2 + 777]
while a < iq.size:
    n = min(sizes[k % 2], iq.size - a)
    nb = rx.process(d_iq[a:a + n], out)
    got.append(out[:nb].cpu().numpy().copy())
    print('call', k, 'samples', n, 'bits', nb, rx.stats())
    a += n; k += 1
got = np.concatenate(got)
# error profile in 8192-bit chunks, re-anchoring after every discontinuity
pos = 70000
while pos + 256 < got.size:
    kk = ref.tobytes().find(got[pos:pos + 256].tobytes()); inv = 0
    if kk < 0:
        kk = ref.tobytes().find((got[pos:pos + 256] ^ 1).tobytes()); inv = 1
    if kk < 0:
        pos += 256; continue
    m = min(got.size - pos, ref.size - kk)
    d = (got[pos:pos + m] ^ inv) != ref[kk:kk + m]
    bad = np.nonzero(np.convolve(d.astype(np.int32), np.ones(512, np.int32), 'valid') > 128)[0]
    run = int(bad[0]) if bad.size else m
    print('  anchor at out bit %d = ref bit %d (inverted %d): clean for %d bits, errors in them %d' % (pos, kk, inv, run, int(d[:run].sum())))
    if run >= m: break
    pos += run + 512
