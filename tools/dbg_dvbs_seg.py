#!/usr/bin/env python3
"""Development aid: how long DVB-S segments (fresh loops, fresh Viterbi) take to deliver correct bits."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import __graft_entry__ as g
import orc_dvbs as od
pkg = g.load_package(); eng = pkg.Engine(0)
rate = int(os.environ.get('RATE', '0'))
nsym = 200000
iq, bits = od.dvbs_iq(rate, nsym, seed=31, esn0_db=9.0, cfo=5e-4, timing=0.3, phase0=0.6)
ref = np.asarray(bits, np.uint8).tobytes()
S, L = 8, 100000            # segments of L samples starting at different offsets
bank = pkg.DvbsDemodBank(eng, S, max_samples=L)
offs = [0, 10000, 23456, 50000, 77777, 100001, 150000, 199998]
tin = [torch.from_numpy(iq[o:o + L]).cuda() for o in offs]
tout = [torch.zeros(L + 4 * 8192, dtype=torch.uint8, device='cuda') for _ in range(S)]
nb = bank.process_batch(tin, tout)
for s in range(S):
    b = tout[s][:nb[s]].cpu().numpy()
    # first position from which 256 bits match the reference somewhere, scanning windows
    first = None
    for p in range(0, max(0, b.size - 256), 512):
        k = ref.find(b[p:p + 256].tobytes())
        if k >= 0:
            first = (p, k)
            break
    tail_ok = ref.find(b[-400:-144].tobytes()) if b.size > 400 else -2
    print('segment at sample %6d: %6d bits out; first good window at bit %s (ref bit %s); expected ref bit ~%d; tail window found: %s' %
          (offs[s], nb[s], first[0] if first else None, first[1] if first else None, offs[s] // 2, tail_ok >= 0))
print(bank.stats()[0].state if hasattr(bank.stats()[0], 'state') else '')
refa = np.asarray(bits, np.uint8)
for s in (4, 5, 0):
    b = tout[s][:nb[s]].cpu().numpy()
    p0 = 1024
    k = ref.find(b[p0:p0 + 256].tobytes())
    if k < 0:
        print('seg', s, 'no anchor'); continue
    a = k - p0
    lo = max(0, -a)
    m = min(b.size, refa.size - a)
    d = (b[lo:m] != refa[a + lo:a + m])
    chunks = [int(d[i:i + 4096].sum()) for i in range(0, m, 4096)]
    print('seg', s, 'errors per 4096-bit chunk:', chunks, 'last error at', int(np.nonzero(d)[0][-1]) if d.any() else None, 'of', m)
