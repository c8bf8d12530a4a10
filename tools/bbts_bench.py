#!/usr/bin/env python3
"""BBFRAME -> TS parser bank timing: S streams x F BBFRAMEs of 8PSK 3/4 normal frames (kbch 48408) per call,
frames resident in HBM.  Prints one JSON line: packets/s, frames/s, GB/s moved (read DFL/8 + write 188 per 188) against HBM."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import __graft_entry__ as g
import orc_bbts as B

pkg = g.load_package()
eng = pkg.Engine(0)
S = int(os.environ.get('STREAMS', '4096'))
F = int(os.environ.get('FRAMES', '4'))
KBCH = int(os.environ.get('KBCH', '48408'))
fb = KBCH // 8
D = fb - 10
rng = np.random.default_rng(0)
nfr = 4 * F
pk = B.ts_packets(nfr * D // 188 + 2, rng)
fr = torch.from_numpy(B.bbframes_from_ts(pk, KBCH, nfr)).cuda()
bank = pkg.BbTsParserBank(eng, S, KBCH, F)
calls = [[fr[k * F:(k + 1) * F].reshape(-1).clone() for _ in range(S)] for k in range(4)]
outs = [torch.zeros(F * fb + 376, dtype=torch.uint8, device='cuda') for _ in range(S)]
for k in range(4):
    nb = bank.process_batch(calls[k], outs)
torch.cuda.synchronize()
reps = int(os.environ.get('REPS', '5'))
t0 = time.perf_counter()
tot = 0
for r in range(reps):
    for k in range(4):
        tot += sum(bank.process_batch(calls[k], outs))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / (4 * reps)
# check: the packets of the last call against the transmitted ones for stream 0 (stream state wrapped around the 4 calls: resync'd by SYNCD)
o = outs[0][:nb[0]].cpu().numpy().reshape(-1, 188)
assert np.all(o[:, 0] == 0x47)
print(json.dumps({'streams': S, 'frames_per_call': F, 'kbch': KBCH, 'ms_per_call': round(dt * 1e3, 3),
                  'frames_per_s': round(S * F / dt), 'ts_packets_per_s': round(tot / 188 / (4 * reps) / dt),
                  'GB_per_s_read_plus_write': round((S * F * D + tot / (4 * reps)) / dt / 1e9, 1), 'includes': 'host arg upload + sync per call'}))
