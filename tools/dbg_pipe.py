#!/usr/bin/env python3
"""Development aid: the bench's pipelined loop with every stream of every step checked against the sent BBFRAMEs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as B
import __graft_entry__ as g
pkg = g.load_package()
eng = pkg.Engine(0)
S = int(os.environ.get('STREAMS', '2048'))
steps = int(os.environ.get('STEPS', '4'))
pipe = int(os.environ.get('PIPE', '1'))
info = pkg.modcod_info(B.MODCOD, bool(B.SHORT), bool(B.PILOTS))
kb = info['kbch'] // 8
blocks, sent = B.make_blocks(1, seed=0, eng=eng, pkg=pkg)
d_blocks = [torch.from_numpy(b).cuda() for b in blocks]
cfg = eng.default_cfg(B.MODCOD, bool(B.SHORT), bool(B.PILOTS), force_ldpc_iters=B.ITERS)
demods = [eng.demod(cfg, max_samples=blocks[0].size) for _ in range(S)]
tin = [d_blocks[s % B.DISTINCT] for s in range(S)]
tout = [torch.zeros(3 * kb, dtype=torch.uint8, device='cuda') for _ in range(S)]
eng.set_pipelined(bool(pipe))
def check(nb, tag):
    bad = []
    for s in range(S):
        got = tout[s][:nb[s]].cpu().numpy().reshape(-1, kb)
        if nb[s] != kb or not all(bytes(x) in sent[s % B.DISTINCT] for x in got):
            bad.append(s)
    rng = []
    for b in bad:
        if rng and b == rng[-1][1] + 1: rng[-1][1] = b
        else: rng.append([b, b])
    print(tag, 'bad streams', len(bad), 'ranges', rng[:24])
    for s in bad[:3]:
        got = tout[s][:nb[s]].cpu().numpy().reshape(-1, kb)
        ref = [np.frombuffer(x, np.uint8) for x in sent[s % B.DISTINCT]]
        d = min(int((got[0] != r).sum()) for r in ref) if len(got) else -1
        print('   stream', s, 'bytes', nb[s], 'min byte diffs vs sent frames', d)
for k in range(B.PREROLL + steps):
    nb = eng.process_batch(demods, tin, tout)
check(nb, 'last step')
empty = [torch.empty(0, dtype=torch.complex64, device='cuda') for _ in range(S)]
nb = eng.process_batch(demods, empty, tout)
check(nb, 'flush')
