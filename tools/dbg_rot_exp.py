import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import __graft_entry__ as g
import orc
pkg = g.load_package(); eng = pkg.Engine(0)
M = 14
info = pkg.modcod_info(M, False, False); kb = info['kbch'] // 8; sym = info['plframe_symbols']
iq, bb, _ = orc.transmit(M, 0, 0, nframes=16, seed=5, esn0_db=16.0, cfo=1e-4, timing=0.3, phase0=0.2, lead_symbols=0, circular=1)
sent = {bytes(b) for b in bb}
block = torch.from_numpy(iq).cuda()
S = 128
NF = 16
rng = np.random.default_rng(0)
# S segments starting at random sample offsets of the periodic stream, NF frames long
long = block.repeat(3)
offs = [int(rng.integers(0, block.numel())) for _ in range(S)]
segs = [long[o:o + NF * 2 * sym].clone() for o in offs]
def run(rot):
    demods = [eng.demod(eng.default_cfg(M, False, False), max_samples=segs[0].numel()) for _ in range(S)]
    tin = [s * complex(np.cos(rot), np.sin(rot)) for s in segs] if rot else segs
    tout = [torch.zeros((NF + 2) * kb, dtype=torch.uint8, device='cuda') for _ in range(S)]
    nb = eng.process_batch(demods, tin, tout)
    res = []
    for i in range(S):
        got = tout[i][:nb[i]].cpu().numpy().reshape(-1, kb)
        ok = [bytes(x) in sent for x in got]
        res.append(ok)
    for d in demods: d.close()
    return res
r0 = run(0.0)
lock0 = [sum(ok[-4:]) == 4 for ok in r0]
first_good = [next((i for i, v in enumerate(ok) if v), None) for ok in r0]
print('rotation 0: segments whose last 4 frames are good:', sum(lock0), 'of', S)
print('first good frame index histogram:', np.bincount([f if f is not None else 15 for f in first_good], minlength=16))
for k in (1, 2, -1):
    rk = run(k * np.pi / 4)
    lk = [sum(ok[-4:]) == 4 for ok in rk]
    fixed = sum(1 for a, b in zip(lock0, lk) if not a and b)
    broke = sum(1 for a, b in zip(lock0, lk) if a and not b)
    print('rotation %+d*45deg: good' % k, sum(lk), 'fixes', fixed, 'of', S - sum(lock0), 'failing; breaks', broke)
rk = run(np.pi / 8)
print('rotation 22.5deg: good', sum(sum(ok[-4:]) == 4 for ok in rk))
