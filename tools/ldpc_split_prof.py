#!/usr/bin/env python3
"""Development aid: cycles of every pseudo-layer of the half-row LDPC decoder (ldpc_split_kernel.hip built with -DLDPC_PROF=2; on the GPU box:
   MODE=cmd CMD="python tools/ldpc_split_prof.py 6 0" bash tools/ab.sh ldpc_split_kernel "-DLDPC_PROF=2")."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
eng = pkg.Engine(0)
rate, short = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (6, 0)
iters = int(os.environ.get('ITERS', '20'))
fi = pkg.fec_info(rate, short); pi = eng.ldpc_plan_info(rate, short)
F = int(os.environ.get('FRAMES', '2048'))
if os.environ.get('SNR'):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import orc
    rng = np.random.default_rng(5)
    base = np.stack([orc.bits_to_llr(orc.encode_frame(rate, short, 200 + k)[1], float(os.environ['SNR']), rng) for k in range(32)])
    llr = torch.from_numpy(base).cuda().repeat((F + 31) // 32, 1)[:F].contiguous()
else:
    llr = torch.randint(-30, 31, (F, fi['ldpc_n']), dtype=torch.int8, device='cuda')
buf = torch.zeros(1024, dtype=torch.int64, device='cuda')
eng.lib.dvbs2gpu_debug_set_prof.argtypes = [C.c_void_p]
eng.lib.dvbs2gpu_debug_set_prof(C.c_void_p(buf.data_ptr()))
eng.ldpc_decode(llr, rate, bool(short), max_trials=iters, force=True)
torch.cuda.synchronize()
full = buf.cpu().numpy().astype(np.float64)
sp = eng.ldpc_split_plan(rate, short)
nfr = F // (pi['cus'] * 2) if F >= pi['cus'] * 2 else 1       # frames workgroup 0 decoded (about)
per = full[128:128 + len(sp['kind'])] / iters / nfr
print('pseudo-layers %d, frames per workgroup ~%d, cycles per iteration: %.0f (+ %.0f outside the sweep)' % (len(sp['kind']), nfr, per.sum(), full[127] / iters / nfr))
cur = -1; line = ''
tot = {}
for i, c in enumerate(per):
    k = 'free' if sp['kind'][i] in (0, 7) and sp['nw'][i] == 12 and (i + 1 == len(per) or sp['layer'][i + 1] != sp['layer'][i]) and (i == 0 or sp['layer'][i - 1] != sp['layer'][i]) else ('walk%d' % sp['kind'][i] if sp['kind'][i] else 'packed')
    tot.setdefault(k, [0, 0.0]); tot[k][0] += 1; tot[k][1] += c
    if sp['layer'][i] != cur:
        if line: print(line)
        cur = sp['layer'][i]; line = 'layer %2d %-6s' % (cur, k)
    line += ' %d:%.0f' % (sp['nw'][i], c)
print(line)
for k, (n, c) in tot.items(): print('%-7s %3d pseudo-layers %8.0f cycles (%.0f each)' % (k, n, c, c / n))

if full[300:332].any():        # -DLDPC_PROF=3 build: waypoints inside the chain layers
    names = ['input', 'publish+record', 'barrier 1', 'walk', 'barrier 2', 'join', 'output']
    nchain = sum(1 for k in sp['kind'] if k == 1)
    for who, wn in ((0, 'thread 0  '), (1, 'thread 384')):
        v = full[300 + 16 * who: 300 + 16 * who + 7] / iters / nfr / nchain
        print('chain layers, %s: ' % wn + '  '.join('%s=%.0f' % (names[i], v[i]) for i in range(7)) + '   sum=%.0f per layer' % v.sum())

if full[200:300].any():        # -DLDPC_PROF=4 build: cycles the claim of the next pseudo-layer's words waits, per pseudo-layer
    for who, off in (('thread 0  ', 200), ('thread 384', 250)):
        v = full[off:off + len(sp['kind'])] / iters / nfr
        print('claim wait, %s: total %.0f cycles per iteration; per pseudo-layer: %s' % (who, v.sum(), ' '.join('%.0f' % x for x in v)))

if full[512:896].any():        # -DLDPC_SPLIT_SPEC_STATS=1 build: passes per kind-8 layer (workgroup 0)
    for i, k in enumerate(sp['kind']):
        if k == 8:
            h = full[512 + 32 * (i % 11): 512 + 32 * (i % 11) + 32]
            print('layer %2d (depth %d): passes  ' % (i, sp['aux'][i] >> 16) + '  '.join('%d:%.1f%%' % (n, 100 * c / h.sum()) for n, c in enumerate(h) if c) + '   mean %.2f' % ((h * np.arange(32)).sum() / h.sum()))
