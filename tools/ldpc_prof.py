#!/usr/bin/env python3
"""Development aid: per-phase cycle breakdown of the LDPC kernel (needs `make -C .../csrc prof`)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
pkg = g.load_package()
pkg.LIB_PATH = os.path.join(ROOT, 'sdrpp-dvbs-demodulator_amd', 'libdvbs2gpu_prof.so')
eng = pkg.Engine(0)
rate, short = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (6, 0)
iters = int(os.environ.get('ITERS', '20'))
fi = pkg.fec_info(rate, short); pi = eng.ldpc_plan_info(rate, short)
F = pi['cus'] * pi['blocks_per_cu'] * 2
llr = torch.randint(-30, 31, (F, fi['ldpc_n']), dtype=torch.int8, device='cuda')
buf = torch.zeros(512, dtype=torch.int64, device='cuda')
eng.lib.dvbs2gpu_debug_set_prof.argtypes = [C.c_void_p]
eng.lib.dvbs2gpu_debug_set_prof(C.c_void_p(buf.data_ptr()))
eng.ldpc_decode(llr, rate, bool(short), max_trials=iters, force=True)
torch.cuda.synchronize()
full = buf.cpu().numpy()
b = full[:96].reshape(6, 16)
names = ['free:S1', 'free:(none)', 'free:S3', 'barrier', 'conf:S1', 'chain:mid', 'conf:S3', 'prefetch/top', 'level:mid', 'ch:publish', 'ch:bar1', 'ch:walk', 'ch:bar2']
print(pi)
for w in range(6):
    print('wave', w, ' '.join('%s=%.0f' % (names[i], b[w, i] / iters) for i in range(len(names))), 'total/iter=%.0f cycles' % ((b[w, :9].sum()) / iters))

if full[127:].any():   # PROF_LEVEL=2 build: one probe per layer
    nl = pi['layers']
    per = full[128:128 + nl] / iters
    print('per-layer cycles (wave 0):', ' '.join('%d:%.0f' % (i, per[i]) for i in range(nl)))
    print('rest of an iteration: %.0f   sum: %.0f' % (full[127] / iters, per.sum() + full[127] / iters))
    if full[65]:
        print('slot layers: level loops %.0f cycles per iteration, %.0f levels -> %.0f cycles per level' % (full[64] / iters, full[65] / iters, full[64] / full[65]))
if full[200:328].any():   # PROF_LEVEL=3 build: waypoints
    names = ['input', 'hand-off', 'barrier1', 'walk', 'barrier2', 'late links', 'output', 'layer barrier+top']
    for kind, kn in ((0, 'free layers'), (1, 'chain layers')):
        for who, wn in ((0, 'wave 0'), (1, 'wave 5')):
            v = full[200 + 64 * kind + 16 * who: 200 + 64 * kind + 16 * who + 16] / iters
            print('%-12s %s: ' % (kn, wn) + '  '.join('%s=%.0f' % (names[i], v[i]) for i in range(8)) + '   sum=%.0f' % v[:8].sum())
            if v[8:11].any(): print('             input phase split: to reads issued=%.0f  first pair done=%.0f  up to last pair=%.0f  (the rest of "input" = last pair + merge)' % (v[8], v[9], v[10]))
