#!/usr/bin/env python3
"""Development aid: LDPC launch with many copies of 4 inputs, optionally with other kernels running beside it on a second
stream; every copy must decode to the same posteriors."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
pkg = g.load_package()
eng = pkg.Engine(0)
rate = 6
fi = pkg.fec_info(rate, False)
torch.manual_seed(1)
base = torch.randint(-30, 31, (4, fi['ldpc_n']), dtype=torch.int8, device='cuda')
load = os.environ.get('LOAD', 'none')
side = torch.cuda.Stream()
a = torch.randn(64 * 1024 * 1024, device='cuda')
m = torch.randn(4096, 4096, device='cuda')
import threading
bank = None
if load == 'dvbs':
    NS = 2048
    bank = pkg.DvbsDemodBank(eng, NS, max_samples=65536)
    iq = [torch.randn(65536, dtype=torch.complex64, device='cuda') * 0.7 for _ in range(4)]
    tin = [iq[i % 4] for i in range(NS)]
    tout = [torch.zeros(80000, dtype=torch.uint8, device='cuda') for _ in range(NS)]
    def side_load():
        for _ in range(3):
            bank.process_batch(tin, tout)
for F in (2048, 4096):
    llr = base.repeat(F // 4, 1).contiguous()
    ref, _, refpost = eng.ldpc_decode(base.clone(), rate, False, max_trials=20, force=True, want_post=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(60):
            if load == 'ew':
                a = a * 1.0001 + 0.5
            elif load == 'mm':
                m = (m @ m) * 1e-4
            elif load == 'sin':
                a = torch.sin(a) * 1.01
    th = None
    if bank is not None:
        th = threading.Thread(target=side_load)
        th.start()
        import time; time.sleep(0.02)
    hard, tri, post = eng.ldpc_decode(llr, rate, False, max_trials=20, force=True, want_post=True)
    torch.cuda.synchronize()
    if th: th.join()
    bad = [i for i in range(F) if not torch.equal(post[i], refpost[i % 4])]
    print(load, F, 'frames differing:', len(bad), bad[:8])
