#!/usr/bin/env python3
"""Development aid: what the per-iteration syndrome check (LDPCDecoder::bad) costs.  Forced mode evaluates it once, after the last
iteration (its result cannot end the loop earlier); the normal mode evaluates it before every iteration, like the reference.  Input:
noise that never converges, so both modes run exactly `ITERS` iterations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
pkg = g.load_package()
eng = pkg.Engine(0)
rate, short = 6, False
iters, F = int(os.environ.get('ITERS', 50)), int(os.environ.get('FRAMES', 4096))
fi = pkg.fec_info(rate, short)
torch.manual_seed(3)
llr = torch.randint(-20, 21, (F, fi['ldpc_n']), dtype=torch.int8, device='cuda')
for force in (True, False):
    eng.ldpc_decode(llr, rate, short, max_trials=iters, force=force)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hard, trials, _ = eng.ldpc_decode(llr, rate, short, max_trials=iters, force=force)
    e1.record()
    torch.cuda.synchronize()
    print('force=%d: %.2f ms for %d frames x %d iterations, trials min/max %d/%d' % (force, e0.elapsed_time(e1), F, iters, int(trials.min()), int(trials.max())))
