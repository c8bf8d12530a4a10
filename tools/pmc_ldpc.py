#!/usr/bin/env python3
"""run one forced LDPC launch (B7, 512 frames, 20 iterations) -- target for rocprofv3 --pmc passes"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
pkg = g.load_package()
eng = pkg.Engine(0)
rate = int(sys.argv[1]) if len(sys.argv) > 1 else 6
short = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
F = int(os.environ.get('FRAMES', '512'))
fi = pkg.fec_info(rate, short)
llr = torch.randint(-30, 31, (F, fi['ldpc_n']), dtype=torch.int8, device='cuda')
eng.ldpc_decode(llr, rate, short, max_trials=int(os.environ.get('ITERS', '20')), force=True)
torch.cuda.synchronize()
