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
if os.environ.get('SNR'):
    # decodable frames (SNR = the BPSK channel's 1 / sigma^2 in dB): what the half-row decoder's speculative layers are fast on; the default, uniform noise, their slow case
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np, orc
    rng = np.random.default_rng(5)
    base = np.stack([orc.bits_to_llr(orc.encode_frame(rate, short, 200 + k)[1], float(os.environ['SNR']), rng) for k in range(32)])
    llr = torch.from_numpy(base).cuda().repeat((F + 31) // 32, 1)[:F].contiguous()
else:
    llr = torch.randint(-30, 31, (F, fi['ldpc_n']), dtype=torch.int8, device='cuda')
eng.ldpc_decode(llr, rate, short, max_trials=int(os.environ.get('ITERS', '20')), force=True)
torch.cuda.synchronize()
