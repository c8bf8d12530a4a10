#!/usr/bin/env python3
"""Time the LDPC kernel alone for several codes (forced iterations) -- development aid."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
eng = pkg.Engine(0)
codes = [(6, 0), (0, 0), (3, 0), (10, 0), (9, 1)] if len(sys.argv) < 2 else [tuple(map(int, a.split(','))) for a in sys.argv[1:]]
iters = int(os.environ.get('ITERS', '50'))
for rate, short in codes:
    fi = pkg.fec_info(rate, short)
    pi = eng.ldpc_plan_info(rate, short)
    F = int(os.environ.get('FRAMES', str(pi['cus'] * pi['blocks_per_cu'] * 4)))
    if os.environ.get('SNR'):
        # decodable frames (SNR = Es/N0 in dB of a BPSK channel): 32 noisy codewords from the CPU encoder, repeated -- what the speculative passes of the half-row decoder
        # (kind 8) are fast on; the default, uniform noise, never converges: their slow case
        import orc
        rng = np.random.default_rng(5)
        base = np.stack([orc.bits_to_llr(orc.encode_frame(rate, short, 200 + k)[1], float(os.environ['SNR']), rng) for k in range(32)])
        llr = torch.from_numpy(base).cuda().repeat((F + 31) // 32, 1)[:F].contiguous()
    else:
        llr = torch.randint(-30, 31, (F, fi['ldpc_n']), dtype=torch.int8, device='cuda')
    eng.ldpc_decode(llr, rate, bool(short), max_trials=2, force=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.ldpc_decode(llr, rate, bool(short), max_trials=iters, force=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_frame_block_us = dt / (F / (pi['cus'] * pi['blocks_per_cu'])) * 1e6
    print('rate', rate, 'short', short, pi, 'F', F, 'ms %.2f' % (dt * 1e3), 'frames/s %.0f' % (F / dt),
          'us/iter/block %.1f' % (per_frame_block_us / iters), 'ns/step %.0f' % (per_frame_block_us / iters / pi['sum_depth'] * 1e3),
          'GB/s(alg) %.0f' % (F * (iters * 4 * pi['edges']) / dt / 1e9))
