#!/usr/bin/env python3
"""ONE fast transponder through the segment receiver (dvbs2gpu_segrx_*): a continuous DVB-S2 8PSK 3/4 normal-frame signal (a periodic
block of the repo's transmitter tiled on the device) in chunks of NSEG * OWN frames; prints Msymbols/s of that single stream and checks
that the returned BBFRAMEs follow the transmitted sequence."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import __graft_entry__ as g
import orc
pkg = g.load_package()
eng = pkg.Engine(0)
MODCOD, SHORT = int(os.environ.get('MODCOD', '14')), 0
NSEG, OWN, WARM = int(os.environ.get('NSEG', '512')), int(os.environ.get('OWN', '8')), int(os.environ.get('WARM', '8'))
PERIOD = 16
info = pkg.modcod_info(MODCOD, False, False)
kb, sym = info['kbch'] // 8, info['plframe_symbols']
iq, bb, _ = orc.transmit(MODCOD, SHORT, 0, nframes=PERIOD, seed=5, esn0_db=float(os.environ.get('ESN0', '16')), cfo=1e-4, timing=0.3, phase0=0.2, lead_symbols=0, circular=1)
index = {bytes(b): k for k, b in enumerate(bb)}
block = torch.from_numpy(iq).cuda()
cfg = eng.default_cfg(MODCOD, False, False)
rx = pkg.SegmentReceiver(eng, cfg, NSEG, OWN, WARM)
nper = rx.chunk_samples // block.numel()
chunk = block.repeat(nper)                      # NSEG*OWN frames, continuous with itself
out = torch.zeros((NSEG * OWN + WARM + 8) * kb, dtype=torch.uint8, device='cuda')
seq, times = [], []
for call in range(int(os.environ.get('CALLS', '3'))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nbytes = rx.process(chunk, out)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
    seq += [index.get(bytes(x), -1) for x in out[:nbytes].cpu().numpy().reshape(-1, kb)]
    print('call', call, 'frames', nbytes // kb, '%.1f ms' % (times[-1] * 1e3), rx.stats())
good = [k for k in seq if k >= 0]
steps = [(b - a) % PERIOD for a, b in zip(good, good[1:])]
dt = min(times[1:]) if len(times) > 1 else times[0]
print(json.dumps({'workload': 'one continuous transponder, MODCOD %d normal frames, %d segments x (%d own + %d warm-up) frames per call' % (MODCOD, NSEG, OWN, WARM),
                  'symbols_per_call': NSEG * OWN * sym, 'ms_per_call': round(dt * 1e3, 1), 'Msymbols_per_s_single_stream': round(NSEG * OWN * sym / dt / 1e6, 1),
                  'frames_returned': len(seq), 'not_a_transmitted_frame': len(seq) - len(good), 'out_of_sequence': sum(1 for s in steps if s != 1)}))
