#!/usr/bin/env python3
"""Per-stream ceiling of the DVB-S2 receive path: ONE stream (and small banks) of 8PSK 3/4 normal frames through
dvbs2gpu_demod_process_batch, IQ resident in HBM, synchronous mode; the serial recurrences (AGC, NCO, Gardner, PLL) set the time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as B
import __graft_entry__ as g
pkg = g.load_package()
eng = pkg.Engine(0)
F = int(os.environ.get('FRAMES', '4'))
info = pkg.modcod_info(B.MODCOD, bool(B.SHORT), bool(B.PILOTS))
kb = info['kbch'] // 8
sym = info['plframe_symbols']
sys.path.insert(0, os.path.join(ROOT, 'tests'))
for S in [int(x) for x in os.environ.get('STREAMS', '1,8,64,512').split(',')]:
    run = B.S2Run(eng, pkg, torch.device('cuda', 0), B.MODCOD, B.SHORT, B.PILOTS, 14.0, S, F, 4, seed=0)
    for _ in range(B.PREROLL_FRAMES // F + 2):
        run.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        nb = run.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('%4d stream(s) x %d frames per call: %.2f ms per call = %.3f Msym/s per stream, %.1f Msym/s total (frames out: %d, check %s)'
          % (S, F, dt * 1e3, F * sym / dt / 1e6, S * F * sym / dt / 1e6, nb[0] // kb, run.check(nb)))
    run.close()
