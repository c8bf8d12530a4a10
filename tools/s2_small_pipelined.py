import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np, torch
import bench as B
import __graft_entry__ as g
pkg = g.load_package()
eng = pkg.Engine(0)
info = pkg.modcod_info(B.MODCOD, bool(B.SHORT), bool(B.PILOTS))
sym = info['plframe_symbols']
for pipe in (False, True):
  for S,F in ((1,4),(1,8),(8,4),(64,1),(64,4)):
    run = B.S2Run(eng, pkg, torch.device('cuda', 0), B.MODCOD, B.SHORT, B.PILOTS, 14.0, S, F, min(S,4), seed=0, iters=16, force=False)
    eng.set_pipelined(pipe)
    for _ in range(B.PREROLL_FRAMES // F + 4):
        run.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 8
    for _ in range(reps):
        nb = run.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('pipelined=%d %3d x %d: %.2f ms per call = %.3f Msym/s per stream, %.1f total' % (pipe, S, F, dt*1e3, F*sym/dt/1e6, S*F*sym/dt/1e6))
    eng.set_pipelined(False)
    run.close()
