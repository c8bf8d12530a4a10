#!/usr/bin/env python3
"""CPU study (oracle only, no GPU): the payload PLL of S2PLLBlock::process (dvbs2_pll.cpp:34-86) in TILES -- every symbol's phase error evaluated in parallel from
guessed loop phases, the ten-instruction recurrence (freq += beta e, clamp, phase += freq + alpha e, wrap) replayed serially over the tile, repeated until the
replay reproduces the phases its errors were evaluated at (that fixed point is the serial result: oracle/s2chain.cpp, pll_tile_study).  Prints the histogram of
evaluation passes per tile for the headline workload and what the scheme would cost per symbol with the engine's measured instruction costs.
usage: tools/pll_tile_study.py [modcod short pilots esn0_db [frames]]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import orc
modcod, short, pilots, esn0 = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (14, 0, 0, 14.0)
frames = int(sys.argv[5]) if len(sys.argv) > 5 else 24
iq, bb, _ = orc.transmit(modcod, short, pilots, nframes=frames, seed=77, esn0_db=esn0, cfo=1e-3, timing=0.3, phase0=0.1, lead_symbols=700)
L = orc._bind_chain()
L.orc_s2rx_pll_study.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
L.orc_s2rx_pll_study.restype = None
L.orc_s2rx_pll_study2.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
L.orc_s2rx_pll_study2.restype = None
for tile in (16, 32, 46, 64, 128):
    rx = orc.OracleRx(orc.default_cfg(modcod, short, pilots, force_ldpc_iters=-1))
    L.orc_s2rx_pll_study(rx.h, tile, None, None)
    rx.process(iq)
    hist = (C.c_longlong * 34)(); mis = C.c_longlong(0)
    L.orc_s2rx_pll_study(rx.h, 0, hist, C.byref(mis))
    h = np.array(list(hist), float)
    n = h.sum()
    mean = (h * np.arange(34)).sum() / max(n, 1)
    # cost model (a lone wave issues one instruction per ~5 cycles; the LUT fetch ~250 cycles; evaluation ~60 instructions, replay ~10 per symbol)
    per_pass = 60 * 5 + 250 + tile * 10 * 5
    print('tile %3d: %6d tiles, fixed point != serial loop in %d, passes per tile: mean %.2f, histogram %s%s' % (tile, n, mis.value, mean,
          ' '.join('%d:%.1f%%' % (i, 100 * h[i] / n) for i in range(34) if h[i]), ''))
    print('          modelled cycles per symbol: %.0f (the serial loop: ~590)' % (mean * per_pass / tile))
    o3 = (C.c_longlong * 3)()
    L.orc_s2rx_pll_study2(rx.h, o3)
    steps, syms, evals = list(o3)
    print('          the form the engine runs (replay restarted at the first symbol whose table cell changed): %.2f evaluation passes per tile, %.2f replay steps per symbol' % (evals / max(n, 1), steps / max(syms, 1)))
