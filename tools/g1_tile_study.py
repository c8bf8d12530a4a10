#!/usr/bin/env python3
"""CPU study (oracle only, no GPU), round 6: would the fixed-point treatment the payload PLL got in round 5 (tools/pll_tile_study.py) pay for the OTHER two serial chains of a stream --
FastAGC (SDR++ loop::FastAGC, call site module_dvbs2_demod.cpp:220) and the timing recovery (gardner.cpp:89-152)?  Per tile the loop's per-sample / per-symbol contribution is evaluated
for all elements at once from guessed loop states, the recurrence proper is replayed serially, and that is repeated until the replay reproduces the guesses (oracle/s2chain.cpp:
agc_tile_study, gardner_tile_study; the fixed point is the serial result -- the study checks it for every tile).  Prints the pass histograms for the headline workload and what the
schemes would cost per symbol with the engine's measured instruction costs.
usage: tools/g1_tile_study.py [modcod short pilots esn0_db [frames]]"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import orc
modcod, short, pilots, esn0 = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (14, 0, 0, 11.0)
frames = int(sys.argv[5]) if len(sys.argv) > 5 else 24
iq, bb, _ = orc.transmit(modcod, short, pilots, nframes=frames, seed=77, esn0_db=esn0, cfo=1e-3, timing=0.3, phase0=0.1, lead_symbols=700)
L = orc._bind_chain()
LL = C.POINTER(C.c_longlong)
L.orc_s2rx_g1_study.argtypes = [C.c_void_p, C.c_int, C.c_int, LL, LL, LL, LL]
L.orc_s2rx_g1_study.restype = None
CYC = 5.0            # a lone wave issues one instruction per ~5 cycles (tools/ubench/lone_wave.hip)
print('workload: MODCOD %d, short %d, pilots %d, Es/N0 %.1f dB, %d PLFRAMEs, carrier offset 1e-3 rad/sample, timing offset 0.3' % (modcod, short, pilots, esn0, frames))
print('FastAGC (serial loop of the engine: 36 instructions per sample incl. an IEEE sqrt, ~240 cycles per sample = ~480 per symbol)')
for tile in (16, 32, 64):
    rx = orc.OracleRx(orc.default_cfg(modcod, short, pilots, force_ldpc_iters=-1))
    L.orc_s2rx_g1_study(rx.h, tile, 0, None, None, None, None)
    rx.process(iq)
    h = (C.c_longlong * 66)(); a2 = (C.c_longlong * 2)()
    L.orc_s2rx_g1_study(rx.h, 0, 0, h, a2, None, None)
    hh = np.array(list(h), float); n = hh.sum()
    mean = (hh * np.arange(66)).sum() / max(n, 1)
    # per pass: evaluation ~40 instructions once per tile (lane = sample) + the replay: subtract, multiply, add, min + the a[k] out of its lane = ~6 instructions per sample
    per_sample = mean * (40.0 / tile + 6.0) * CYC
    print('  tile %2d samples: %6d tiles, fixed point != serial loop in %d, passes per tile: mean %.2f  %s' % (tile, n, a2[0], mean,
          ' '.join('%d:%.1f%%' % (i, 100 * hh[i] / n) for i in range(66) if hh[i] / n >= 0.005)))
    print('                   modelled cycles per sample: %.0f (serial loop ~240) -> per symbol %.0f' % (per_sample, 2 * per_sample))
print('timing recovery (candidate-table form of the engine: ~68 instructions + one LDS round trip per symbol, ~500 cycles; resolver + producer form ~600)')
for tile in (8, 16, 32):
    rx = orc.OracleRx(orc.default_cfg(modcod, short, pilots, force_ldpc_iters=-1))
    L.orc_s2rx_g1_study(rx.h, 0, tile, None, None, None, None)
    rx.process(iq)
    g = (C.c_longlong * 34)(); g5 = (C.c_longlong * 5)()
    L.orc_s2rx_g1_study(rx.h, 0, 0, None, None, g, g5)
    gg = np.array(list(g), float); n = gg.sum()
    mism, syms, evals, steps, tiles = list(g5)
    mean = evals / max(tiles, 1)
    # per evaluation pass: ~110 instructions once per tile (lane = symbol: three 8-tap complex interpolants + the error); replay ~26 instructions per symbol (two advances, two floors, arm, compare)
    per_symbol = (mean * 110.0 / tile + (steps / max(syms, 1)) * 26.0) * CYC
    print('  tile %2d symbols: %6d tiles, result != serial loop in %d, evaluation passes per tile: mean %.2f  %s; replay steps per symbol %.3f' % (tile, n, mism, mean,
          ' '.join('%d:%.1f%%' % (i, 100 * gg[i] / n) for i in range(34) if gg[i] / n >= 0.005), steps / max(syms, 1)))
    print('                   modelled cycles per symbol: %.0f' % per_symbol)
