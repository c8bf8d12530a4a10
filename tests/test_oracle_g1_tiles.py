"""CPU-only: the parallel-in-time forms of FastAGC and of the timing recovery studied in tools/g1_tile_study.py (oracle/s2chain.cpp: agc_tile_study, gardner_tile_study) -- the
per-sample amplitude / the per-symbol timing error evaluated for a whole tile at once from guessed loop states, the recurrence proper (SDR++ loop::FastAGC; gardner.cpp:141-152)
replayed in the reference's order, repeated until the replay reproduces the guesses -- must end, for every tile, in exactly the state the serial loop reaches; and the numbers of
passes must be what profiles/r06_g1_tile_study.txt reports (about three for the AGC, between one and two for the timing recovery)."""
import ctypes as C

import numpy as np
import orc


def test_agc_and_timing_recovery_tile_fixed_points_are_the_serial_loops():
    iq, bb, _ = orc.transmit(14, 1, 0, nframes=6, seed=5, esn0_db=12.0, cfo=1e-3, timing=0.3, phase0=0.1, lead_symbols=500)
    L = orc._bind_chain()
    LL = C.POINTER(C.c_longlong)
    L.orc_s2rx_g1_study.argtypes = [C.c_void_p, C.c_int, C.c_int, LL, LL, LL, LL]
    L.orc_s2rx_g1_study.restype = None
    ref = orc.OracleRx(orc.default_cfg(14, 1, 0)).process(iq)
    for agc_tile, gd_tile in ((16, 8), (64, 32)):
        rx = orc.OracleRx(orc.default_cfg(14, 1, 0))
        L.orc_s2rx_g1_study(rx.h, agc_tile, gd_tile, None, None, None, None)
        out = rx.process(iq)
        assert np.array_equal(out, ref)                       # the study leaves the receiver untouched
        h = (C.c_longlong * 66)(); a2 = (C.c_longlong * 2)(); g = (C.c_longlong * 34)(); g5 = (C.c_longlong * 5)()
        L.orc_s2rx_g1_study(rx.h, 0, 0, h, a2, g, g5)
        h = np.array(list(h)); g = np.array(list(g))
        assert h.sum() > 100 and a2[0] == 0 and a2[1] > 1000
        mean = (h * np.arange(66)).sum() / h.sum()
        assert 2.0 <= mean < 5.0 and h[65] == 0, (agc_tile, mean)
        mism, syms, evals, steps, tiles = list(g5)
        assert tiles > 100 and mism == 0 and steps == syms and g[33] == 0
        assert 1.0 <= evals / tiles < 3.0, (gd_tile, evals / tiles)
