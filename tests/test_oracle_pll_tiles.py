"""CPU-only: the parallel-in-time form of the payload PLL studied in tools/pll_tile_study.py (oracle/s2chain.cpp: pll_tile_study) -- evaluating every symbol's phase
error from guessed loop phases, replaying the recurrence of dvbs2_pll.cpp:81 over the tile, repeating to the fixed point -- must end, for every tile, in exactly the
loop state the serial loop reaches (that is what makes the scheme bit-exact by construction); and it needs more than one pass per tile (the errors do depend on the
phases), fewer than the tile has symbols."""
import ctypes as C

import numpy as np
import orc


def test_pll_tile_fixed_point_is_the_serial_loop():
    iq, bb, _ = orc.transmit(14, 1, 0, nframes=6, seed=5, esn0_db=12.0, cfo=1e-3, timing=0.3, phase0=0.1, lead_symbols=500)
    L = orc._bind_chain()
    L.orc_s2rx_pll_study.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.orc_s2rx_pll_study.restype = None
    ref = orc.OracleRx(orc.default_cfg(14, 1, 0)).process(iq)
    for tile in (16, 64):
        rx = orc.OracleRx(orc.default_cfg(14, 1, 0))
        L.orc_s2rx_pll_study(rx.h, tile, None, None)
        out = rx.process(iq)
        assert np.array_equal(out, ref)                       # the study leaves the receiver untouched
        hist = (C.c_longlong * 34)(); mis = C.c_longlong(-1)
        L.orc_s2rx_pll_study(rx.h, 0, hist, C.byref(mis))
        h = np.array(list(hist))
        assert h.sum() > 100 and mis.value == 0
        mean = (h * np.arange(34)).sum() / h.sum()
        assert 1.5 < mean < tile / 2 and h[33] == 0, (tile, mean, list(h))
