#!/usr/bin/env python3
"""Sensitivity of the RESTATED receiver (CPU oracle only -- the GPU engine equals it bit for bit): frames decoded of N vs Es/N0 for
MODCOD 4 (QPSK 1/2 normal), 14 (8PSK 3/4 normal) and 27 (32APSK 8/9 short + pilots), with carrier offset 0 and 1e-3 rad/sample, after a
pre-roll, together with the levels along the chain that decide whether the demapper sees what it was built for: level at the AGC
output, radius of the 1-sps symbols after the RRC filter, and how many LLRs sit at the int8 rails.

    python tools/sensitivity.py [--frames 24] [--preroll 16] [--jobs 8] [--out profiles/r03_sensitivity.json]

What to read off: the implementation loss against the waterfall of an ideal receiver with the same LDPC code, and
whether it comes from the loops (CFO 0 vs 1e-3) or from the level chain (AGC set point 1.0 -> RRC gain -> LUT domain +-0.75 per axis,
constellation.cpp:272-322)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

CASES = {   # modcod: (short, pilots, Es/N0 grid)
    4: (0, 0, [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 8.0]),
    14: (0, 0, [7.0, 8.0, 9.0, 10.0, 11.0, 12.0, 14.0, 16.0]),
    27: (1, 1, [13.0, 14.0, 15.0, 16.0, 17.0, 18.0, 20.0, 24.0]),
}
# Es/N0 at which the code itself reaches quasi-error-free operation on an ideal AWGN receiver (EN 302 307-1 table 13)
IDEAL = {4: 1.00, 14: 7.91, 27: 14.28}


def point(args):
    modcod, short, pilots, esn0, cfo, nframes, preroll, trials = args
    import orc
    total = preroll + nframes
    iq, bb, _ = orc.transmit(modcod, short, pilots, nframes=total, seed=100 + modcod, esn0_db=esn0, cfo=cfo, timing=0.3, phase0=0.1, lead_symbols=500)
    rx = orc.OracleRx(orc.default_cfg(modcod, short, pilots, max_ldpc_trials=trials))
    mp = orc.modcod_params(modcod, short, pilots)
    sent = {bytes(b): i for i, b in enumerate(bb)}
    chunk = 2 * mp['plframe']
    good = seen = 0
    rad, clip, llrn, agc_lvl = [], 0, 0, []
    in_rms = float(np.sqrt(np.mean(np.abs(iq[2000:]) ** 2)))
    for a in range(0, iq.size, chunk):
        out = rx.process(iq[a:a + chunk])
        measured = a >= preroll * chunk
        for x in out:
            k = sent.get(bytes(x))
            if measured:
                seen += 1
                good += k is not None
        if measured:
            s = rx.tap(0)
            if s.size:
                rad.append(float(np.mean(np.abs(s))))
            l = rx.tap(3)
            if l.size:
                clip += int((np.abs(l.astype(np.int16)) >= 127).sum())
                llrn += int(l.size)
            agc_lvl.append(float(rx.L.orc_s2rx_agc_gain(rx.h)) * in_rms)
    return dict(modcod=modcod, esn0_db=esn0, cfo=cfo, frames_sent_after_preroll=nframes, frames_delivered=seen, frames_decoded=good,
                agc_output_rms=round(float(np.mean(agc_lvl)), 4) if agc_lvl else None,
                symbol_radius_1sps=round(float(np.mean(rad)), 4) if rad else None,
                llr_clipped_fraction=round(clip / llrn, 4) if llrn else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=24)
    ap.add_argument('--preroll', type=int, default=16)
    ap.add_argument('--trials', type=int, default=25)
    ap.add_argument('--jobs', type=int, default=8)
    ap.add_argument('--modcods', default='4,14,27')
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r03_sensitivity.json'))
    a = ap.parse_args()
    import orc
    if not hasattr(orc._bind_chain(), 'orc_s2rx_agc_gain'):
        raise SystemExit('liboracle.so is stale: make -C oracle')
    orc._bind_chain().orc_s2rx_agc_gain.restype = __import__('ctypes').c_float
    orc._bind_chain().orc_s2rx_agc_gain.argtypes = [__import__('ctypes').c_void_p]
    work = []
    for m in [int(x) for x in a.modcods.split(',')]:
        short, pilots, grid = CASES[m]
        for cfo in (0.0, 1e-3):
            for e in grid:
                work.append((m, short, pilots, e, cfo, a.frames, a.preroll, a.trials))
    t0 = time.time()
    from multiprocessing import Pool
    with Pool(a.jobs, initializer=_init) as pool:
        res = pool.map(point, work, chunksize=1)
    out = dict(note='CPU oracle (oracle/s2chain.cpp) = what the GPU engine computes bit for bit; timing offset 0.3 sample, 500 lead symbols, '
                    '%d pre-roll frames not counted, %d frames counted per point, max %d LDPC iterations' % (a.preroll, a.frames, a.trials),
               ideal_receiver_esn0_db=IDEAL, points=res, seconds=round(time.time() - t0, 1))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, 'w'), indent=1)
    for m in sorted({r['modcod'] for r in res}):
        print('MODCOD %d (ideal receiver: %.2f dB)' % (m, IDEAL[m]))
        print('  Es/N0   cfo=0: decoded/N  AGC out  radius  LLR@rail | cfo=1e-3: decoded/N  AGC out  radius  LLR@rail')
        for e in CASES[m][2]:
            r0 = next(r for r in res if r['modcod'] == m and r['esn0_db'] == e and r['cfo'] == 0.0)
            r1 = next(r for r in res if r['modcod'] == m and r['esn0_db'] == e and r['cfo'] != 0.0)
            print('  %5.1f   %9s %8s %7s %8s | %13s %8s %7s %8s' % (
                e, '%d/%d' % (r0['frames_decoded'], a.frames), r0['agc_output_rms'], r0['symbol_radius_1sps'], r0['llr_clipped_fraction'],
                '%d/%d' % (r1['frames_decoded'], a.frames), r1['agc_output_rms'], r1['symbol_radius_1sps'], r1['llr_clipped_fraction']))


def _init():
    import orc
    import ctypes
    L = orc._bind_chain()
    L.orc_s2rx_agc_gain.restype = ctypes.c_float
    L.orc_s2rx_agc_gain.argtypes = [ctypes.c_void_p]


if __name__ == '__main__':
    main()
