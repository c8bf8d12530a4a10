import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, orc
import __graft_entry__ as g
pkg = g.load_package(); eng = pkg.Engine(0)
modcod, short, pilots, esn0, nframes, chunk = 14, 1, 0, 16.0, 8, 20000
iq, bb, _ = orc.transmit(modcod, short, pilots, nframes=nframes, seed=modcod, esn0_db=esn0, cfo=1e-3, timing=0.3, phase0=0.1, lead_symbols=700)
rx = orc.OracleRx(orc.default_cfg(modcod, short, pilots))
dm = eng.demod(eng.default_cfg(modcod, bool(short), bool(pilots)), max_samples=chunk)
for a in range(0, iq.size, chunk):
    part = iq[a:a+chunk]
    o = rx.process(part); gg = dm.process(part)
    so, sg = rx.tap(0), dm.tap(0)
    e = np.abs(so - sg)
    print('call', a // chunk, 'nsym', so.size, 'sym err max %.2e' % (e.max() if e.size else 0), 'frac>1e-4 %.4f' % ((e > 1e-4).mean() if e.size else 0), 'nco', rx.L.orc_s2rx_nco_freq(rx.h), dm.nco_freq())
    po, pg = rx.tap(2), dm.tap(2)
    if po.size:
        pl = rx.mp['plframe']
        for f in range(po.size // pl):
            x, y = po[f*pl:(f+1)*pl], pg[f*pl:(f+1)*pl]
            fo, fg = rx.tap(1)[f*pl:(f+1)*pl], dm.tap(1)[f*pl:(f+1)*pl]
            ang = np.angle(y * np.conj(x))
            blocks = [float(np.abs(ang[i:i+600]).max()) for i in range(0, pl, 600)]
            print('   frame', f, 'in err max %.2e' % np.abs(fo - fg).max(), 'pll angle diff per block', ' '.join('%.1e' % b for b in blocks))
