import sys, os, importlib
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
pkg = importlib.import_module('sdrpp-dvbs-demodulator_amd')
import orc_dvbs as od, orc_dvbs_tail as ot
from orc_dvbs import P
eng = pkg.Engine()
npk = 1400
obits, ts = ot.dvbs_outer_tx(npk, seed=61)
enc = od.cc_encode(obits)
nsym = enc.size // 2
iq = np.zeros(2 * nsym, np.complex64)
od.LF().orc_dvbs_modulate(P(np.ascontiguousarray(enc)), nsym, 12.0, 1e-4, 0.3, 0.2, 7, P(iq))
nseg, own, warm = 4, 49152, 32768
rx = pkg.DvbsSegmentReceiver(eng, nseg, own, warm)
d_iq = torch.from_numpy(iq).cuda()
bits = torch.zeros(2 * nseg * own * 2 + 4 * 65536, dtype=torch.uint8, device='cuda')
allb = []
for a in range(0, iq.size, rx.chunk_samples):
    nb = rx.process(d_iq[a:a + rx.chunk_samples], bits)
    print(a, nb, rx.stats())
    allb.append(bits[:nb].cpu().numpy().copy())
got = np.concatenate(allb)
ref = obits.astype(np.uint8)
# locate blocks
pos = 0
rb = ref.tobytes()
for a in range(0, got.size - 4096, 20000):
    k = rb.find(got[a:a+256].tobytes()); ki = rb.find((got[a:a+256]^1).tobytes())
    print(a, k - a if k >= 0 else None, ki - a if ki >= 0 else None)
