"""Development aid: config D of bench.py alone (DVB-S QPSK 1/2 receiver bank: 4096 / 64 / 1 carriers), without its CPU leg."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import torch
    import __graft_entry__ as g
    import orc_dvbs as od
    pkg = g.load_package()
    eng = pkg.Engine(0)
    dev = torch.device('cuda:0')
    nsym = 65536
    iq, _ = od.dvbs_iq(0, nsym, seed=1, esn0_db=12.0, cfo=1e-3, timing=0.3)
    for S in [int(a) for a in (sys.argv[1:] or ['4096', '64', '1'])]:
        bank = pkg.DvbsDemodBank(eng, S, max_samples=iq.size)
        d_iq = torch.from_numpy(iq).to(dev)
        tin = [d_iq for _ in range(S)]
        tout = [torch.zeros(iq.size + 4 * 8192, dtype=torch.uint8, device=dev) for _ in range(S)]
        bank.process_batch(tin, tout)
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            bank.process_batch(tin, tout)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        st = bank.stats()[0]
        print('%d carriers: %.2f ms per %d symbols = %.2f Msym/s  locked=%s' % (S, dt * 1e3, nsym, S * nsym / dt / 1e6, st.state == 1 and st.rate == 0))
        bank.close()


if __name__ == '__main__':
    main()
