"""DVB-S inner-code micro-benchmark on the GPU: Viterbi_DVBS batch throughput per rate (HIP events)."""
import argparse
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--streams', type=int, default=4096)
    ap.add_argument('--blocks', type=int, default=8)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g
    import orc_dvbs as od
    pkg = g.load_package()
    eng = pkg.Engine(0)
    S, nb = args.streams, args.blocks
    for rate in range(5):
        soft, _ = od.dvbs_tx(rate, nb * 8192, seed=rate, sigma=15.0)
        d = torch.from_numpy(soft.reshape(1, nb, 8192)).cuda().repeat(S, 1, 1).contiguous()
        vit = pkg.ViterbiBatch(eng, S)
        vit.work(d[:, :1].contiguous())          # acquisition block (52 trial decodes per stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        bits, nbits, stats = vit.work(d)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        assert int(stats[:, :, 1].min()) == 1 and int(stats[:, :, 2].min()) == rate
        nsym = S * nb * 4096
        print('rate %s: %d streams x %d blocks in %.2f ms = %.1f Msym/s, %.2f Gbit/s decoded' %
              (od.RATE_NAMES[rate], S, nb, ms, nsym / ms / 1e3, float(nbits.sum()) / ms / 1e6))
        vit.reset()
        torch.cuda.synchronize()
        e0.record()
        vit.work(d[:, :1].contiguous())
        e1.record()
        torch.cuda.synchronize()
        print('   acquisition block (IDLE search + first decode): %.2f ms' % e0.elapsed_time(e1))
        vit.close()




def demod_bench():
    """whole DVB-S receive path (front end + slicer + Viterbi), BASELINE config D shape: rate 1/2 QPSK at 2 sps"""
    import time
    import torch
    import __graft_entry__ as g
    import orc_dvbs as od
    pkg = g.load_package()
    eng = pkg.Engine(0)
    nsym = 65536
    iq, _ = od.dvbs_iq(0, nsym, seed=1, esn0_db=12.0, cfo=1e-3, timing=0.3)
    for S in [int(x) for x in os.environ.get("DVBS_BANK_STREAMS", "1,64,1024").split(",")]:
        bank = pkg.DvbsDemodBank(eng, S, max_samples=iq.size)
        tin = [torch.from_numpy(iq).cuda() for _ in range(S)]
        tout = [torch.zeros(iq.size + 4 * 8192, dtype=torch.uint8, device='cuda') for _ in range(S)]
        bank.process_batch(tin, tout)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            nb = bank.process_batch(tin, tout)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        st = bank.stats()[0]
        print('DVB-S bank: %d streams x %d symbols per call: %.2f ms = %.2f Msym/s total, %.3f Msym/s per stream (lock %d rate %d ber %.3f, %d bits out)'
              % (S, nsym, dt * 1e3, S * nsym / dt / 1e6, nsym / dt / 1e6, st.state, st.rate, st.ber, nb[0]))
        if os.environ.get('DVBS_WITH_TAIL'):      # the whole of DVBSDemod::process: + TS deframer, Forney, RS(204,188), energy dispersal
            import orc_dvbs_tail as ot
            ncall = 5
            obits, ts = ot.dvbs_outer_tx(ncall * nsym // 1632 + 2, seed=3)          # a continuous outer-coded stream, one chunk per call
            enc = od.cc_encode(obits)
            ns2 = enc.size // 2
            iq2 = np.zeros(2 * ns2, np.complex64)
            od.LF().orc_dvbs_modulate(od.P(np.ascontiguousarray(enc)), ns2, 12.0, 1e-3, 0.3, 0.2, 7, od.P(iq2))
            d2 = torch.from_numpy(iq2).cuda()
            bank.reset()
            tail = pkg.DvbsTailBank(eng, S, max_bits=2 * nsym + 4 * 8192)
            tts = [torch.zeros(188 * 8 * 16, dtype=torch.uint8, device='cuda') for _ in range(S)]
            def step(k):
                part = d2[2 * nsym * k:2 * nsym * (k + 1)]
                nb = bank.process_batch([part] * S, tout)
                return tail.process_batch([tout[i][:nb[i]] for i in range(S)], tts)
            step(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tot = 0
            for k in range(1, ncall):
                nby = step(k)
                tot += nby[S - 1]
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t0) / (ncall - 1)
            sent = {bytes(t) for t in ts}
            last = tts[S - 1][:nby[S - 1]].cpu().numpy().reshape(-1, 188)
            print('   with the tail (IQ -> TS packets, continuous outer-coded stream): %.2f ms per call = %.2f Msym/s total, %d TS packets per stream in %d calls, last call: %d of %d are transmitted ones'
                  % (dt2 * 1e3, S * nsym / dt2 / 1e6, tot // 188, ncall - 1, sum(bytes(x) in sent for x in last), len(last)))
            tail.close()
        bank.close()


if __name__ == '__main__':
    if os.environ.get('DVBS_DEMOD_BENCH'):
        demod_bench()
    else:
        main()
