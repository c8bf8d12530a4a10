"""A poisoned input -- NaN / infinite samples -- must neither hang the GPU nor overrun anything: every serial loop of the receive chains has a bounded
trip count whatever its state holds (the written-out loops of round 4 keep the bounds of the C++ forms they replace).  What comes out of such a call is
not specified; what is checked: the call returns, a reset brings the receiver back, and a clean signal decodes as before."""
import numpy as np
import pytest
import orc
import orc_dvbs as od

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def _poison(iq, rng):
    x = iq.copy()
    n = x.size
    x[n // 7] = np.nan
    x[n // 3:n // 3 + 50] = complex(np.inf, -np.inf)
    x[n // 2] = complex(0.0, np.nan)
    x[(2 * n) // 3:(2 * n) // 3 + 200] *= 1e30
    idx = rng.integers(0, n, 40)
    x[idx] = np.nan
    return x


@pytest.mark.parametrize('carriers', [1, 70])
def test_dvbs_receiver_survives_nan_and_inf(engine, pkg, carriers):
    import torch
    rng = np.random.default_rng(3)
    iq, _ = od.dvbs_iq(0, 30000, seed=4, esn0_db=12.0, cfo=1e-3, timing=0.3)
    bad = _poison(iq, rng)
    bank = pkg.DvbsDemodBank(engine, carriers, max_samples=iq.size)
    dev = torch.device('cuda')
    outs = [torch.zeros(iq.size + 4 * 8192, dtype=torch.uint8, device=dev) for _ in range(carriers)]
    d_bad, d_ok = torch.from_numpy(bad).to(dev), torch.from_numpy(iq).to(dev)
    for _ in range(2):
        bank.process_batch([d_bad for _ in range(carriers)], outs)          # returns: that is the test
    torch.cuda.synchronize()
    ref = pkg.DvbsDemodBank(engine, 1, max_samples=iq.size)
    o1 = [torch.zeros(iq.size + 4 * 8192, dtype=torch.uint8, device=dev)]
    n_ref = [ref.process_batch([d_ok], o1)[0] for _ in range(2)]
    bank.reset()
    n_new = [bank.process_batch([d_ok for _ in range(carriers)], outs) for _ in range(2)]
    assert [x[0] for x in n_new] == n_ref and [x[-1] for x in n_new] == n_ref
    assert torch.equal(outs[0][:n_ref[1]], o1[0][:n_ref[1]]) and torch.equal(outs[-1][:n_ref[1]], o1[0][:n_ref[1]])
    bank.close(); ref.close()


@pytest.mark.parametrize('streams', [1, 9, 300])
def test_s2_receiver_survives_nan_and_inf(engine, pkg, streams):
    import torch
    rng = np.random.default_rng(5)
    iq, bbs, _ = orc.transmit(4, 1, 0, nframes=14, seed=21, esn0_db=22.0, cfo=1e-4, timing=0.2, phase0=0.3)
    bad = _poison(iq, rng)
    dev = torch.device('cuda')
    dms = [engine.demod(engine.default_cfg(4, True, False), max_samples=iq.size) for _ in range(streams)]
    kb = pkg.modcod_info(4, True, False)['kbch'] // 8
    outs = [torch.zeros(20 * kb, dtype=torch.uint8, device=dev) for _ in range(streams)]
    d_bad, d_ok = torch.from_numpy(bad).to(dev), torch.from_numpy(iq).to(dev)
    for pipelined in (False, True):
        engine.set_pipelined(pipelined)
        try:
            for _ in range(2):
                engine.process_batch(dms, [d_bad for _ in range(streams)], outs)      # returns: that is the test
            engine.process_batch(dms, [torch.empty(0, dtype=torch.complex64, device=dev) for _ in range(streams)], outs)
        finally:
            engine.set_pipelined(False)
    torch.cuda.synchronize()
    for d in dms:
        d.reset()
    nb = engine.process_batch(dms, [d_ok for _ in range(streams)], outs)
    fresh = engine.demod(engine.default_cfg(4, True, False), max_samples=iq.size)
    o1 = [torch.zeros(20 * kb, dtype=torch.uint8, device=dev)]
    n1 = engine.process_batch([fresh], [d_ok], o1)[0]
    sent = {bytes(b) for b in bbs}
    fr1 = o1[0][:n1].cpu().numpy().reshape(-1, kb)
    assert len(fr1) >= 10 and sum(bytes(f) in sent for f in fr1) >= len(fr1) - 2            # (the first frame or two come out of the acquisition)
    for s in (0, streams - 1):
        assert nb[s] == n1 and torch.equal(outs[s][:n1], o1[0][:n1]), s                     # a reset receiver = a new one
    fresh.close()
    for d in dms:
        d.close()
