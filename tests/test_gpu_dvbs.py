"""GPU parity tests for the DVB-S inner-code path (rows a18-a20), through the C ABI, against the CPU oracle (bit-exact)."""
import numpy as np
import pytest
import orc_dvbs as od
from orc_dvbs import P, VP

pytestmark = pytest.mark.gpu


def test_slice_matches_oracle(engine, pkg):
    import torch
    rng = np.random.default_rng(1)
    n = 8192 * 3 // 2
    iq = rng.normal(0, 0.9, 2 * n).astype(np.float32)
    iq[:10] = [1.27, -1.27, 1.2700001, -1.2700001, 1.28, -1.28, 0.0099999, -0.0099999, 50.0, -50.0]
    soft = pkg.dvbs_slice(engine, torch.from_numpy(iq).view(torch.complex64).cuda()).cpu().numpy()
    o = od.L()
    h = VP(o.orc_dvbs_slicer_create())
    exp = np.zeros(3 * 8192, np.int8)
    got_n = o.orc_dvbs_slicer_process(h, n, P(iq), P(exp))
    o.orc_dvbs_slicer_destroy(h)
    assert got_n == 3 * 8192 and (soft == exp).all()
    assert pkg.dvbs_slice(engine, torch.zeros(0, dtype=torch.complex64, device='cuda')).numel() == 0


@pytest.mark.parametrize('frame', [54, 56, 57, 59, 114, 1024, 1366, 1699, 4096, 6799])     # (frame + 6 steps in chunks of 60 = 10 rotations of the state layout: every remainder 0..5, a chunk that is exactly full)
def test_ccdec_chained_blocks_bit_exact(engine, pkg, frame):
    import torch
    rng = np.random.default_rng(frame)
    S, nblk = 6, 4
    stride = 2 * frame                      # consecutive blocks overlap by the 12-byte tail, like the reference's buffers
    Lb = stride * nblk + 64
    soft = np.zeros((S, Lb), np.uint8)
    for s in range(S):
        bits = rng.integers(0, 2, frame * nblk + 64, dtype=np.uint8)
        enc = od.cc_encode(bits)[:Lb]
        sigma = [10, 25, 40, 60, 90, 1e-3][s]
        x = np.where(enc > 0, 127 + 35, 127 - 35) + rng.normal(0, sigma, Lb)
        x = np.clip(np.rint(x), 0, 255).astype(np.uint8)
        if s == 4:
            x[rng.random(Lb) < 0.3] = 128   # erasures
        soft[s] = x
    soft[5] = rng.integers(0, 256, Lb, dtype=np.uint8)   # pure garbage: exercises the uint8 wrap-around of the metrics
    dec = pkg.CcDecoderBatch(engine, S, frame)
    got = dec.work(torch.from_numpy(soft).cuda(), nblk, stride).cpu().numpy()
    got2 = dec.work(torch.from_numpy(soft).cuda(), nblk, stride).cpu().numpy()   # second call continues the chain
    o = od.L()
    for s in range(S):
        h = VP(o.orc_ccdec_create(frame))
        for rep, g in ((0, got), (1, got2)):
            for b in range(nblk):
                out = np.zeros(frame, np.uint8)
                o.orc_ccdec_work(h, P(soft[s, stride * b:]), P(out))
                assert (g[s, b] == out).all(), (s, rep, b, int((g[s, b] != out).sum()))
        o.orc_ccdec_destroy(h)
    dec.close()


def _viterbi_case_streams(nb):
    """list of (name, soft int8 [nb, 8192])"""
    rng = np.random.default_rng(77)
    cases = []
    for rate in range(5):
        for drop, rot in ((0, False), (1, True), (3, False)):
            if rate in (2, 4):
                drop = 2 * (drop > 0)
            soft, _ = od.dvbs_tx(rate, nb * 8192, seed=300 + 10 * rate + drop + rot, drop=drop, rot90=rot, sigma=12.0)
            cases.append((f'rate{rate}_d{drop}_r{int(rot)}', soft.reshape(nb, 8192)))
    cases.append(('noise', rng.integers(-70, 71, (nb, 8192)).astype(np.int8)))
    cases.append(('full_range_noise', rng.integers(-128, 128, (nb, 8192)).astype(np.int8)))
    # marginal SNR: BER hovers around the threshold, watchdog counting matters
    soft, _ = od.dvbs_tx(0, nb * 8192, seed=501, sigma=30.0)
    cases.append(('marginal_12', soft.reshape(nb, 8192)))
    soft, _ = od.dvbs_tx(4, nb * 8192, seed=502, sigma=17.0)
    cases.append(('marginal_78', soft.reshape(nb, 8192)))
    # signal, then noise, then a different rate: lock -> watchdog -> IDLE -> re-lock
    a, _ = od.dvbs_tx(1, nb * 8192, seed=503)
    b, _ = od.dvbs_tx(3, nb * 8192, seed=504, rot90=True)
    x = a.reshape(nb, 8192).copy()
    x[2:5] = rng.integers(-70, 71, (3, 8192))
    x[5:] = b.reshape(nb, 8192)[5:]
    cases.append(('relock', x))
    return cases


def _check_viterbi(names, soft, gb, gn, gs, thr, max_outsync):
    for s, name in enumerate(names):
        v = od.OracleViterbi(thr, max_outsync)
        eb, en, es = v.work(soft[s])
        assert (gn[s] == en).all(), (name, gn[s], en)
        assert (gs[s] == es).all(), (name, gs[s], es)
        for b in range(soft.shape[1]):
            n = int(en[b])
            if es[b, 2] == 3:
                n = min(n, 6799)   # rate 5/6: the reference's decoder leaves [6799, nbits) unwritten
            assert (gb[s, b, :n] == eb[b, :n]).all(), (name, b, int((gb[s, b, :n] != eb[b, :n]).sum()))


def test_viterbi_dvbs_bit_exact(engine, pkg):
    import torch
    nb = 8
    cases = _viterbi_case_streams(nb)
    soft = np.stack([c[1] for c in cases])
    vit = pkg.ViterbiBatch(engine, len(cases), 0.15, 3)     # short watchdog so the relock case goes through IDLE again
    d = torch.from_numpy(soft).cuda()
    # two calls of 3 + 5 blocks: state must carry across calls exactly like one call of 8
    gb1, gn1, gs1 = vit.work(d[:, :3].contiguous())
    gb2, gn2, gs2 = vit.work(d[:, 3:].contiguous())
    gb = torch.cat([gb1, gb2], 1).cpu().numpy(); gn = torch.cat([gn1, gn2], 1).cpu().numpy(); gs = torch.cat([gs1, gs2], 1).cpu().numpy()
    _check_viterbi([c[0] for c in cases], soft, gb, gn, gs, 0.15, 3)
    # sanity on the scenario itself: every coded stream locked to its rate, noise never did
    for s, (name, _) in enumerate(cases):
        if name.startswith('rate'):
            assert (gs[s, :, 1] == 1).all() and (gs[s, :, 2] == int(name[4])).all(), name
        if 'noise' in name:
            assert (gn[s] == 0).all()
    relock = gs[[c[0] for c in cases].index('relock')]
    assert relock[0, 2] == 1 and relock[-1, 2] == 3 and 0 in relock[:, 1]
    # reset brings the handle back to a fresh decoder
    vit.reset()
    gb3, gn3, gs3 = vit.work(d[:, :3].contiguous())
    assert (gn3.cpu().numpy() == gn[:, :3]).all() and (gs3.cpu().numpy() == gs[:, :3]).all()
    vit.close()


def test_viterbi_dvbs_default_watchdog(engine, pkg):
    import torch
    nb = 4
    soft = np.stack([od.dvbs_tx(r, nb * 8192, seed=600 + r, sigma=20.0)[0].reshape(nb, 8192) for r in range(5)])
    vit = pkg.ViterbiBatch(engine, 5)
    gb, gn, gs = [x.cpu().numpy() for x in vit.work(torch.from_numpy(soft).cuda())]
    _check_viterbi([f'r{r}' for r in range(5)], soft, gb, gn, gs, 0.15, 20)
    vit.close()


def test_viterbi_many_streams_identical(engine, pkg):
    """size-independent property at scale: 1024 streams fed the same blocks produce identical output"""
    import torch
    soft, bits = od.dvbs_tx(2, 2 * 8192, seed=9)
    S = 1024
    d = torch.from_numpy(soft.reshape(1, 2, 8192)).cuda().repeat(S, 1, 1).contiguous()
    vit = pkg.ViterbiBatch(engine, S)
    gb, gn, gs = vit.work(d)
    assert bool((gb == gb[0:1]).all()) and bool((gn == gn[0:1]).all()) and bool((gs == gs[0:1]).all())
    g0 = gb[0, 0, :6144].cpu().numpy()
    assert (g0[:6000] == bits[:6000]).mean() > 0.999
    vit.close()


def test_forney_bit_exact_across_calls(engine, pkg):
    import torch
    rng = np.random.default_rng(3)
    S = 5
    f = pkg.ForneyBatch(engine, S)
    o = od.L()
    hs = [VP(o.orc_forney_create()) for _ in range(S)]
    for groups in (1, 3, 1, 7):
        x = rng.integers(0, 256, (S, groups * 1632), dtype=np.uint8)
        got = f.deinterleave(torch.from_numpy(x).cuda()).cpu().numpy()
        for s in range(S):
            for g in range(groups):
                e = np.zeros(1632, np.uint8)
                o.orc_forney_deinterleave(hs[s], P(np.ascontiguousarray(x[s, g * 1632:(g + 1) * 1632])), P(e))
                assert (got[s, g * 1632:(g + 1) * 1632] == e).all(), (groups, s, g)
    for h in hs:
        o.orc_forney_destroy(h)
    f.close()


def test_dvbs_error_codes(engine, pkg):
    import ctypes as C
    lib = engine.lib
    h = C.c_void_p()
    assert lib.dvbs2gpu_viterbi_create(engine.h, 0, 0.15, 20, C.byref(h)) == pkg.ERR_ARG
    assert lib.dvbs2gpu_ccdec_create(engine.h, 4, 2, C.byref(h)) == pkg.ERR_ARG
    assert lib.dvbs2gpu_forney_create(None, 4, C.byref(h)) == pkg.ERR_ARG
    f = pkg.ForneyBatch(engine, 1)
    assert lib.dvbs2gpu_forney_deinterleave_batch(f.h, None, 1632, None, None) == pkg.ERR_ARG
    assert lib.dvbs2gpu_forney_deinterleave_batch(f.h, None, 7, None, None) == pkg.ERR_ARG
    assert lib.dvbs2gpu_forney_deinterleave_batch(f.h, None, 0, None, None) == 0
    f.close()
