"""CPU tests of the DVB-S tail oracle (oracle/dvbs_tail.cpp), pinned against the reference's own deframer, libcorrect
Reed-Solomon decoder + DVBSReedSolomon wrapper and descrambler compiled in place (oracle/_ref)."""
import ctypes as C
import numpy as np
import pytest
import orc
import orc_dvbs_tail as ot
from orc_dvbs import P, VP


def _ref():
    r = orc.ref()
    if r is None or not hasattr(r, 'ref_tsdef_create'):
        pytest.skip('oracle/_ref not built')
    for n in ('ref_tsdef_create', 'ref_dvbsrs_create', 'ref_dvbsdescr_create'):
        getattr(r, n).restype = VP
    r.ref_tsdef_work.argtypes = [VP, VP, C.c_int, VP]
    r.ref_dvbsrs_decode.argtypes = [VP, VP]
    r.ref_dvbsdescr_work.argtypes = [VP, VP]
    return r


def test_rs_wrapper_matches_reference_incl_failures_and_stale_output():
    o, r = ot.L(), _ref()
    rng = np.random.default_rng(1)
    ho, hr = VP(o.orc_dvbsrs_create()), VP(r.ref_dvbsrs_create())
    nerr_seq = [0, 1, 3, 8, 9, 12, 0, 8, 20, 40, 2, 9, 9, 0, 7, 8, 8, 10, 11, 0]
    for rep in range(6):
        for ne in nerr_seq:
            msg = rng.integers(0, 256, 188, dtype=np.uint8)
            cw = ot.rs_encode_204(msg)
            pos = rng.choice(204, ne, replace=False)
            cw[pos] ^= rng.integers(1, 256, ne, dtype=np.uint8)
            a, b = cw.copy(), cw.copy()
            ea, eb = o.orc_dvbsrs_decode(ho, P(a)), r.ref_dvbsrs_decode(hr, P(b))
            assert ea == eb and np.array_equal(a, b), (rep, ne, ea, eb)
            if ne <= 8:
                assert np.array_equal(a[:188], msg)
    # parity-only and last-parity-byte errors (location 0 / log aliasing)
    for pos in ([203], [188], [203, 0], [202, 203, 5]):
        msg = rng.integers(0, 256, 188, dtype=np.uint8)
        cw = ot.rs_encode_204(msg)
        cw[pos] ^= 0x5a
        a, b = cw.copy(), cw.copy()
        assert o.orc_dvbsrs_decode(ho, P(a)) == r.ref_dvbsrs_decode(hr, P(b)) and np.array_equal(a, b)
        assert np.array_equal(a[:188], msg)
    o.orc_dvbsrs_destroy(ho)


def test_own_rs_encoder_produces_codewords():
    o = ot.L()
    rng = np.random.default_rng(2)
    for _ in range(5):
        msg = rng.integers(0, 256, 188, dtype=np.uint8)
        cw = ot.rs_encode_204(msg)
        enc = np.concatenate([np.zeros(51, np.uint8), cw])
        out = np.zeros(239, np.uint8)
        assert o.orc_rs255_decode(P(enc), P(out)) == 239 and np.array_equal(out[51:], msg)


def test_descrambler_matches_reference():
    o, r = ot.L(), _ref()
    rng = np.random.default_rng(3)
    ho, hr = VP(o.orc_dvbsdescr_create()), VP(r.ref_dvbsdescr_create())
    for rep in range(12):
        frm = rng.integers(0, 256, 1632, dtype=np.uint8)
        for k in range(8):
            frm[204 * k] = 0xB8 if (rep % 3 != 0 and k == (rep % 8)) else 0x47
        a, b = frm.copy(), frm.copy()
        o.orc_dvbsdescr_work(ho, P(a)); r.ref_dvbsdescr_work(hr, P(b))
        assert np.array_equal(a, b), rep
    o.orc_dvbsdescr_destroy(ho)


def test_deframer_matches_reference_and_finds_frames():
    o, r = ot.L(), _ref()
    bits, ts = ot.dvbs_outer_tx(48, seed=4)
    rng = np.random.default_rng(5)
    stream = np.concatenate([rng.integers(0, 2, 777, dtype=np.uint8), bits, 1 - bits[:1632 * 8 * 2], rng.integers(0, 2, 500, dtype=np.uint8)])
    flips = rng.random(stream.size) < 0.002
    stream = (stream ^ flips).astype(np.uint8)
    ho, hr = VP(o.orc_tsdef_create()), VP(r.ref_tsdef_create())
    pos, total = 0, 0
    for chunk in (4096, 4096, 13056, 5000, 60000, 1, 7, stream.size):
        seg = np.ascontiguousarray(stream[pos:pos + chunk])
        if seg.size == 0:
            break
        oa, ob = np.zeros(1632 * 64, np.uint8), np.zeros(1632 * 64, np.uint8)
        ea = np.zeros(2, np.int32)
        na = o.orc_tsdef_work(ho, P(seg), seg.size, P(oa), P(ea))
        nb = r.ref_tsdef_work(hr, P(seg), seg.size, P(ob))
        assert na == nb and np.array_equal(oa[:1632 * na], ob[:1632 * nb]), (pos, na, nb)
        total += na
        pos += chunk
    assert total >= 5 + 1          # 6 normal-polarity frames minus edge effects + inverted ones
    o.orc_tsdef_destroy(ho)


def test_full_tail_recovers_ts_packets():
    """deframer -> Forney de-interleaver -> RS -> descrambler on the output of the numpy transmitter"""
    import orc_dvbs as od
    o, ol = ot.L(), od.L()
    bits, ts = ot.dvbs_outer_tx(64, seed=6)
    rng = np.random.default_rng(7)
    bits = (bits ^ (rng.random(bits.size) < 0.001)).astype(np.uint8)
    hd, hf, hr, hs = VP(o.orc_tsdef_create()), VP(ol.orc_forney_create()), VP(o.orc_dvbsrs_create()), VP(o.orc_dvbsdescr_create())
    frames = np.zeros(1632 * 16, np.uint8)
    nf = o.orc_tsdef_work(hd, P(bits), bits.size, P(frames), None)
    assert nf == 8
    got = []
    for k in range(nf):
        f = np.ascontiguousarray(frames[1632 * k:1632 * (k + 1)])
        d = np.zeros(1632, np.uint8)
        ol.orc_forney_deinterleave(hf, P(f), P(d))
        for i in range(8):
            pkt = np.ascontiguousarray(d[204 * i:204 * (i + 1)])
            o.orc_dvbsrs_decode(hr, P(pkt))
            d[204 * i:204 * i + 188] = pkt[:188]
        o.orc_dvbsdescr_work(hs, P(d))
        got.append(np.stack([d[204 * i:204 * i + 188] for i in range(8)]))
    got = np.concatenate(got)
    # the de-interleaver needs 11 packets of history: later packets are the transmitted TS packets
    hits = 0
    for g in got[16:]:
        hits += any(np.array_equal(g, t) for t in ts)
    assert hits >= len(got[16:]) - 1, (hits, len(got))


def test_tail_golden_vectors():
    """committed vectors generated from the reference (tests/golden/make_golden_dvbs.py)"""
    import json, os, hashlib
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'dvbs_golden.json')) as f:
        G = json.load(f)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    o = ot.L()
    rng = np.random.default_rng(G['rs']['seed'])
    h = VP(o.orc_dvbsrs_create())
    outs, errs = [], []
    for ne in [0, 1, 3, 8, 9, 12, 0, 8, 20, 40, 2, 9, 9, 0, 7, 8, 8, 10, 11, 0] * 2:
        msg = rng.integers(0, 256, 188, dtype=np.uint8)
        cw = ot.rs_encode_204(msg)
        pos = rng.choice(204, ne, replace=False)
        cw[pos] ^= rng.integers(1, 256, ne, dtype=np.uint8)
        errs.append(int(o.orc_dvbsrs_decode(h, P(cw))))
        outs.append(cw.copy())
    assert errs == G['rs']['errors'] and sha(np.concatenate(outs)) == G['rs']['sha']
    rng = np.random.default_rng(G['descramble']['seed'])
    h = VP(o.orc_dvbsdescr_create())
    outs = []
    for rep in range(12):
        frm = rng.integers(0, 256, 1632, dtype=np.uint8)
        for k in range(8):
            frm[204 * k] = 0xB8 if (rep % 3 != 0 and k == (rep % 8)) else 0x47
        o.orc_dvbsdescr_work(h, P(frm))
        outs.append(frm.copy())
    assert sha(np.concatenate(outs)) == G['descramble']['sha']
    bits, _ = ot.dvbs_outer_tx(48, seed=G['deframer']['seeds'][0])
    rng = np.random.default_rng(G['deframer']['seeds'][1])
    stream = np.concatenate([rng.integers(0, 2, 777, dtype=np.uint8), bits, 1 - bits[:1632 * 8 * 2]])
    stream = (stream ^ (rng.random(stream.size) < 0.002)).astype(np.uint8)
    h = VP(o.orc_tsdef_create())
    out = np.zeros(1632 * 32, np.uint8)
    nf = o.orc_tsdef_work(h, P(stream), int(stream.size), P(out), None)
    assert nf == G['deframer']['frames'] and sha(out[:1632 * nf]) == G['deframer']['sha']
