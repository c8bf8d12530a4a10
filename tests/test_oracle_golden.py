"""The CPU oracle against the committed golden vectors (tests/golden/, produced by the REFERENCE's own FEC
code -- see tests/golden/make_golden.py) and, when oracle/_ref is built, against the reference directly."""
import hashlib
import json
import os

import numpy as np
import pytest

import orc

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, 'golden', 'fec_golden.json')))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ldpc_input(rate, short, seed, snr_db):
    rng = np.random.default_rng(seed)
    _, bits = orc.encode_frame(rate, short, seed)
    return orc.bits_to_llr(bits, snr_db, rng)


@pytest.mark.parametrize('case', G['ldpc'], ids=lambda c: '%s%s-%s-%d' % ('C' if c['short'] else 'B', c['rate'] + 1, c['kind'], c['trials']))
def test_ldpc_golden(case):
    llr = ldpc_input(case['rate'], case['short'], case['seed'], case['snr_db'])
    assert sha(llr) == case['in_sha'], 'input generator drifted; regenerate the goldens'
    ret = orc.lib().orc_ldpc_decode(case['rate'], case['short'], llr, case['trials'], 0)
    assert ret == case['ret']
    assert sha(llr) == case['out_sha']


def test_ldpc_golden_full_vector():
    c = G['ldpc_full']
    llr = np.load(os.path.join(HERE, 'golden', c['infile']))
    want = np.load(os.path.join(HERE, 'golden', c['outfile']))
    ret = orc.lib().orc_ldpc_decode(c['rate'], c['short'], llr, c['trials'], 0)
    assert ret == c['ret'] and np.array_equal(llr, want)


@pytest.mark.parametrize('case', G['bch'], ids=lambda c: 'r%d-s%d-e%d' % (c['rate'], c['short'], len(c['errors'])))
def test_bch_golden(case):
    p = orc.fec_params(case['rate'], case['short'])
    fr = np.zeros(p['K'] // 8, np.uint8)
    orc.lib().orc_make_bbframe(fr, p['kbch'], case['frame_seed'])
    orc.lib().orc_bch_encode(case['rate'], case['short'], fr)       # own encoder == reference encoder (checked below)
    for x in case['errors']:
        fr[x // 8] ^= 1 << (7 - x % 8)
    assert sha(fr) == case['in_sha']
    ret = orc.lib().orc_bch_decode(case['rate'], case['short'], fr)
    assert ret == case['ret'] and sha(fr) == case['out_sha']


def test_bch_golden_full_vectors():
    c = G['bch_full']
    fin = np.load(os.path.join(HERE, 'golden', c['infile']))
    fout = np.load(os.path.join(HERE, 'golden', c['outfile']))
    for k in range(fin.shape[0]):
        fr = fin[k].copy()
        assert orc.lib().orc_bch_decode(c['rate'], c['short'], fr) == c['ret'][k]
        assert np.array_equal(fr, fout[k])


def test_bb_prbs_golden():
    seq = np.zeros(58192 // 8, np.uint8)
    orc.lib().orc_bb_prbs(seq, seq.size)
    assert [int(x) for x in seq[:256]] == G['bb_prbs_first_256']
    assert sha(seq) == G['bb_prbs_sha_7274']


@pytest.mark.parametrize('case', G['deinterleave'], ids=lambda c: 'c%d-r%d-s%d' % (c['constel'], c['rate'], c['short']))
def test_deinterleave_golden(case):
    orc._bind_chain()
    n = 16200 if case['short'] else 64800
    src = (np.arange(n) * 7 % 251).astype(np.int8)
    dst = np.zeros(n, np.int8)
    orc.lib().orc_s2_deinterleave(case['constel'], case['rate'], case['short'], src, dst)
    assert sha(dst) == case['out_sha']


# ------------------------------------------------------------------ direct comparison when the reference build is present
needs_ref = pytest.mark.skipif(orc.ref() is None, reason='oracle/_ref not built (reference sources absent)')


@needs_ref
@pytest.mark.parametrize('rate,short', [(6, 0), (3, 0), (0, 1), (6, 1), (9, 1)])
def test_ldpc_oracle_equals_reference_on_random_frames(rate, short):
    rng = np.random.default_rng(rate + 50 * short)
    p = orc.fec_params(rate, short)
    for k in range(3):
        llr = rng.integers(-60, 61, p['N']).astype(np.int8) if k == 0 else ldpc_input(rate, short, 9000 + k, 3.0 + rate * 0.45)
        a, b = llr.copy(), llr.copy()
        ra = orc.lib().orc_ldpc_decode(rate, short, a, 10, 0)
        rb = orc.ref().ref_ldpc_decode(rate, short, b, 10)
        assert ra == rb and np.array_equal(a, b)


@needs_ref
def test_bch_oracle_equals_reference_incl_miscorrection_paths():
    rng = np.random.default_rng(3)
    for rate, short in [(6, 0), (9, 0), (3, 1)]:
        p = orc.fec_params(rate, short)
        nb = p['K'] // 8
        base = np.zeros(nb, np.uint8)
        orc.lib().orc_make_bbframe(base, p['kbch'], 1)
        orc.lib().orc_bch_encode(rate, short, base)
        ref_enc = np.zeros(nb, np.uint8)
        orc.lib().orc_make_bbframe(ref_enc, p['kbch'], 1)
        orc.ref().ref_bch_encode(rate, short, ref_enc)
        assert np.array_equal(base, ref_enc)                       # own encoder == reference encoder
        for ne in [1, 2, 3, p['t'], p['t'] + 1, p['t'] + 3, 30]:
            for rep in range(6):
                fr = base.copy()
                for x in rng.choice(p['K'], ne, replace=False):
                    fr[x // 8] ^= 1 << (7 - x % 8)
                a, b = fr.copy(), fr.copy()
                assert orc.lib().orc_bch_decode(rate, short, a) == orc.ref().ref_bch_decode(rate, short, b)
                assert np.array_equal(a, b)


def test_transmitter_receiver_round_trip_on_cpu():
    """own TX -> oracle RX recovers the BBFRAMEs (QPSK 1/2 short, clean channel with offsets)"""
    iq, bb, _ = orc.transmit(4, 1, 0, nframes=6, seed=3, esn0_db=12.0, cfo=1e-3, timing=0.3, phase0=0.1, lead_symbols=500)
    rx = orc.OracleRx(orc.default_cfg(4, 1, 0))
    out = rx.process(iq)
    sent = {bytes(b) for b in bb}
    assert sum(bytes(o) in sent for o in out) >= 3
    # BBHEADER CRC-8 as the consumer checks it (bbframe_ts_parser.cpp:70-83): 80 bits divide to zero
    for o in out:
        if bytes(o) in sent:
            crc = 0
            for n in range(80):
                b = ((o[n // 8] >> (7 - n % 8)) & 1) ^ (crc & 1)
                crc >>= 1
                if b:
                    crc ^= 0xAB
            assert crc == 0
