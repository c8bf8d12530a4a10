"""GPU parity tests for the DVB-S tail (TS deframer -> Forney -> RS(204,188) -> energy dispersal -> TS packets) through the C ABI,
bit-exact against the CPU oracle (oracle/dvbs_tail.cpp, itself pinned against the reference sources)."""
import numpy as np
import pytest
import orc_dvbs as od
import orc_dvbs_tail as ot
from orc_dvbs import P, VP
from orc_dvbs_tail import OracleTail

pytestmark = pytest.mark.gpu


def _streams(nbits):
    rng = np.random.default_rng(21)
    cases = []
    bits, ts = ot.dvbs_outer_tx(64, seed=30)
    cases.append(('clean', bits[:nbits]))
    b2, _ = ot.dvbs_outer_tx(64, seed=31)
    cases.append(('ber1e-3', (b2 ^ (rng.random(b2.size) < 1e-3))[:nbits].astype(np.uint8)))
    b3, _ = ot.dvbs_outer_tx(64, seed=32)
    cases.append(('ber8e-3_rs_failures', (b3 ^ (rng.random(b3.size) < 8e-3))[:nbits].astype(np.uint8)))      # many packets beyond t = 8
    b4, _ = ot.dvbs_outer_tx(64, seed=33)
    cases.append(('inverted', (1 - b4)[:nbits].astype(np.uint8)))
    b5, _ = ot.dvbs_outer_tx(64, seed=34)
    cases.append(('offset_777', np.concatenate([rng.integers(0, 2, 777, dtype=np.uint8), b5])[:nbits]))
    cases.append(('noise', rng.integers(0, 2, nbits, dtype=np.uint8)))
    b6, _ = ot.dvbs_outer_tx(64, seed=35)
    x = b6.copy()
    x[30000:30000 + 13056] = rng.integers(0, 2, 13056)            # a burst wipes one frame: sync lost for one group
    cases.append(('burst', x[:nbits]))
    return cases


def test_tail_bit_exact_vs_oracle_over_chunked_calls(engine, pkg):
    import torch
    nbits = 64 * 204 * 8
    cases = _streams(nbits)
    S = len(cases)
    bank = pkg.DvbsTailBank(engine, S, max_bits=70000)
    oracles = [OracleTail() for _ in range(S)]
    pos = 0
    total = [0] * S
    for chunk in (4096, 6144, 13056, 1, 40001, 7, 65536, nbits):
        tin, exp = [], []
        for s in range(S):
            seg = cases[s][1][pos:pos + chunk]
            tin.append(torch.from_numpy(np.ascontiguousarray(seg)).cuda())
            exp.append(oracles[s].process(seg))
        tout = [torch.zeros(188 * 8 * 8, dtype=torch.uint8, device='cuda') for _ in range(S)]
        nb = bank.process_batch(tin, tout)
        for s in range(S):
            e, nf = exp[s]
            assert nb[s] == e.size, (cases[s][0], pos, nb[s], e.size)
            assert np.array_equal(tout[s][:nb[s]].cpu().numpy(), e), (cases[s][0], pos)
            st = bank.stats(s)
            assert st['frames'] == nf
            if nf:
                assert st['rs_errors'] == oracles[s].last_rs, (cases[s][0], st, oracles[s].last_rs)
            assert (st['errors_nor'], st['errors_inv']) == tuple(int(x) for x in oracles[s].errs), cases[s][0]
            total[s] += nf
        pos += chunk
        if pos >= nbits:
            break
    assert total[0] >= 7 and total[5] == 0                     # the clean stream found its frames, noise never syncs
    bank.close()


def test_tail_recovers_transmitted_ts_packets(engine, pkg):
    import torch
    bits, ts = ot.dvbs_outer_tx(96, seed=50)
    rng = np.random.default_rng(51)
    bits = (bits ^ (rng.random(bits.size) < 2e-3)).astype(np.uint8)
    bank = pkg.DvbsTailBank(engine, 1, max_bits=bits.size)
    out = torch.zeros(188 * 96, dtype=torch.uint8, device='cuda')
    nb = bank.process_batch([torch.from_numpy(bits).cuda()], [out])
    got = out[:nb[0]].cpu().numpy().reshape(-1, 188)
    assert len(got) == 96 and (got[:, 0] == 0x47).all()
    sent = {bytes(t) for t in ts}
    hits = sum(bytes(g) in sent for g in got[16:])               # the de-interleaver needs 11 packets of history
    assert hits == len(got) - 16
    bank.close()


def test_dvbs_end_to_end_iq_to_ts(engine, pkg):
    """IQ -> QPSK_ALT front end -> slicer -> Viterbi -> deframer -> de-interleaver -> RS -> descrambler == transmitted TS packets"""
    import torch
    npk = 240
    obits, ts = ot.dvbs_outer_tx(npk, seed=60)
    # inner code: rate 1/2 mother code over the outer bit stream, then QPSK
    enc = od.cc_encode(obits)
    nsym = enc.size // 2
    iq = np.zeros(2 * nsym, np.complex64)
    od.LF().orc_dvbs_modulate(P(np.ascontiguousarray(enc)), nsym, 9.0, 5e-4, 0.3, 0.2, 7, P(iq))
    rx = pkg.DvbsDemodBank(engine, 1, max_samples=65536)
    tail = pkg.DvbsTailBank(engine, 1, max_bits=80000)
    got = []
    for p in range(0, iq.size, 65536):
        bits = rx.process(iq[p:p + 65536])
        out = torch.zeros(188 * 64, dtype=torch.uint8, device='cuda')
        nb = tail.process_batch([torch.from_numpy(bits).cuda()], [out])
        got.append(out[:nb[0]].cpu().numpy())
    got = np.concatenate(got).reshape(-1, 188)
    sent = {bytes(t) for t in ts}
    hits = sum(bytes(g) in sent for g in got)
    # minus loop acquisition / Viterbi lock at the start and the interleaver's 11-packet history
    assert len(got) >= 120 and hits >= len(got) - 24, (len(got), hits)
    rx.close(); tail.close()


def test_dvbs_segment_receiver_to_ts_packets(engine, pkg):
    """one DVB-S carrier, IQ -> dvbs2gpu_dvbs_segrx_* (segments demodulated side by side, bit streams joined) -> dvbs2gpu_dvbs_tail_* ->
    TS packets: after the start-up every packet that comes out is one that was sent, consecutive and without a gap (the joined stream is
    one bit stream: the deframer keeps its lock across the joins)"""
    import torch
    npk = 1400
    obits, ts = ot.dvbs_outer_tx(npk, seed=61)
    enc = od.cc_encode(obits)
    nsym = enc.size // 2
    iq = np.zeros(2 * nsym, np.complex64)
    od.LF().orc_dvbs_modulate(P(np.ascontiguousarray(enc)), nsym, 12.0, 1e-4, 0.3, 0.2, 7, P(iq))
    nseg, own, warm = 4, 49152, 32768
    rx = pkg.DvbsSegmentReceiver(engine, nseg, own, warm)
    tail = pkg.DvbsTailBank(engine, 1, max_bits=2 * nseg * own * 2 + 4 * 65536)
    d_iq = torch.from_numpy(iq).cuda()
    bits = torch.zeros(2 * nseg * own * 2 + 4 * 65536, dtype=torch.uint8, device='cuda')
    out = torch.zeros(188 * 1024, dtype=torch.uint8, device='cuda')
    got, unmatched = [], 0
    for k, a in enumerate(range(0, iq.size, rx.chunk_samples)):
        nb = rx.process(d_iq[a:a + rx.chunk_samples], bits)
        unmatched += rx.stats()['unmatched'] if k else 0        # (the very first segment starts cold: its end may not have settled)
        nby = tail.process_batch([bits[:nb]], [out])
        got.append(out[:nby[0]].cpu().numpy().copy())
    got = np.concatenate(got).reshape(-1, 188)
    index = {bytes(t): i for i, t in enumerate(ts)}
    seq = [index.get(bytes(g), -1) for g in got]
    assert unmatched == 0 and len(seq) >= npk - 200, (unmatched, len(seq))
    bad = [i for i in range(1, len(seq)) if seq[i] < 0 or seq[i] != seq[i - 1] + 1]
    # everything after the first call (its 218 k bits = 133 packets) is the transmitted packet sequence, consecutive, without a gap or a repeat
    assert not bad or bad[-1] < 140, bad[-10:]
    assert seq[-1] >= npk - 32, seq[-1]                          # (the interleaver holds 11 packets back, the deframer works in groups of 8)
    rx.close(); tail.close()
