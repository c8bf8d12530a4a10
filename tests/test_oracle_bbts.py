"""CPU tests of the oracle's BBFRAME -> TS / GSE parser restatement (oracle/bbframe_ts.cpp).  The reference's translation unit
cannot be compiled here (it needs SDR++ core's <dsp/stream.h>), so these are semantic round trips through this repo's own
transmitter side: PARITY UNPINNED for this row."""
import numpy as np
import pytest

import orc_bbts as B

KBCH = 14232          # short frame 8/9 (config C), 1779 bytes per BBFRAME


def test_header_crc_is_what_the_parser_checks():
    for dfl in (0, 8, 1504, KBCH - 80):
        h = B.bbheader(3, dfl, 48, upl_bits=1504, sync=0x47)
        assert B.L().orc_bbts_crc8_bits(h.ctypes.data, 80) == 0
        h[4] ^= 1
        assert B.L().orc_bbts_crc8_bits(h.ctypes.data, 80) != 0


@pytest.mark.parametrize('kbch,dfl_bytes', [(KBCH, None), (KBCH, 1000), (48408, None), (3072, None)])
def test_ts_round_trip_across_calls(kbch, dfl_bytes):
    rng = np.random.default_rng(1)
    nfr = 12
    D = dfl_bytes if dfl_bytes is not None else kbch // 8 - 10
    pk = B.ts_packets(nfr * D // 188 + 2, rng)
    fr = B.bbframes_from_ts(pk, kbch, nfr, dfl_bytes)
    p = B.OracleBbTs(kbch)
    out = []
    for a, b in ((0, 1), (1, 4), (4, 4), (4, 6), (6, 12)):
        out.append(p.work(fr[a:b]))
    out = np.concatenate(out)
    n = (nfr * D - 1) // 188                      # complete packets after the first (skipped) CRC byte
    assert out.size == n * 188
    assert np.array_equal(out.reshape(-1, 188), pk[:n])
    st = p.stats()
    assert st['synched'] == 1 and st['ts_gs'] == 3 and st['upl'] == 1504 and st['dfl'] == D * 8 and st['last_bb_cnt'] == 6 and st['last_bb_proc'] == 6
    assert st['count'] == (nfr * D - 1) % 188


def test_ts_sync_from_the_middle_and_resync_after_a_bad_header():
    rng = np.random.default_rng(2)
    kbch, nfr = KBCH, 10
    D = kbch // 8 - 10
    pk = B.ts_packets(nfr * D // 188 + 2, rng)
    fr = B.bbframes_from_ts(pk, kbch, nfr)
    # start at frame 3: SYNCD points at the first packet start inside the frame
    p = B.OracleBbTs(kbch)
    out = p.work(fr[3:]).reshape(-1, 188)
    k0 = -(-3 * D // 188)
    assert np.array_equal(out, pk[k0:k0 + len(out)]) and len(out) == (nfr * D - k0 * 188 - 1) // 188
    # a header CRC failure in frame 4 drops synchronisation; frame 5 re-enters at its SYNCD
    bad = fr.copy()
    bad[4, 2] ^= 0x10
    p = B.OracleBbTs(kbch)
    out = p.work(bad).reshape(-1, 188)
    n_before = (4 * D - 1) // 188
    k1 = -(-5 * D // 188)
    want = np.concatenate([pk[:n_before], pk[k1:k1 + (nfr * D - k1 * 188 - 1) // 188]])
    assert np.array_equal(out, want)
    assert p.stats()['last_bb_proc'] == nfr - 1


def test_header_validation_rules():
    kbch = KBCH
    fb = kbch // 8
    p = B.OracleBbTs(kbch)
    fr = np.zeros((1, fb), np.uint8)
    for dfl, syncd, ok in (((fb - 10) * 8, 0, True), ((fb - 10) * 8 + 8, 0, False), (1504, 1504 - 8, False), (1504, 1504 - 16, True),
                           (1500, 0, False), (0, 0, False), (8, 0, False), (16, 0, True)):
        fr[0, :10] = B.bbheader(3, dfl, syncd, upl_bits=1504, sync=0x47)
        p.set_frame_size(kbch)
        p.work(fr)
        assert p.stats()['synched'] == int(ok), (dfl, syncd)


def test_short_data_field_replaces_the_carried_partial():
    # DFL/8 < 188 while a partial is pending: the reference overwrites the partial (bbframe_ts_parser.cpp:201-205)
    rng = np.random.default_rng(3)
    kbch = 3072
    fb = kbch // 8
    fr = rng.integers(0, 256, (3, fb), dtype=np.uint8)
    fr[0, :10] = B.bbheader(3, 300 * 8, 0, 1504, 0x47)
    fr[1, :10] = B.bbheader(3, 100 * 8, 0, 1504, 0x47)
    fr[2, :10] = B.bbheader(3, 300 * 8, 0, 1504, 0x47)
    p = B.OracleBbTs(kbch)
    out = p.work(fr)
    # frame 0: skip 1, 299 bytes -> 1 packet + 111 carried; frame 1: 100 bytes replace it; frame 2: 88 + 212 -> 2 packets, 24 carried
    assert out.size == 3 * 188 and p.stats()['count'] == 24
    assert np.array_equal(out[1:188], fr[0, 11:11 + 187])
    assert np.array_equal(out[189:189 + 100], fr[1, 10:110]) and np.array_equal(out[289:376], fr[2, 10:97])
    assert np.array_equal(out[377:564], fr[2, 98:98 + 187])


def _presync(p, kbch):
    # the parser skips SYNCD/8 + 1 bytes when it (re)synchronises, GSE frames included: an all-padding frame absorbs that
    z = np.zeros(kbch // 8, np.uint8)
    z[:10] = B.bbheader(1, (kbch // 8 - 10) * 8, 0)
    assert p.work(z).size == 0 and p.stats()['synched'] == 1


def test_gse_complete_and_fragmented_pdus():
    rng = np.random.default_rng(4)
    kbch = KBCH
    p = B.OracleBbTs(kbch)
    _presync(p, kbch)
    pdu1 = rng.integers(0, 256, 300, dtype=np.uint8)
    pdu2 = rng.integers(0, 256, 1200, dtype=np.uint8)
    pdu3 = rng.integers(0, 256, 77, dtype=np.uint8)
    lab = bytes(range(1, 7))
    fa = B.gse_fragments(0x86DD, pdu2, [400, 900], frag_id=9, label=lab)
    fb_ = B.gse_fragments(0x0800, pdu1, [120], frag_id=3)
    f1 = B.gse_bbframe([B.gse_complete(0x0800, pdu1, label=lab), fa[0], fb_[0]], kbch)
    f2 = B.gse_bbframe([fa[1], B.gse_complete(0x1234, pdu3), fb_[1]], kbch)
    f3 = B.gse_bbframe([fa[2]], kbch)
    out = p.work(np.stack([f1, f2, f3]))
    want = b'\0\0\x08\0' + bytes(pdu1) + b'\0\0' + bytes(pdu3) + b'\0\0\x08\0' + bytes(pdu1) + b'\0\0\x86\xdd' + bytes(pdu2)
    assert bytes(out) == want
    assert p.stats()['last_gse_crc_err'] == 0 and p.stats()['ts_gs'] == 1
    # CRC-32 failure: nothing comes out, the flag is raised
    fc = B.gse_fragments(0x0800, pdu1, [100], frag_id=5, corrupt_crc=True)
    out = p.work(B.gse_bbframe(fc, kbch))
    assert out.size == 0 and p.stats()['last_gse_crc_err'] == 1


def test_gse_first_frame_after_sync_loss_is_parsed_one_byte_late():
    # reference quirk (bbframe_ts_parser.cpp:159-170 applies to GSE frames too): documented, reproduced
    rng = np.random.default_rng(5)
    kbch = KBCH
    p = B.OracleBbTs(kbch)
    pdu = rng.integers(0, 256, 64, dtype=np.uint8)
    f = B.gse_bbframe([B.gse_complete(0x0800, pdu)], kbch)
    first = p.work(f)
    second = p.work(f)
    assert bytes(second) == b'\0\0\x08\0' + bytes(pdu) and bytes(first) != bytes(second)


def test_output_space_rules():
    rng = np.random.default_rng(6)
    kbch = KBCH
    D = kbch // 8 - 10
    pk = B.ts_packets(4 * D // 188 + 2, rng)
    fr = B.bbframes_from_ts(pk, kbch, 4)
    p = B.OracleBbTs(kbch)
    assert p.work(fr, cap=3 * 188) is None          # the reference's undefined "BUFF OVF!" path
    p = B.OracleBbTs(kbch)
    out = p.work(fr, cap=fr.size + 376)
    assert out.size == ((4 * D - 1) // 188) * 188


def test_golden_vectors():
    """tests/golden/bbts_golden.json (generator: tests/golden/make_golden_bbts.py): clean round trips whose expected output is the
    transmitted packet sequence, and fuzzed sequences anchored on the restatement's own output (regression only: parity unpinned)"""
    import hashlib
    import json
    import os
    G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'bbts_golden.json')))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    for c in G['ts_round_trip']:
        rng = np.random.default_rng(c['seed'])
        D = c['dfl_bytes'] if c['dfl_bytes'] is not None else c['kbch'] // 8 - 10
        pk = B.ts_packets(c['nframes'] * D // 188 + 2, rng)
        fr = B.bbframes_from_ts(pk, c['kbch'], c['nframes'], c['dfl_bytes'])
        assert sha(fr) == c['sha256_in']
        p = B.OracleBbTs(c['kbch'])
        out = np.concatenate([p.work(fr[:4]), p.work(fr[4:])])
        assert sha(out) == c['sha256_out'] and out.size == 188 * c['packets_out']
    for c in G['fuzz']:
        rng = np.random.default_rng(c['seed'])
        p = B.OracleBbTs(c['kbch'])
        for call in range(c['calls']):
            fr = B.fuzz_frames(rng, c['kbch'], int(rng.integers(0, 6)), ts_gs_choices=tuple(c['ts_gs_choices']), p_bad=0.2)
            o = p.work(fr, cap=fr.size + 376)
            st = p.stats()
            assert sha(o) == c['sha256_out_per_call'][call]
            assert [st['synched'], st['last_bb_proc'], st['last_gse_crc_err'], st['ts_gs'], int(o.size)] == c['state_per_call'][call]
