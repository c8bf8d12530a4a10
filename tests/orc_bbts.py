"""Helpers for the BBFRAME -> TS / GSE parser: ctypes access to oracle/bbframe_ts.cpp and a small transmitter side in numpy
(TS packets -> user packets with CRC-8 -> BBFRAMEs with BBHEADER, EN 302 307-1 5.1.4-5.1.6; GSE packets / fragments with CRC-32,
TS 102 606).  Test infrastructure only."""
import ctypes as C
import numpy as np
from orc import lib

VP = C.c_void_p
_b = False
STAT_KEYS = ('ts_gs', 'sis_mis', 'ccm_acm', 'issyi', 'npd', 'ro', 'isi', 'upl', 'dfl', 'sync', 'syncd',
             'last_gse_crc_err', 'last_bb_cnt', 'last_bb_proc', 'last_ts_errs', 'synched', 'count')


def L():
    global _b
    l = lib()
    if not _b:
        l.orc_bbts_create.restype = VP
        l.orc_bbts_create.argtypes = [C.c_int]
        l.orc_bbts_destroy.restype = None
        l.orc_bbts_destroy.argtypes = [VP]
        l.orc_bbts_set_frame_size.restype = None
        l.orc_bbts_set_frame_size.argtypes = [VP, C.c_int]
        l.orc_bbts_work.argtypes = [VP, VP, C.c_int, VP, C.c_int]
        l.orc_bbts_get_stats.restype = None
        l.orc_bbts_get_stats.argtypes = [VP, VP]
        l.orc_bbts_crc8_bits.restype = C.c_uint
        l.orc_bbts_crc8_bits.argtypes = [VP, C.c_int]
        _b = True
    return l


class OracleBbTs:
    def __init__(self, kbch_bits):
        self.kbch = kbch_bits
        self.h = L().orc_bbts_create(kbch_bits)

    def __del__(self):
        if getattr(self, 'h', None):
            L().orc_bbts_destroy(self.h)
            self.h = None

    def set_frame_size(self, kbch_bits):
        L().orc_bbts_set_frame_size(self.h, kbch_bits)
        self.kbch = kbch_bits

    def work(self, bbframes, cap=None):
        bb = np.ascontiguousarray(bbframes, np.uint8).reshape(-1)
        cnt = bb.size // (self.kbch // 8)
        cap = cap if cap is not None else bb.size + 376
        out = np.zeros(max(cap, 1), np.uint8)
        n = L().orc_bbts_work(self.h, bb.ctypes.data, cnt, out.ctypes.data, cap)
        if n < 0:
            return None
        return out[:n].copy()

    def stats(self):
        a = np.zeros(17, np.int32)
        L().orc_bbts_get_stats(self.h, a.ctypes.data)
        return dict(zip(STAT_KEYS, [int(x) for x in a]))


def crc8(data):
    """CRC-8 x^8+x^7+x^6+x^4+x^2+1, MSB first, zero initial state (EN 302 307-1 5.1.4 / 5.1.6)"""
    crc = 0
    for byte in bytes(data):
        for k in range(7, -1, -1):
            fb = ((byte >> k) & 1) ^ (crc >> 7)
            crc = (crc << 1) & 0xff
            if fb:
                crc ^= 0xD5
    return crc


def bbheader(ts_gs, dfl_bits, syncd_bits=0, upl_bits=0, sync=0, sis=1, ccm=1, issyi=0, npd=0, ro=0, isi=0, good_crc=True):
    h = np.zeros(10, np.uint8)
    h[0] = (ts_gs << 6) | (sis << 5) | (ccm << 4) | (issyi << 3) | (npd << 2) | ro
    h[1] = isi
    h[2], h[3] = upl_bits >> 8, upl_bits & 0xff
    h[4], h[5] = dfl_bits >> 8, dfl_bits & 0xff
    h[6] = sync
    h[7], h[8] = syncd_bits >> 8, syncd_bits & 0xff
    h[9] = crc8(h[:9]) ^ (0 if good_crc else 0x5a)
    return h


def ts_packets(n, rng):
    p = rng.integers(0, 256, (n, 188), dtype=np.uint8)
    p[:, 0] = 0x47
    return p


def bbframes_from_ts(packets, kbch_bits, nframes, dfl_bytes=None):
    """Slice the user-packet stream (sync byte of packet k replaced by the CRC-8 of packet k-1's 187 bytes) into data fields."""
    fb = kbch_bits // 8
    D = dfl_bytes if dfl_bytes is not None else fb - 10
    ups = packets.copy()
    for k in range(len(ups)):
        ups[k, 0] = crc8(packets[k - 1, 1:]) if k else 0
    stream = ups.reshape(-1)
    assert stream.size >= nframes * D
    frames = np.zeros((nframes, fb), np.uint8)
    for f in range(nframes):
        off = f * D
        syncd = (-off) % 188
        frames[f, :10] = bbheader(3, D * 8, syncd * 8, upl_bits=1504, sync=0x47)
        frames[f, 10:10 + D] = stream[off:off + D]
    return frames


def crc32_mpeg(data, crc=0xffffffff):
    for byte in bytes(data):
        crc ^= byte << 24
        for _ in range(8):
            crc = ((crc << 1) ^ 0x04c11db7) & 0xffffffff if crc & 0x80000000 else (crc << 1) & 0xffffffff
    return crc


def gse_complete(proto, pdu, label=None):
    """one unfragmented GSE packet; label: 6 bytes (label type 00) or None (label type 10, broadcast)"""
    lt = 0 if label is not None else 2
    body = bytes([proto >> 8, proto & 0xff]) + (bytes(label) if label is not None else b'') + bytes(pdu)
    n = len(body)
    return bytes([0xC0 | (lt << 4) | (n >> 8), n & 0xff]) + body


def gse_fragments(proto, pdu, cuts, frag_id, label=None, corrupt_crc=False):
    """PDU cut at the byte positions `cuts` into START / middle / END packets"""
    lt = 0 if label is not None else 2
    lab = bytes(label) if label is not None else b''
    total = 2 + len(lab) + len(pdu)
    tl = bytes([total >> 8, total & 0xff])
    pr = bytes([proto >> 8, proto & 0xff])
    crc = crc32_mpeg(tl + pr + lab + bytes(pdu))
    if corrupt_crc:
        crc ^= 0x1000
    tail = bytes(pdu) + bytes([(crc >> 24) & 0xff, (crc >> 16) & 0xff, (crc >> 8) & 0xff, crc & 0xff])
    pieces = [tail[a:b] for a, b in zip([0] + list(cuts), list(cuts) + [len(tail)])]
    out = []
    for i, pc in enumerate(pieces):
        if i == 0:
            body = bytes([frag_id]) + tl + pr + lab + pc
            h = 0x80 | (lt << 4)
        elif i == len(pieces) - 1:
            body = bytes([frag_id]) + pc
            h = 0x40 | 0x30
        else:
            body = bytes([frag_id]) + pc
            h = 0x30
        n = len(body)
        out.append(bytes([h | (n >> 8), n & 0xff]) + body)
    return out


def gse_bbframe(gse_packets, kbch_bits):
    """GSE packets back to back in one data field, zero padding after them"""
    fb = kbch_bits // 8
    data = b''.join(gse_packets)
    dfl = fb - 10
    assert len(data) <= dfl
    fr = np.zeros(fb, np.uint8)
    fr[:10] = bbheader(1, dfl * 8, 0, upl_bits=0, sync=0)
    fr[10:10 + len(data)] = np.frombuffer(data, np.uint8)
    return fr


def fuzz_frames(rng, kbch_bits, nframes, ts_gs_choices=(3,), p_bad=0.15):
    """random data fields behind mostly-valid random BBHEADERs (some with bad CRC / DFL / SYNCD)"""
    fb = kbch_bits // 8
    fr = rng.integers(0, 256, (nframes, fb), dtype=np.uint8)
    for f in range(nframes):
        r = rng.random()
        dfl = int(rng.integers(0, (fb - 10) + 1)) * 8
        if rng.random() < 0.5:
            dfl = (fb - 10) * 8
        if rng.random() < 0.1:
            dfl = int(rng.integers(0, 400)) * 8
        syncd = int(rng.integers(0, 188)) * 8
        good = True
        if r < p_bad / 3:
            good = False
        elif r < 2 * p_bad / 3:
            dfl += int(rng.integers(1, 8))
        elif r < p_bad:
            syncd = dfl + int(rng.integers(0, 100))
        tg = int(rng.choice(ts_gs_choices))
        fr[f, :10] = bbheader(tg, dfl & 0xffff, syncd & 0xffff, upl_bits=0 if tg == 1 else 1504, sync=0x47, good_crc=good,
                              issyi=int(rng.random() < 0.03), npd=int(rng.random() < 0.03), sis=int(rng.random() < 0.8), isi=int(rng.integers(0, 256)))
    return fr
