"""Helpers for the DVB-S tail (TS deframer, RS(204,188), energy dispersal): ctypes access to oracle/dvbs_tail.cpp and a small
DVB-S outer transmitter in numpy (TS packets -> energy dispersal -> RS(204,188) -> Forney interleaver -> bits).
Test infrastructure only."""
import ctypes as C
import numpy as np
from orc import lib, ref
from orc_dvbs import P, VP
import orc_dvbs as _od

_b = False


def L():
    global _b
    l = lib()
    if not _b:
        for n in ('orc_tsdef_create', 'orc_dvbsrs_create', 'orc_dvbsdescr_create'):
            getattr(l, n).restype = VP
        for n in ('orc_tsdef_destroy', 'orc_dvbsrs_destroy', 'orc_dvbsdescr_destroy'):
            getattr(l, n).restype = None
            getattr(l, n).argtypes = [VP]
        l.orc_tsdef_work.argtypes = [VP, VP, C.c_int, VP, VP]
        l.orc_dvbsrs_decode.argtypes = [VP, VP]
        l.orc_rs255_decode.argtypes = [VP, VP]
        l.orc_dvbsdescr_work.argtypes = [VP, VP]
        _b = True
    return l


# ---- GF(256), primitive polynomial 0x11d
_EXP = np.zeros(512, np.int32)
_LOG = np.zeros(256, np.int32)
_e = 1
_EXP[0] = 1
for _i in range(1, 512):
    _e *= 2
    if _e > 255:
        _e ^= 0x11d
    _EXP[_i] = _e
    if _i < 255:
        _LOG[_e] = _i


def gmul(a, b):
    return 0 if a == 0 or b == 0 else int(_EXP[_LOG[a] + _LOG[b]])


def _generator():
    g = [1]
    for i in range(16):           # roots alpha^0 .. alpha^15 (fcr 0, gap 1)
        r = int(_EXP[i])
        ng = [0] * (len(g) + 1)
        for k, c in enumerate(g):   # g(x) * (x + r), coefficients highest order first
            ng[k] ^= c
            ng[k + 1] ^= gmul(c, r)
        g = ng
    return g


_GEN = _generator()


def rs_encode_204(msg188):
    """systematic RS(255,239) shortened to (204,188): returns 204 bytes"""
    rem = [0] * 16
    for b in msg188:
        fb = int(b) ^ rem[0]
        rem = rem[1:] + [0]
        if fb:
            for k in range(16):
                rem[k] ^= gmul(_GEN[k + 1], fb)
    return np.concatenate([np.asarray(msg188, np.uint8), np.asarray(rem, np.uint8)])


def prbs_bytes(n):
    """energy-dispersal bytes after a reset (reg = 0xa9), dvbs_scrambling.h:14-26"""
    reg, out = 0xa9, np.zeros(n, np.uint8)
    for i in range(n):
        v = 0
        for _ in range(8):
            fb = ((reg >> 13) ^ (reg >> 14)) & 1
            reg = ((reg << 1) | fb) & 0x7fff
            v = (v << 1) | fb
        out[i] = v
    return out


def forney_interleave(data, state=None):
    """inverse of the reference de-interleaver: branch b (= byte index mod 12) is delayed by 17*b*12 bytes; state = history"""
    n = len(data)
    hist = np.zeros(17 * 11 * 12, np.uint8) if state is None else state
    full = np.concatenate([hist, data])
    idx = np.arange(n)
    src = idx + len(hist) - 17 * 12 * (idx % 12)
    out = full[src]
    return out, full[-len(hist):]


def dvbs_outer_tx(npackets, seed):
    """npackets (multiple of 8) random TS packets -> (bits uint8 [npackets*204*8], ts [npackets, 188])"""
    rng = np.random.default_rng(seed)
    ts = rng.integers(0, 256, (npackets, 188), dtype=np.uint8)
    ts[:, 0] = 0x47
    pr = prbs_bytes(8 * 188)
    coded = np.zeros((npackets, 204), np.uint8)
    for p in range(npackets):
        k = p % 8
        pkt = ts[p].copy()
        if k == 0:
            pkt[0] = 0xB8
            pkt[1:] ^= pr[0:187]
        else:
            pkt[1:] ^= pr[188 * k:188 * k + 187]       # one PRBS byte is skipped over every non-inverted sync byte
        coded[p] = rs_encode_204(pkt)
    inter, _ = forney_interleave(coded.reshape(-1))
    bits = np.unpackbits(inter)
    return bits, ts


class OracleTail:
    """DVBSDemod::process after vit.process, frame k at stride 1632"""

    def __init__(self):
        self.o, self.ol = L(), _od.L()
        self.hd, self.hf = VP(self.o.orc_tsdef_create()), VP(self.ol.orc_forney_create())
        self.hr, self.hs = VP(self.o.orc_dvbsrs_create()), VP(self.o.orc_dvbsdescr_create())
        self.last_rs = [0] * 8
        self.errs = np.zeros(2, np.int32)

    def process(self, bits):
        bits = np.ascontiguousarray(bits, np.uint8)
        frames = np.zeros(1632 * (bits.size // 13056 + 4), np.uint8)
        nf = self.o.orc_tsdef_work(self.hd, P(bits), bits.size, P(frames), P(self.errs)) if bits.size else 0
        out = []
        for k in range(nf):
            f = np.ascontiguousarray(frames[1632 * k:1632 * (k + 1)])
            d = np.zeros(1632, np.uint8)
            self.ol.orc_forney_deinterleave(self.hf, P(f), P(d))
            for i in range(8):
                pkt = np.ascontiguousarray(d[204 * i:204 * (i + 1)])
                self.last_rs[i] = self.o.orc_dvbsrs_decode(self.hr, P(pkt))
                d[204 * i:204 * i + 188] = pkt[:188]
            self.o.orc_dvbsdescr_work(self.hs, P(d))
            for i in range(8):
                out.append(d[204 * i:204 * i + 188].copy())
        return (np.concatenate(out) if out else np.zeros(0, np.uint8)), nf
