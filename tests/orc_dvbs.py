"""ctypes helpers for the DVB-S part of the CPU oracle (oracle/dvbs.cpp) + a small DVB-S inner-code transmitter
(convolutional encoder of the oracle, puncturing derived from the oracle's own de-puncturers).  Test infrastructure only."""
import ctypes as C
import numpy as np
from orc import lib, ref

VP = C.c_void_p
RATE_NAMES = ['1/2', '2/3', '3/4', '5/6', '7/8']


def P(a):
    return a.ctypes.data_as(VP)


_bound = False


def L():
    global _bound
    l = lib()
    if not _bound:
        for n in ('orc_dvbs_slicer_create', 'orc_depunc_create', 'orc_ccdec_create', 'orc_ccenc_create', 'orc_vitdvbs_create', 'orc_forney_create'):
            getattr(l, n).restype = VP
        l.orc_vitdvbs_create.argtypes = [C.c_float, C.c_int, C.c_int]
        for n in ('orc_dvbs_slicer_destroy', 'orc_depunc_destroy', 'orc_ccdec_destroy', 'orc_ccenc_destroy', 'orc_vitdvbs_destroy', 'orc_forney_destroy'):
            getattr(l, n).restype = None
            getattr(l, n).argtypes = [VP]
        l.orc_dvbs_slicer_process.argtypes = [VP, C.c_int, VP, VP]
        l.orc_rotate_soft.argtypes = [VP, C.c_int, C.c_int, C.c_int]
        l.orc_signed_to_unsigned.argtypes = [VP, VP, C.c_int]
        l.orc_depuncture_34.argtypes = [VP, VP, C.c_int, C.c_int]
        l.orc_depuncture_78.argtypes = [VP, VP, C.c_int, C.c_int]
        l.orc_depunc_static.argtypes = [VP, VP, VP, C.c_int, C.c_int]
        l.orc_depunc_set_shift.argtypes = [VP, C.c_int]
        l.orc_depunc_cont.argtypes = [VP, VP, VP, C.c_int]
        l.orc_ccdec_work.argtypes = [VP, VP, VP]
        l.orc_ccenc_work.argtypes = [VP, VP, VP]
        l.orc_vitdvbs_work.argtypes = [VP, VP, C.c_int, VP, VP]
        l.orc_forney_deinterleave.argtypes = [VP, VP, VP]
        _bound = True
    return l


def cc_encode(bits):
    """r=1/2 K=7 mother code (X = poly 79, Y = poly 109 in the reference's bit order); returns [2n] bits"""
    l = L()
    bits = np.ascontiguousarray(bits, np.uint8)
    h = VP(l.orc_ccenc_create(len(bits)))
    out = np.zeros(2 * len(bits), np.uint8)
    l.orc_ccenc_work(h, P(bits), P(out))
    l.orc_ccenc_destroy(h)
    return out


def puncture_keep_mask(rate, n_tx):
    """mask over mother-code positions that are transmitted for `rate`, derived by de-puncturing n_tx dummy softs with the
    oracle's own de-puncturer at shift 0 (erasures = 128 mark the punctured positions)"""
    l = L()
    x = np.ones(n_tx, np.uint8)
    out = np.full(4 * n_tx + 16, 255, np.uint8)
    if rate == 0:
        return np.ones(n_tx, bool)
    if rate == 1 or rate == 3:
        h = VP(l.orc_depunc_create(3 if rate == 1 else 6))
        n = l.orc_depunc_static(h, P(x), P(out), n_tx, 0)
        l.orc_depunc_destroy(h)
    elif rate == 2:
        n = l.orc_depuncture_34(P(x), P(out), n_tx, 0)
    else:
        n = l.orc_depuncture_78(P(x), P(out), n_tx, 0)
    return out[:n] != 128


def dvbs_tx(rate, n_soft, seed, amp=40.0, sigma=8.0, drop=0, rot90=False):
    """n_soft int8 soft bits of a DVB-S inner-coded random bit stream at puncturing `rate`, with `drop` leading softs
    removed (decoder must find the shift) and optionally rotated so that the decoder needs PHASE_90.  Returns (soft, bits)."""
    rng = np.random.default_rng(seed)
    need = n_soft + drop + 64
    keep = puncture_keep_mask(rate, need)
    nbits = len(keep) // 2 + 8
    bits = rng.integers(0, 2, nbits, dtype=np.uint8)
    mother = cc_encode(bits)[:len(keep)]
    tx = mother[keep][drop:drop + n_soft].astype(np.float64)
    soft = (2 * tx - 1) * amp + rng.normal(0, sigma, n_soft)
    soft = np.clip(np.trunc(soft), -127, 127).astype(np.int8)
    if rot90:
        # the decoder's PHASE_90 maps (a, b) -> (b, -a); pre-rotate with the inverse: (I, Q) -> (-Q, I)
        s2 = soft.reshape(-1, 2)
        soft = np.stack([-s2[:, 1], s2[:, 0]], axis=1).reshape(-1).astype(np.int8)
    return soft, bits


class OracleViterbi:
    def __init__(self, thr=0.15, max_outsync=20):
        self.l = L()
        self.h = VP(self.l.orc_vitdvbs_create(thr, max_outsync, 8192))

    def __del__(self):
        try:
            self.l.orc_vitdvbs_destroy(self.h)
        except Exception:
            pass

    def work(self, soft_blocks):
        """soft_blocks int8 [nblocks, 8192] -> bits uint8 [nblocks, 8192] (zero beyond nbits), nbits [nblocks], stats [nblocks, 5]"""
        nb = soft_blocks.shape[0]
        bits = np.zeros((nb, 8192), np.uint8)
        nbits = np.zeros(nb, np.int32)
        stats = np.zeros((nb, 5), np.int32)
        st = np.zeros(4, np.float32)
        for b in range(nb):
            x = np.ascontiguousarray(soft_blocks[b]).copy()
            o = np.zeros(8192 + 64, np.uint8)
            nbits[b] = self.l.orc_vitdvbs_work(self.h, P(x), 8192, P(o), P(st))
            bits[b] = o[:8192]
            ps = int(st[3])
            stats[b] = [np.float32(st[0]).view(np.int32), int(st[1]), int(st[2]), ps // 16, ps % 16]
        return bits, nbits, stats


# ------------------------------------------------------------------ DVB-S front end (oracle/dvbs_fe.cpp)
class QpskAltCfg(C.Structure):
    _fields_ = [('symbolrate', C.c_double), ('samplerate', C.c_double), ('rrc_taps', C.c_int), ('rrc_alpha', C.c_float),
                ('agc_rate', C.c_float), ('costas_bw', C.c_float), ('fll_bw', C.c_float), ('omega_gain', C.c_float),
                ('mu_gain', C.c_float), ('omega_rel_limit', C.c_float)]


_fe_bound = False


def LF():
    global _fe_bound
    l = L()
    if not _fe_bound:
        l.orc_qpskalt_default_cfg.argtypes = [C.POINTER(QpskAltCfg)]
        l.orc_qpskalt_default_cfg.restype = None
        l.orc_qpskalt_create.argtypes = [C.POINTER(QpskAltCfg)]
        l.orc_qpskalt_create.restype = VP
        l.orc_qpskalt_destroy.argtypes = [VP]
        l.orc_qpskalt_destroy.restype = None
        l.orc_qpskalt_process.argtypes = [VP, C.c_int, VP, VP]
        l.orc_qpskalt_stage.argtypes = [VP, C.c_int, C.c_int, VP, VP]
        l.orc_qpskalt_state.argtypes = [VP, VP]
        l.orc_qpskalt_state.restype = None
        l.orc_dvbs_modulate.argtypes = [VP, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint64, VP]
        l.orc_dvbs_modulate.restype = None
        _fe_bound = True
    return l


def qpsk_alt_default_cfg(**kw):
    c = QpskAltCfg()
    LF().orc_qpskalt_default_cfg(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def dvbs_tx_bits(rate, n_tx_bits, seed):
    """transmitted (punctured) bit stream of a random message: returns (tx bits uint8 [n_tx_bits], info bits)"""
    rng = np.random.default_rng(seed)
    keep = puncture_keep_mask(rate, n_tx_bits + 64)
    nbits = len(keep) // 2 + 8
    bits = rng.integers(0, 2, nbits, dtype=np.uint8)
    mother = cc_encode(bits)[:len(keep)]
    return np.ascontiguousarray(mother[keep][:n_tx_bits]), bits


def dvbs_iq(rate, nsym, seed, esn0_db=12.0, cfo=0.0, timing=0.0, phase0=0.0):
    """complex64 [2*nsym] at 2 sps of a DVB-S inner-coded QPSK stream (no RS / interleaver: inner code only)"""
    tx, bits = dvbs_tx_bits(rate, 2 * nsym, seed)
    out = np.zeros(2 * nsym, np.complex64)
    LF().orc_dvbs_modulate(P(tx), nsym, esn0_db, cfo, timing, phase0, seed, P(out))
    return out, bits


class OracleQpskAlt:
    def __init__(self, cfg=None):
        self.l = LF()
        self.cfg = cfg or qpsk_alt_default_cfg()
        self.h = VP(self.l.orc_qpskalt_create(C.byref(self.cfg)))

    def __del__(self):
        try:
            self.l.orc_qpskalt_destroy(self.h)
        except Exception:
            pass

    def process(self, iq):
        iq = np.ascontiguousarray(iq, np.complex64)
        out = np.zeros(iq.size // 2 + 64, np.complex64)
        n = self.l.orc_qpskalt_process(self.h, iq.size, P(iq), P(out))
        return out[:n]

    def stage(self, which, x):
        x = np.ascontiguousarray(x, np.complex64)
        out = np.zeros(x.size + 64, np.complex64)
        n = self.l.orc_qpskalt_stage(self.h, which, x.size, P(x), P(out))
        return out[:n]

    def state(self):
        s = np.zeros(8, np.float32)
        self.l.orc_qpskalt_state(self.h, P(s))
        return s
