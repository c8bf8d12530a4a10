"""ctypes loaders for the CPU oracle (oracle/liboracle.so) and, when it has been built, the compiled
reference FEC (oracle/_ref/libdvbs2ref.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, 'oracle')

RATE_NAMES = ['1/4', '1/3', '2/5', '1/2', '3/5', '2/3', '3/4', '4/5', '5/6', '8/9', '9/10']
ALL_CODES = [(r, 0) for r in range(11)] + [(r, 1) for r in range(10)]

_i8p = np.ctypeslib.ndpointer(np.int8, flags='C_CONTIGUOUS')
_u8p = np.ctypeslib.ndpointer(np.uint8, flags='C_CONTIGUOUS')
_u16p = np.ctypeslib.ndpointer(np.uint16, flags='C_CONTIGUOUS')
_i32p = np.ctypeslib.ndpointer(np.int32, flags='C_CONTIGUOUS')
_f32p = np.ctypeslib.ndpointer(np.float32, flags='C_CONTIGUOUS')


def build_oracle():
    so = os.path.join(ORACLE_DIR, 'liboracle.so')
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith(('.cpp', '.h'))]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(['make', '-C', ORACLE_DIR, '-s', 'liboracle.so'])
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_oracle())
        L.orc_fec_params.argtypes = [C.c_int, C.c_int, _i32p]
        L.orc_modcod_params.argtypes = [C.c_int, C.c_int, C.c_int, _i32p, _f32p]
        L.orc_ldpc_edges.argtypes = [C.c_int, C.c_int]
        L.orc_ldpc_decode.argtypes = [C.c_int, C.c_int, _i8p, C.c_int, C.c_int]
        L.orc_ldpc_encode.argtypes = [C.c_int, C.c_int, _u8p]
        L.orc_bch_decode.argtypes = [C.c_int, C.c_int, _u8p]
        L.orc_bch_syndromes.argtypes = [C.c_int, C.c_int, _u8p, _u16p]
        L.orc_bch_encode.argtypes = [C.c_int, C.c_int, _u8p]
        L.orc_bch_encode.restype = None
        L.orc_bb_prbs.argtypes = [_u8p, C.c_int]
        L.orc_bb_prbs.restype = None
        L.orc_bb_descramble.argtypes = [_u8p, C.c_int]
        L.orc_bb_descramble.restype = None
        L.orc_hard_pack.argtypes = [_i8p, C.c_int, _u8p]
        L.orc_hard_pack.restype = None
        L.orc_make_bbframe.argtypes = [_u8p, C.c_int, C.c_uint64]
        L.orc_make_bbframe.restype = None
        L.orc_fec_encode_frame.argtypes = [C.c_int, C.c_int, C.c_uint64, _u8p, _u8p]
        L.orc_fec_decode_frame.argtypes = [C.c_int, C.c_int, _i8p, C.c_int, C.c_int, _u8p, _i32p]
        _lib = L
    return _lib


_ref = None


def ref():
    """Compiled reference FEC, or None when oracle/_ref has not been built (GPU box without prebuilt)."""
    global _ref
    if _ref is None:
        so = os.path.join(ORACLE_DIR, '_ref', 'libdvbs2ref.so')
        if not os.path.exists(so) and os.path.isdir('/root/reference/src/demod'):
            subprocess.check_call(['make', '-C', ORACLE_DIR, '-s', 'ref'])
        if not os.path.exists(so):
            return None
        L = C.CDLL(so)
        L.ref_ldpc_decode.argtypes = [C.c_int, C.c_int, _i8p, C.c_int]
        L.ref_ldpc_decode_many.argtypes = [C.c_int, C.c_int, _i8p, C.c_int, C.c_int, _i32p]
        L.ref_ldpc_decode_many.restype = None
        L.ref_ldpc_decode_simd16.argtypes = [C.c_int, C.c_int, _i8p, C.c_int, C.c_int]
        L.ref_bch_decode.argtypes = [C.c_int, C.c_int, _u8p]
        L.ref_bch_decode_many.argtypes = [C.c_int, C.c_int, _u8p, C.c_int, C.c_int, _i32p]
        L.ref_bch_decode_many.restype = None
        L.ref_bch_encode.argtypes = [C.c_int, C.c_int, _u8p]
        L.ref_bch_encode.restype = None
        L.ref_bb_descramble.argtypes = [C.c_int, C.c_int, _u8p]
        L.ref_bb_descramble.restype = None
        L.ref_deinterleave.argtypes = [C.c_int, C.c_int, C.c_int, _i8p, _i8p]
        L.ref_deinterleave.restype = None
        if hasattr(L, 'ref_fec16_create'):       # (a prebuilt oracle/_ref of an older checkout may lack these)
            L.ref_fec16_create.argtypes = [C.c_int, C.c_int]
            L.ref_fec16_create.restype = C.c_void_p
            L.ref_fec16_destroy.argtypes = [C.c_void_p]
            L.ref_fec16_destroy.restype = None
            L.ref_fec16_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
            L.ref_build_flags.restype = C.c_char_p
        _ref = L
    return _ref


def fec_params(rate, short):
    out = np.zeros(6, np.int32)
    if lib().orc_fec_params(rate, short, out) != 0:
        return None
    return dict(code_index=int(out[0]), N=int(out[1]), K=int(out[2]), kbch=int(out[3]), m=int(out[4]), t=int(out[5]))


def encode_frame(rate, short, seed):
    """-> (bbframe bytes kbch/8, code bits N of 0/1)"""
    p = fec_params(rate, short)
    bb = np.zeros(p['kbch'] // 8, np.uint8)
    bits = np.zeros(p['N'], np.uint8)
    assert lib().orc_fec_encode_frame(rate, short, seed, bb, bits) == 0
    return bb, bits


def bits_to_llr(bits, snr_db, rng, scale=None, mod='bpsk'):
    """BPSK-over-AWGN int8 LLRs, negative = bit 1 (module_dvbs2_demod.cpp:360)."""
    x = 1.0 - 2.0 * bits.astype(np.float64)
    sigma = 10 ** (-snr_db / 20.0)
    y = x + sigma * rng.standard_normal(bits.shape)
    if scale is None:
        scale = 8.0
    l = np.clip(np.rint(y * scale * 2 / (sigma * sigma) / 8.0), -127, 127)
    return l.astype(np.int8)


# ------------------------------------------------------------------------------- DVB-S2 chain (oracle/s2chain.cpp)
class DemodCfg(C.Structure):
    """Same layout as dvbs2gpu_demod_cfg (include/dvbs2gpu.h) and orc::DemodCfg."""
    _fields_ = [('symbolrate', C.c_double), ('samplerate', C.c_double), ('agc_rate', C.c_float), ('rrc_alpha', C.c_float),
                ('rrc_taps', C.c_int), ('loop_bw', C.c_float), ('fll_bw', C.c_float), ('clock_omega_gain', C.c_float),
                ('clock_mu_gain', C.c_float), ('omega_rel_limit', C.c_float), ('modcod', C.c_int), ('shortframes', C.c_int),
                ('pilots', C.c_int), ('sof_threshold', C.c_float), ('max_ldpc_trials', C.c_int), ('force_ldpc_iters', C.c_int),
                ('acm_vcm', C.c_int), ('soft_plsc', C.c_int), ('pilot_aided', C.c_int)]


class TxCfg(C.Structure):
    _fields_ = [('modcod', C.c_int), ('shortframes', C.c_int), ('pilots', C.c_int), ('nframes', C.c_int), ('seed', C.c_uint64),
                ('esn0_db', C.c_double), ('cfo', C.c_double), ('timing', C.c_double), ('phase0', C.c_double), ('lead_symbols', C.c_int),
                ('circular', C.c_int), ('nsamples', C.c_int), ('vcm_n', C.c_int), ('vcm_pls', C.c_int * 64)]


class FrameStats(C.Structure):
    _fields_ = [('best_match', C.c_float), ('detect_modcod', C.c_int), ('detect_short', C.c_int), ('detect_pilots', C.c_int),
                ('fed_err', C.c_float), ('ldpc_trials', C.c_int), ('bch_corr', C.c_int), ('bbframe_bytes', C.c_int)]


def _bind_chain():
    L = lib()
    if getattr(L, '_chain_bound', False):
        return L
    L.orc_default_cfg.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(DemodCfg)]
    L.orc_default_cfg.restype = None
    L.orc_s2rx_create.argtypes = [C.POINTER(DemodCfg)]
    L.orc_s2rx_create.restype = C.c_void_p
    L.orc_s2rx_destroy.argtypes = [C.c_void_p]
    L.orc_s2rx_destroy.restype = None
    L.orc_s2rx_reset.argtypes = [C.c_void_p]
    L.orc_s2rx_process.argtypes = [C.c_void_p, C.c_int, _f32p, _u8p, C.c_int]
    L.orc_s2rx_tap.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.orc_s2rx_nco_freq.argtypes = [C.c_void_p]
    L.orc_s2rx_nco_freq.restype = C.c_float
    for name in ('orc_s2rx_agc', 'orc_s2rx_nco', 'orc_s2rx_rrc'):
        getattr(L, name).argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
        getattr(L, name).restype = None
    L.orc_s2rx_gardner.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p]
    L.orc_s2rx_fed.argtypes = [C.c_void_p, _f32p]
    L.orc_s2rx_fed.restype = C.c_float
    L.orc_s2rx_pll.argtypes = [C.c_void_p, _f32p, _f32p]
    L.orc_s2rx_pll.restype = None
    L.orc_s2rx_plhdr.argtypes = [C.c_void_p, _f32p, _f32p, _i32p]
    L.orc_s2rx_plhdr.restype = None
    L.orc_s2rx_to_soft.argtypes = [C.c_void_p, _f32p, _i8p]
    L.orc_s2rx_to_soft.restype = None
    L.orc_s2_transmit.argtypes = [C.POINTER(TxCfg), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_constellation_lut.argtypes = [C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.orc_constellation_soft_calc.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, _f32p, _i8p, _f32p]
    L.orc_constellation_soft_calc.restype = None
    L.orc_constellation_points.argtypes = [C.c_int, C.c_float, C.c_float, _f32p]
    L.orc_constellation_points.restype = None
    L.orc_math_eval.argtypes = [C.c_int, C.c_int, _f32p, C.c_void_p, _f32p, _f32p]
    L.orc_math_eval.restype = None
    L.orc_s2_deinterleave.argtypes = [C.c_int, C.c_int, C.c_int, _i8p, _i8p]
    L.orc_s2_deinterleave.restype = None
    L.orc_pl_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_pl_tables.restype = None
    L.orc_rrc_taps.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, _f32p]
    L.orc_rrc_taps.restype = None
    L.orc_gardner_bank.argtypes = [_f32p]
    L.orc_gardner_bank.restype = None
    L._chain_bound = True
    return L


def default_cfg(modcod, short=0, pilots=0, **kw):
    L = _bind_chain()
    c = DemodCfg()
    L.orc_default_cfg(modcod, short, pilots, C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def modcod_params(modcod, short=0, pilots=0):
    out = np.zeros(8, np.int32)
    g = np.zeros(2, np.float32)
    if lib().orc_modcod_params(modcod, short, pilots, out, g) != 0:
        return None
    keys = ['constel', 'bits', 'rate', 'slots', 'pilot_blocks', 'plframe', 'N', 'kbch']
    d = dict(zip(keys, [int(x) for x in out]))
    d['g1'], d['g2'] = float(g[0]), float(g[1])
    return d


def pls_info(pls):
    """(plframe symbols, kbch/8 bytes) of a PLS code modcod << 2 | short << 1 | pilots (modcod 0: dummy PLFRAME, no BBFRAME)"""
    if pls >> 2 == 0:
        return 3330, 0
    mp = modcod_params(pls >> 2, (pls >> 1) & 1, pls & 1)
    return mp['plframe'], mp['kbch'] // 8


def transmit_vcm(pls_list, nframes, seed=1, esn0_db=200.0, cfo=0.0, timing=0.0, phase0=0.0, lead_symbols=0):
    """ACM/VCM stream: frame f carries PLS code pls_list[f % len] -> (iq, [bbframe bytes per frame or None for dummy frames])"""
    L = _bind_chain()
    t = TxCfg(0, 0, 0, nframes, seed, esn0_db, cfo, timing, phase0, lead_symbols, 0, 0, len(pls_list), (C.c_int * 64)(*pls_list))
    sizes = [pls_info(pls_list[f % len(pls_list)]) for f in range(nframes)]
    nsym = lead_symbols + sum(s[0] for s in sizes)
    iq = np.zeros(2 * nsym, np.complex64)
    bb = np.zeros(sum(s[1] for s in sizes) + 8, np.uint8)
    n = L.orc_s2_transmit(C.byref(t), iq.ctypes.data, iq.size, bb.ctypes.data, None, 0)
    assert n == iq.size, (n, iq.size)
    out, pos = [], 0
    for plf, kb in sizes:
        out.append(bb[pos:pos + kb].copy() if kb else None)
        pos += kb
    return iq, out


def transmit(modcod, short=0, pilots=0, nframes=2, seed=1, esn0_db=200.0, cfo=0.0, timing=0.0, phase0=0.0, lead_symbols=0, circular=0, nsamples=0):
    """-> (iq complex64 [n], bbframes uint8 [nframes, kbch/8], symbols complex64); nsamples != 0: resampled to that many samples
    (sampling-clock error), default exactly 2 per symbol"""
    L = _bind_chain()
    t = TxCfg(modcod, short, pilots, nframes, seed, esn0_db, cfo, timing, phase0, lead_symbols, circular, nsamples, 0)
    mp = modcod_params(modcod, short, pilots)
    nsym = lead_symbols + nframes * mp['plframe']
    iq = np.zeros(nsamples if nsamples else 2 * nsym, np.complex64)
    bb = np.zeros((nframes, mp['kbch'] // 8), np.uint8)
    syms = np.zeros(nsym, np.complex64)
    n = L.orc_s2_transmit(C.byref(t), iq.ctypes.data, iq.size, bb.ctypes.data, syms.ctypes.data, syms.size)
    assert n == iq.size, (n, iq.size)
    return iq, bb, syms


class OracleRx:
    def __init__(self, cfg):
        self.L = _bind_chain()
        self.cfg = cfg
        self.h = self.L.orc_s2rx_create(C.byref(cfg))
        assert self.h
        self.mp = modcod_params(cfg.modcod, cfg.shortframes, cfg.pilots)

    def __del__(self):
        if getattr(self, 'h', None):
            self.L.orc_s2rx_destroy(self.h)
            self.h = None

    def process(self, iq):
        iq = np.ascontiguousarray(iq, np.complex64)
        if self.cfg.acm_vcm:
            return self.process_vcm(iq)
        cap = (iq.size // (2 * self.mp['plframe']) + 3) * (self.mp['kbch'] // 8)
        out = np.zeros(cap, np.uint8)
        n = self.L.orc_s2rx_process(self.h, iq.size, iq.view(np.float32), out, cap)
        return out[:n].reshape(-1, self.mp['kbch'] // 8)

    def process_vcm(self, iq):
        """ACM/VCM mode: -> list of BBFRAMEs (variable sizes, from the per-frame statistics' bbframe_bytes)"""
        iq = np.ascontiguousarray(iq, np.complex64)
        cap = iq.size // 2 + 65536           # (a frame yields fewer bytes than it has symbols)
        out = np.zeros(cap, np.uint8)
        n = self.L.orc_s2rx_process(self.h, iq.size, iq.view(np.float32), out, cap)
        frames, pos = [], 0
        for st in self.tap(4):
            if st.bbframe_bytes:
                frames.append(out[pos:pos + st.bbframe_bytes].copy())
                pos += st.bbframe_bytes
        assert pos == n, (pos, n)
        return frames

    def tap(self, which):
        n = self.L.orc_s2rx_tap(self.h, which, None)
        if which in (0, 1, 2):
            a = np.zeros(n, np.complex64)
        elif which == 3:
            a = np.zeros(n, np.int8)
        else:
            a = (FrameStats * n)()
            self.L.orc_s2rx_tap(self.h, which, C.cast(a, C.c_void_p))
            return list(a)
        if n:
            self.L.orc_s2rx_tap(self.h, which, a.ctypes.data)
        return a


def math_eval(func, a, b=None):
    """host evaluation of include/dvbs2gpu_math.h (0 sincos, 1 atan2(a, b), 2 exp, 3 log, 4 LLR clamp) -> (out0, out1)"""
    _bind_chain()
    a = np.ascontiguousarray(a, np.float32)
    o0, o1 = np.zeros_like(a), np.zeros_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, np.float32)
        bp = b.ctypes.data
    lib().orc_math_eval(int(func), a.size, a, bp, o0, o1)
    return o0, o1
