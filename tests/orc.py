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
