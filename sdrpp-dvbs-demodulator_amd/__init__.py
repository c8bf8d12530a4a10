"""sdrpp-dvbs-demodulator_amd -- Python plumbing over libdvbs2gpu.so (C ABI: include/dvbs2gpu.h).

The product is the HIP library; this module only loads it (ctypes), hands it device pointers of torch
tensors and the current HIP stream, and raises if the library or a GPU is missing -- there is no CPU
fallback and nothing here touches oracle/.

Import with  importlib.import_module("sdrpp-dvbs-demodulator_amd")  (the directory name has hyphens),
or use the helper  `from __graft_entry__ import load_package`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdvbs2gpu.so')

ERR_NAMES = {-1: 'ERR_ARG', -2: 'ERR_MODCOD', -3: 'ERR_HIP', -4: 'ERR_NODEVICE', -5: 'ERR_CAPACITY'}


class Dvbs2GpuError(RuntimeError):
    def __init__(self, code, text):
        super().__init__('dvbs2gpu %s (%d): %s' % (ERR_NAMES.get(code, '?'), code, text))
        self.code = code


class ModcodInfo(C.Structure):
    _fields_ = [('constellation', C.c_int32), ('bits_per_symbol', C.c_int32), ('rate', C.c_int32), ('slots', C.c_int32),
                ('pilot_blocks', C.c_int32), ('plframe_symbols', C.c_int32), ('ldpc_n', C.c_int32), ('ldpc_k', C.c_int32),
                ('kbch', C.c_int32), ('bch_t', C.c_int32), ('ldpc_edges', C.c_int32), ('g1', C.c_float), ('g2', C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# name -> (restype, argtypes); every symbol declared in include/dvbs2gpu.h
_vp = C.c_void_p
_i = C.c_int
PROTOTYPES = {
    'dvbs2gpu_version': (C.c_char_p, []),
    'dvbs2gpu_last_error': (C.c_char_p, []),
    'dvbs2gpu_create': (_i, [_i, C.POINTER(_vp)]),
    'dvbs2gpu_destroy': (None, [_vp]),
    'dvbs2gpu_modcod_info_get': (_i, [_i, _i, _i, C.POINTER(ModcodInfo)]),
    'dvbs2gpu_fec_info_get': (_i, [_i, _i, C.POINTER(ModcodInfo)]),
    'dvbs2gpu_ldpc_plan_info': (_i, [_vp, _i, _i, C.POINTER(C.c_int32)]),
    'dvbs2gpu_ldpc_decode_batch': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'dvbs2gpu_bch_decode_batch': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
    'dvbs2gpu_bb_descramble_batch': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
    'dvbs2gpu_fec_decode_batch': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
}

_lib = None


def load_library():
    """dlopen libdvbs2gpu.so (built in-tree by __graft_entry__.build()).  Fails loudly if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError('%s not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                              '(hipcc --offload-arch=gfx950); there is no CPU fallback' % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def modcod_info(modcod, shortframes=False, pilots=False):
    info = ModcodInfo()
    rc = load_library().dvbs2gpu_modcod_info_get(int(modcod), int(bool(shortframes)), int(bool(pilots)), C.byref(info))
    if rc != 0:
        raise Dvbs2GpuError(rc, load_library().dvbs2gpu_last_error().decode())
    return info.as_dict()


def fec_info(rate, shortframes=False):
    info = ModcodInfo()
    rc = load_library().dvbs2gpu_fec_info_get(int(rate), int(bool(shortframes)), C.byref(info))
    if rc != 0:
        raise Dvbs2GpuError(rc, load_library().dvbs2gpu_last_error().decode())
    return info.as_dict()


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    """One engine context on one GPU.  All tensor arguments are torch CUDA tensors on that GPU."""

    def __init__(self, device=0):
        import torch
        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise Dvbs2GpuError(-4, 'no GPU visible to torch; the engine has no CPU fallback')
        self.device = torch.device('cuda', device)
        h = C.c_void_p()
        self._check(self.lib.dvbs2gpu_create(int(device), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, 'h', None):
            self.lib.dvbs2gpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise Dvbs2GpuError(rc, self.lib.dvbs2gpu_last_error().decode())
        return rc

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def ldpc_plan_info(self, rate, shortframes=False):
        out = (C.c_int32 * 8)()
        self._check(self.lib.dvbs2gpu_ldpc_plan_info(self.h, int(rate), int(bool(shortframes)), out))
        keys = ['layers', 'max_deg', 'rec_dwords', 'sum_depth', 'blocks_per_cu', 'cus', 'edges', 'conflict_layers']
        return dict(zip(keys, list(out)))

    # ---- FEC stages --------------------------------------------------------------------------------
    def ldpc_decode(self, llr, rate, shortframes=False, max_trials=25, force=False, want_post=False):
        """llr int8 [F, N] -> (hard uint8 [F, K/8], trials int32 [F], post int8 [F, N] or None)"""
        t = self.torch
        fi = fec_info(rate, shortframes)
        assert llr.dtype == t.int8 and llr.is_cuda and llr.is_contiguous() and llr.shape[1] == fi['ldpc_n']
        F = llr.shape[0]
        hard = t.empty((F, fi['ldpc_k'] // 8), dtype=t.uint8, device=llr.device)
        trials = t.empty((F,), dtype=t.int32, device=llr.device)
        post = t.empty_like(llr) if want_post else None
        self._check(self.lib.dvbs2gpu_ldpc_decode_batch(self.h, int(rate), int(bool(shortframes)), _ptr(llr), F, int(max_trials),
                                                        int(bool(force)), _ptr(hard), _ptr(post), _ptr(trials), self._stream()))
        return hard, trials, post

    def bch_decode(self, frames, rate, shortframes=False):
        """frames uint8 [F, K/8], corrected in place -> corrections int32 [F]"""
        t = self.torch
        fi = fec_info(rate, shortframes)
        assert frames.dtype == t.uint8 and frames.is_cuda and frames.is_contiguous() and frames.shape[1] == fi['ldpc_k'] // 8
        corr = t.empty((frames.shape[0],), dtype=t.int32, device=frames.device)
        self._check(self.lib.dvbs2gpu_bch_decode_batch(self.h, int(rate), int(bool(shortframes)), _ptr(frames), frames.shape[0],
                                                       _ptr(corr), self._stream()))
        return corr

    def bb_descramble(self, frames, rate, shortframes=False):
        t = self.torch
        fi = fec_info(rate, shortframes)
        assert frames.dtype == t.uint8 and frames.is_cuda and frames.is_contiguous() and frames.shape[1] == fi['ldpc_k'] // 8
        out = t.empty((frames.shape[0], fi['kbch'] // 8), dtype=t.uint8, device=frames.device)
        self._check(self.lib.dvbs2gpu_bb_descramble_batch(self.h, int(rate), int(bool(shortframes)), _ptr(frames), frames.shape[0],
                                                          _ptr(out), self._stream()))
        return out

    def fec_decode(self, llr, rate, shortframes=False, max_trials=25, force=False, out=None, trials=None, corr=None):
        """llr int8 [F, N] -> (bbframes uint8 [F, kbch/8], trials int32 [F], bch corrections int32 [F])"""
        t = self.torch
        fi = fec_info(rate, shortframes)
        assert llr.dtype == t.int8 and llr.is_cuda and llr.is_contiguous() and llr.shape[1] == fi['ldpc_n']
        F = llr.shape[0]
        if out is None:
            out = t.empty((F, fi['kbch'] // 8), dtype=t.uint8, device=llr.device)
        if trials is None:
            trials = t.empty((F,), dtype=t.int32, device=llr.device)
        if corr is None:
            corr = t.empty((F,), dtype=t.int32, device=llr.device)
        self._check(self.lib.dvbs2gpu_fec_decode_batch(self.h, int(rate), int(bool(shortframes)), _ptr(llr), F, int(max_trials),
                                                       int(bool(force)), _ptr(out), _ptr(trials), _ptr(corr), self._stream()))
        return out, trials, corr
