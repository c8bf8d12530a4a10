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
# DVBS2GPU_LIB: development aid (tools/ab.sh runs the bench against a variant build under /tmp without touching the in-tree library)
LIB_PATH = os.environ.get('DVBS2GPU_LIB') or os.path.join(_HERE, 'libdvbs2gpu.so')

ERR_ARG, ERR_MODCOD, ERR_HIP, ERR_NODEVICE, ERR_CAPACITY = -1, -2, -3, -4, -5
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


class DemodCfg(C.Structure):
    """dvbs2gpu_demod_cfg"""
    _fields_ = [('symbolrate', C.c_double), ('samplerate', C.c_double), ('agc_rate', C.c_float), ('rrc_alpha', C.c_float),
                ('rrc_taps', C.c_int32), ('loop_bw', C.c_float), ('fll_bw', C.c_float), ('clock_omega_gain', C.c_float),
                ('clock_mu_gain', C.c_float), ('omega_rel_limit', C.c_float), ('modcod', C.c_int32), ('shortframes', C.c_int32),
                ('pilots', C.c_int32), ('sof_threshold', C.c_float), ('max_ldpc_trials', C.c_int32), ('force_ldpc_iters', C.c_int32),
                ('acm_vcm', C.c_int32), ('soft_plsc', C.c_int32), ('pilot_aided', C.c_int32)]


STAGE_NAMES = ('frontend', 'rrc', 'plsync', 'loops', 'demap', 'ldpc', 'bch', 'deliver')


class StageTimes(C.Structure):
    """dvbs2gpu_stage_times"""
    _fields_ = [('ms', C.c_double * 8), ('launches', C.c_int64 * 8), ('units', C.c_int64 * 8)]


class FrameStats(C.Structure):
    """dvbs2gpu_frame_stats"""
    _fields_ = [('pl_sync_best_match', C.c_float), ('detected_modcod', C.c_int32), ('detected_shortframes', C.c_int32),
                ('detected_pilots', C.c_int32), ('coarse_freq_err', C.c_float), ('ldpc_trials', C.c_int32), ('bch_corrections', C.c_int32),
                ('bbframe_bytes', C.c_int32)]


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
    'dvbs2gpu_set_option': (_i, [_vp, C.c_char_p, _i]),
    'dvbs2gpu_preinit': (_i, []),
    'dvbs2gpu_device_count': (_i, []),
    'dvbs2gpu_fleet_plan': (_i, [C.POINTER(C.c_int32), C.POINTER(C.c_double), _i, _i, C.c_double, C.POINTER(C.c_int32)]),
    'dvbs2gpu_fleet_create': (_i, [C.POINTER(_i), _i, C.POINTER(_vp)]),
    'dvbs2gpu_fleet_destroy': (None, [_vp]),
    'dvbs2gpu_fleet_size': (_i, [_vp]),
    'dvbs2gpu_fleet_assign': (_i, [_vp, _vp, _i, _i, C.c_double, C.POINTER(C.c_int32)]),
    'dvbs2gpu_fleet_set_pipelined': (_i, [_vp, _i]),
    'dvbs2gpu_fleet_reset': (_i, [_vp]),
    'dvbs2gpu_fleet_process_batch': (_i, [_vp, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp), _i, C.POINTER(_i)]),
    'dvbs2gpu_fleet_get_stats': (_i, [_vp, _i, C.POINTER(FrameStats), _i]),
    'dvbs2gpu_get_state': (_i, [_vp, C.c_char_p, C.POINTER(C.c_longlong)]),
    'dvbs2gpu_debug_last_fec_job': (_i, [_vp, _i, C.POINTER(C.c_longlong), C.POINTER(C.c_int32), C.POINTER(_vp), _i]),
    'dvbs2gpu_ldpc_plan_dump': (_i, [_i, _i, _vp, _vp, _vp, C.POINTER(C.c_int32)]),
    'dvbs2gpu_ldpc_plan_info': (_i, [_vp, _i, _i, C.POINTER(C.c_int32)]),
    'dvbs2gpu_ldpc_decoder_form': (_i, [_vp, _i, _i]),
    'dvbs2gpu_ldpc_wave_plan_dump': (_i, [_i, _i, _vp, _vp, _vp, C.POINTER(C.c_int32)]),
    'dvbs2gpu_ldpc_addr_table_dump': (_i, [_i, _i, _vp, C.POINTER(C.c_int32)]),
    'dvbs2gpu_ldpc_split_plan_dump': (_i, [_i, _i, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32)]),
    'dvbs2gpu_ldpc_decode_batch': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'dvbs2gpu_bch_decode_batch': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
    'dvbs2gpu_bb_descramble_batch': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
    'dvbs2gpu_fec_decode_batch': (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'dvbs2gpu_demap_batch': (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    'dvbs2gpu_deinterleave_batch': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
    'dvbs2gpu_set_stage_timing': (_i, [_vp, _i]),
    'dvbs2gpu_get_stage_times': (_i, [_vp, _vp]),
    'dvbs2gpu_math_eval': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'dvbs2gpu_demod_default_cfg': (None, [_i, _i, _i, C.POINTER(DemodCfg)]),
    'dvbs2gpu_demod_create': (_i, [_vp, C.POINTER(DemodCfg), _i, C.POINTER(_vp)]),
    'dvbs2gpu_demod_destroy': (None, [_vp]),
    'dvbs2gpu_demod_reset': (_i, [_vp]),
    'dvbs2gpu_demod_set_params': (_i, [_vp, _i, _i, _i, C.c_float, _i]),
    'dvbs2gpu_demod_get_kbch': (_i, [_vp]),
    'dvbs2gpu_demod_process': (_i, [_vp, _i, _vp, _vp, _i]),
    'dvbs2gpu_demod_process_batch': (_i, [C.POINTER(_vp), _i, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp), _i, C.POINTER(_i)]),
    'dvbs2gpu_set_pipelined': (_i, [_vp, _i]),
    'dvbs2gpu_demod_get_stats': (_i, [_vp, C.POINTER(FrameStats), _i]),
    'dvbs2gpu_demod_get_nco_freq': (C.c_float, [_vp]),
    'dvbs2gpu_demod_get_tap': (_i, [_vp, _i, _vp, _i]),
    # DVB-S inner code
    'dvbs2gpu_dvbs_slice': (_i, [_vp, _vp, _i, _vp, _vp]),
    'dvbs2gpu_ccdec_create': (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    'dvbs2gpu_ccdec_destroy': (None, [_vp]),
    'dvbs2gpu_ccdec_work_batch': (_i, [_vp, _vp, C.c_int64, _i, _i, _vp, _vp]),
    'dvbs2gpu_viterbi_create': (_i, [_vp, _i, C.c_float, _i, C.POINTER(_vp)]),
    'dvbs2gpu_viterbi_reset': (_i, [_vp]),
    'dvbs2gpu_viterbi_destroy': (None, [_vp]),
    'dvbs2gpu_viterbi_work_batch': (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'dvbs2gpu_forney_create': (_i, [_vp, _i, C.POINTER(_vp)]),
    'dvbs2gpu_forney_destroy': (None, [_vp]),
    'dvbs2gpu_forney_deinterleave_batch': (_i, [_vp, _vp, _i, _vp, _vp]),
    'dvbs2gpu_dvbs_demod_default_cfg': (None, [_vp]),
    'dvbs2gpu_dvbs_demod_create': (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    'dvbs2gpu_dvbs_demod_reset': (_i, [_vp]),
    'dvbs2gpu_dvbs_demod_destroy': (None, [_vp]),
    'dvbs2gpu_dvbs_demod_process': (_i, [_vp, _i, _vp, _vp, _i]),
    'dvbs2gpu_dvbs_demod_process_batch': (_i, [_vp, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp), _i, C.POINTER(_i)]),
    'dvbs2gpu_dvbs_demod_get_stats': (_i, [_vp, _vp]),
    'dvbs2gpu_dvbs_demod_get_tap': (_i, [_vp, _i, _i, _vp, _i]),
    'dvbs2gpu_dvbs_tail_create': (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    'dvbs2gpu_dvbs_tail_reset': (_i, [_vp]),
    'dvbs2gpu_dvbs_tail_destroy': (None, [_vp]),
    'dvbs2gpu_dvbs_tail_process_batch': (_i, [_vp, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp), _i, C.POINTER(_i), _vp]),
    'dvbs2gpu_dvbs_tail_get_stats': (_i, [_vp, _i, C.POINTER(C.c_int32)]),
    'dvbs2gpu_dvbs_tail_get_tap': (_i, [_vp, _i, _i, _vp, _i]),
    'dvbs2gpu_dvbs_tail_rs_stage': (_i, [_vp, _vp, _i, _i, _vp, _i]),
    'dvbs2gpu_dvbs_depuncture': (_i, [_vp, _i, _i, _vp, _i, _vp, _i, C.POINTER(C.c_int32)]),
    'dvbs2gpu_demod_get_frame_positions': (_i, [_vp, C.POINTER(C.c_int64), _i]),
    'dvbs2gpu_segrx_create': (_i, [_vp, C.POINTER(DemodCfg), _i, _i, _i, C.POINTER(_vp)]),
    'dvbs2gpu_segrx_reset': (_i, [_vp]),
    'dvbs2gpu_segrx_destroy': (None, [_vp]),
    'dvbs2gpu_segrx_chunk_samples': (C.c_longlong, [_vp]),
    'dvbs2gpu_segrx_process': (_i, [_vp, _vp, C.c_longlong, _vp, C.c_longlong]),
    'dvbs2gpu_segrx_get_stats': (_i, [_vp, C.POINTER(C.c_int32)]),
    'dvbs2gpu_dvbs_segrx_create': (_i, [_vp, _vp, _i, _i, _i, C.POINTER(_vp)]),
    'dvbs2gpu_dvbs_segrx_reset': (_i, [_vp]),
    'dvbs2gpu_dvbs_segrx_destroy': (None, [_vp]),
    'dvbs2gpu_dvbs_segrx_chunk_samples': (C.c_longlong, [_vp]),
    'dvbs2gpu_dvbs_segrx_process': (_i, [_vp, _vp, C.c_longlong, _vp, C.c_longlong]),
    'dvbs2gpu_dvbs_segrx_get_stats': (_i, [_vp, C.POINTER(C.c_int32)]),
    'dvbs2gpu_dvbs_process_ts': (_i, [_vp, _vp, _i, _vp, _vp, _i]),
    'dvbs2gpu_dvbs_segrx_find_join': (C.c_longlong, [_vp, C.c_longlong, _vp, C.c_longlong, C.POINTER(C.c_int)]),
    'dvbs2gpu_bbts_create': (_i, [_vp, _i, _i, _i, C.POINTER(_vp)]),
    'dvbs2gpu_bbts_set_frame_size': (_i, [_vp, _i]),
    'dvbs2gpu_bbts_destroy': (None, [_vp]),
    'dvbs2gpu_bbts_process_batch': (_i, [_vp, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_vp), _i, C.POINTER(_i), _vp]),
    'dvbs2gpu_bbts_work': (_i, [_vp, _vp, _i, _vp, _i]),
    'dvbs2gpu_bbts_get_stats': (_i, [_vp, _i, C.POINTER(C.c_int32), _i]),
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


def ldpc_split_plan(rate, shortframes=False):
    """Host-only: the half-row LDPC decoder's plan (csrc/ldpc_split_plan.h) as numpy arrays; None for codes that decoder does not take."""
    import numpy as np
    lib = load_library()
    cnt = (C.c_int32 * 6)()
    rc = lib.dvbs2gpu_ldpc_split_plan_dump(int(rate), int(bool(shortframes)), None, None, None, None, cnt)
    if rc != 0:
        raise Dvbs2GpuError(rc, lib.dvbs2gpu_last_error().decode())
    npl, npw, hs, rec_total, nwords, rec_dw = list(cnt)
    if npl == 0:
        return None
    layers = np.zeros((npl, 4), np.uint32); table = np.zeros(nwords, np.uint32)
    row_of = np.zeros((npl, 384), np.int32); layer_of = np.zeros(npl, np.int32)
    rc = lib.dvbs2gpu_ldpc_split_plan_dump(int(rate), int(bool(shortframes)), layers.ctypes.data, table.ctypes.data, row_of.ctypes.data, layer_of.ctypes.data, cnt)
    if rc != 0:
        raise Dvbs2GpuError(rc, lib.dvbs2gpu_last_error().decode())
    return {'npl': npl, 'npw': npw, 'hs': hs, 'rec_total': rec_total, 'rec_dwords': rec_dw, 'kind': (layers[:, 0] & 0xff).astype(int), 'nw': ((layers[:, 0] >> 8) & 0xff).astype(int),
            'nc': ((layers[:, 0] >> 16) & 15).astype(int), 'noprev': ((layers[:, 0] >> 20) & 1).astype(int), 'aux': layers[:, 1].astype(int), 'rec_off': layers[:, 2].astype(int),
            'ent_off': layers[:, 3].astype(int), 'table': table[:npl * 768 * npw].reshape(npl, 768, npw), 'words': table, 'row_of': row_of, 'layer': layer_of}


class FleetEntry(C.Structure):
    """dvbs2gpu_fleet_entry"""
    _fields_ = [('cfg', DemodCfg), ('weight', C.c_double), ('max_samples', C.c_int32), ('reserved', C.c_int32)]


def fleet_plan(modcods, weights, world, tolerance=0.25):
    """Host-only: the fleet's placement rule (dvbs2gpu_fleet_plan) -> member index of every transponder"""
    lib = load_library()
    n = len(modcods)
    m = (C.c_int32 * max(n, 1))(*[int(x) for x in modcods])
    w = (C.c_double * max(n, 1))(*[float(x) for x in weights])
    out = (C.c_int32 * max(n, 1))()
    rc = lib.dvbs2gpu_fleet_plan(m, w, n, int(world), float(tolerance), out)
    if rc != 0:
        raise Dvbs2GpuError(rc, lib.dvbs2gpu_last_error().decode())
    return [int(out[i]) for i in range(n)]


class Fleet:
    """dvbs2gpu_fleet_*: a transponder table over several (logical) devices behind the C ABI; host buffers in, BBFRAMEs out in table order"""

    def __init__(self, devices):
        self.lib = load_library()
        h = _vp()
        d = (_i * len(devices))(*[int(x) for x in devices])
        rc = self.lib.dvbs2gpu_fleet_create(d, len(devices), C.byref(h))
        if rc != 0:
            raise Dvbs2GpuError(rc, self.lib.dvbs2gpu_last_error().decode())
        self.h, self.nt, self.cap = h, 0, 0

    def _check(self, rc):
        if rc < 0:
            raise Dvbs2GpuError(rc, self.lib.dvbs2gpu_last_error().decode())
        return rc

    def assign(self, cfgs, max_samples, out_cap, weights=None, tolerance=0.25):
        """cfgs: one DemodCfg per transponder -> member index of every transponder"""
        n = len(cfgs)
        tab = (FleetEntry * max(n, 1))()
        for i, c in enumerate(cfgs):
            tab[i].cfg = c
            tab[i].weight = float(weights[i]) if weights is not None else 0.0
            tab[i].max_samples = int(max_samples)
        out = (C.c_int32 * max(n, 1))()
        self._check(self.lib.dvbs2gpu_fleet_assign(self.h, C.cast(tab, _vp), n, int(out_cap), float(tolerance), out))
        self.nt, self.cap = n, int(out_cap)
        return [int(out[i]) for i in range(n)]

    def set_pipelined(self, on):
        self._check(self.lib.dvbs2gpu_fleet_set_pipelined(self.h, int(bool(on))))

    def process(self, iqs):
        """iqs: one numpy complex64 array per transponder (may be empty) -> list of numpy uint8 arrays (BBFRAME bytes, table order)"""
        import numpy as np
        iqs = [np.ascontiguousarray(x, np.complex64) for x in iqs]
        outs = [np.zeros(self.cap, np.uint8) for _ in range(self.nt)]
        pin = (_vp * self.nt)(*[x.ctypes.data for x in iqs])
        cnt = (_i * self.nt)(*[int(x.size) for x in iqs])
        pout = (_vp * self.nt)(*[o.ctypes.data for o in outs])
        nb = (_i * self.nt)()
        self._check(self.lib.dvbs2gpu_fleet_process_batch(self.h, pin, cnt, pout, self.cap, nb))
        return [outs[i][:nb[i]] for i in range(self.nt)]

    def close(self):
        if getattr(self, 'h', None):
            self.lib.dvbs2gpu_fleet_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    """One engine context on one GPU.  All tensor arguments are torch CUDA tensors on that GPU."""

    def __init__(self, device=0, options=None):
        """options: development / test options of the context, {name: int} (dvbs2gpu_set_option)"""
        import torch
        self.torch = torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise Dvbs2GpuError(-4, 'no GPU visible to torch; the engine has no CPU fallback')
        self.device = torch.device('cuda', device)
        h = C.c_void_p()
        self._check(self.lib.dvbs2gpu_create(int(device), C.byref(h)))
        self.h = h
        for name, value in (options or {}).items():
            self.set_option(name, value)

    def set_option(self, name, value):
        self._check(self.lib.dvbs2gpu_set_option(self.h, str(name).encode(), int(value)))

    def get_state(self, name):
        """read-only introspection (dvbs2gpu_get_state): 'kernel_launches', 'g_prio_duty', 'stage_pipeline_on', 'fec_part_on', ..."""
        v = C.c_longlong()
        self._check(self.lib.dvbs2gpu_get_state(self.h, str(name).encode(), C.byref(v)))
        return int(v.value)

    def last_fec_job(self, slot=0):
        """the pipelined CCM decoder job group `slot` delivered last (dvbs2gpu_debug_last_fec_job): dict with the device pointers of its LLRs / BBFRAMEs, the
        frame count, N, kb, the code, and `first` (first pooled frame of every stream of the job) + `handles` (the streams' demod handles)"""
        import numpy as np
        o = (C.c_longlong * 10)()
        self._check(self.lib.dvbs2gpu_debug_last_fec_job(self.h, int(slot), o, None, None, 0))
        n = int(o[3])
        first = (C.c_int32 * (n + 1))()
        hs = (_vp * max(n, 1))()
        self._check(self.lib.dvbs2gpu_debug_last_fec_job(self.h, int(slot), o, first, hs, n + 1))
        return dict(d_llr=int(o[0]), d_bb=int(o[1]), nf=int(o[2]), n=n, N=int(o[4]), kb=int(o[5]), rate=int(o[6]), short=int(o[7]), max_trials=int(o[8]), force=int(o[9]),
                    first=np.array(first[:n + 1], dtype=np.int64), handles=[hs[i] for i in range(n)])

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

    def ldpc_decoder_form(self, rate, shortframes=False):
        """0 lane per row, 1 wave per frame, 2 half a row per lane: the kernel that serves the code under this engine's options"""
        r = self.lib.dvbs2gpu_ldpc_decoder_form(self.h, int(rate), int(bool(shortframes)))
        if r < 0:
            self._check(r)
        return r

    def ldpc_split_plan(self, rate, shortframes=False):
        return ldpc_split_plan(rate, shortframes)

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

    def demap(self, frames, modcod, shortframes=False, pilots=False):
        """frames complex64 [F, plframe] (PLL output) -> LLR int8 [F, N]"""
        t = self.torch
        mi = modcod_info(modcod, shortframes, pilots)
        assert frames.dtype == t.complex64 and frames.is_cuda and frames.is_contiguous() and frames.shape[1] == mi['plframe_symbols']
        llr = t.empty((frames.shape[0], mi['ldpc_n']), dtype=t.int8, device=frames.device)
        self._check(self.lib.dvbs2gpu_demap_batch(self.h, int(modcod), int(bool(shortframes)), int(bool(pilots)), _ptr(frames),
                                                  frames.shape[0], _ptr(llr), self._stream()))
        return llr

    def deinterleave(self, llr, modcod, shortframes=False):
        """S2Deinterleaver::deinterleave on int8 [F, N] frames (the demapper's index function as its own stage)"""
        t = self.torch
        assert llr.dtype == t.int8 and llr.is_cuda and llr.is_contiguous() and llr.dim() == 2
        out = t.empty_like(llr)
        self._check(self.lib.dvbs2gpu_deinterleave_batch(self.h, int(modcod), int(bool(shortframes)), _ptr(llr), llr.shape[0], _ptr(out), self._stream()))
        return out

    def dvbs_depuncture(self, period, mode, data, state4, fill=0):
        """stage entry: the Viterbi kernel's de-puncturers / soft rotation on host bytes -> (output bytes incl. one byte beyond the
        returned count, count, new state4)"""
        import numpy as np
        data = np.ascontiguousarray(data).view(np.uint8)
        out = np.full(2 * data.size + 64, fill, np.uint8)
        st = (C.c_int32 * 4)(*[int(x) for x in state4])
        n = self._check(self.lib.dvbs2gpu_dvbs_depuncture(self.h, int(period), int(mode), C.c_void_p(data.ctypes.data), data.size,
                                                          C.c_void_p(out.ctypes.data), out.size, st))
        return out, n, list(st)

    def math_eval(self, func, a, b=None):
        """include/dvbs2gpu_math.h evaluated on the device (0 sincos, 1 atan2(a, b), 2 exp, 3 log, 4 LLR clamp) -> (out0, out1)"""
        t = self.torch
        assert a.dtype == t.float32 and a.is_cuda and a.is_contiguous()
        o0, o1 = t.empty_like(a), t.empty_like(a)
        self._check(self.lib.dvbs2gpu_math_eval(self.h, int(func), a.numel(), _ptr(a), _ptr(b) if b is not None else None, _ptr(o0),
                                                _ptr(o1), self._stream()))
        return o0, o1

    def set_stage_timing(self, on):
        self._check(self.lib.dvbs2gpu_set_stage_timing(self.h, int(bool(on))))

    def stage_times(self):
        """per-stage device times since the previous call: {stage: (ms, launches, units)}"""
        st = StageTimes()
        self._check(self.lib.dvbs2gpu_get_stage_times(self.h, C.byref(st)))
        return {STAGE_NAMES[i]: (st.ms[i], st.launches[i], st.units[i]) for i in range(8)}

    def set_pipelined(self, on):
        """FEC of call k overlaps the front end of call k+1; BBFRAMEs are delivered one process_batch call later"""
        self._check(self.lib.dvbs2gpu_set_pipelined(self.h, int(bool(on))))

    def default_cfg(self, modcod, shortframes=False, pilots=False, **kw):
        c = DemodCfg()
        self.lib.dvbs2gpu_demod_default_cfg(int(modcod), int(bool(shortframes)), int(bool(pilots)), C.byref(c))
        for k, v in kw.items():
            setattr(c, k, v)
        return c

    def demod(self, cfg, max_samples=1000000):
        return Demod(self, cfg, max_samples)

    def process_batch(self, demods, iq_tensors, out_tensors):
        """One call for many streams: iq_tensors[i] complex64 CUDA 1-D, out_tensors[i] uint8 CUDA buffers.
        Returns the list of byte counts."""
        n = len(demods)
        hs = (C.c_void_p * n)(*[d.h for d in demods])
        iq = (C.c_void_p * n)(*[t.data_ptr() for t in iq_tensors])
        cnt = (C.c_int * n)(*[int(t.numel()) for t in iq_tensors])
        out = (C.c_void_p * n)(*[t.data_ptr() for t in out_tensors])
        nb = (C.c_int * n)()
        cap = min(int(t.numel()) for t in out_tensors)
        self._check(self.lib.dvbs2gpu_demod_process_batch(hs, n, iq, cnt, out, cap, nb))
        return list(nb)

    def prepare_batch(self, demods, iq_tensors, out_tensors):
        """the argument arrays of process_batch built once for a fixed set of buffers (thousands of streams: building them per call costs
        tens of milliseconds of Python); returns a callable that runs one call and returns the byte counts (numpy int32)"""
        import numpy as np
        n = len(demods)
        hs = (C.c_void_p * n)(*[d.h for d in demods])
        iq = (C.c_void_p * n)(*[t.data_ptr() for t in iq_tensors])
        cnt = (C.c_int * n)(*[int(t.numel()) for t in iq_tensors])
        out = (C.c_void_p * n)(*[t.data_ptr() for t in out_tensors])
        nb = (C.c_int * n)()
        cap = min(int(t.numel()) for t in out_tensors)
        keep = (list(iq_tensors), list(out_tensors))

        def run(_keep=keep):
            self._check(self.lib.dvbs2gpu_demod_process_batch(hs, n, iq, cnt, out, cap, nb))
            return np.frombuffer(nb, dtype=np.int32).copy()
        return run


class Demod:
    """One DVB-S2 transponder stream: mirror of DVBS2Demod (init/process/reset/setDemodParams/getKBCH)."""

    def __init__(self, engine, cfg, max_samples=1000000):
        self.eng = engine
        self.lib = engine.lib
        self.cfg = cfg
        self.info = modcod_info(cfg.modcod, cfg.shortframes, cfg.pilots)
        h = C.c_void_p()
        engine._check(self.lib.dvbs2gpu_demod_create(engine.h, C.byref(cfg), int(max_samples), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, 'h', None):
            self.lib.dvbs2gpu_demod_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.eng._check(self.lib.dvbs2gpu_demod_reset(self.h))

    def get_kbch(self):
        return self.lib.dvbs2gpu_demod_get_kbch(self.h)

    def set_params(self, modcod, shortframes, pilots, sof_threshold=0.6, max_ldpc_trials=16):
        """DVBS2Demod::setDemodParams (module_dvbs2_demod.cpp:118-168)"""
        self.eng._check(self.lib.dvbs2gpu_demod_set_params(self.h, int(modcod), int(bool(shortframes)), int(bool(pilots)), float(sof_threshold), int(max_ldpc_trials)))
        self.info = modcod_info(modcod, bool(shortframes), bool(pilots))

    def process(self, iq):
        """iq: numpy complex64 1-D (host, 2 sps) -> numpy uint8 [frames, kbch/8]"""
        import numpy as np
        iq = np.ascontiguousarray(iq, np.complex64)
        if self.cfg.acm_vcm:
            return self.process_vcm(iq)
        kb = self.info['kbch'] // 8
        cap = (iq.size // (2 * self.info['plframe_symbols']) + 4) * kb
        out = np.zeros(cap, np.uint8)
        n = self.lib.dvbs2gpu_demod_process(self.h, int(iq.size), C.c_void_p(iq.ctypes.data), C.c_void_p(out.ctypes.data), cap)
        self.eng._check(n)
        return out[:n].reshape(-1, kb)

    def process_vcm(self, iq):
        """ACM/VCM mode: -> list of BBFRAMEs (numpy uint8, sizes from the per-frame statistics' bbframe_bytes)"""
        import numpy as np
        iq = np.ascontiguousarray(iq, np.complex64)
        cap = iq.size // 2 + 65536 + 40000
        out = np.zeros(cap, np.uint8)
        n = self.eng._check(self.lib.dvbs2gpu_demod_process(self.h, int(iq.size), C.c_void_p(iq.ctypes.data), C.c_void_p(out.ctypes.data), cap))
        frames, pos = [], 0
        for st in self.stats():
            if st.bbframe_bytes:
                frames.append(out[pos:pos + st.bbframe_bytes].copy())
                pos += st.bbframe_bytes
        assert pos == n, (pos, n)
        return frames

    def stats(self):
        n = self.lib.dvbs2gpu_demod_get_stats(self.h, None, 0)
        arr = (FrameStats * max(n, 1))()
        self.lib.dvbs2gpu_demod_get_stats(self.h, arr, n)
        return [arr[i] for i in range(n)]

    def nco_freq(self):
        return float(self.lib.dvbs2gpu_demod_get_nco_freq(self.h))

    def tap(self, which):
        import numpy as np
        n = self.eng._check(self.lib.dvbs2gpu_demod_get_tap(self.h, which, None, 0))
        a = np.zeros(n, np.int8 if which == 3 else np.complex64)
        if n:
            self.eng._check(self.lib.dvbs2gpu_demod_get_tap(self.h, which, C.c_void_p(a.ctypes.data), n))
        return a


class _Handle:
    _destroy = None

    def close(self):
        if getattr(self, 'h', None):
            getattr(self.lib, self._destroy)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CcDecoderBatch(_Handle):
    """`nstreams` chained CCDecoder objects (dvbs/viterbi/cc_decoder.cpp) of `frame_size` bits."""
    _destroy = 'dvbs2gpu_ccdec_destroy'

    def __init__(self, engine, nstreams, frame_size):
        self.eng, self.lib, self.nstreams, self.frame_size = engine, engine.lib, nstreams, frame_size
        h = C.c_void_p()
        engine._check(self.lib.dvbs2gpu_ccdec_create(engine.h, nstreams, frame_size, C.byref(h)))
        self.h = h

    def work(self, soft, nblocks, block_stride):
        """soft uint8 CUDA [nstreams, L]; block b of a stream starts at b*block_stride.  -> bits uint8 [nstreams, nblocks, frame]"""
        t = self.eng.torch
        assert soft.dtype == t.uint8 and soft.is_cuda and soft.is_contiguous() and soft.shape[0] == self.nstreams
        assert (nblocks - 1) * block_stride + 2 * (self.frame_size + 6) <= soft.shape[1]
        bits = t.empty((self.nstreams, nblocks, self.frame_size), dtype=t.uint8, device=soft.device)
        self.eng._check(self.lib.dvbs2gpu_ccdec_work_batch(self.h, _ptr(soft), soft.shape[1], block_stride, nblocks, _ptr(bits), self.eng._stream()))
        return bits


class ViterbiStats(C.Structure):
    _fields_ = [('ber', C.c_float), ('state', C.c_int32), ('rate', C.c_int32), ('phase', C.c_int32), ('shift', C.c_int32)]


class ViterbiBatch(_Handle):
    """`nstreams` Viterbi_DVBS objects (dvbs/viterbi_all.cpp) as DVBSDemod::init creates them."""
    _destroy = 'dvbs2gpu_viterbi_destroy'

    def __init__(self, engine, nstreams, ber_threshold=0.15, max_outsync=20):
        self.eng, self.lib, self.nstreams = engine, engine.lib, nstreams
        h = C.c_void_p()
        engine._check(self.lib.dvbs2gpu_viterbi_create(engine.h, nstreams, ber_threshold, max_outsync, C.byref(h)))
        self.h = h

    def reset(self):
        self.eng._check(self.lib.dvbs2gpu_viterbi_reset(self.h))

    def work(self, soft):
        """soft int8 CUDA [nstreams, nblocks, 8192] -> (bits uint8 [nstreams, nblocks, 8192], nbits int32 [nstreams, nblocks],
        stats int32 [nstreams, nblocks, 5] (ber as float bits, state, rate, phase, shift))"""
        t = self.eng.torch
        assert soft.dtype == t.int8 and soft.is_cuda and soft.is_contiguous() and soft.shape[0] == self.nstreams and soft.shape[2] == 8192
        nb = soft.shape[1]
        bits = t.zeros((self.nstreams, nb, 8192), dtype=t.uint8, device=soft.device)
        nbits = t.zeros((self.nstreams, nb), dtype=t.int32, device=soft.device)
        stats = t.zeros((self.nstreams, nb, 5), dtype=t.int32, device=soft.device)
        self.eng._check(self.lib.dvbs2gpu_viterbi_work_batch(self.h, _ptr(soft), nb, _ptr(bits), _ptr(nbits), _ptr(stats), self.eng._stream()))
        return bits, nbits, stats


class ForneyBatch(_Handle):
    """`nstreams` DVBSInterleaving de-interleavers (dvbs/dvbs_interleaving.h)."""
    _destroy = 'dvbs2gpu_forney_destroy'

    def __init__(self, engine, nstreams):
        self.eng, self.lib, self.nstreams = engine, engine.lib, nstreams
        h = C.c_void_p()
        engine._check(self.lib.dvbs2gpu_forney_create(engine.h, nstreams, C.byref(h)))
        self.h = h

    def deinterleave(self, data):
        """data uint8 CUDA [nstreams, nbytes] -> same shape"""
        t = self.eng.torch
        assert data.dtype == t.uint8 and data.is_cuda and data.is_contiguous() and data.shape[0] == self.nstreams
        out = t.empty_like(data)
        self.eng._check(self.lib.dvbs2gpu_forney_deinterleave_batch(self.h, _ptr(data), data.shape[1], _ptr(out), self.eng._stream()))
        return out


def dvbs_slice(engine, iq):
    """iq complex64 CUDA 1-D -> int8 [2n] soft bits (DVBSymToSoftBlock conversion)"""
    t = engine.torch
    assert iq.dtype == t.complex64 and iq.is_cuda and iq.is_contiguous()
    out = t.empty(2 * iq.numel(), dtype=t.int8, device=iq.device)
    engine._check(engine.lib.dvbs2gpu_dvbs_slice(engine.h, _ptr(iq), iq.numel(), _ptr(out), engine._stream()))
    return out


class DvbsCfg(C.Structure):
    _fields_ = [('symbolrate', C.c_double), ('samplerate', C.c_double), ('agc_rate', C.c_float), ('rrc_alpha', C.c_float),
                ('rrc_taps', C.c_int32), ('loop_bw', C.c_float), ('fll_bw', C.c_float), ('clock_omega_gain', C.c_float),
                ('clock_mu_gain', C.c_float), ('omega_rel_limit', C.c_float), ('viterbi_ber_threshold', C.c_float),
                ('viterbi_max_outsync', C.c_int32)]


class DvbsDemodBank(_Handle):
    """`nstreams` DVB-S receivers: mirror of DVBSDemod from the input samples to the Viterbi output (module_dvbs_demod.cpp:78-81)."""
    _destroy = 'dvbs2gpu_dvbs_demod_destroy'

    def __init__(self, engine, nstreams=1, max_samples=1 << 20, **kw):
        self.eng, self.lib, self.nstreams, self.max_samples = engine, engine.lib, nstreams, max_samples
        self.cfg = DvbsCfg()
        self.lib.dvbs2gpu_dvbs_demod_default_cfg(C.byref(self.cfg))
        for k, v in kw.items():
            setattr(self.cfg, k, v)
        h = C.c_void_p()
        engine._check(self.lib.dvbs2gpu_dvbs_demod_create(engine.h, C.byref(self.cfg), nstreams, max_samples, C.byref(h)))
        self.h = h

    def reset(self):
        self.eng._check(self.lib.dvbs2gpu_dvbs_demod_reset(self.h))

    def process(self, iq):
        """single-stream bank: numpy complex64 (host) -> numpy uint8 decoded bits"""
        import numpy as np
        iq = np.ascontiguousarray(iq, np.complex64)
        out = np.zeros(iq.size + 4 * 8192, np.uint8)
        n = self.eng._check(self.lib.dvbs2gpu_dvbs_demod_process(self.h, int(iq.size), C.c_void_p(iq.ctypes.data), C.c_void_p(out.ctypes.data), out.size))
        return out[:n]

    def process_batch(self, iq_tensors, out_tensors):
        n = self.nstreams
        iq = (C.c_void_p * n)(*[t.data_ptr() for t in iq_tensors])
        cnt = (C.c_int * n)(*[int(t.numel()) for t in iq_tensors])
        out = (C.c_void_p * n)(*[t.data_ptr() for t in out_tensors])
        nb = (C.c_int * n)()
        cap = min(int(t.numel()) for t in out_tensors)
        self.eng._check(self.lib.dvbs2gpu_dvbs_demod_process_batch(self.h, iq, cnt, out, cap, nb))
        return list(nb)

    def stats(self):
        arr = (ViterbiStats * self.nstreams)()
        self.eng._check(self.lib.dvbs2gpu_dvbs_demod_get_stats(self.h, arr))
        return [arr[i] for i in range(self.nstreams)]

    def symbols(self, stream=0):
        import numpy as np
        n = self.eng._check(self.lib.dvbs2gpu_dvbs_demod_get_tap(self.h, stream, 0, None, 0))
        a = np.zeros(n, np.complex64)
        if n:
            self.eng._check(self.lib.dvbs2gpu_dvbs_demod_get_tap(self.h, stream, 0, C.c_void_p(a.ctypes.data), n))
        return a

    def loop_state(self, stream=0):
        import numpy as np
        a = np.zeros(8, np.float32)
        self.eng._check(self.lib.dvbs2gpu_dvbs_demod_get_tap(self.h, stream, 1, C.c_void_p(a.ctypes.data), 8))
        return a


class DvbsTailBank(_Handle):
    """TS deframer + Forney de-interleaver + RS(204,188) + energy dispersal for `nstreams` DVB-S streams (module_dvbs_demod.cpp:82-99)."""
    _destroy = 'dvbs2gpu_dvbs_tail_destroy'

    def __init__(self, engine, nstreams=1, max_bits=1 << 18):
        self.eng, self.lib, self.nstreams, self.max_bits = engine, engine.lib, nstreams, max_bits
        h = C.c_void_p()
        engine._check(self.lib.dvbs2gpu_dvbs_tail_create(engine.h, nstreams, max_bits, C.byref(h)))
        self.h = h

    def reset(self):
        self.eng._check(self.lib.dvbs2gpu_dvbs_tail_reset(self.h))

    def process_batch(self, bit_tensors, out_tensors):
        """bit_tensors[i]: uint8 CUDA 1-D (one bit per byte); out_tensors[i]: uint8 CUDA buffers -> list of byte counts"""
        n = self.nstreams
        pin = (C.c_void_p * n)(*[t.data_ptr() for t in bit_tensors])
        cnt = (C.c_int * n)(*[int(t.numel()) for t in bit_tensors])
        pout = (C.c_void_p * n)(*[t.data_ptr() for t in out_tensors])
        nb = (C.c_int * n)()
        cap = min(int(t.numel()) for t in out_tensors)
        self.eng._check(self.lib.dvbs2gpu_dvbs_tail_process_batch(self.h, pin, cnt, pout, cap, nb, self.eng._stream()))
        return list(nb)

    def stats(self, stream=0):
        a = (C.c_int32 * 11)()
        self.eng._check(self.lib.dvbs2gpu_dvbs_tail_get_stats(self.h, stream, a))
        return {'frames': a[0], 'errors_nor': a[1], 'errors_inv': a[2], 'rs_errors': list(a[3:11])}

    def tap(self, which, stream=0):
        """stage taps of the last call: 0 deframed frames, 1 packets after Forney + RS, 2 RS status, 3 RS error counts (int32)"""
        import numpy as np
        n = self.eng._check(self.lib.dvbs2gpu_dvbs_tail_get_tap(self.h, stream, which, None, 0))
        a = np.zeros(n, np.uint8)
        if n:
            self.eng._check(self.lib.dvbs2gpu_dvbs_tail_get_tap(self.h, stream, which, C.c_void_p(a.ctypes.data), n))
        return a.view(np.int32) if which == 3 else a

    def rs_stage(self, packets, skip_rs=False):
        """packets: numpy uint8 [n, 204] (n a multiple of 8) -> TS bytes uint8 [n, 188] (stage entry, stream 0)"""
        import numpy as np
        packets = np.ascontiguousarray(packets, np.uint8)
        out = np.zeros(packets.shape[0] * 188, np.uint8)
        n = self.eng._check(self.lib.dvbs2gpu_dvbs_tail_rs_stage(self.h, C.c_void_p(packets.ctypes.data), packets.shape[0], int(bool(skip_rs)),
                                                                 C.c_void_p(out.ctypes.data), out.size))
        return out[:n].reshape(-1, 188)


class BbTsParserBank(_Handle):
    """BBFRAME -> MPEG-TS / GSE parser for `nstreams` DVB-S2 streams: dsp::dvbs2::BBFrameTSParser (bbframe_ts_parser.cpp:104-390),
    the consumer of DVBS2Demod's output in the reference's sink handler (main.cpp:532-558)."""
    _destroy = 'dvbs2gpu_bbts_destroy'
    HEADER_FIELDS = ('ts_gs', 'sis_mis', 'ccm_acm', 'issyi', 'npd', 'ro', 'isi', 'upl', 'dfl', 'sync', 'syncd')

    def __init__(self, engine, nstreams=1, kbch_bits=48408, max_frames=16):
        self.eng, self.lib, self.nstreams, self.kbch, self.max_frames = engine, engine.lib, nstreams, kbch_bits, max_frames
        h = C.c_void_p()
        engine._check(self.lib.dvbs2gpu_bbts_create(engine.h, nstreams, kbch_bits, max_frames, C.byref(h)))
        self.h = h

    def set_frame_size(self, kbch_bits):
        self.eng._check(self.lib.dvbs2gpu_bbts_set_frame_size(self.h, kbch_bits))
        self.kbch = kbch_bits

    def process_batch(self, bb_tensors, out_tensors):
        """bb_tensors[i]: uint8 CUDA, a whole number of kbch/8-byte BBFRAMEs; out_tensors[i]: uint8 CUDA buffers -> byte counts"""
        n, fb = self.nstreams, self.kbch // 8
        pin = (C.c_void_p * n)(*[t.data_ptr() for t in bb_tensors])
        cnt = (C.c_int * n)(*[int(t.numel()) // fb for t in bb_tensors])
        pout = (C.c_void_p * n)(*[t.data_ptr() for t in out_tensors])
        nb = (C.c_int * n)()
        cap = min(int(t.numel()) for t in out_tensors)
        self.eng._check(self.lib.dvbs2gpu_bbts_process_batch(self.h, pin, cnt, pout, cap, nb, self.eng._stream()))
        return list(nb)

    def work(self, bbframes, cap=None):
        """BBFrameTSParser::work on host buffers (bank of one stream): numpy uint8 in -> numpy uint8 out"""
        import numpy as np
        bb = np.ascontiguousarray(bbframes, np.uint8).reshape(-1)
        cnt = bb.size // (self.kbch // 8)
        cap = cap if cap is not None else bb.size + 376
        out = np.zeros(max(cap, 1), np.uint8)
        n = self.eng._check(self.lib.dvbs2gpu_bbts_work(self.h, C.c_void_p(bb.ctypes.data), cnt, C.c_void_p(out.ctypes.data), cap))
        return out[:n].copy()

    def stats(self, stream=0):
        a = (C.c_int32 * 17)()
        self.eng._check(self.lib.dvbs2gpu_bbts_get_stats(self.h, stream, a, 17))
        d = {k: a[i] for i, k in enumerate(self.HEADER_FIELDS)}
        d.update(last_gse_crc_err=a[11], last_bb_cnt=a[12], last_bb_proc=a[13], last_ts_errs=a[14], synched=a[15], count=a[16])
        return d


class SegmentReceiver(_Handle):
    """One fast transponder on the many-stream engine: a long chunk of one continuous IQ stream is cut into overlapping segments
    that run as independent streams and are stitched back in order (include/dvbs2gpu.h, segment receiver)."""
    _destroy = 'dvbs2gpu_segrx_destroy'

    def __init__(self, engine, cfg, nsegments, own_frames, warm_frames):
        self.eng, self.lib = engine, engine.lib
        h = C.c_void_p()
        engine._check(self.lib.dvbs2gpu_segrx_create(engine.h, C.byref(cfg), nsegments, own_frames, warm_frames, C.byref(h)))
        self.h = h
        self.chunk_samples = int(self.lib.dvbs2gpu_segrx_chunk_samples(self.h))

    def reset(self):
        self.eng._check(self.lib.dvbs2gpu_segrx_reset(self.h))

    def process(self, iq, out):
        """iq: complex64 CUDA 1-D (continues the stream), out: uint8 CUDA buffer -> bytes written"""
        return self.eng._check(self.lib.dvbs2gpu_segrx_process(self.h, C.c_void_p(iq.data_ptr()), int(iq.numel()), C.c_void_p(out.data_ptr()), int(out.numel())))

    def stats(self):
        a = (C.c_int32 * 3)()
        self.eng._check(self.lib.dvbs2gpu_segrx_get_stats(self.h, a))
        return {'sightings': a[0], 'returned': a[1], 'warmup_only': a[2]}


class DvbsSegmentReceiver(_Handle):
    """One fast DVB-S carrier: overlapping segments through the receiver bank, decoded bit streams joined in order
    (include/dvbs2gpu.h, DVB-S segment receiver)."""
    _destroy = 'dvbs2gpu_dvbs_segrx_destroy'

    def __init__(self, engine, nsegments, own_symbols, warm_symbols, **kw):
        self.eng, self.lib = engine, engine.lib
        self.cfg = DvbsCfg()
        self.lib.dvbs2gpu_dvbs_demod_default_cfg(C.byref(self.cfg))
        for k, v in kw.items():
            setattr(self.cfg, k, v)
        h = C.c_void_p()
        engine._check(self.lib.dvbs2gpu_dvbs_segrx_create(engine.h, C.byref(self.cfg), nsegments, own_symbols, warm_symbols, C.byref(h)))
        self.h = h
        self.chunk_samples = int(self.lib.dvbs2gpu_dvbs_segrx_chunk_samples(self.h))

    def reset(self):
        self.eng._check(self.lib.dvbs2gpu_dvbs_segrx_reset(self.h))

    def process(self, iq, out):
        """iq: complex64 CUDA 1-D (continues the stream), out: uint8 CUDA buffer -> number of bits written (one per byte)"""
        return self.eng._check(self.lib.dvbs2gpu_dvbs_segrx_process(self.h, C.c_void_p(iq.data_ptr()), int(iq.numel()), C.c_void_p(out.data_ptr()), int(out.numel())))

    def stats(self):
        a = (C.c_int32 * 4)()
        self.eng._check(self.lib.dvbs2gpu_dvbs_segrx_get_stats(self.h, a))
        return {'segments': a[0], 'matched': a[1], 'unmatched': a[2], 'bits': a[3]}
