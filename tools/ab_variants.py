import sys, os, time, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, __graft_entry__ as g
pkg = g.load_package()
keep = ['dvbs2gpu_version','dvbs2gpu_last_error','dvbs2gpu_create','dvbs2gpu_destroy','dvbs2gpu_fec_info_get','dvbs2gpu_ldpc_plan_info','dvbs2gpu_ldpc_decode_batch']
pkg.PROTOTYPES = {k: v for k, v in pkg.PROTOTYPES.items() if k in keep}
libs = sorted(glob.glob(os.path.join(ROOT, 'sdrpp-dvbs-demodulator_amd', 'libv_*.so')))
codes = [(6, 0), (10, 0), (9, 1), (3, 0), (4, 0)]
res = {}
for rep in range(2):
    for lib in libs:
        pkg._lib = None; pkg.LIB_PATH = lib
        eng = pkg.Engine(0)
        for rate, short in codes:
            fi = pkg.fec_info(rate, short); pi = eng.ldpc_plan_info(rate, short)
            F = pi['cus'] * pi['blocks_per_cu'] * 2
            llr = torch.randint(-30, 31, (F, fi['ldpc_n']), dtype=torch.int8, device='cuda')
            eng.ldpc_decode(llr, rate, bool(short), max_trials=2, force=True); torch.cuda.synchronize()
            t0 = time.perf_counter(); eng.ldpc_decode(llr, rate, bool(short), max_trials=30, force=True); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res.setdefault((os.path.basename(lib), rate, short), []).append(dt / 2 / 30 * 1e6)
        eng.close()
for k in sorted(res):
    print(k, ' '.join('%.1f' % x for x in res[k]))
