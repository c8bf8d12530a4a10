"""Development aid: where a syndrome check of the half-row decoder spends its time (-DLDPC_PROF=5 build, normal mode on input that does not converge)."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
eng = pkg.Engine(0)
fi = pkg.fec_info(6, 0)
F = 2048
llr = torch.randint(-30, 31, (F, fi['ldpc_n']), dtype=torch.int8, device='cuda')
buf = torch.zeros(1024, dtype=torch.int64, device='cuda')
eng.lib.dvbs2gpu_debug_set_prof.argtypes = [C.c_void_p]
eng.lib.dvbs2gpu_debug_set_prof(C.c_void_p(buf.data_ptr()))
eng.ldpc_decode(llr, 6, False, max_trials=20, force=False)
torch.cuda.synchronize()
v = buf.cpu().numpy().astype(np.float64)
n = max(v[404], 1)
print('checks %d: sign pack %.0f | barrier + store drain %.0f | syndromes %.0f | flags + two barriers %.0f cycles each' % (n, v[400] / n, v[401] / n, v[402] / n, v[403] / n))
