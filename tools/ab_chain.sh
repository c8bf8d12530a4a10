python - <<'PY'
import sys, os, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, __graft_entry__ as g
pkg=g.load_package()
for lib in ['libdvbs2gpu.so','libdvbs2gpu_nochain.so']:
    pkg._lib=None; pkg.LIB_PATH=os.path.join('sdrpp-dvbs-demodulator_amd',lib)
    eng=pkg.Engine(0)
    llr=torch.randint(-30,31,(2048,64800),dtype=torch.int8,device='cuda')
    eng.ldpc_decode(llr,6,False,max_trials=2,force=True); torch.cuda.synchronize()
    for rep in range(2):
        t0=time.perf_counter(); eng.ldpc_decode(llr,6,False,max_trials=50,force=True); torch.cuda.synchronize(); dt=time.perf_counter()-t0
        print(lib,'ms %.2f'%(dt*1e3),'us/iter/block %.1f'%(dt/4/50*1e6))
    eng.close()
PY
