#!/usr/bin/env python3
"""Development aid, on the GPU box: bench.py's drop-in call lines alone (dvbs2gpu_demod_process, host buffers, ONE stream) -- for options in DVBS2GPU_OPTIONS.
usage: DVBS2GPU_OPTIONS=fe_slices=4 python tools/dropin_bench.py [samples per call ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench                      # (sets GPU_MAX_HW_QUEUES before torch initialises HIP)
import torch
import __graft_entry__ as g
pkg = g.load_package()
dev = torch.device('cuda', 0)
eng = pkg.Engine(0)
for ns in ([int(a) for a in sys.argv[1:]] or [8192, 65536]):
    r = bench.dropin_calls(eng, pkg, dev, ns)
    print(ns, {k: r[k] for k in ('ms_per_call', 'ms_per_call_median', 'msym_s', 'kernel_launches_per_call', 'frames_delivered', 'frames_equal_to_transmitted')})
eng.close()
