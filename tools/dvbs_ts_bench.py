"""Development aid: config D of bench.py alone (DVB-S QPSK 1/2, IQ -> TS packets: receiver bank + tail bank; 4096 carriers and one), with its checks."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import __graft_entry__ as g
import bench
pkg = g.load_package()
eng = pkg.Engine(0)
r = bench.secondary_dvbs(eng, pkg, torch.device('cuda:0'))
for k, v in r.items():
    print(k, v)
