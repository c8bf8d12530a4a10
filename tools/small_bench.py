"""Development aid: the two small-batch lines of bench.py alone (64 transponders x 1 PLFRAME, 1 transponder x 4 PLFRAMEs; synchronous calls)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import __graft_entry__ as g
import bench
pkg = g.load_package()
eng = pkg.Engine(0)
for S, F in (() if len(sys.argv) > 1 and sys.argv[1] == 'vcm' else ((64, 1), (1, 4), (16, 4))):
    r = bench.small_batch(eng, pkg, torch.device('cuda:0'), S, F)
    print(S, F, 'ms_per_call', r['ms_per_call'], 'per stream', r['msym_s_per_stream'], 'equal', r['frames_equal_to_transmitted'], '/', r['frames_delivered'], r['stage_ms_per_call'])
if len(sys.argv) > 1 and sys.argv[1] == 'vcm':
    r = bench.secondary_vcm(eng, pkg, torch.device("cuda:0"))
    print('ACM/VCM', {k: r[k] for k in r if k in ('value', 'ms_per_call', 'msym_s_per_stream', 'frames_checked_last_call', 'frames_equal_to_transmitted')})
