import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, bench as B, __graft_entry__ as g
pkg = g.load_package(); eng = pkg.Engine(0); eng.set_stage_timing(True)
dev = torch.device('cuda', 0)
for S, F in ((1, 4), (1, 8), (64, 1), (64, 8), (512, 1)):
    print(json.dumps(B.small_batch(eng, pkg, dev, S, F)))
