#!/usr/bin/env python3
"""Development aid (GPU box): randomized soak of the throughput mode.  Streams of several CCM MODCODs and ACM/VCM streams; every call takes a random subset
in random order (so batches change between single-configuration, mixed and ACM/VCM shapes, streams come and go); the same schedule runs synchronously and
pipelined, and what the pipelined run delivers must be what the synchronous run produced one call earlier for every stream present in both calls.
usage: python tools/stress_pipelined.py [calls=120] [seed=1] [copies=1]   (copies > 25 reaches the big-batch flow: front-end prepass + one host thread per group)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import orc
import __graft_entry__ as g

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
copies = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pkg = g.load_package()
eng = pkg.Engine(0)
rng = np.random.default_rng(seed)
ccm = [(4, 1, 0), (14, 1, 0), (11, 0, 0), (6, 1, 1), (14, 1, 0), (13, 0, 0), (27, 1, 1), (4, 1, 0)]
vcm_lists = [[(4 << 2) | 2, (14 << 2) | 2, 0, (6 << 2) | 2 | 1], [(13 << 2) | 2, (12 << 2) | 2]]
sigs, cfgs, kinds = [], [], []
for s, (m, sh, p) in enumerate(ccm):
    iq, _, _ = orc.transmit(m, sh, p, nframes=(12 if sh else 5) * 3, seed=5000 + s, esn0_db=22.0, cfo=1e-4 * s, timing=0.03 * s, phase0=0.1)
    sigs.append(iq); cfgs.append(dict(modcod=m, shortframes=bool(sh), pilots=bool(p))); kinds.append('ccm')
for k, pl in enumerate(vcm_lists):
    iq, _ = orc.transmit_vcm(pl, 60, seed=6000 + k, esn0_db=28.0, cfo=1e-4, timing=0.1 * k, lead_symbols=300)
    sigs.append(iq); cfgs.append(dict(modcod=4, shortframes=True, pilots=False, acm_vcm=1)); kinds.append('vcm')
sigs, cfgs, kinds = sigs * copies, cfgs * copies, kinds * copies
S = len(sigs)
dev_sig = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in sigs[:len(sigs) // copies]]
chunk = 40000
schedule = []
for c in range(calls):
    n = int(rng.integers(1, S + 1))
    schedule.append(list(rng.permutation(S)[:n]))
schedule.append(list(range(S)))          # everybody once more ...
cap = chunk // 2 + 60000


def run(pipelined):
    dms = [eng.demod(eng.default_cfg(c['modcod'], c['shortframes'], c['pilots'], **({'acm_vcm': 1} if c.get('acm_vcm') else {})), max_samples=chunk) for c in cfgs]
    tout = [torch.zeros(cap, dtype=torch.uint8, device='cuda') for _ in range(S)]
    pos = [0] * S
    eng.set_pipelined(pipelined)
    outs = []
    try:
        for ci, who in enumerate(schedule + ([list(range(S))] if pipelined else [])):
            flush = ci > len(schedule) - 1
            tin = []
            for s in who:
                if flush:
                    tin.append(torch.empty(0, dtype=torch.complex64, device='cuda'))
                else:
                    a = (pos[s] + 2 * (s // len(dev_sig)) * 977) % (sigs[s].size - chunk)
                    a -= a & 1
                    n = int(rng_len[ci][s])
                    tin.append(dev_sig[s % len(dev_sig)][a:a + n].clone())
                    pos[s] += n
            if os.environ.get('STRESS_INPUT_SYNC') == '1': torch.cuda.current_stream().synchronize()       # (the inputs -- torch's clones on the null stream -- complete before the call)
            nb = eng.process_batch([dms[s] for s in who], tin, [tout[s] for s in who])
            if os.environ.get('STRESS_SYNC') == '1': torch.cuda.synchronize()       # (debugging: no FEC job of call k beside the front end of call k + 1)
            outs.append({s: (tout[s][:nb[k]].cpu().numpy().copy(), [(x.ldpc_trials, x.bch_corrections, x.detected_modcod, x.bbframe_bytes) for x in dms[s].stats()]) for k, s in enumerate(who)})
    finally:
        eng.set_pipelined(False)
        for d in dms:
            d.close()
    return outs


# per (call, stream) sample counts, fixed for both runs: mostly whole chunks, sometimes short or empty calls
rng_len = [[int(rng.choice([chunk, chunk, chunk, 20000, 2000, 0])) & ~1 for _ in range(S)] for _ in range(len(schedule) + 1)]
if os.environ.get('STRESS_PIPE_FIRST') == '1':      # (the pipelined run on the engine's FIRST streams: which hardware queues streams share depends on their creation order)
    pipe = run(True)
    sync = run(False)
else:
    sync = run(False)
    pipe = run(True)
full = schedule + [list(range(S))]
bytes_ok = frames = dropped = 0
for c in range(1, len(full)):
    for s in full[c]:
        if s in full[c - 1]:
            assert np.array_equal(pipe[c][s][0], sync[c - 1][s][0]), ('bytes', c, s, kinds[s])
            assert pipe[c][s][1] == sync[c - 1][s][1], ('stats', c, s, kinds[s], cfgs[s], 'pipelined', pipe[c][s][1], 'synchronous', sync[c - 1][s][1])
            bytes_ok += sync[c - 1][s][0].size; frames += len(sync[c - 1][s][1])
        else:
            assert pipe[c][s][0].size == 0 and pipe[c][s][1] == [], ('joined', c, s)
for c in range(len(full) - 1):
    dropped += sum(len(sync[c][s][1]) for s in full[c] if s not in full[c + 1])
print('calls', len(full), 'streams', S, 'frames compared', frames, 'bytes', bytes_ok, 'frames dropped by leaving streams', dropped, 'OK')
