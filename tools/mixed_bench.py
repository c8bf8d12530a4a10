#!/usr/bin/env python3
"""BASELINE config 4 shape on one GPU: independent transponders with mixed QPSK / 8PSK MODCODs (cycled over 4, 6, 7, 11, 12, 13, 14, 15,
normal frames, 50 forced LDPC iterations) in ONE pipelined batch: eight configuration groups, one FEC job per group and call.
Prints Msymbols/s over all streams and checks every delivered frame of the last step against the transmitted ones."""
import os, sys, time, json
os.environ.setdefault('GPU_MAX_HW_QUEUES', '12')   # HIP runs at most this many streams concurrently (default 4): one per configuration group + FEC + front end
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench as B
import __graft_entry__ as g
pkg = g.load_package()
eng = pkg.Engine(0)
MODCODS = [4, 6, 7, 11, 12, 13, 14, 15]
ESN0 = {4: 8.0, 6: 10.0, 7: 11.0, 11: 14.0, 12: 12.0, 13: 13.0, 14: 14.0, 15: 16.0}
S = int(os.environ.get('STREAMS', '4096'))
steps = int(os.environ.get('STEPS', '6'))
per = S // len(MODCODS)
demods, tin, tout, sents, kbs, syms = [], [], [], [], [], []
for m in MODCODS:
    B.MODCOD, B.ESN0_DB, B.PREROLL = m, ESN0[m], 24
    info = pkg.modcod_info(m, False, False)
    blocks, sent = B.make_blocks(1, seed=m, eng=eng, pkg=pkg)
    d_blocks = [torch.from_numpy(b).cuda() for b in blocks]
    cfg = eng.default_cfg(m, False, False, force_ldpc_iters=B.ITERS)
    for s in range(per):
        demods.append(eng.demod(cfg, max_samples=blocks[0].size))
        tin.append(d_blocks[s % B.DISTINCT])
        sents.append(sent[s % B.DISTINCT])
        kbs.append(info['kbch'] // 8)
        syms.append(info['plframe_symbols'])
cap = 3 * max(kbs)
tout = [torch.zeros(cap, dtype=torch.uint8, device='cuda') for _ in demods]
for pipelined in (False, True):
    eng.set_pipelined(pipelined)
    for _ in range(26 if not pipelined else 3):
        eng.process_batch(demods, tin, tout)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        nb = eng.process_batch(demods, tin, tout)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    bad = 0
    for i in range(len(demods)):
        got = tout[i][:nb[i]].cpu().numpy().reshape(-1, kbs[i])
        if nb[i] != kbs[i] or not all(bytes(x) in sents[i] for x in got):
            bad += 1
    if pipelined:
        eng.process_batch(demods, [torch.empty(0, dtype=torch.complex64, device='cuda') for _ in demods], tout)
    print(json.dumps({'workload': 'mixed MODCODs %s, %d streams (%d per MODCOD), 1 PLFRAME per stream per step, 50 forced LDPC iterations' % (MODCODS, len(demods), per),
                      'pipelined': pipelined, 'ms_per_step': round(dt * 1e3, 2), 'Msymbols_per_s': round(sum(syms) / dt / 1e6, 1), 'streams_not_bit_exact': bad}))
eng.set_pipelined(False)
