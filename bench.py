#!/usr/bin/env python3
"""Headline benchmark: Msymbols/s of the DVB-S2 hot path on MI355X.

Workload (BASELINE.json configs[2], the one the metric is quoted on): DVB-S2 8PSK 3/4 normal FECFRAME
(MODCOD 14, PLFRAME 21 690 symbols, LDPC table B7, BCH t=12), LDPC forced to exactly 50 layered
iterations per frame (no early exit), synthetic frames resident in HBM when the timed region starts.
One step = one pass of the hot path over one batch of `--frames` PLFRAMEs per GPU.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel:
the LDPC decoder, HIP-event timed in here) and `cpu_baseline` (reference FEC compiled from
/root/reference into oracle/_ref when that build travelled, else the oracle port; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

MODCOD = 14          # 8PSK 3/4
RATE, SHORT = 6, 0
ITERS = 50
HBM_PEAK_GBS = 8000.0


def make_batch(nframes, seed):
    """Synthetic receive-side input for the stages that exist: int8 LLRs of valid codewords in AWGN
    (Es/N0 about threshold + 1 dB), built by the repo's own transmitter (oracle encoder, test/bench
    infrastructure).  A few distinct frames are tiled to the batch size."""
    import orc
    rng = np.random.default_rng(seed)
    p = orc.fec_params(RATE, SHORT)
    distinct = 16
    llr = np.zeros((distinct, p['N']), np.int8)
    bbs = np.zeros((distinct, p['kbch'] // 8), np.uint8)
    for f in range(distinct):
        bb, bits = orc.encode_frame(RATE, SHORT, 0xD5B2 + seed * 1000 + f)
        llr[f] = orc.bits_to_llr(bits, 5.7, rng)
        bbs[f] = bb
    reps = (nframes + distinct - 1) // distinct
    return p, np.tile(llr, (reps, 1))[:nframes], np.tile(bbs, (reps, 1))[:nframes]


def cpu_baseline(budget_s=12.0):
    """Reference (or oracle-port) FEC timed on this box's host cores: LDPC 50 iterations on noise-only
    LLRs (never converges => exactly 50 sweeps, same work as the forced GPU run) + BCH + descramble."""
    import orc
    p = orc.fec_params(RATE, SHORT)
    sym_per_frame = 21690
    ncores = os.cpu_count() or 1
    R = orc.ref()
    rng = np.random.default_rng(1)
    results = {}
    if R is not None:
        kind = 'reference'
        noise = rng.integers(-40, 41, size=(16, p['N'])).astype(np.int8)
        counts = [0] * ncores
        stop = time.time() + budget_s

        def worker(i):
            buf = noise.copy()
            fr = np.zeros(p['K'] // 8, np.uint8)
            while time.time() < stop:
                b = buf.copy()
                R.ref_ldpc_decode_simd16(RATE, SHORT, b, ITERS, 1)       # 16 frames, one per SSE4.1 int8 lane
                for f in range(16):
                    orc.lib().orc_hard_pack(b[f], p['K'], fr)
                    R.ref_bch_decode(RATE, SHORT, fr)
                    R.ref_bb_descramble(RATE, SHORT, fr)
                counts[i] += 16
        t0 = time.time()
        th = [threading.Thread(target=worker, args=(i,)) for i in range(ncores)]
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.time() - t0
        frames = sum(counts)
        value = frames * sym_per_frame / dt / 1e6
        # as-wired variant: one frame per decode call, single thread (bbframe_ldpc.cpp:123-139)
        b = noise[0].copy()
        t1 = time.time()
        n1 = 0
        while time.time() - t1 < 2.0:
            x = b.copy()
            R.ref_ldpc_decode(RATE, SHORT, x, ITERS)
            n1 += 1
        aswired = n1 * sym_per_frame / (time.time() - t1) / 1e6
        sample = ('%d frames: reference LDPC (16 frames/call in the 16 int8 SSE4.1 lanes, 50 iterations, noise LLRs) '
                  '+ reference BCH + descrambler, %d threads, %.1f s' % (frames, ncores, dt))
        results = dict(value=round(value, 4), unit='Msymbols/s', cores=ncores, kind=kind, sample=sample,
                       as_wired_1thread_msym_s=round(aswired, 4))
    else:
        kind = 'port'
        noise = rng.integers(-40, 41, size=(p['N'],)).astype(np.int8)
        counts = [0] * ncores
        stop = time.time() + budget_s

        def worker(i):
            bb = np.zeros(p['kbch'] // 8, np.uint8)
            c = np.zeros(1, np.int32)
            while time.time() < stop:
                x = noise.copy()
                orc.lib().orc_fec_decode_frame(RATE, SHORT, x, ITERS, 1, bb, c)
                counts[i] += 1
        t0 = time.time()
        th = [threading.Thread(target=worker, args=(i,)) for i in range(ncores)]
        [t.start() for t in th]
        [t.join() for t in th]
        dt = time.time() - t0
        frames = sum(counts)
        value = frames * sym_per_frame / dt / 1e6
        sample = '%d frames: oracle scalar port LDPC 50 iterations + BCH + descrambler, %d threads, %.1f s' % (frames, ncores, dt)
        results = dict(value=round(value, 4), unit='Msymbols/s', cores=ncores, kind=kind, sample=sample)
    return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--frames', type=int, default=4096, help='PLFRAMEs per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    import __graft_entry__ as g
    pkg = g.load_package()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        torch.cuda.set_device(0)
        local_rank = 0
    dev = torch.device('cuda', local_rank)
    eng = pkg.Engine(local_rank)

    info = pkg.modcod_info(MODCOD, False, False)
    p, llr_h, bbs_h = make_batch(args.frames, seed=rank)
    F = args.frames
    llr = torch.from_numpy(llr_h).to(dev)
    out = torch.empty((F, p['kbch'] // 8), dtype=torch.uint8, device=dev)
    trials = torch.empty((F,), dtype=torch.int32, device=dev)
    corr = torch.empty((F,), dtype=torch.int32, device=dev)

    def step():
        eng.fec_decode(llr, RATE, False, max_trials=ITERS, force=True, out=out, trials=trials, corr=corr)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # parity of the timed output: bit-exact BBFRAMEs (what the transmitter sent)
    ok = bool(np.array_equal(out.cpu().numpy(), bbs_h))

    # dominant kernel (LDPC) timed alone with HIP events on the launch stream
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    nk = max(3, min(args.steps, 10))
    hard = torch.empty((F, p['K'] // 8), dtype=torch.uint8, device=dev)
    eng.lib.dvbs2gpu_ldpc_decode_batch(eng.h, RATE, 0, llr.data_ptr(), F, ITERS, 1, hard.data_ptr(), None, trials.data_ptr(), eng._stream())
    torch.cuda.synchronize()
    e0.record()
    for _ in range(nk):
        eng.lib.dvbs2gpu_ldpc_decode_batch(eng.h, RATE, 0, llr.data_ptr(), F, ITERS, 1, hard.data_ptr(), None, trials.data_ptr(), eng._stream())
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / nk
    bytes_per_frame = ITERS * 4 * info['ldpc_edges'] + info['ldpc_n'] + info['kbch'] // 8
    achieved = bytes_per_frame * F / (k_ms * 1e-3) / 1e9

    if rank == 0:
        sym = info['plframe_symbols']
        value = world * F * args.steps * sym / dt / 1e6
        line = {
            'metric': 'Msymbols/s demod+FEC, DVB-S2 8PSK 3/4 normal-frame @50 LDPC iters',
            'value': round(value, 3), 'unit': 'Msymbols/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'int8', 'data': 'synthetic',
            'config': {'workload': 'DVB-S2 8PSK 3/4 normal FECFRAME (MODCOD 14), pilots off, 27.5 Msym/s-class stream, '
                                   '50 forced LDPC iterations',
                       'stages': 'LLR(int8) -> LDPC -> BCH -> BB-descramble -> BBFRAME (front-end stages not yet in the timed path)',
                       'frames_per_gpu_per_step': F, 'symbols_per_frame': sym, 'parallelism': 'frames sharded over GPUs, no collective',
                       'output_bit_exact': ok},
            'roofline': {'bound': 'hbm', 'kernel': 'ldpc_decode_kernel<12,4>', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': None,
                         'kernel_ms': round(k_ms, 4), 'algorithmic_bytes_per_frame': bytes_per_frame},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == '__main__':
    main()
