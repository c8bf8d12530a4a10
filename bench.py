#!/usr/bin/env python3
"""Headline benchmark: Msymbols/s of the DVB-S2 demod + FEC hot path on MI355X.

Workload (BASELINE.json configs[2], the one the metric is quoted on): DVB-S2 8PSK 3/4 normal FECFRAME
(MODCOD 14, PLFRAME 21 690 symbols, LDPC table B7, BCH t=12), pilots off, LDPC forced to exactly 50 layered
iterations per frame (no early exit).  `--streams` independent transponder streams per GPU, each a
continuous synthetic 27.5 Msym/s-class signal (2 samples/symbol complex64 IQ, RRC 0.35, timing offset
0.3 sample, AWGN Es/N0 14 dB) of `--frames` PLFRAMEs per step, resident in HBM when the timed region
starts.  One step = one dvbs2gpu_demod_process_batch call = the whole hot path
  IQ -> AGC -> NCO -> Gardner -> RRC -> /2 -> PL sync -> FED/PLL/PLHDR -> soft demap + de-interleave
     -> LDPC (50 it) -> BCH -> BB descramble -> BBFRAMEs
over every stream; the streams keep their loop state from step to step (the IQ block is periodic, so the
signal is seamless).  value = PLFRAME symbols consumed per second, all GPUs.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel = the LDPC decoder, HIP-event timed in
here on the stream it is launched on) and `cpu_baseline` (rank 0, N=1 only: the reference's own FEC code
compiled into oracle/_ref when that build travelled, else the oracle port, plus the oracle front end).
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
WORKLOAD = ('DVB-S2 8PSK 3/4 normal FECFRAME (MODCOD 14), pilots off, 27.5 Msym/s-class streams at 2 sps, '
            'Es/N0 14 dB, 50 forced LDPC iterations, IQ in -> BBFRAMEs out')
PILOTS = 0           # (tools/config_sweep.py re-runs main() with other MODCOD / RATE / SHORT / PILOTS / ESN0_DB values)
ITERS = 50
ESN0_DB = 14.0
HBM_PEAK_GBS = 8000.0
DISTINCT = 4         # distinct periodic IQ blocks; stream s replays block s % DISTINCT
PREROLL = 10         # untimed frames per stream before the warm-up: loop acquisition (AGC, Gardner, PL sync, PLL)


def make_blocks(frames, seed, eng=None, pkg=None):
    """periodic IQ blocks built by the repo's own transmitter (oracle/s2chain.cpp, bench/test infrastructure).
    The reference's decision-directed PLL can sit in a pi/4 false lock for many frames on 8PSK (the oracle and
    the engine agree on that frame by frame), so candidate blocks are pre-rolled through the engine and only blocks on
    which the demodulator has converged (delivers the transmitted BBFRAMEs) are used for the timed streams."""
    import orc
    cands = []
    ncand = 4 * DISTINCT
    for b in range(ncand):
        iq, bb, _ = orc.transmit(MODCOD, SHORT, PILOTS, nframes=frames, seed=0xD5B2 + 64 * seed + b, esn0_db=ESN0_DB, cfo=0.0, timing=0.3,
                                 phase0=0.1, lead_symbols=0, circular=1)
        cands.append((iq, {bytes(x) for x in bb}))
    if eng is None:
        return [c[0] for c in cands[:DISTINCT]], [c[1] for c in cands[:DISTINCT]]
    import torch
    info = pkg.modcod_info(MODCOD, bool(SHORT), bool(PILOTS))
    kb = info['kbch'] // 8
    cfg = eng.default_cfg(MODCOD, bool(SHORT), bool(PILOTS), force_ldpc_iters=0)
    demods = [eng.demod(cfg, max_samples=cands[0][0].size) for _ in range(ncand)]
    tin = [torch.from_numpy(c[0]).cuda() for c in cands]
    tout = [torch.zeros((frames + 2) * kb, dtype=torch.uint8, device='cuda') for _ in range(ncand)]
    good = [0] * ncand
    for step in range(PREROLL):
        nb = eng.process_batch(demods, tin, tout)
        for i in range(ncand):
            got = tout[i][:nb[i]].cpu().numpy().reshape(-1, kb)
            ok = nb[i] == frames * kb and all(bytes(x) in cands[i][1] for x in got)
            good[i] = good[i] + 1 if ok else 0
    for d in demods:
        d.close()
    keep = [i for i in range(ncand) if good[i] >= 3][:DISTINCT]
    if not keep:
        raise RuntimeError('none of %d candidate blocks converged in %d frames' % (ncand, PREROLL))
    while len(keep) < DISTINCT:        # (never seen: 5..10 of 12 converge for the seeds of ranks 0..7) reuse what converged
        keep.append(keep[len(keep) % len(set(keep))])
    return [cands[i][0] for i in keep], [cands[i][1] for i in keep]


def single_transponder(eng, pkg, nseg=512, own=8, warm=8, calls=3):
    """side measurement (not `value`): ONE continuous 8PSK 3/4 transponder through the segment receiver (dvbs2gpu_segrx_*): its IQ is
    cut into overlapping segments that run as independent streams and are stitched back in order -- the answer to "a stream's loops are
    serial" for a single 27.5 Msym/s carrier.  LDPC with early exit (this is a receive path, not the forced-iteration stress)."""
    import torch
    import orc
    period = 16
    info = pkg.modcod_info(MODCOD, bool(SHORT), bool(PILOTS))
    kb, sym = info['kbch'] // 8, info['plframe_symbols']
    iq, bb, _ = orc.transmit(MODCOD, SHORT, PILOTS, nframes=period, seed=5, esn0_db=16.0, cfo=1e-4, timing=0.3, phase0=0.2, lead_symbols=0, circular=1)
    index = {bytes(b): k for k, b in enumerate(bb)}
    rx = pkg.SegmentReceiver(eng, eng.default_cfg(MODCOD, bool(SHORT), bool(PILOTS)), nseg, own, warm)
    chunk = torch.from_numpy(iq).cuda().repeat(rx.chunk_samples // iq.size)
    out = torch.zeros((nseg * own + warm + 8) * kb, dtype=torch.uint8, device='cuda')
    seq, times = [], []
    for _ in range(calls):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nbytes = rx.process(chunk, out)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        seq += [index.get(bytes(x), -1) for x in out[:nbytes].cpu().numpy().reshape(-1, kb)]
    rx.close()
    good = [k for k in seq if k >= 0]
    steps = [(b - a) % period for a, b in zip(good, good[1:])]
    dt = min(times[1:])
    return {'value': round(nseg * own * sym / dt / 1e6, 1), 'unit': 'Msymbols/s of ONE continuous stream', 'ms_per_call': round(dt * 1e3, 1),
            'segments': nseg, 'own_frames': own, 'warmup_frames': warm, 'chunk_symbols': nseg * own * sym,
            'frames_returned': len(seq), 'frames_not_transmitted_ones': len(seq) - len(good), 'frames_out_of_sequence': sum(1 for s in steps if s != 1),
            'realtime_factor_at_27.5_Msym_s': round(nseg * own * sym / dt / 27.5e6, 2)}


def cpu_baseline(frames_block, budget_s=10.0):
    """CPU path on this box's host cores.  FEC: the reference's own code (oracle/_ref: LDPC with 16 frames in the
    16 int8 SSE4.1 lanes of one call, 50 iterations on noise LLRs = never converges = the sweeps of the forced GPU
    run plus the reference's syndrome check before every iteration, which the forced GPU mode evaluates once -- see
    roofline.kernel_ms_normal_mode_same_iterations for the GPU doing the same --, + BCH + descrambler) when it travelled, else the oracle's scalar port.  Front end: the oracle restatement
    (the reference's float blocks need SDR++/VOLK, not buildable here).  Both legs run on all cores; the
    per-symbol times add."""
    import orc
    p = orc.fec_params(RATE, SHORT)
    sym_per_frame = orc.modcod_params(MODCOD, SHORT, PILOTS)['plframe']
    ncores = os.cpu_count() or 1
    R = orc.ref()
    rng = np.random.default_rng(1)

    def run_threads(fn):
        counts = [0] * ncores
        stop = time.time() + budget_s / 2

        def worker(i):
            while time.time() < stop:
                counts[i] += fn(i)
        t0 = time.time()
        th = [threading.Thread(target=worker, args=(i,)) for i in range(ncores)]
        [t.start() for t in th]
        [t.join() for t in th]
        return sum(counts), time.time() - t0

    # ---- FEC leg
    if R is not None:
        kind = 'reference'
        noise = rng.integers(-40, 41, size=(16, p['N'])).astype(np.int8)

        def fec(i):
            b = noise.copy()
            fr = np.zeros(p['K'] // 8, np.uint8)
            R.ref_ldpc_decode_simd16(RATE, SHORT, b, ITERS, 1)
            for f in range(16):
                orc.lib().orc_hard_pack(b[f], p['K'], fr)
                R.ref_bch_decode(RATE, SHORT, fr)
                R.ref_bb_descramble(RATE, SHORT, fr)
            return 16
    else:
        kind = 'port'
        noise = rng.integers(-40, 41, size=(p['N'],)).astype(np.int8)

        def fec(i):
            bb = np.zeros(p['kbch'] // 8, np.uint8)
            c = np.zeros(1, np.int32)
            x = noise.copy()
            orc.lib().orc_fec_decode_frame(RATE, SHORT, x, ITERS, 1, bb, c)
            return 1
    nfec, tfec = run_threads(fec)
    fec_frames_per_s = nfec / tfec
    # ---- front-end leg (oracle; FEC skipped with force_ldpc_iters = -1)
    blk = frames_block[0]
    rxs = [orc.OracleRx(orc.default_cfg(MODCOD, SHORT, PILOTS, force_ldpc_iters=-1)) for _ in range(ncores)]

    def fe(i):
        rxs[i].process(blk)
        return blk.size // 2
    nsym, tfe = run_threads(fe)
    fe_sym_per_s = nsym / tfe
    t_per_sym = 1.0 / fe_sym_per_s + 1.0 / (fec_frames_per_s * sym_per_frame)
    out = dict(value=round(1e-6 / t_per_sym, 4), unit='Msymbols/s', cores=ncores, kind=kind,
               sample='%d threads: FEC leg %d frames in %.1f s (%s LDPC 50 it + BCH + descramble = %.3f Msym/s), front-end leg '
                      '%d symbols in %.1f s (oracle AGC..demap = %.3f Msym/s); per-symbol times added'
                      % (ncores, nfec, tfec, 'reference 16-lane SSE4.1' if kind == 'reference' else 'oracle scalar',
                         fec_frames_per_s * sym_per_frame / 1e6, nsym, tfe, fe_sym_per_s / 1e6),
               fec_only_msym_s=round(fec_frames_per_s * sym_per_frame / 1e6, 4), frontend_only_msym_s=round(fe_sym_per_s / 1e6, 4))
    if R is not None:
        # as-wired variant: one frame per decode call, single thread (bbframe_ldpc.cpp:123-139)
        b = noise[0].copy()
        t1 = time.time()
        n1 = 0
        while time.time() - t1 < 1.5:
            x = b.copy()
            R.ref_ldpc_decode(RATE, SHORT, x, ITERS)
            n1 += 1
        out['fec_as_wired_1thread_msym_s'] = round(n1 * sym_per_frame / (time.time() - t1) / 1e6, 4)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--streams', type=int, default=4096, help='transponder streams per GPU')
    ap.add_argument('--frames', type=int, default=1, help='PLFRAMEs per stream per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-aux', action='store_true', help='skip the single-transponder (segment receiver) side measurement')
    ap.add_argument('--no-pipeline', action='store_true', help='run the FEC inside the call that produced the frames (no overlap with the next front end)')
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
        # DVBS2GPU_BENCH_BACKEND=gloo: development aid to exercise the multi-rank path on a box with fewer GPUs than ranks
        backend = os.environ.get('DVBS2GPU_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            local_rank = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend)
    else:
        local_rank = 0
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local_rank)
    eng = pkg.Engine(local_rank)

    info = pkg.modcod_info(MODCOD, bool(SHORT), bool(PILOTS))
    sym = info['plframe_symbols']
    kb = info['kbch'] // 8
    S, F = args.streams, args.frames
    blocks, sent = make_blocks(F, seed=rank, eng=eng, pkg=pkg)
    d_blocks = [torch.from_numpy(b).to(dev) for b in blocks]
    nsamp = blocks[0].size
    cfg = eng.default_cfg(MODCOD, bool(SHORT), bool(PILOTS), force_ldpc_iters=ITERS)
    demods = [eng.demod(cfg, max_samples=nsamp) for _ in range(S)]
    tin = [d_blocks[s % DISTINCT] for s in range(S)]
    tout = [torch.zeros((F + 2) * kb, dtype=torch.uint8, device=dev) for _ in range(S)]

    # throughput mode: FEC of step k overlaps the front end of step k+1 (two HIP streams); BBFRAMEs arrive one step later
    pipelined = not args.no_pipeline
    eng.set_pipelined(pipelined)

    def step():
        return eng.process_batch(demods, tin, tout)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(PREROLL):          # acquisition, untimed and not counted as warm-up
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    nb = None
    for _ in range(args.steps):
        nb = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    def check_out(nbytes):
        ok = all(n == F * kb for n in nbytes)
        ex = ok
        if ok:
            for s in list(range(0, S, max(1, S // 16))) + list(range(min(S, DISTINCT))):
                got = tout[s][:nbytes[s]].cpu().numpy().reshape(-1, kb)
                ex = ex and all(bytes(x) in sent[s % DISTINCT] for x in got)
        return ok, ex

    # parity of the timed output: every stream delivered F frames per step, each bit-exact one of the BBFRAMEs sent
    frames_ok, exact = check_out(nb)
    if pipelined:
        # collect the frames of the last timed step (zero-sample call), same check; then leave the mode
        empty = [torch.empty(0, dtype=torch.complex64, device=dev) for _ in range(S)]
        nb_last = eng.process_batch(demods, empty, tout)
        ok2, ex2 = check_out(nb_last)
        frames_ok, exact = frames_ok and ok2, exact and ex2
        eng.set_pipelined(False)

    # dominant kernel (LDPC) alone, HIP events on its launch stream, same batch size as inside a step
    nfr = S * F
    llr = torch.randint(-40, 41, (nfr, info['ldpc_n']), dtype=torch.int8, device=dev)
    hard = torch.empty((nfr, info['ldpc_k'] // 8), dtype=torch.uint8, device=dev)
    tri = torch.empty((nfr,), dtype=torch.int32, device=dev)

    def ldpc_only():
        eng.lib.dvbs2gpu_ldpc_decode_batch(eng.h, RATE, SHORT, llr.data_ptr(), nfr, ITERS, 1, hard.data_ptr(), None, tri.data_ptr(), eng._stream())
    ldpc_only()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    nk = 3
    e0.record()
    for _ in range(nk):
        ldpc_only()
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / nk
    # the same launch in the normal (early-exit) mode: this noise never converges, so it also runs ITERS iterations, with the reference's
    # syndrome check (LDPCDecoder::bad) before every one; the forced mode evaluates the check once, after the last iteration, because its
    # result cannot end the loop earlier
    def ldpc_checked():
        eng.lib.dvbs2gpu_ldpc_decode_batch(eng.h, RATE, SHORT, llr.data_ptr(), nfr, ITERS, 0, hard.data_ptr(), None, tri.data_ptr(), eng._stream())
    ldpc_checked()
    torch.cuda.synchronize()
    e0.record()
    ldpc_checked()
    e1.record()
    torch.cuda.synchronize()
    k_ms_checked = e0.elapsed_time(e1)
    checked_all_ran = bool((tri == -1).all().item())
    bytes_per_frame = ITERS * 4 * info['ldpc_edges'] + info['ldpc_n'] + info['kbch'] // 8
    achieved = bytes_per_frame * nfr / (k_ms * 1e-3) / 1e9
    plan = eng.ldpc_plan_info(RATE, bool(SHORT))
    # PMC traffic of this very launch shape, collected offline (counters need their own rocprofv3 passes) and committed
    traffic, traffic_note = None, None
    tp = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_ldpc_traffic.json')
    if os.path.exists(tp):
        tj = json.load(open(tp))
        if tj['launch']['frames'] == nfr and tj['launch']['iterations'] == ITERS:
            traffic = tj['traffic_bytes_per_launch']
            traffic_note = tj['source'] + '; ' + tj['correction']

    if rank == 0:
        value = world * S * F * args.steps * sym / dt / 1e6
        line = {
            'metric': 'Msymbols/s demod+FEC, DVB-S2 8PSK 3/4 normal-frame @50 LDPC iters',
            'value': round(value, 3), 'unit': 'Msymbols/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'int8 (FEC) / f32 (demod)', 'data': 'synthetic',
            'config': {'workload': WORKLOAD,
                       'streams_per_gpu': S, 'frames_per_stream_per_step': F, 'symbols_per_frame': sym,
                       'parallelism': 'independent transponder streams sharded over GPUs, no data-path collective',
                       'all_frames_delivered': frames_ok, 'output_bit_exact': exact,
                       'fec_pipelined_across_steps': pipelined},
            'roofline': {'bound': 'hbm', 'kernel': 'ldpc_decode_kernel<%d,%d,%s>' % (plan['max_deg'], plan['rec_dwords'], 'true' if plan.get('irregular') else 'false'), 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                         'traffic_unit': 'bytes per launch (fabric-side FETCH_SIZE x2 + WRITE_SIZE)', 'traffic_source': traffic_note,
                         'algorithmic_bytes_per_launch': bytes_per_frame * nfr,
                         'kernel_ms': round(k_ms, 4), 'frames_per_launch': nfr, 'algorithmic_bytes_per_frame': bytes_per_frame,
                         'ldpc_share_of_step': round(k_ms / (dt / args.steps * 1e3), 3),
                         'syndrome_check': 'forced mode (the headline): evaluated once, after the last iteration; normal mode: before every iteration, as in the reference',
                         'kernel_ms_normal_mode_same_iterations': round(k_ms_checked, 4) if checked_all_ran else None,
                         'note': 'posteriors stay in LDS; the message records (132 MB live) bounce through the Infinity Cache, whose hits the fabric-side counters include',
                         'plan': plan},
        }
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(blocks)
        if world == 1 and not args.no_aux:
            line['single_transponder'] = single_transponder(eng, pkg)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for d in demods:
        d.close()
    eng.close()


if __name__ == '__main__':
    main()
