"""Context options that select an alternative flow (DVBS2GPU_OPTIONS in the environment reaches every context a process creates): each one is forced in a child process over
the chain tests that exercise the flow it changes -- every flow must give the bytes of the default one (the tests compare with the oracle)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    # frame loops behind the PL sync only (no speculation ahead of it) for small banks
    (dict(DVBS2GPU_OPTIONS='loops_ahead=0'), 'front_end_is_bit_identical or tiny_and_empty or realignment or (demod_end_to_end_vs_oracle and (4-1-0 or 14-1-0))'),
    # small mixed batches through round 3's flow: a shared front-end pass, then a host thread + HIP stream per configuration group
    (dict(DVBS2GPU_OPTIONS='mixed_groups=1'), 'mixed'),
    # ... and the one-launch-per-stage flow with its FEC jobs in line on ONE side stream
    (dict(DVBS2GPU_OPTIONS='mix_fec_streams=1'), 'mixed'),
    # the throughput mode's big FEC jobs on the partition stream (a subset of the compute units; by default a rule decides per batch), jobs of other flows on the plain FEC stream behind them
    (dict(DVBS2GPU_OPTIONS='fec_part=1'), 'pipelined'),
    # all post stages of a slice on one stream (default for big banks: the frame loops on a stream of their own beside the next slice's RRC)
    (dict(DVBS2GPU_OPTIONS='stage_loops_stream=0'), 'pipelined'),
]


@pytest.mark.gpu
@pytest.mark.parametrize('env,sel', CASES, ids=[next(iter(e.values())) for e, _ in CASES])
def test_alternative_flows_give_the_same_bytes(env, sel):
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_s2chain.py'), '-m', 'gpu', '-x', '-q', '-k', sel],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_partition_stream_beside_null_stream_work_of_the_host(pkg):
    """the partition stream (hipExtStreamCreateWithCUMask) is a default-flag, BLOCKING stream: work the host enqueues on the legacy null stream serialises with the decoder
    jobs on it (INTEGRATION.md tells hosts to keep to explicit non-blocking streams) -- but it must never change results: the same pipelined calls with fec_part = 1 and a
    null-stream operation of the host (a torch default-stream kernel and a plain hipMemcpy-style copy) between every two calls deliver what the default flow delivers"""
    import numpy as np
    import torch
    import orc
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    S, chunk = 6, 40000
    sigs = [orc.transmit(14, 1, 0, nframes=7, seed=90 + i, esn0_db=15.0, cfo=1e-4, timing=0.1 * i, phase0=0.2, lead_symbols=300 + 50 * i)[0] for i in range(S)]
    got = []
    for opts, poke in (({}, False), ({'fec_part': 1}, True)):
        eng = pkg.Engine(0, options=opts)
        cfg = eng.default_cfg(14, True, False, max_ldpc_trials=16)
        dms = [eng.demod(cfg, max_samples=chunk) for _ in range(S)]
        outs = [torch.zeros(1 << 18, dtype=torch.uint8, device='cuda') for _ in range(S)]
        eng.set_pipelined(True)
        per = [bytearray() for _ in range(S)]
        junk = torch.zeros(1 << 20, device='cuda')
        for a in list(range(0, max(x.size for x in sigs), chunk)) + [None]:
            parts = [torch.from_numpy(np.ascontiguousarray(x[a:a + chunk] if a is not None else x[:0])).cuda() for x in sigs]
            nb = eng.process_batch(dms, parts, outs)
            for i in range(S):
                per[i] += outs[i][:nb[i]].cpu().numpy().tobytes()
            if poke:
                junk.add_(1.0)                                  # a kernel on the null stream
                junk[:1024].copy_(torch.ones(1024))             # and a host -> device copy on it
        eng.set_pipelined(False)
        for d in dms:
            d.close()
        eng.close()
        got.append([bytes(x) for x in per])
    assert got[0] == got[1] and all(len(x) > 0 for x in got[0])
