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
]


@pytest.mark.gpu
@pytest.mark.parametrize('env,sel', CASES, ids=[next(iter(e.values())) for e, _ in CASES])
def test_alternative_flows_give_the_same_bytes(env, sel):
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_s2chain.py'), '-m', 'gpu', '-x', '-q', '-k', sel],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
