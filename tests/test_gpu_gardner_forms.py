"""The three forms of the timing-recovery kernel (csrc/s2_rx_kernels.hip: one wave with 8 lanes per stream, resolver + producer waves, candidate
tables) are selected by bank size and by which stream of the pipelined step is critical; every one must be bit-identical to the oracle.  A form
is forced through the context option gardner_form (DVBS2GPU_OPTIONS reaches every context of a process), so each runs the chain tests that
exercise the front end in a child process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize('form', [1, 2, 4])
def test_chain_is_bit_identical_with_every_timing_recovery_form(form):
    env = dict(os.environ, DVBS2GPU_OPTIONS='gardner_form=%d' % form)
    sel = 'front_end_is_bit_identical or time_sliced_front_end or tiny_and_empty or (demod_end_to_end_vs_oracle and (4-1-0 or 6-1-1 or 14-1-0))'
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_s2chain.py'), '-m', 'gpu', '-x', '-q', '-k', sel],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize('skew', [4, 40])
def test_candidate_tables_fall_back_to_the_dot_products_bit_exactly(skew):
    """form 4 never leaves its candidate tables on these signals (the arm moves by a few hundredths per symbol); with the prediction skewed
    by 4 arms the resolver is at the edge of the table and leaves it whenever the arm drifts, with 40 it computes EVERY symbol itself --
    the chain must come out bit-identical either way"""
    env = dict(os.environ, DVBS2GPU_OPTIONS='gardner_form=4,gardner_cand_skew=%d' % skew)
    sel = 'front_end_is_bit_identical or time_sliced_front_end or tiny_and_empty or (demod_end_to_end_vs_oracle and (4-1-0 or 14-1-0))'
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_s2chain.py'), '-m', 'gpu', '-x', '-q', '-k', sel],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
