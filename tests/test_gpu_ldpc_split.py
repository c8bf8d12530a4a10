"""The half-row LDPC decoder (csrc/ldpc_split_kernel.hip: two lanes per row, one frame per workgroup) serves the normal frames it takes (rates 1/4, 2/5, 1/2, 3/5, 2/3, 3/4) by default; the
context option ldpc_split = 0 hands them back to the lane-per-row decoder (csrc/ldpc_kernel.hip).  BOTH are run here on engines of their own: posteriors, trial
counts and hard decisions must equal the oracle's -- early exit, iteration limit, forced iterations, erasures, saturating garbage -- and many frames must flow
through the persistent grid's work counter.  The half-row decoder's layers with shared bits are speculative (an ATTEMPT at the plain row update where the layer before changed no
shared posterior, else a chain walk / passes): a third engine runs it with every attempt made to fail (context option ldpc_split_fail_attempts), so the fall-back path is held
to the same bits."""
import os

import numpy as np
import pytest
import orc
from test_gpu_fec import MARGINAL_SNR, make_llrs, oracle_ldpc

pytestmark = pytest.mark.gpu
CODES = [(6, 0), (3, 0), (4, 0), (5, 0), (2, 0), (0, 0)]          # rate 3/4 (14 links per row), 1/2 (7: half 1 carries a neutral slot), 3/5 (11), 2/3 (10; LAYER 0 -- with the row that has no previous
                                                                  # parity bit -- is one with two shared pairs, 90 levels: speculative passes), 2/5 (6; layer 0 a chain layer), 1/4 (4: two slots per half)


@pytest.fixture(scope='module', params=[{'ldpc_split': 1}, {'ldpc_split': 0}, {'ldpc_split': 1, 'ldpc_split_fail_attempts': 1}], ids=['half_row', 'lane_per_row', 'half_row_attempts_fail'])
def split_engine(pkg, request):
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    eng = pkg.Engine(0, options=request.param)
    for r, _ in CODES:
        assert eng.ldpc_decoder_form(r, False) == (2 if request.param['ldpc_split'] else 0)
    yield eng
    eng.close()


@pytest.mark.parametrize('rate,short', CODES)
def test_half_row_decoder_bit_exact(split_engine, pkg, rate, short):
    import torch
    assert pkg.ldpc_split_plan(rate, short) is not None
    rng = np.random.default_rng(300 + rate)
    m = MARGINAL_SNR[rate]
    snrs = [m + 3.0, m + 0.6, m + 0.3, m, -8.0]
    p, llr, _ = make_llrs(rate, short, 8, rng, snrs)
    llr[0, ::7] = 0
    llr[5] = rng.integers(-128, 128, size=p['N']).astype(np.int8)        # saturating garbage
    for force, mt in ((0, 12), (1, 9), (0, 0), (0, 1)):
        want_post, want_trials = oracle_ldpc(rate, short, llr, mt, force)
        hard, trials, post = split_engine.ldpc_decode(torch.from_numpy(llr).cuda(), rate, bool(short), max_trials=mt, force=bool(force), want_post=True)
        torch.cuda.synchronize()
        assert np.array_equal(trials.cpu().numpy(), want_trials), (force, mt)
        assert np.array_equal(post.cpu().numpy(), want_post), (force, mt)
        want_hard = np.packbits((want_post[:, :p['K']] < 0).astype(np.uint8), axis=1)
        assert np.array_equal(hard.cpu().numpy(), want_hard), (force, mt)
    # frames that have long converged (the attempts' home ground): 30 forced iterations at a comfortable level
    p2, llr2, _ = make_llrs(rate, short, 4, rng, [m + 4.0, m + 6.0])
    want_post, want_trials = oracle_ldpc(rate, short, llr2, 30, 1)
    hard, trials, post = split_engine.ldpc_decode(torch.from_numpy(llr2).cuda(), rate, bool(short), max_trials=30, force=True, want_post=True)
    torch.cuda.synchronize()
    assert np.array_equal(trials.cpu().numpy(), want_trials) and np.array_equal(post.cpu().numpy(), want_post)


def test_half_row_decoder_many_frames(split_engine):
    """more frames than resident workgroups: the work counter hands every frame to exactly one workgroup, every frame slot is reused with clean records"""
    import torch
    rate, short = 6, 0
    rng = np.random.default_rng(19)
    p, llr4, _ = make_llrs(rate, short, 4, rng, [MARGINAL_SNR[rate] + 0.5, -8.0])
    reps = 300
    llr = np.tile(llr4, (reps, 1))
    want_post, want_trials = oracle_ldpc(rate, short, llr4, 5)
    hard, trials, post = split_engine.ldpc_decode(torch.from_numpy(llr).cuda(), rate, False, max_trials=5, want_post=True)
    torch.cuda.synchronize()
    assert np.array_equal(trials.cpu().numpy(), np.tile(want_trials, reps))
    assert np.array_equal(post.cpu().numpy().reshape(reps, 4, -1), np.broadcast_to(want_post, (reps,) + want_post.shape))


@pytest.mark.parametrize('rate', [6, 4, 3, 5, 2, 0])
def test_half_row_decoder_soak_against_the_lane_per_row_decoder(pkg, rate):
    """many frames over a spread of noise levels (frames that converge at once, slowly, never), normal and forced mode, several iteration limits: the half-row decoder -- attempts,
    passes, walks, whatever each frame makes it take -- and the lane-per-row decoder must agree bit for bit (posteriors, trial counts, hard decisions); the lane-per-row decoder
    is held to the oracle by the tests above and by test_gpu_fec.py"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    rng = np.random.default_rng(77 + rate)
    m = MARGINAL_SNR[rate]
    snrs = list(np.linspace(m - 1.0, m + 6.0, 24)) + [-8.0]
    p, llr, _ = make_llrs(rate, 0, 200, rng, snrs)
    llr[7] = rng.integers(-128, 128, size=p['N']).astype(np.int8)
    llr[9, ::5] = 0
    a = pkg.Engine(0, options={'ldpc_split': 1})
    b = pkg.Engine(0, options={'ldpc_split': 0})
    try:
        x = torch.from_numpy(llr).cuda()
        for force, mt in ((0, 25), (1, 40), (1, 3), (0, 6)):
            ha, ta, pa = a.ldpc_decode(x, rate, False, max_trials=mt, force=bool(force), want_post=True)
            hb, tb, pb = b.ldpc_decode(x, rate, False, max_trials=mt, force=bool(force), want_post=True)
            torch.cuda.synchronize()
            assert torch.equal(ta, tb), (force, mt)
            assert torch.equal(pa, pb), (force, mt)
            assert torch.equal(ha, hb), (force, mt)
    finally:
        a.close(); b.close()
