"""The wave-per-frame LDPC decoder (csrc/ldpc_wave_kernel.hip) serves the short codes it is faster on; here it is FORCED for every short
code (and the lane-per-row decoder forced for all of them as well), on fresh engines: posteriors, trial counts and hard decisions must
equal the oracle's -- early exit, iteration limit, forced iterations, erasures -- and the reference's golden outputs."""
import os

import numpy as np
import pytest
import orc
from test_gpu_fec import MARGINAL_SNR, make_llrs, oracle_ldpc

pytestmark = pytest.mark.gpu
SHORT = [(r, s) for r, s in orc.ALL_CODES if s]


@pytest.fixture(scope='module', params=[1, 0], ids=['wave-per-frame', 'lane-per-row'])
def forced_engine(request, pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    eng = pkg.Engine(0, options={'ldpc_wave': request.param})      # (read when an engine first builds a code's plan)
    yield eng
    eng.close()


@pytest.mark.parametrize('rate,short', SHORT)
def test_short_codes_bit_exact_with_either_decoder(forced_engine, rate, short):
    import torch
    rng = np.random.default_rng(100 + rate)
    m = MARGINAL_SNR[rate]
    snrs = [m + 3.0, m + 0.6, m + 0.3, m, -8.0]
    p, llr, _ = make_llrs(rate, short, 20, rng, snrs)
    llr[0, ::7] = 0
    llr[7] = rng.integers(-128, 128, size=p['N']).astype(np.int8)        # saturating garbage
    for force, mt in ((0, 12), (1, 9), (0, 0)):
        want_post, want_trials = oracle_ldpc(rate, short, llr, mt, force)
        hard, trials, post = forced_engine.ldpc_decode(torch.from_numpy(llr).cuda(), rate, True, max_trials=mt, force=bool(force), want_post=True)
        torch.cuda.synchronize()
        assert np.array_equal(trials.cpu().numpy(), want_trials), (force, mt)
        assert np.array_equal(post.cpu().numpy(), want_post), (force, mt)
        want_hard = np.packbits((want_post[:, :p['K']] < 0).astype(np.uint8), axis=1)
        assert np.array_equal(hard.cpu().numpy(), want_hard), (force, mt)


def test_many_frames_through_the_persistent_grid(forced_engine):
    """more frames than resident waves: the dynamic frame counter hands every frame to exactly one wave"""
    import torch
    rate, short = 9, 1
    rng = np.random.default_rng(9)
    p, llr8, _ = make_llrs(rate, short, 8, rng, [MARGINAL_SNR[rate] + 0.5, -8.0])
    reps = 700
    llr = np.tile(llr8, (reps, 1))
    want_post, want_trials = oracle_ldpc(rate, short, llr8, 6)
    hard, trials, post = forced_engine.ldpc_decode(torch.from_numpy(llr).cuda(), rate, True, max_trials=6, want_post=True)
    torch.cuda.synchronize()
    assert np.array_equal(trials.cpu().numpy(), np.tile(want_trials, reps))
    assert np.array_equal(post.cpu().numpy(), np.tile(want_post, (reps, 1)))
