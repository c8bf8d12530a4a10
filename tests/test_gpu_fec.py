"""GPU parity tests for the FEC stages, through the C ABI, against the CPU oracle (bit-exact)."""
import numpy as np
import pytest
import orc

pytestmark = pytest.mark.gpu

# snr_db for orc.bits_to_llr giving a handful of iterations per code rate (BPSK, real noise)
MARGINAL_SNR = {0: 0.7, 1: 1.2, 2: 1.3, 3: 2.2, 4: 2.8, 5: 4.1, 6: 4.7, 7: 5.2, 8: 5.7, 9: 6.6, 10: 6.8}


def make_llrs(rate, short, nframes, rng, snrs):
    p = orc.fec_params(rate, short)
    llr = np.zeros((nframes, p['N']), np.int8)
    bbs = np.zeros((nframes, p['kbch'] // 8), np.uint8)
    for f in range(nframes):
        bb, bits = orc.encode_frame(rate, short, 1000 * rate + f)
        llr[f] = orc.bits_to_llr(bits, snrs[f % len(snrs)], rng)
        bbs[f] = bb
    return p, llr, bbs


def oracle_ldpc(rate, short, llr, max_trials, force=0):
    post = llr.copy()
    trials = np.zeros(llr.shape[0], np.int32)
    for f in range(llr.shape[0]):
        trials[f] = orc.lib().orc_ldpc_decode(rate, short, post[f], max_trials, force)
    return post, trials


@pytest.mark.parametrize('rate,short', orc.ALL_CODES)
def test_ldpc_posteriors_bit_exact_all_codes(engine, rate, short):
    import torch
    rng = np.random.default_rng(rate * 2 + short)
    m = MARGINAL_SNR[rate]
    snrs = [m + 3.0, m + 0.6, m + 0.3, m, -8.0]   # easy, marginal x3, hopeless (runs to max_trials)
    nf = 5 if not short else 10
    p, llr, _ = make_llrs(rate, short, nf, rng, snrs)
    # a frame with exact zeros (erasures) must also follow the reference's "zero = failed check" rule
    llr[0, ::7] = 0
    want_post, want_trials = oracle_ldpc(rate, short, llr, 12)
    hard, trials, post = engine.ldpc_decode(torch.from_numpy(llr).cuda(), rate, bool(short), max_trials=12, want_post=True)
    torch.cuda.synchronize()
    assert np.array_equal(trials.cpu().numpy(), want_trials)
    assert np.array_equal(post.cpu().numpy(), want_post)          # int8 soft outputs: exact (tolerance 0 <= 1e-4)
    want_hard = np.packbits((want_post[:, :p['K']] < 0).astype(np.uint8), axis=1)
    assert np.array_equal(hard.cpu().numpy(), want_hard)


def test_ldpc_forced_iterations_and_zero_trials(engine):
    import torch
    rate, short = 6, 0
    rng = np.random.default_rng(5)
    p, llr, _ = make_llrs(rate, short, 3, rng, [30.0, MARGINAL_SNR[rate] + 0.5, -8.0])
    for force, mt in ((1, 7), (0, 0), (0, 1)):
        want_post, want_trials = oracle_ldpc(rate, short, llr, mt, force)
        hard, trials, post = engine.ldpc_decode(torch.from_numpy(llr).cuda(), rate, False, max_trials=mt, force=bool(force), want_post=True)
        torch.cuda.synchronize()
        assert np.array_equal(trials.cpu().numpy(), want_trials), (force, mt)
        assert np.array_equal(post.cpu().numpy(), want_post), (force, mt)


@pytest.mark.parametrize('rate,short', [(6, 0), (5, 0), (9, 0), (3, 1), (9, 1)])
def test_bch_decode_bit_exact(engine, rate, short):
    import torch
    rng = np.random.default_rng(11 + rate)
    p = orc.fec_params(rate, short)
    nb = p['K'] // 8
    t = p['t']
    cases = [0, 1, 2, 3, 4, t - 1, t, t + 1, t + 2, 2 * t, 40]
    frames = np.zeros((len(cases) * 3, nb), np.uint8)
    for n, ne in enumerate(cases * 3):
        fr = np.zeros(nb, np.uint8)
        orc.lib().orc_make_bbframe(fr, p['kbch'], 50 + n)
        orc.lib().orc_bch_encode(rate, short, fr)
        pos = rng.choice(p['K'], ne, replace=False)
        if n % 3 == 1 and ne:
            pos[0] = p['K'] - 1 - int(rng.integers(0, 100))   # one error in the parity bits
        for x in np.unique(pos):
            fr[x // 8] ^= 1 << (7 - x % 8)
        frames[n] = fr
    want = frames.copy()
    want_ret = np.zeros(frames.shape[0], np.int32)
    for n in range(frames.shape[0]):
        want_ret[n] = orc.lib().orc_bch_decode(rate, short, want[n])
    d = torch.from_numpy(frames).cuda()
    corr = engine.bch_decode(d, rate, bool(short))
    torch.cuda.synchronize()
    assert np.array_equal(corr.cpu().numpy(), want_ret)
    assert np.array_equal(d.cpu().numpy(), want)


@pytest.mark.parametrize('rate,short', [(6, 0), (3, 0), (9, 1), (0, 1)])
def test_fec_chain_recovers_bbframes(engine, rate, short):
    import torch
    rng = np.random.default_rng(3)
    m = MARGINAL_SNR[rate]
    nf = 6
    p, llr, bbs = make_llrs(rate, short, nf, rng, [m + 1.0, m + 0.5])
    out, trials, corr = engine.fec_decode(torch.from_numpy(llr).cuda(), rate, bool(short), max_trials=25)
    torch.cuda.synchronize()
    exp = np.zeros_like(bbs)
    exp_t = np.zeros(nf, np.int32)
    exp_c = np.zeros(nf, np.int32)
    for f in range(nf):
        l = llr[f].copy()
        c = np.zeros(1, np.int32)
        exp_t[f] = orc.lib().orc_fec_decode_frame(rate, short, l, 25, 0, exp[f], c)
        exp_c[f] = c[0]
    assert np.array_equal(out.cpu().numpy(), exp)
    assert np.array_equal(trials.cpu().numpy(), exp_t)
    assert np.array_equal(corr.cpu().numpy(), exp_c)
    assert np.array_equal(out.cpu().numpy(), bbs)      # and it is what the transmitter sent


def test_ldpc_batch_larger_than_grid_and_empty(engine):
    """more frames than resident workgroups (persistent loop) and the empty batch"""
    import torch
    rate, short = 9, 1
    rng = np.random.default_rng(9)
    nf = 1500
    p, base, _ = make_llrs(rate, short, 8, rng, [MARGINAL_SNR[rate] + 1.0, MARGINAL_SNR[rate] + 0.3])
    llr = np.tile(base, (nf // 8 + 1, 1))[:nf].copy()
    want_post, want_trials = oracle_ldpc(rate, short, base, 10)
    hard, trials, post = engine.ldpc_decode(torch.from_numpy(llr).cuda(), rate, True, max_trials=10, want_post=True)
    torch.cuda.synchronize()
    post = post.cpu().numpy(); trials = trials.cpu().numpy()
    for f in range(nf):
        assert trials[f] == want_trials[f % 8]
    assert np.array_equal(post[:8], want_post) and np.array_equal(post[-8:], np.roll(want_post, -((nf - 8) % 8), axis=0))
    e = torch.empty((0, p['N']), dtype=torch.int8, device='cuda')
    h, t, _ = engine.ldpc_decode(e, rate, True)
    assert h.shape[0] == 0


def test_fec_round_trip_at_the_headline_batch_size(engine):
    """BASELINE's headline shape -- 4096 normal frames of rate 3/4, 50 forced iterations, one launch -- through size-independent properties:
    encode -> independent noise per frame -> decode returns every transmitted BBFRAME; frames with identical input (the noise repeats every
    1024 frames) come out identical wherever they sit in the batch; the first frames equal the oracle's output bit for bit"""
    import torch
    rate, short, nf, ndistinct = 6, 0, 4096, 16
    p = orc.fec_params(rate, short)
    enc = [orc.encode_frame(rate, short, 500 + k) for k in range(ndistinct)]
    bbs = np.stack([e[0] for e in enc])
    bits = torch.from_numpy(np.stack([e[1] for e in enc])).cuda()
    g = torch.Generator(device='cuda'); g.manual_seed(11)
    sigma = 10 ** (-(MARGINAL_SNR[rate] + 1.0) / 20.0)
    noise = torch.randn((1024, p['N']), generator=g, device='cuda').repeat(4, 1)
    x = (1.0 - 2.0 * bits.float()).repeat(nf // ndistinct, 1)
    llr = torch.clamp(torch.round((x + sigma * noise) * (2.0 / (sigma * sigma))), -127, 127).to(torch.int8)
    out, trials, corr = engine.fec_decode(llr, rate, False, max_trials=50, force=True)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    want = np.tile(bbs, (nf // ndistinct, 1))
    assert np.array_equal(out, want), int((out != want).any(axis=1).sum())
    assert int((trials != 50).sum()) == 0 and int((corr < 0).sum()) == 0
    hard, _, post = engine.ldpc_decode(llr, rate, False, max_trials=50, force=True, want_post=True)
    post = post.cpu().numpy()
    for k in range(1, 4):
        assert np.array_equal(post[:1024], post[1024 * k:1024 * (k + 1)]), k
    want_post, _ = oracle_ldpc(rate, short, llr[:3].cpu().numpy(), 50, force=1)
    assert np.array_equal(post[:3], want_post)
