"""GPU parity tests of the DVB-S2 receive chain (front end, PL sync, PLL/PLHDR, demapper) through the C ABI
against the CPU oracle on the same synthetic IQ.

Tolerance: NONE.  Every fp32 operation of the float stages is a separately rounded IEEE operation in the same order on
both sides (-ffp-contract=off), and sin/cos/atan2/exp/log are the engine's own straight-line definitions
(include/dvbs2gpu_math.h, tests/test_det_math.py shows gfx950 and x86-64 evaluate them to the same bits), so symbols,
aligned frames, PLL outputs, LLRs, per-frame statistics and BBFRAMEs are compared for EQUALITY, call by call, with the
carrier / timing loops active.  (Round 1 used the device libm and could only compare statistically: the sign-directed
Gardner TED and the LUT-quantised PLL turn a 1-ULP phasor difference into a different polyphase arm or LUT cell.)"""
import numpy as np
import pytest
import orc

pytestmark = pytest.mark.gpu

def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.dtype.kind in 'fc':
        return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    return np.array_equal(a, b)


@pytest.mark.parametrize('modcod,short,pilots', [(4, 1, 0), (14, 1, 0), (12, 1, 0), (19, 1, 0), (6, 1, 1), (14, 1, 1),
                                                 (19, 0, 0), (12, 0, 1), (4, 0, 1), (21, 0, 1)])
def test_demap_bit_exact(engine, modcod, short, pilots):
    """LUT demap + de-interleave on arbitrary symbols: exact (same LUT, same double index math).  QPSK, 8PSK (3/5: columns reversed) and
    16APSK normal frames take the four-symbols-per-lane form, 16APSK short frames (columns of 4050 bytes) and symbols at an address that is
    not a multiple of 16 the byte form"""
    import torch
    mp = orc.modcod_params(modcod, short, pilots)
    rng = np.random.default_rng(modcod)
    F = 3
    fr = (rng.standard_normal((F, mp['plframe'])) + 1j * rng.standard_normal((F, mp['plframe']))).astype(np.complex64) * 0.6
    fr[0, 90:200] = 0                      # exact zeros, and samples far outside the LUT
    fr[1, 90:150] *= 50
    rx = orc.OracleRx(orc.default_cfg(modcod, short, pilots))
    want = np.zeros((F, mp['N']), np.int8)
    for f in range(F):
        rx.L.orc_s2rx_to_soft(rx.h, np.ascontiguousarray(fr[f]).view(np.float32), want[f])
    got = engine.demap(torch.from_numpy(fr).cuda(), modcod, bool(short), bool(pilots))
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    # the same frames one sample further into a buffer (8 bytes off the 16-byte grid)
    buf = torch.zeros(F * mp['plframe'] + 1, dtype=torch.complex64, device='cuda')
    buf[1:] = torch.from_numpy(fr).cuda().reshape(-1)
    got2 = engine.demap(buf[1:].reshape(F, mp['plframe']), modcod, bool(short), bool(pilots))
    assert np.array_equal(got2.cpu().numpy(), want)


@pytest.mark.parametrize('modcod,short,pilots', [(27, 1, 1), (24, 0, 0), (28, 0, 1)])
def test_demap_32apsk_bit_exact(engine, modcod, short, pilots):
    """32APSK is computed per symbol with exp/log (constellation.cpp:205-261,319-321) -- the shared fp64 definitions on both
    sides: the int8 LLRs are equal, including subnormal exp() results and the all-underflow (non-finite) case"""
    import torch
    mp = orc.modcod_params(modcod, short, pilots)
    rng = np.random.default_rng(1)
    F = 3
    fr = (rng.standard_normal((F, mp['plframe'])) + 1j * rng.standard_normal((F, mp['plframe']))).astype(np.complex64) * 0.7
    fr[1, 90:400] *= 4          # far outside the constellation: exp() results go subnormal
    fr[2, 90:300] *= 40         # every exp() underflows to 0: log(0) - log(0) is NaN -> LLR 0 by definition
    fr[2, 300:400] = 0
    rx = orc.OracleRx(orc.default_cfg(modcod, short, pilots))
    want = np.zeros((F, mp['N']), np.int8)
    for f in range(F):
        rx.L.orc_s2rx_to_soft(rx.h, np.ascontiguousarray(fr[f]).view(np.float32), want[f])
    got = engine.demap(torch.from_numpy(fr).cuda(), modcod, bool(short), bool(pilots)).cpu().numpy()
    assert np.array_equal(got, want), int((got != want).sum())
    assert np.abs(want[0].astype(np.int32)).max() > 60           # (the comparison is not about zeros)


CASES = [
    # modcod, short, pilots, esn0, nframes, chunk, frames that must decode, max LDPC trials
    # (how many frames decode is the reference receiver's acquisition behaviour -- its decision-directed loops are slow and fragile
    #  at low SNR with a carrier offset; the oracle shows the same, frame by frame -- parity is the equality of everything)
    (4, 1, 0, 12.0, 8, 7919, 4, 16),
    (14, 1, 0, 16.0, 16, 20000, 4, 16),
    (14, 0, 0, 16.0, 4, 50000, 1, 16),
    (19, 1, 0, 30.0, 10, 1000000, 1, 16),   # the reference's APSK path is fragile (AGC set point vs demapper prescale)
    (6, 1, 1, 12.0, 8, 3001, 4, 16),
    (4, 0, 0, 10.0, 8, 40001, 4, 50),       # BASELINE config 2: QPSK 1/2 NORMAL frames, 50 iterations
    (4, 0, 0, 3.0, 5, 40001, 0, 50),        # ... near threshold (threshold + 2 dB: the loops do not settle within 5 frames)
    (14, 0, 0, 14.0, 6, 30011, 2, 50),      # BASELINE config 3 at the bench's SNR
    (14, 0, 0, 9.0, 3, 30011, 0, 50),       # ... at threshold + 1 dB
    (27, 1, 1, 15.0, 12, 6007, 0, 50),      # BASELINE config 5 stand-in: 32APSK 8/9 short + pilots at threshold - 1 dB: every frame
                                            # runs into the iteration limit (max-iter stress)
    (27, 1, 1, 30.0, 12, 6007, 4, 50),      # ... and the same chain decoding
    # the lowest Es/N0 at which at least half the frames decode once the loops have settled (tools/sensitivity.py, DESIGN.md section 6:
    # the restated receiver -- the reference's distance-metric LUT demapper with its halving clamp -- stands 3 to 6 dB off the codes'
    # thresholds): equality with the oracle where frames DO decode, not only where none does
    (4, 0, 0, 7.0, 12, 40001, 5, 50),
    (14, 0, 0, 11.0, 24, 30011, 8, 50, 3),
    # ... and a signal (noise seed 14) on which this receiver settles on a wrong 8PSK rotation and delivers 23 frames of which none is a
    # transmitted one -- about one stream in eleven does at this Es/N0 (bench: fraction_equal_to_transmitted); which streams do depends
    # on the last bit of a phasor (with the separately rounded sin/cos of rounds 1-2 this very signal decoded): equality through a false lock
    (14, 0, 0, 11.0, 24, 30011, 0, 50),
    (27, 1, 1, 17.5, 24, 6007, 4, 50),      # (the loops take ~18 short frames to settle on 32APSK: the last 5 decode)
]
CASES = [c if len(c) == 9 else c + (c[0],) for c in CASES]      # last field: seed of payload + noise (default: the MODCOD number)


@pytest.mark.parametrize('modcod,short,pilots,esn0,nframes,chunk,min_good,trials,seed', CASES)
def test_demod_end_to_end_vs_oracle(engine, modcod, short, pilots, esn0, nframes, chunk, min_good, trials, seed):
    """IQ with carrier offset 1e-3 rad/sample, timing offset 0.3 samples and a phase offset through both receivers in
    chunks: every tap and every output byte of every call must be EQUAL"""
    iq, bb, _ = orc.transmit(modcod, short, pilots, nframes=nframes, seed=seed, esn0_db=esn0, cfo=1e-3, timing=0.3, phase0=0.1,
                             lead_symbols=700)
    rx = orc.OracleRx(orc.default_cfg(modcod, short, pilots, max_ldpc_trials=trials))
    dm = engine.demod(engine.default_cfg(modcod, bool(short), bool(pilots), max_ldpc_trials=trials), max_samples=max(chunk, 4096))
    outs_g, decoded, nfr_seen, maxed = [], [], 0, 0
    for ncall, a in enumerate(range(0, iq.size, chunk)):
        part = iq[a:a + chunk]
        o = rx.process(part)
        g = dm.process(part)
        assert same_bits(rx.tap(0), dm.tap(0)), ('1-sps symbols', ncall)
        assert same_bits(rx.tap(1), dm.tap(1)), ('aligned frames', ncall)
        assert same_bits(rx.tap(2), dm.tap(2)), ('PLL output', ncall)
        assert same_bits(rx.tap(3), dm.tap(3)), ('LLRs', ncall)
        st_o, st_g = rx.tap(4), dm.stats()
        assert len(st_o) == len(st_g)
        for a_, b_ in zip(st_o, st_g):
            assert np.float32(a_.best_match).view(np.uint32) == np.float32(b_.pl_sync_best_match).view(np.uint32)
            assert np.float32(a_.fed_err).view(np.uint32) == np.float32(b_.coarse_freq_err).view(np.uint32)
            assert (a_.detect_modcod, a_.detect_short, a_.detect_pilots) == (b_.detected_modcod, b_.detected_shortframes, b_.detected_pilots)
            assert (a_.ldpc_trials, a_.bch_corr) == (b_.ldpc_trials, b_.bch_corrections)
            decoded.append(a_.bch_corr >= 0)          # (a valid BCH codeword; the LDPC check may still have failed on parity bits)
            maxed += a_.ldpc_trials < 0
        nfr_seen += len(st_g)
        assert o.shape == g.shape and np.array_equal(o, g), ('BBFRAMEs', ncall)
        outs_g.append(g)
        assert np.float32(dm.nco_freq()).view(np.uint32) == np.float32(rx.L.orc_s2rx_nco_freq(rx.h)).view(np.uint32)
    G = np.concatenate(outs_g)
    assert G.shape[0] >= nframes - 3 and len(decoded) == G.shape[0] == nfr_seen
    dec = np.array(decoded, bool)
    assert dec.sum() >= min_good, ('decodable frames', int(dec.sum()))
    sent = {bytes(b) for b in bb}
    assert all(bytes(x) in sent for x in G[dec])
    if modcod == 27 and esn0 < 16:
        assert maxed >= nfr_seen - 2, ('config 5 stand-in is meant to run into the iteration limit', maxed, nfr_seen)
    dm.close()


@pytest.mark.gpu
@pytest.mark.parametrize('cut', [2 * 5000 + 1, 2 * 16233])
def test_realignment_in_mid_stream_equals_oracle(engine, cut):
    """the transmission loses `cut` samples in mid-stream: the PL sync finds its next window misaligned and realigns (dvbs2_pl_sync.cpp:145-164)
    -- with the frame loops of a small bank running AHEAD of the PL sync this is where what they did on the window that turned out not to be
    a frame has to be given up.  Every tap and every output equal to the oracle's, call by call, across the break"""
    modcod, short, pilots = 14, 1, 0
    iq, bb, _ = orc.transmit(modcod, short, pilots, nframes=14, seed=33, esn0_db=16.0, cfo=1e-3, timing=0.3, phase0=0.1, lead_symbols=211)
    half = (iq.size // 2) | 1
    iq = np.concatenate([iq[:half], iq[half + cut:]])
    rx = orc.OracleRx(orc.default_cfg(modcod, short, pilots, max_ldpc_trials=16))
    dm = engine.demod(engine.default_cfg(modcod, True, False, max_ldpc_trials=16), max_samples=16384)
    good = 0
    sent = {bytes(b) for b in bb}
    for ncall, a in enumerate(range(0, iq.size, 9001)):
        part = iq[a:a + 9001]
        o = rx.process(part)
        g = dm.process(part)
        for t in range(4):
            assert same_bits(rx.tap(t), dm.tap(t)), ('tap', t, 'call', ncall)
        assert o.shape == g.shape and np.array_equal(o, g), ('BBFRAMEs', ncall)
        assert np.float32(dm.nco_freq()).view(np.uint32) == np.float32(rx.L.orc_s2rx_nco_freq(rx.h)).view(np.uint32)
        good += sum(bytes(x) in sent for x in g)
    assert good >= 3, good          # (the frames before the break; after it the reference receiver needs many frames to settle again)
    dm.close()


@pytest.mark.gpu
def test_tiny_and_empty_calls_equal_oracle(engine):
    """calls of 0 .. 40 samples (time slices without a sample, periods of the timing recovery with fewer samples than a tile, calls that
    end between the two outputs of a symbol), then ordinary ones: every tap and every output equal to the oracle's, call by call"""
    modcod, short, pilots = 4, 1, 0
    iq, bb, _ = orc.transmit(modcod, short, pilots, nframes=4, seed=91, esn0_db=12.0, cfo=1e-3, timing=0.3, phase0=0.1, lead_symbols=150)
    rx = orc.OracleRx(orc.default_cfg(modcod, short, pilots, max_ldpc_trials=16))
    dm = engine.demod(engine.default_cfg(modcod, True, False, max_ldpc_trials=16), max_samples=8192)
    rng = np.random.default_rng(5)
    cuts = [0]
    while cuts[-1] < 2500:
        cuts.append(cuts[-1] + int(rng.choice([0, 1, 2, 3, 5, 8, 15, 16, 17, 31, 33, 40])))
    while cuts[-1] < iq.size:
        cuts.append(min(iq.size, cuts[-1] + 7001))
    good = 0
    for ncall, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        part = iq[a:b]
        o = rx.process(part)
        g = dm.process(part)
        for t in range(4):
            assert same_bits(rx.tap(t), dm.tap(t)), ('tap', t, 'call', ncall, 'samples', b - a)
        assert o.shape == g.shape and np.array_equal(o, g), ('BBFRAMEs', ncall)
        assert np.float32(dm.nco_freq()).view(np.uint32) == np.float32(rx.L.orc_s2rx_nco_freq(rx.h)).view(np.uint32)
        good += g.shape[0]
    assert good >= 2
    dm.close()


@pytest.mark.parametrize('chunk', [3001, 777, 65536])
def test_front_end_is_bit_identical_without_nco_feedback(engine, chunk):
    """fll_bw = 0 keeps the NCO at frequency 0: AGC, Gardner, RRC, decimator and PL sync alone, call by call, for any
    chunking (odd sizes exercise the decimator phase and the delay lines across calls)."""
    modcod, short, pilots = 6, 1, 1
    iq, bb, _ = orc.transmit(modcod, short, pilots, nframes=5, seed=9, esn0_db=15.0, cfo=0.0, timing=0.37, phase0=0.1, lead_symbols=333)
    rx = orc.OracleRx(orc.default_cfg(modcod, short, pilots, fll_bw=0.0))
    dm = engine.demod(engine.default_cfg(modcod, True, True, fll_bw=0.0), max_samples=max(chunk, 4096))
    nfr = 0
    for a in range(0, iq.size, chunk):
        part = iq[a:a + chunk]
        rx.process(part); dm.process(part)
        so, sg = rx.tap(0), dm.tap(0)
        assert so.size == sg.size and np.array_equal(so.view(np.uint32), sg.view(np.uint32)), ('symbols differ in call', a // chunk)
        fo, fg = rx.tap(1), dm.tap(1)
        assert fo.size == fg.size and np.array_equal(fo.view(np.uint32), fg.view(np.uint32))
        so_, sg_ = rx.tap(4), dm.stats()
        for x, y in zip(so_, sg_):
            assert x.best_match == y.pl_sync_best_match and x.fed_err == y.coarse_freq_err
        nfr += len(sg_)
    assert nfr >= 3
    dm.close()


def test_demod_batch_of_streams_matches_single(engine):
    """process_batch over several independent streams == each stream on its own handle"""
    import torch
    modcod, short, pilots = 14, 1, 0
    S = 5
    iqs, ref = [], []
    for s in range(S):
        iq, bb, _ = orc.transmit(modcod, short, pilots, nframes=5, seed=100 + s, esn0_db=18.0, cfo=5e-4 * s, timing=0.1 * s,
                                 phase0=0.05, lead_symbols=300 + 11 * s)
        iqs.append(iq)
        d = engine.demod(engine.default_cfg(modcod, True, False), max_samples=iq.size)
        ref.append(d.process(iq))
        d.close()
    dms = [engine.demod(engine.default_cfg(modcod, True, False), max_samples=max(i.size for i in iqs)) for _ in range(S)]
    tin = [torch.from_numpy(i).cuda() for i in iqs]
    kb = dms[0].info['kbch'] // 8
    tout = [torch.zeros(8 * kb, dtype=torch.uint8, device='cuda') for _ in range(S)]
    nb = engine.process_batch(dms, tin, tout)
    for s in range(S):
        got = tout[s][:nb[s]].cpu().numpy().reshape(-1, kb)
        assert np.array_equal(got, ref[s])
    for d in dms:
        d.close()


def test_demod_error_codes(engine, pkg):
    c = engine.default_cfg(14)
    c.modcod = 0
    with pytest.raises(pkg.Dvbs2GpuError) as e:
        engine.demod(c)
    assert e.value.code == -2
    c = engine.default_cfg(11, True)     # short-frame 9/10 does not exist
    with pytest.raises(pkg.Dvbs2GpuError):
        engine.demod(c)
    # configurations the buffers are not sized for are refused at creation (the timing loop's output rate is bounded by omega_rel_limit)
    for kw in (dict(omega_rel_limit=0.2), dict(omega_rel_limit=-0.1), dict(symbolrate=0.0), dict(rrc_taps=0)):
        with pytest.raises(pkg.Dvbs2GpuError) as e:
            engine.demod(engine.default_cfg(14, **kw))
        assert e.value.code == pkg.ERR_ARG, kw


def test_pipelined_batch_delivers_the_same_frames_one_call_later(engine, pkg):
    """throughput mode (dvbs2gpu_set_pipelined): FEC of call k overlaps call k+1; outputs and stats are those of the
    synchronous mode, shifted by one call; a zero-sample call collects the tail"""
    import torch
    S, calls = 6, 5
    cfgkw = dict(modcod=11, short=0, pilots=0)
    iqs = []
    for s in range(S):
        iq, bb, _ = orc.transmit(11, 0, 0, nframes=calls, seed=900 + s, esn0_db=12.0, cfo=1e-3, timing=0.25, phase0=0.3)
        iqs.append(iq)
    info = pkg.modcod_info(11, False, False)
    kb = info['kbch'] // 8
    chunk = iqs[0].size // calls
    cfg = engine.default_cfg(11, False, False)

    def run(pipelined):
        demods = [engine.demod(cfg, max_samples=chunk) for _ in range(S)]
        tout = [torch.zeros(4 * kb, dtype=torch.uint8, device='cuda') for _ in range(S)]
        engine.set_pipelined(pipelined)
        outs = []
        try:
            for c in range(calls + (1 if pipelined else 0)):
                if c < calls:
                    tin = [torch.from_numpy(iqs[s][c * chunk:(c + 1) * chunk]).cuda() for s in range(S)]
                else:
                    tin = [torch.empty(0, dtype=torch.complex64, device='cuda') for _ in range(S)]
                nb = engine.process_batch(demods, tin, tout)
                outs.append(([tout[s][:nb[s]].cpu().numpy().copy() for s in range(S)],
                             [[(x.ldpc_trials, x.bch_corrections, x.detected_modcod) for x in d.stats()] for d in demods]))
        finally:
            engine.set_pipelined(False)
            for d in demods:
                d.close()
        return outs

    sync = run(False)
    pipe = run(True)
    assert all(len(x) == 0 for x in pipe[0][0])                     # nothing can be ready in the first call
    total = 0
    for c in range(calls):
        for s in range(S):
            assert np.array_equal(pipe[c + 1][0][s], sync[c][0][s]), (c, s)
            assert pipe[c + 1][1][s] == sync[c][1][s], (c, s)
            total += len(sync[c][0][s])
    assert total >= S * (calls - 2) * kb


def test_pipelined_call_waits_for_inputs_the_host_is_still_writing_on_the_null_stream(engine, pkg):
    """throughput mode: the call runs on a non-blocking stream of its own, the host's producer of the input buffers (here torch: a long chain of kernels on the legacy
    null stream, ending in the copy that fills the buffer the call is given) on the null stream -- the call must start behind it, as the synchronous mode does by running
    on the null stream itself.  (Found by tools/stress_pipelined.py on fresh streams in round 6: frames of busy calls came out with LDPC trials -1 and BCH corrections
    where the synchronous run had 0 / 0 -- the front end had read input that was still being written.)"""
    import torch
    S, calls = 4, 4
    iqs = [orc.transmit(14, 1, 0, nframes=6 * calls, seed=7700 + s, esn0_db=22.0, cfo=2e-4, timing=0.1 * s, phase0=0.2)[0] for s in range(S)]
    chunk = min(x.size for x in iqs) // calls
    chunk -= chunk & 1
    kb = pkg.modcod_info(14, True, False)['kbch'] // 8
    cfg = engine.default_cfg(14, True, False)
    dev = [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in iqs]
    ballast = torch.zeros(64 << 20, dtype=torch.float32, device='cuda')
    torch.cuda.synchronize()

    def run(eng, pipelined):
        demods = [eng.demod(cfg, max_samples=chunk) for _ in range(S)]
        tout = [torch.zeros(16 * kb, dtype=torch.uint8, device='cuda') for _ in range(S)]
        tin = [torch.zeros(chunk, dtype=torch.complex64, device='cuda') for _ in range(S)]
        eng.set_pipelined(pipelined)
        outs = []
        try:
            for c in range(calls + (1 if pipelined else 0)):
                if c < calls:
                    for _ in range(6):
                        ballast.add_(1.0)                                   # ~1.5 ms of null-stream work in front of the copies ...
                    for s in range(S):
                        tin[s].zero_()                                      # ... which first wipe the buffers (a call that does not wait demodulates zeros / half a buffer)
                        tin[s].copy_(dev[s][c * chunk:(c + 1) * chunk])
                    nb = eng.process_batch(demods, tin, tout)               # (no synchronisation by the host in between)
                else:
                    nb = eng.process_batch(demods, [torch.empty(0, dtype=torch.complex64, device='cuda') for _ in range(S)], tout)
                outs.append(([tout[s][:nb[s]].cpu().numpy().copy() for s in range(S)],
                             [[(x.ldpc_trials, x.bch_corrections) for x in d.stats()] for d in demods]))
        finally:
            eng.set_pipelined(False)
            for d in demods:
                d.close()
        return outs

    sync = run(engine, False)
    # (an engine of its own for the throughput mode: its streams are the first it creates -- streams that happen to share a hardware queue with the null stream hide the race)
    fresh = pkg.Engine(0)
    try:
        pipe = run(fresh, True)
    finally:
        fresh.close()
    frames = 0
    for c in range(calls):
        for s in range(S):
            assert np.array_equal(pipe[c + 1][0][s], sync[c][0][s]), (c, s)
            assert pipe[c + 1][1][s] == sync[c][1][s], (c, s)
            frames += len(sync[c][1][s])
    assert frames >= S * calls * 3


def test_mixed_modcod_batch_matches_single(engine):
    """BASELINE config 4 shape: one process_batch call over transponders with DIFFERENT MODCODs (QPSK and 8PSK, normal and short
    frames): streams are grouped per configuration inside the call; every stream == its own single-stream handle"""
    import torch
    specs = [(4, 1, 0), (14, 1, 0), (6, 1, 0), (13, 0, 0), (4, 1, 0), (12, 1, 0), (14, 1, 0), (11, 0, 0)]
    iqs, ref, dms = [], [], []
    for s, (modcod, short, pilots) in enumerate(specs):
        iq, bb, _ = orc.transmit(modcod, short, pilots, nframes=4 if short else 2, seed=700 + s, esn0_db=14.0, cfo=2e-4 * s, timing=0.07 * s,
                                 phase0=0.05, lead_symbols=200 + 13 * s)
        iqs.append(iq)
        cfg = engine.default_cfg(modcod, bool(short), bool(pilots))
        d = engine.demod(cfg, max_samples=iq.size)
        ref.append(d.process(iq))
        d.close()
        dms.append(engine.demod(engine.default_cfg(modcod, bool(short), bool(pilots)), max_samples=iq.size))
    tin = [torch.from_numpy(i).cuda() for i in iqs]
    cap = max(d.info['kbch'] // 8 for d in dms) * 8
    tout = [torch.zeros(cap, dtype=torch.uint8, device='cuda') for _ in specs]
    nb = engine.process_batch(dms, tin, tout)
    total = 0
    for s, d in enumerate(dms):
        kb = d.info['kbch'] // 8
        got = tout[s][:nb[s]].cpu().numpy().reshape(-1, kb)
        assert np.array_equal(got, ref[s]), (s, specs[s])
        total += len(got)
    assert total >= len(specs)             # frames did come out
    for d in dms:
        d.close()


def test_mixed_batch_of_all_four_constellations_with_and_without_pilots(engine):
    """the one-launch-per-stage flow of mixed batches (process_mixed: per-stream parameter tables) with everything a table entry can differ in -- QPSK, 8PSK,
    16APSK and 32APSK (the closest-point search instead of the error LUT, the byte form of the demapper), normal and short frames, pilots on and off,
    several calls per stream: every stream == its own single-stream handle, call by call, bytes and per-frame statistics"""
    import torch
    specs = [(4, 1, 1), (14, 0, 1), (19, 1, 0), (27, 1, 1), (6, 1, 0), (21, 1, 1), (24, 1, 0), (14, 1, 0), (27, 1, 0), (11, 0, 0)]
    calls = 3
    iqs = []
    for s, (modcod, short, pilots) in enumerate(specs):
        iq, bb, _ = orc.transmit(modcod, short, pilots, nframes=6 if short else 3, seed=1700 + s, esn0_db=24.0, cfo=1.5e-4 * s, timing=0.06 * s,
                                 phase0=0.05 * s, lead_symbols=150 + 17 * s)
        iqs.append(iq)
    parts = lambda x, c: np.ascontiguousarray(x[c * ((x.size // calls) & ~1):(c + 1) * ((x.size // calls) & ~1)] if c < calls - 1 else x[c * ((x.size // calls) & ~1):])
    ref = []
    for s, (modcod, short, pilots) in enumerate(specs):
        d = engine.demod(engine.default_cfg(modcod, bool(short), bool(pilots)), max_samples=iqs[s].size)
        r = []
        for c in range(calls):
            out = np.concatenate([np.asarray(x).reshape(-1) for x in [d.process(parts(iqs[s], c))]] or [np.zeros(0, np.uint8)])
            r.append((out, [(x.ldpc_trials, x.bch_corrections, x.detected_modcod, x.detected_pilots) for x in d.stats()], d.tap(2).copy(), d.tap(3).copy()))
        ref.append(r)
        d.close()
    dms = [engine.demod(engine.default_cfg(m, bool(sh), bool(p)), max_samples=iqs[s].size) for s, (m, sh, p) in enumerate(specs)]
    cap = max(d.info['kbch'] // 8 for d in dms) * 8
    tout = [torch.zeros(cap, dtype=torch.uint8, device='cuda') for _ in specs]
    total = multi = 0
    for c in range(calls):
        nb = engine.process_batch(dms, [torch.from_numpy(parts(iqs[s], c)).cuda() for s in range(len(specs))], tout)
        for s, d in enumerate(dms):
            assert np.array_equal(tout[s][:nb[s]].cpu().numpy(), ref[s][c][0]), (c, s, specs[s])
            assert [(x.ldpc_trials, x.bch_corrections, x.detected_modcod, x.detected_pilots) for x in d.stats()] == ref[s][c][1], (c, s, specs[s])
            # the constellation / LLR taps of a stream of a mixed batch: every frame of the call, not just the first (the batch keeps the streams' frames in slots
            # of its LONGEST PLFRAME)
            assert same_bits(d.tap(2), ref[s][c][2]), (c, s, specs[s])
            assert np.array_equal(d.tap(3), ref[s][c][3]), (c, s, specs[s])
            total += nb[s]
            multi += len(ref[s][c][1]) > 1
    assert total > 40000 and multi >= 4        # (calls with several frames per stream did occur)
    for d in dms:
        d.close()


def test_pipelined_mode_with_and_without_the_stage_pipeline(engine, pkg):
    """throughput mode: a call runs its RRC / PL sync / frame loops either behind the timing-recovery slices (stage pipeline) or after the
    last one, as the balancer sees fit -- so the flow may change from call to call.  Always, never and every other call
    (context option stage_pipeline = 2) must deliver the same bytes and statistics, call by call"""
    import os, torch
    S, calls = 5, 6
    iqs = [orc.transmit(13, 0, 1, nframes=2 * calls, seed=940 + s, esn0_db=13.0, cfo=7e-4, timing=0.1 * s, phase0=0.2, lead_symbols=100 + 41 * s)[0] for s in range(S)]
    kb = pkg.modcod_info(13, False, True)['kbch'] // 8
    chunk = min(i.size for i in iqs) // calls

    def run(eng):
        cfg = eng.default_cfg(13, False, True)
        demods = [eng.demod(cfg, max_samples=chunk) for _ in range(S)]
        tout = [torch.zeros(6 * kb, dtype=torch.uint8, device='cuda') for _ in range(S)]
        eng.set_pipelined(True)
        outs = []
        try:
            for c in range(calls + 1):
                tin = [torch.from_numpy(np.ascontiguousarray(iqs[s][c * chunk:(c + 1) * chunk])).cuda() if c < calls
                       else torch.empty(0, dtype=torch.complex64, device='cuda') for s in range(S)]
                nb = eng.process_batch(demods, tin, tout)
                outs.append(([tout[s][:nb[s]].cpu().numpy().copy() for s in range(S)],
                             [[(x.ldpc_trials, x.bch_corrections, x.detected_modcod, x.coarse_freq_err, x.pl_sync_best_match) for x in d.stats()] for d in demods],
                             [d.nco_freq() for d in demods]))
        finally:
            eng.set_pipelined(False)
            for d in demods:
                d.close()
        return outs

    res = []
    for opts in ({'stage_pipeline': 0}, {'stage_min_duty': -1}, {'stage_pipeline': 2}):
        e2 = pkg.Engine(0, options=opts)
        res.append(run(e2))
        e2.close()
    assert sum(len(x) for c in res[0] for x in c[0]) >= S * (2 * calls - 4) * kb
    for other in res[1:]:
        for a, b in zip(res[0], other):
            for x, y in zip(a[0], b[0]):
                assert np.array_equal(x, y)
            assert a[1] == b[1] and a[2] == b[2]


def test_pipelined_full_load_every_stream_bit_exact(engine, pkg):
    """Throughput mode at the bench's load (2048 streams, 8PSK 3/4 normal frames, 50 forced LDPC iterations): the decoder of call k
    shares the CUs with the front-end kernels of call k+1 for several multi-frame rounds per workgroup.  Every byte every stream delivers
    in the overlapped steps and in the flush must equal what the SYNCHRONOUS mode delivers for the same input one call earlier, and
    (bench.FrameChecker: device-side hash + full byte compare) nearly all frames are transmitted ones, in sequence -- the rest are streams
    whose loops have not settled yet, identically in both modes.  (Regression: a hand-scheduled LDPC chain-walk loop was bit-exact alone
    and wrong only while front-end kernels were co-resident.)"""
    import torch
    import bench as B
    S, F, calls = 2048, 2, B.PREROLL_FRAMES // 2 + 3

    def run_mode(pipelined):
        run = B.S2Run(engine, pkg, torch.device('cuda', 0), B.MODCOD, B.SHORT, B.PILOTS, 14.0, S, F, 16, seed=5)
        engine.set_pipelined(pipelined)
        keep = []
        try:
            for c in range(calls + (1 if pipelined else 0)):
                nb = run.step() if c < calls else run.flush()
                if c >= calls - 3:
                    keep.append((np.asarray(nb).copy(), run.out.clone(), run.check(nb)))
        finally:
            engine.set_pipelined(False)
            run.close()
        return keep

    sync, pipe = run_mode(False), run_mode(True)
    for k in range(3):                       # pipelined call c + 1 delivers what synchronous call c does
        nb_s, out_s, chk_s = sync[k]
        nb_p, out_p, chk_p = pipe[k + 1]
        assert np.array_equal(nb_s, nb_p) and chk_s == chk_p, (k, chk_s, chk_p)
        cap = out_s.shape[1]
        valid = torch.arange(cap, device=out_s.device)[None, :] < torch.as_tensor(nb_s, device=out_s.device)[:, None]
        assert bool(((out_s == out_p) | ~valid).all()), k
        assert chk_p['delivered'] >= S * F - S // 8 and chk_p['equal'] >= 0.95 * chk_p['delivered'] and chk_p['out_of_order'] == 0, chk_p

    # ORACLE spot check under that load: sixteen of the 2048 streams go through the CPU restatement of DVBS2Demod::process (module_dvbs2_demod.cpp:216-372) with
    # the same inputs, call by call; what the pipelined engine delivered for them -- while the decoder and the whole bank's front end shared the compute units --
    # must be the oracle's BBFRAMEs byte for byte (front end incl. loop state across 15 calls, demapper, 50 forced LDPC iterations, BCH, descrambler)
    from concurrent.futures import ThreadPoolExecutor
    sample = [0, 1, 63, 64, 65, 127, 128, 500, 777, 1000, 1023, 1024, 1535, 2000, 2046, 2047]
    ref_run = B.S2Run(engine, pkg, torch.device('cuda', 0), B.MODCOD, B.SHORT, B.PILOTS, 14.0, S, F, 16, seed=5)      # (the same inputs: seeded)
    blocks = {s: ref_run.iq[s].cpu().numpy() for s in sample}
    kb, N = ref_run.kb, ref_run.info['ldpc_n']
    ref_run.close()
    rate = pkg.modcod_info(B.MODCOD, bool(B.SHORT), bool(B.PILOTS))['rate']
    assert rate == B.RATE

    def oracle_stream(s):
        rx = orc.OracleRx(orc.default_cfg(B.MODCOD, B.SHORT, B.PILOTS, force_ldpc_iters=-1))      # (-1: the front end alone; the FEC of the compared calls follows below)
        kept = []
        for c in range(calls):
            rx.process(blocks[s])
            if c >= calls - 3:
                llr = rx.tap(3).reshape(-1, N)
                frames = np.zeros((len(llr), kb), np.uint8)
                for f in range(len(llr)):
                    corr = np.zeros(1, np.int32)
                    x = llr[f].copy()
                    orc.lib().orc_fec_decode_frame(rate, B.SHORT, x, B.ITERS, 1, frames[f], corr)
                kept.append(frames)
        return kept

    with ThreadPoolExecutor(max_workers=16) as ex:
        want = dict(zip(sample, ex.map(oracle_stream, sample)))
    compared = 0
    for k in range(3):
        nb_p, out_p, _ = pipe[k + 1]
        for s in sample:
            got = out_p[s][:int(nb_p[s])].cpu().numpy().reshape(-1, kb)
            assert got.shape == want[s][k].shape and np.array_equal(got, want[s][k]), ('oracle vs pipelined engine under full load', k, s)
            compared += len(got)
    assert compared >= 3 * len(sample) * (F - 1)


def test_every_qpsk_and_8psk_modcod_in_one_mixed_batch(engine, pkg):
    """MODCODs 1..17 with normal and short frames (9/10 has no short frame) as 32 streams of ONE batch: every LDPC code of both
    frame sizes goes through the full chain side by side; once the loops have settled every delivered BBFRAME is one that was sent."""
    import torch
    cases = [(m, s) for m in range(1, 18) for s in (0, 1) if not (s == 1 and m in (11, 17))]
    nfr = {0: 7, 1: 20}
    iqs, sent, demods, kbs = [], [], [], []
    for m, s in cases:
        iq, bb, _ = orc.transmit(m, s, 0, nframes=nfr[s], seed=100 + 2 * m + s, esn0_db=25.0, lead_symbols=300)   # (no carrier / timing offsets: this is about the codes, not the loops' acquisition)
        iqs.append(torch.from_numpy(iq).cuda())
        sent.append({bytes(b) for b in bb})
        kbs.append(bb.shape[1])
        demods.append(engine.demod(engine.default_cfg(m, bool(s), False), max_samples=iq.size))
    cap = max((nfr[s] + 1) * kb for (m, s), kb in zip(cases, kbs))            # (one capacity for the whole batch)
    tout = [torch.zeros(cap, dtype=torch.uint8, device='cuda') for _ in kbs]
    try:
        nb = engine.process_batch(demods, iqs, tout)
    finally:
        for d in demods:
            d.close()
    for i, (m, s) in enumerate(cases):
        got = tout[i][:nb[i]].cpu().numpy().reshape(-1, kbs[i])
        good = [bytes(x) in sent[i] for x in got]          # (frames caught during loop acquisition are delivered too, as the reference does)
        assert len(got) >= nfr[s] - 3 and all(good[-3:]), (m, s, good)


def test_pipelined_mixed_modcod_batch_equals_synchronous(engine, pkg):
    """throughput mode with several FEC groups in flight at once (QPSK and 8PSK MODCODs, both frame sizes, in one batch: one job per
    configuration group and call on the FEC stream): frames, their order and the per-frame stats of every stream equal the synchronous
    mode's, one call later"""
    import torch
    cases = [(4, 0), (6, 1), (11, 0), (13, 0), (14, 1), (16, 0), (9, 1), (2, 0), (14, 1)]
    nfr = {0: 6, 1: 16}
    calls = 3
    iqs, kbs = [], []
    for k, (m, s) in enumerate(cases):
        iq, bb, _ = orc.transmit(m, s, 0, nframes=nfr[s], seed=300 + k, esn0_db=25.0, lead_symbols=200)
        iqs.append(iq)
        kbs.append(bb.shape[1])
    cap = max((nfr[s] + 1) * kb for (m, s), kb in zip(cases, kbs))

    def run(pipelined):
        demods = [engine.demod(engine.default_cfg(m, bool(s), False), max_samples=iq.size) for (m, s), iq in zip(cases, iqs)]
        tout = [torch.zeros(cap, dtype=torch.uint8, device='cuda') for _ in cases]
        engine.set_pipelined(pipelined)
        outs = []
        try:
            for c in range(calls + (1 if pipelined else 0)):
                tin = []
                for iq in iqs:
                    n = (iq.size // calls) & ~1
                    part = iq[c * n:(c + 1) * n] if c < calls - 1 else (iq[c * n:] if c == calls - 1 else iq[:0])
                    tin.append(torch.from_numpy(np.ascontiguousarray(part)).cuda() if part.size else torch.empty(0, dtype=torch.complex64, device='cuda'))
                nb = engine.process_batch(demods, tin, tout)
                outs.append(([tout[i][:nb[i]].cpu().numpy().copy() for i in range(len(cases))],
                             [[(x.ldpc_trials, x.bch_corrections, x.detected_modcod) for x in d.stats()] for d in demods]))
        finally:
            engine.set_pipelined(False)
            for d in demods:
                d.close()
        return outs

    sync, pipe = run(False), run(True)
    total = 0
    for c in range(calls):
        for i in range(len(cases)):
            assert np.array_equal(pipe[c + 1][0][i], sync[c][0][i]), (c, cases[i])
            assert pipe[c + 1][1][i] == sync[c][1][i], (c, cases[i])
            total += sync[c][0][i].size // kbs[i]
    assert all(x.size == 0 for x in pipe[0][0]) and total >= sum(nfr[s] for m, s in cases) - 3 * len(cases)


def test_one_long_stream_cut_into_overlapping_segments(engine, pkg):
    """a single transponder is a serial chain (0.69 Msym/s per stream), so a fast one has to be cut into
    overlapping segments that run as independent streams, each re-acquiring its loops on a warm-up prefix.  One continuous QPSK 3/4
    short-frame signal with carrier, phase and timing offsets is cut at arbitrary sample positions; behind its warm-up every segment must
    deliver exactly its run of the transmitted BBFRAMEs, so the union is the transmitted sequence."""
    import torch
    m, s = 6, 1
    total, own, warm = 8 + 6 * 10, 10, 8
    iq, bb, _ = orc.transmit(m, s, 0, nframes=total, seed=77, esn0_db=14.0, cfo=3e-4, timing=0.3, phase0=0.4, lead_symbols=123)
    index = {bytes(b): k for k, b in enumerate(bb)}
    assert len(index) == total
    plf = pkg.modcod_info(m, True, False)['plframe_symbols']
    kb = bb.shape[1]
    spf = 2 * plf                                            # samples per frame
    nseg = (total - warm) // own
    starts = [int((warm + g * own - warm - 0.37) * spf) + 2 * 123 for g in range(nseg)]      # not frame aligned
    ends = [int((warm + (g + 1) * own + 1.2) * spf) + 2 * 123 for g in range(nseg)]           # (one frame beyond: PL sync confirms a frame by the next header)
    segs = [iq[max(a, 0):min(b, iq.size)] for a, b in zip(starts, ends)]
    demods = [engine.demod(engine.default_cfg(m, True, False), max_samples=max(x.size for x in segs)) for _ in segs]
    tout = [torch.zeros((own + warm + 4) * kb, dtype=torch.uint8, device='cuda') for _ in segs]
    try:
        nb = engine.process_batch(demods, [torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in segs], tout)
    finally:
        for d in demods:
            d.close()
    covered = set()
    for g in range(nseg):
        got = [index.get(bytes(x), -1) for x in tout[g][:nb[g]].cpu().numpy().reshape(-1, kb)]
        good = [k for k in got if k >= 0]
        assert good == list(range(good[0], good[0] + len(good))), (g, got)          # a contiguous run, in order
        lo, hi = warm + g * own, warm + (g + 1) * own                                    # the segment's own frames
        assert set(range(lo, min(hi, total - 1))) <= set(good), (g, lo, hi, got)
        covered |= {k for k in good if lo <= k < hi}
    assert covered >= set(range(warm, min(warm + nseg * own, total - 1)))               # the union is the transmitted sequence


@pytest.mark.parametrize('modcod,short,nseg,own,warm,total', [(6, 1, 6, 10, 8, 150), (14, 1, 5, 12, 10, 140)])
def test_segment_receiver_returns_one_continuous_stream_in_order(engine, pkg, modcod, short, nseg, own, warm, total):
    """dvbs2gpu_segrx_*: one continuous signal (carrier, phase and timing offsets) handed over in chunks of different lengths; the
    returned BBFRAMEs are the transmitted ones, in order, without a duplicate or a gap (the loops' first acquisition at the very
    start of the stream and the frames still waiting in the last chunk's tail aside)."""
    import torch
    iq, bb, _ = orc.transmit(modcod, short, 0, nframes=total, seed=91 + modcod, esn0_db=16.0, cfo=2e-4, timing=0.3, phase0=0.7, lead_symbols=211)
    index = {bytes(b): k for k, b in enumerate(bb)}
    assert len(index) == total
    kb = bb.shape[1]
    rx = pkg.SegmentReceiver(engine, engine.default_cfg(modcod, bool(short), False), nseg, own, warm)
    d_iq = torch.from_numpy(iq).cuda()
    out = torch.zeros((nseg * own + warm + 8) * kb, dtype=torch.uint8, device='cuda')
    got, a, k = [], 0, 0
    sizes = [rx.chunk_samples, rx.chunk_samples // 3 + 1001, rx.chunk_samples]
    while a < iq.size:
        n = min(sizes[k % len(sizes)], iq.size - a)
        nbytes = rx.process(d_iq[a:a + n], out)
        got += [index.get(bytes(x), -1) for x in out[:nbytes].cpu().numpy().reshape(-1, kb)]
        a += n
        k += 1
    rx.close()
    good = [g for g in got if g >= 0]
    bad_positions = [i for i, g in enumerate(got) if g < 0]
    assert all(i < 12 for i in bad_positions), (bad_positions, got[:20])            # only while the loops first acquire
    assert good == list(range(good[0], good[0] + len(good))), got                     # in order, no duplicate, no gap
    assert good[0] <= 12 and good[-1] >= total - 4, (good[0], good[-1])


def test_segment_receiver_with_calls_shorter_than_its_history(engine, pkg):
    """chunks of 3 frames (shorter than the warm-up history the receiver keeps): every call re-acquires on the kept history and returns
    only what the previous calls have not; still one ordered, gap-free sequence.  reset() starts a new stream."""
    import torch
    m, s, total = 6, 1, 70
    iq, bb, _ = orc.transmit(m, s, 0, nframes=total, seed=17, esn0_db=16.0, cfo=2e-4, timing=0.3, phase0=1.1, lead_symbols=77)
    index = {bytes(b): k for k, b in enumerate(bb)}
    kb = bb.shape[1]
    spf = 2 * pkg.modcod_info(m, True, False)['plframe_symbols']
    rx = pkg.SegmentReceiver(engine, engine.default_cfg(m, True, False), 2, 8, 8)
    d_iq = torch.from_numpy(iq).cuda()
    out = torch.zeros(40 * kb, dtype=torch.uint8, device='cuda')
    for rep in range(2):
        got = []
        for a in range(0, iq.size, 3 * spf + 10):
            nbytes = rx.process(d_iq[a:a + 3 * spf + 10], out)
            got += [index.get(bytes(x), -1) for x in out[:nbytes].cpu().numpy().reshape(-1, kb)]
        good = [g for g in got if g >= 0]
        assert all(i < 12 for i, g in enumerate(got) if g < 0), got
        assert good == list(range(good[0], good[0] + len(good))) and good[0] <= 12 and good[-1] >= total - 4, got
        rx.reset()
    with pytest.raises(pkg.Dvbs2GpuError):
        rx.process(torch.zeros(rx.chunk_samples + 2, dtype=torch.complex64, device='cuda'), out)
    rx.close()


def test_reset_and_set_params(engine, pkg):
    """control plane of the handle: reset() (DVBS2Demod::reset, module_dvbs2_demod.cpp:98-116) makes the next run identical to a fresh
    handle's; set_params (setDemodParams, :118-168) switches MODCOD / frame size between calls and restarts the PL sync buffer; a bad
    MODCOD is refused with ERR_MODCOD and leaves the configuration alone"""
    iq_a, bb_a, _ = orc.transmit(4, 1, 0, nframes=8, seed=501, esn0_db=20.0, lead_symbols=150)
    iq_b, bb_b, _ = orc.transmit(14, 1, 0, nframes=10, seed=502, esn0_db=25.0, lead_symbols=90)
    d = engine.demod(engine.default_cfg(4, True, False), max_samples=max(iq_a.size, iq_b.size))
    first = d.process(iq_a)
    st1 = [(x.ldpc_trials, x.bch_corrections, x.detected_modcod) for x in d.stats()]
    d.reset()
    again = d.process(iq_a)
    assert np.array_equal(first, again) and st1 == [(x.ldpc_trials, x.bch_corrections, x.detected_modcod) for x in d.stats()]
    fresh = engine.demod(engine.default_cfg(4, True, False), max_samples=iq_a.size)
    assert np.array_equal(first, fresh.process(iq_a))
    fresh.close()
    sent_a = {bytes(x) for x in bb_a}
    assert sum(bytes(x) in sent_a for x in first) >= 5
    kb_a = d.get_kbch()
    with pytest.raises(pkg.Dvbs2GpuError) as ei:
        d.set_params(29, True, False)
    assert ei.value.code == pkg.ERR_MODCOD and d.get_kbch() == kb_a
    d.set_params(14, True, False)
    assert d.get_kbch() == bb_b.shape[1] * 8 and d.get_kbch() != kb_a
    out = d.process(iq_b)
    sent_b = {bytes(x) for x in bb_b}
    good = [bytes(x) in sent_b for x in out]
    assert out.shape[1] == bb_b.shape[1] and sum(good) >= 5 and all(good[-3:]), good
    d.close()


def test_segment_receiver_argument_checks(engine, pkg):
    cfg = engine.default_cfg(14, False, False)
    for nseg, own, warm in ((0, 4, 2), (2, 2, 4), (2, 4, 0), (1, 200000, 1)):       # the last: a segment longer than the batch entry's int counts
        with pytest.raises(pkg.Dvbs2GpuError) as e:
            pkg.SegmentReceiver(engine, cfg, nseg, own, warm)
        assert e.value.code == pkg.ERR_ARG
    with pytest.raises(pkg.Dvbs2GpuError) as e:
        bad = engine.default_cfg(14, False, False)
        bad.modcod = 77
        pkg.SegmentReceiver(engine, bad, 2, 4, 2)
    assert e.value.code == pkg.ERR_MODCOD


# ------------------------------------------------------------------------------------------------ extensions (include/dvbs2gpu.h)
VCM_PLS = [(4 << 2) | 2, (14 << 2) | 2, (6 << 2) | 2 | 1, 0, (19 << 2) | 2, (27 << 2) | 2 | 1, 13 << 2, (12 << 2) | 2]


def test_pipelined_batches_that_change_between_mixed_and_single_configuration(engine, pkg):
    """throughput mode, the stream set AND its shape change from call to call: a mixed batch (one launch per stage, one FEC job per LDPC code), then streams of
    one configuration only (the plain group flow), then another mix ... -- whatever flow started a job, the next call collects it for the streams it still holds"""
    import torch
    specs = [(11, 0), (11, 0), (4, 1), (14, 1), (4, 1)]
    calls = 5
    iqs, kbs = [], []
    for s, (m, sh) in enumerate(specs):
        iq, bb, _ = orc.transmit(m, sh, 0, nframes=(calls + 1) * (3 if sh else 1), seed=2300 + s, esn0_db=16.0, cfo=1e-4 * s, timing=0.05 * s, phase0=0.1)
        iqs.append(iq); kbs.append(bb.shape[1])
    chunk = [(x.size // (calls + 1)) & ~1 for x in iqs]
    schedule = [[0, 1, 2, 3], [0, 1], [1, 2, 4], [4, 2], [0, 1, 2, 3, 4]]

    def run(pipelined):
        dms = [engine.demod(engine.default_cfg(m, bool(sh), False), max_samples=chunk[s]) for s, (m, sh) in enumerate(specs)]
        tout = [torch.zeros(8 * max(kbs), dtype=torch.uint8, device='cuda') for s in range(len(specs))]
        fed = [0] * len(specs)
        engine.set_pipelined(pipelined)
        outs = []
        try:
            for who in schedule + ([[0, 1, 2, 3, 4]] if pipelined else []):
                last = len(outs) >= len(schedule)
                tin = []
                for s in who:
                    tin.append(torch.empty(0, dtype=torch.complex64, device='cuda') if last else torch.from_numpy(iqs[s][fed[s] * chunk[s]:(fed[s] + 1) * chunk[s]]).cuda())
                    fed[s] += 0 if last else 1
                nb = engine.process_batch([dms[s] for s in who], tin, [tout[s] for s in who])
                outs.append({s: (tout[s][:nb[k]].cpu().numpy().copy(), [(x.ldpc_trials, x.bch_corrections, x.detected_modcod) for x in dms[s].stats()]) for k, s in enumerate(who)})
        finally:
            engine.set_pipelined(False)
            for d in dms:
                d.close()
        return outs

    sync, pipe = run(False), run(True)
    full = schedule + [[0, 1, 2, 3, 4]]
    frames = 0
    for c in range(1, len(full)):
        for s in full[c]:
            if s in full[c - 1]:
                assert np.array_equal(pipe[c][s][0], sync[c - 1][s][0]) and pipe[c][s][1] == sync[c - 1][s][1], (c, s)
                frames += len(sync[c - 1][s][1])
            else:
                assert pipe[c][s][0].size == 0 and pipe[c][s][1] == [], (c, s)
    assert all(v[0].size == 0 for v in pipe[0].values()) and frames >= 8


@pytest.mark.parametrize('esn0,cfo,chunk', [(100.0, 0.0, 50000), (22.0, 1e-3, 30011), (22.0, 1e-3, 1000000)])
def test_acm_vcm_stream_cycling_modcods_equals_oracle(engine, esn0, cfo, chunk):
    """ONE stream whose frames cycle over seven MODCODs (QPSK, 8PSK, 16APSK, 32APSK; short and normal frames; with
    and without pilots) and a dummy PLFRAME.  The framing follows the PLS code of every frame; symbols, aligned frames, PLL output, LLRs,
    per-frame statistics (MODCOD, SOF quality, FED estimate, LDPC trials, BCH corrections, BBFRAME size) and the BBFRAME bytes must
    EQUAL the oracle's, call by call, and the decoded frames are the transmitted ones in order."""
    iq, bbs = orc.transmit_vcm(VCM_PLS, 26, seed=3, esn0_db=esn0, cfo=cfo, timing=0.3, phase0=0.2, lead_symbols=500)
    rx = orc.OracleRx(orc.default_cfg(4, 1, 0, acm_vcm=1, max_ldpc_trials=25))
    dm = engine.demod(engine.default_cfg(4, True, False, acm_vcm=1, max_ldpc_trials=25), max_samples=max(min(chunk, iq.size), 4096))
    got_all, mods = [], []
    for ncall, a in enumerate(range(0, iq.size, chunk)):
        part = iq[a:a + chunk]
        o = rx.process(part)
        g = dm.process(part)
        for t in range(4):
            assert same_bits(rx.tap(t), dm.tap(t)), ('tap', t, ncall)
        st_o, st_g = rx.tap(4), dm.stats()
        assert len(st_o) == len(st_g), ncall
        for a_, b_ in zip(st_o, st_g):
            assert np.float32(a_.best_match).view(np.uint32) == np.float32(b_.pl_sync_best_match).view(np.uint32)
            assert np.float32(a_.fed_err).view(np.uint32) == np.float32(b_.coarse_freq_err).view(np.uint32)
            assert (a_.detect_modcod, a_.detect_short, a_.detect_pilots, a_.ldpc_trials, a_.bch_corr, a_.bbframe_bytes) == \
                   (b_.detected_modcod, b_.detected_shortframes, b_.detected_pilots, b_.ldpc_trials, b_.bch_corrections, b_.bbframe_bytes)
            mods.append(b_.detected_modcod)
        assert len(o) == len(g) and all(np.array_equal(x, y) for x, y in zip(o, g)), ('BBFRAMEs', ncall)
        got_all += g
    sent = [bytes(b) for b in bbs if b is not None]
    idx = [sent.index(bytes(x)) if bytes(x) in sent else -1 for x in got_all]
    good = [k for k in idx if k >= 0]
    assert len(good) >= len(sent) - 9 and good == list(range(good[0], good[0] + len(good))), idx      # (the acquisition window swallows the first frames)
    assert len(set(mods)) >= 6 and 0 in mods, mods                                                    # dummy PLFRAMEs were recognised and skipped
    dm.close()


def test_acm_vcm_batch_of_streams_and_error_paths(engine, pkg):
    """several ACM/VCM streams and a CCM stream in one process_batch call == each on its own handle"""
    import torch
    lists = [VCM_PLS, [(6 << 2) | 2, (14 << 2) | 2], [(13 << 2) | 2 | 1, 0, (4 << 2)]]
    iqs, ref, dms = [], [], []
    for k, pl in enumerate(lists):
        iq, _ = orc.transmit_vcm(pl, 16, seed=20 + k, esn0_db=25.0, cfo=2e-4 * k, timing=0.1 * k, lead_symbols=300)
        iqs.append(iq)
        d = engine.demod(engine.default_cfg(4, True, False, acm_vcm=1), max_samples=iq.size)
        ref.append(np.concatenate(d.process(iq) or [np.zeros(0, np.uint8)]))
        d.close()
        dms.append(engine.demod(engine.default_cfg(4, True, False, acm_vcm=1), max_samples=iq.size))
    iq, _, _ = orc.transmit(14, 1, 0, nframes=6, seed=31, esn0_db=25.0, lead_symbols=200)
    iqs.append(iq)
    d = engine.demod(engine.default_cfg(14, True, False), max_samples=iq.size)
    ref.append(d.process(iq).reshape(-1))
    d.close()
    dms.append(engine.demod(engine.default_cfg(14, True, False), max_samples=iq.size))
    tin = [torch.from_numpy(i).cuda() for i in iqs]
    tout = [torch.zeros(max(i.size for i in iqs) // 2 + 100000, dtype=torch.uint8, device='cuda') for _ in iqs]
    nb = engine.process_batch(dms, tin, tout)
    for s in range(len(iqs)):
        assert nb[s] == ref[s].size and np.array_equal(tout[s][:nb[s]].cpu().numpy(), ref[s]), s
    assert sum(nb) > 20000
    for d in dms:
        d.close()


def test_acm_vcm_streams_in_the_pipelined_mode_equal_synchronous_one_call_later(engine, pkg):
    """throughput mode with ACM/VCM streams (one FEC job per LDPC code present in a call, on the FEC stream beside the next call's front end) and a
    CCM stream in the same batch: BBFRAMEs of differing size, their order and the per-frame stats equal the synchronous mode's, one call later"""
    import torch
    lists = [VCM_PLS, [(6 << 2) | 2, (14 << 2) | 2], [(13 << 2) | 2 | 1, 0, (4 << 2)]]
    iqs = []
    for k, pl in enumerate(lists):
        iq, _ = orc.transmit_vcm(pl, 16, seed=20 + k, esn0_db=25.0, cfo=2e-4 * k, timing=0.1 * k, lead_symbols=300)
        iqs.append(iq)
    iq, _, _ = orc.transmit(14, 1, 0, nframes=6, seed=31, esn0_db=25.0, lead_symbols=200)
    iqs.append(iq)
    calls = 3
    cap = max(i.size for i in iqs) // 2 + 100000

    def run(pipelined):
        dms = [engine.demod(engine.default_cfg(4, True, False, acm_vcm=1), max_samples=i.size) for i in iqs[:3]]
        dms.append(engine.demod(engine.default_cfg(14, True, False), max_samples=iqs[3].size))
        tout = [torch.zeros(cap, dtype=torch.uint8, device='cuda') for _ in iqs]
        engine.set_pipelined(pipelined)
        outs = []
        try:
            for c in range(calls + (1 if pipelined else 0)):
                tin = []
                for x in iqs:
                    n = (x.size // calls) & ~1
                    part = x[c * n:(c + 1) * n] if c < calls - 1 else (x[c * n:] if c == calls - 1 else x[:0])
                    tin.append(torch.from_numpy(np.ascontiguousarray(part)).cuda() if part.size else torch.empty(0, dtype=torch.complex64, device='cuda'))
                nb = engine.process_batch(dms, tin, tout)
                outs.append(([tout[i][:nb[i]].cpu().numpy().copy() for i in range(len(iqs))],
                             [[(x.ldpc_trials, x.bch_corrections, x.detected_modcod, x.detected_shortframes, x.bbframe_bytes) for x in d.stats()] for d in dms]))
        finally:
            engine.set_pipelined(False)
            for d in dms:
                d.close()
        return outs

    sync, pipe = run(False), run(True)
    assert all(x.size == 0 for x in pipe[0][0]) and all(len(x) == 0 for x in pipe[0][1])
    total = 0
    for c in range(calls):
        for i in range(len(iqs)):
            assert np.array_equal(pipe[c + 1][0][i], sync[c][0][i]), (c, i)
            assert pipe[c + 1][1][i] == sync[c][1][i], (c, i)
            total += sync[c][0][i].size
    assert total > 20000


def test_pipelined_batch_whose_streams_come_and_go(engine, pkg):
    """throughput mode with a stream set that changes from call to call: a stream that joins starts with nothing pending, the streams that stay get the
    frames of the previous call whatever their new position in the batch, a stream that is absent from the call after its own loses that call's
    frames (the output buffers are the collecting call's) and nothing else"""
    import torch
    S, calls = 5, 5
    iqs = [orc.transmit(11, 0, 0, nframes=calls, seed=700 + s, esn0_db=12.0, cfo=1e-3, timing=0.25, phase0=0.3)[0] for s in range(S)]
    kb = pkg.modcod_info(11, False, False)['kbch'] // 8
    chunk = iqs[0].size // calls
    cfg = engine.default_cfg(11, False, False)
    schedule = [[0, 1, 2, 3], [1, 2, 3, 4], [3, 4, 1, 2], [0, 1, 2, 3, 4], [4, 0]]

    def run(pipelined):
        demods = [engine.demod(cfg, max_samples=chunk) for _ in range(S)]
        tout = [torch.zeros(4 * kb, dtype=torch.uint8, device='cuda') for _ in range(S)]
        fed = [0] * S                                  # chunks a stream has consumed so far
        engine.set_pipelined(pipelined)
        outs = []
        try:
            for who in schedule + ([[0, 1, 2, 3, 4]] if pipelined else []):
                last = len(outs) >= len(schedule)
                tin = []
                for s in who:
                    tin.append(torch.empty(0, dtype=torch.complex64, device='cuda') if last else torch.from_numpy(iqs[s][fed[s] * chunk:(fed[s] + 1) * chunk]).cuda())
                    fed[s] += 0 if last else 1
                nb = engine.process_batch([demods[s] for s in who], tin, [tout[s] for s in who])
                outs.append({s: (tout[s][:nb[k]].cpu().numpy().copy(), [(x.ldpc_trials, x.bch_corrections) for x in demods[s].stats()]) for k, s in enumerate(who)})
        finally:
            engine.set_pipelined(False)
            for d in demods:
                d.close()
        return outs

    sync, pipe = run(False), run(True)
    full = schedule + [[0, 1, 2, 3, 4]]
    frames = 0
    for c in range(1, len(full)):
        for s in full[c]:
            if s in full[c - 1]:
                assert np.array_equal(pipe[c][s][0], sync[c - 1][s][0]) and pipe[c][s][1] == sync[c - 1][s][1], (c, s)
                frames += len(sync[c - 1][s][1])
            else:
                assert pipe[c][s][0].size == 0 and pipe[c][s][1] == [], (c, s)      # joined in this call: nothing pending
    assert all(v[0].size == 0 for v in pipe[0].values()) and frames >= 8


def test_pipelined_handle_destroyed_and_recreated_between_two_calls(engine, pkg):
    """throughput mode: a stream is closed while the FEC job of its last call is still pending, and a new handle is created right away (the allocator likes to hand
    out the same address).  The new stream joins with NOTHING pending -- it must not inherit the closed stream's frames, byte count or statistics --, the streams
    that stay get theirs"""
    import torch
    S, calls = 3, 4
    iqs = [orc.transmit(11, 0, 0, nframes=calls, seed=900 + s, esn0_db=12.0, cfo=1e-3, timing=0.25, phase0=0.3)[0] for s in range(S + calls)]
    kb = pkg.modcod_info(11, False, False)['kbch'] // 8
    chunk = iqs[0].size // calls
    cfg = engine.default_cfg(11, False, False)
    demods = [engine.demod(cfg, max_samples=chunk) for _ in range(S)]
    tout = [torch.zeros(4 * kb, dtype=torch.uint8, device='cuda') for _ in range(S)]
    src = list(range(S)); fed = [0] * S
    engine.set_pipelined(True)
    try:
        got_frames = 0
        for c in range(calls):
            tin = [torch.from_numpy(iqs[src[k]][fed[k] * chunk:(fed[k] + 1) * chunk]).cuda() for k in range(S)]
            nb = engine.process_batch(demods, tin, tout)
            for k in range(S): fed[k] += 1
            if c >= 1:
                assert nb[S - 1] == 0 and demods[S - 1].stats() == [], (c, nb)       # the handle created after the previous call: nothing pending
                got_frames += sum(nb[:S - 1])
            # the last stream leaves with its job in flight; its successor takes its place (and, usually, its address)
            demods[S - 1].close()
            demods[S - 1] = engine.demod(cfg, max_samples=chunk)
            src[S - 1] = S + c; fed[S - 1] = 0
        assert got_frames >= 2 * kb
    finally:
        engine.set_pipelined(False)
        for d in demods:
            d.close()


@pytest.mark.parametrize('modcod,short,pilots,esn0,flags', [(14, 1, 1, 12.0, dict(pilot_aided=1)), (27, 1, 1, 18.0, dict(pilot_aided=1, soft_plsc=1)),
                                                            (4, 1, 0, 6.0, dict(soft_plsc=1)), (14, 1, 0, 14.0, dict(pilot_aided=1, soft_plsc=1))])
def test_soft_plsc_and_pilot_aided_modes_equal_oracle(engine, modcod, short, pilots, esn0, flags):
    """SURVEY 8(f) rank 4, behind flags (default off = the reference's behaviour, tested above): soft ML decode of the PLS code in the PLHDR
    demodulator and the block phase estimate from header / pilot symbols in the PLL.  Same equality bar against the oracle."""
    iq, bb, _ = orc.transmit(modcod, short, pilots, nframes=14, seed=70 + modcod, esn0_db=esn0, cfo=1e-3, timing=0.3, phase0=0.1, lead_symbols=400)
    rx = orc.OracleRx(orc.default_cfg(modcod, short, pilots, max_ldpc_trials=30, **flags))
    dm = engine.demod(engine.default_cfg(modcod, bool(short), bool(pilots), max_ldpc_trials=30, **flags), max_samples=20000)
    ndec = 0
    for ncall, a in enumerate(range(0, iq.size, 20000)):
        part = iq[a:a + 20000]
        o, g = rx.process(part), dm.process(part)
        for t in range(4):
            assert same_bits(rx.tap(t), dm.tap(t)), ('tap', t, ncall)
        for a_, b_ in zip(rx.tap(4), dm.stats()):
            assert (a_.detect_modcod, a_.detect_short, a_.detect_pilots, a_.ldpc_trials, a_.bch_corr) == \
                   (b_.detected_modcod, b_.detected_shortframes, b_.detected_pilots, b_.ldpc_trials, b_.bch_corrections)
        assert np.array_equal(o, g)
        ndec += len(g)
    assert ndec >= 8
    dm.close()


@pytest.mark.gpu
def test_time_sliced_front_end_changes_nothing(engine, pkg):
    """the AGC/NCO and timing-recovery stages keep their state in the stream record, so running a call's samples as 1, 4 (default) or 8
    time slices -- the AGC of slice c+1 beside the Gardner loop of slice c on an auxiliary stream -- must give the same bytes, call by call
    (context options, set on fresh engines)"""
    import os, torch
    modcod, S = 14, 6
    iqs = []
    for s in range(S):
        iq, _, _ = orc.transmit(modcod, 1, 0, nframes=6, seed=500 + s, esn0_db=16.0, cfo=4e-4 * (s + 1), timing=0.13 * s, phase0=0.1, lead_symbols=200 + 37 * s)
        iqs.append(iq)
    n = min(i.size for i in iqs)
    cuts = [0, n // 3 + 5, 2 * n // 3 - 11, n]

    def run(eng):
        dms = [eng.demod(eng.default_cfg(modcod, True, False), max_samples=n) for _ in range(S)]
        kb = dms[0].info['kbch'] // 8
        out = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            tin = [torch.from_numpy(np.ascontiguousarray(i[a:b])).cuda() for i in iqs]
            tout = [torch.zeros(8 * kb, dtype=torch.uint8, device='cuda') for _ in range(S)]
            nb = eng.process_batch(dms, tin, tout)
            out.append([tout[s][:nb[s]].cpu().numpy().copy() for s in range(S)])
            out.append([np.array([d.nco_freq()], np.float32) for d in dms])
        for d in dms:
            d.close()
        return out

    ref = run(engine)
    assert sum(x.size for x in ref[0] + ref[2] + ref[4]) > 0
    # ... and so do the later stages: with the stage pipeline (default) the RRC decimator, the PL-sync walk and the frame loops run behind
    # every timing-recovery slice, frames in per-stream slots; without it (context option stage_pipeline = 0) after the last slice on frames the host
    # pooled; stage_loops fixes how many of the slices are followed by a frame-loop launch; stage_post_stream = 0 keeps
    # them on the AGC's stream instead of a third one
    for opts in ({'fe_slices': 1}, {'fe_slices': 8}, {'stage_pipeline': 0}, {'stage_pipeline': 0, 'fe_slices': 1}, {'stage_loops': 4},
                 {'stage_loops': 3, 'fe_slices': 8}, {'stage_loops': 1}, {'stage_post_stream': 0, 'fe_slices': 4}, {'fe_slices': 4}):
        e2 = pkg.Engine(0, options=opts)
        got = run(e2)
        e2.close()
        for a, b in zip(ref, got):
            for x, y in zip(a, b):
                assert np.array_equal(x, y), opts
