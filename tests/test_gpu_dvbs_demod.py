"""GPU parity tests for the DVB-S receive path (a17 front end + a18 slicer + a19 Viterbi) through the C ABI, against the CPU
oracle (oracle/dvbs_fe.cpp, PARITY UNPINNED for the float stages: SDR++/VOLK are not available, see its header).

Exactness: every stage -- AGC, band-edge/RRC FIRs, the interpolator dot products (same documented summation order), the timing
loop, and the FLL / Costas rotations by the engine's own sin/cos (include/dvbs2gpu_math.h, evaluated to the same bits on both
sides) -- is BIT-IDENTICAL to the oracle for any chunking, with all loops active: symbols, loop state and decoded bits are
compared for equality."""
import numpy as np
import pytest
import orc_dvbs as od

pytestmark = pytest.mark.gpu


def _oracle_chain(iq, chunks, **cfgkw):
    rx = od.OracleQpskAlt(od.qpsk_alt_default_cfg(**cfgkw))
    syms, pos = [], 0
    for c in chunks:
        syms.append(rx.process(iq[pos:pos + c]))
        pos += c
    return np.concatenate(syms), rx


def test_frontend_bit_exact_without_fll(engine, pkg):
    """fll_bw = 0 and costas bw = 0: phases stay 0, cos/sin(0) exact -> the whole front end must be bit-identical"""
    iq, _ = od.dvbs_iq(0, 20000, seed=1, esn0_db=10.0, timing=0.37)
    chunks = [3001, 777, 16384, 8192, 11646]
    assert sum(chunks) == iq.size
    exp, rx = _oracle_chain(iq, chunks, fll_bw=0.0, costas_bw=0.0)
    bank = pkg.DvbsDemodBank(engine, 1, max_samples=20000, fll_bw=0.0, loop_bw=0.0)
    got, pos = [], 0
    for c in chunks:
        bank.process(iq[pos:pos + c])
        got.append(bank.symbols())
        pos += c
    got = np.concatenate(got)
    assert got.size == exp.size
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), float(np.abs(got - exp).max())
    st = bank.loop_state()
    assert np.array_equal(st.view(np.uint32), rx.state().view(np.uint32)), (st, rx.state())
    bank.close()


@pytest.mark.parametrize('costas_bw', [0.0, None])
def test_fll_and_costas_active_bit_exact(engine, pkg, costas_bw):
    """FLL active (the band-edge FIRs sit in the loop: systolic array over the lanes), with and without the Costas loop: the
    rotations use the shared sin/cos definition -> symbols and loop state equal the oracle's, chunked arbitrarily"""
    iq, _ = od.dvbs_iq(0, 24576, seed=2, esn0_db=12.0, cfo=2e-3, timing=0.2)
    chunks = [5000, 16384, 777, 16384, iq.size - 5000 - 16384 - 777 - 16384]
    assert 0 < chunks[-1] <= 16384
    okw = {} if costas_bw is None else {'costas_bw': costas_bw}
    gkw = {} if costas_bw is None else {'loop_bw': costas_bw}
    exp, rx = _oracle_chain(iq, chunks, **okw)
    bank = pkg.DvbsDemodBank(engine, 1, max_samples=16384, **gkw)
    got, pos = [], 0
    for c in chunks:
        bank.process(iq[pos:pos + c])
        got.append(bank.symbols())
        pos += c
    got = np.concatenate(got)
    assert got.size == exp.size and np.array_equal(got.view(np.uint32), exp.view(np.uint32)), float(np.abs(got[:min(got.size, exp.size)] - exp[:min(got.size, exp.size)]).max())
    st, es = bank.loop_state(), rx.state()
    assert np.array_equal(st.view(np.uint32), es.view(np.uint32)), (st, es)
    assert abs(abs(st[2]) - 2e-3) < 5e-4                                     # and the FLL did find the carrier offset (loop frequency = -offset)
    bank.close()


@pytest.mark.parametrize('ppm', [1500.0, -1200.0])
def test_timing_phase_sweeping_through_the_whole_bank_bit_exact(engine, pkg, ppm):
    """a sampling clock that is off by ~0.1 %: the timing phase walks through all 256 rows of the interpolator bank and over its ends every few hundred
    symbols -- the written-out per-symbol loop of dvbs_fd_kernel hands every symbol at the ends of the bank (one-sided derivative) and every re-centring
    of its LDS row window to the general form and takes over again behind it; odd chunk sizes (tiles of 1..3 samples, an empty call) on top"""
    iq0, _ = od.dvbs_iq(0, 30000, seed=11, esn0_db=14.0, cfo=5e-4, timing=0.1)
    t = np.arange(int(iq0.size / (1 + ppm * 1e-6)) - 2) * (1 + ppm * 1e-6)
    k = t.astype(np.int64)
    f = (t - k).astype(np.float32)
    iq = (iq0[k] * (1 - f) + iq0[k + 1] * f).astype(np.complex64)                     # (any resampler will do: both sides get the same samples)
    chunks = [1, 2, 3, 0, 7001, 255, 256, 257, 16384, 12288]
    chunks.append(min(16384, iq.size - sum(chunks)))
    iq = iq[:sum(chunks)]
    exp, rx = _oracle_chain(iq, chunks)
    bank = pkg.DvbsDemodBank(engine, 1, max_samples=16384)
    got, pos = [], 0
    for c in chunks:
        bank.process(iq[pos:pos + c])
        got.append(bank.symbols())
        pos += c
    got = np.concatenate(got)
    assert got.size == exp.size and np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    st, es = bank.loop_state(), rx.state()
    assert np.array_equal(st.view(np.uint32), es.view(np.uint32)), (st, es)
    assert abs(exp.size - iq.size / 2 * (1 + ppm * 1e-6)) < 64                       # the loop followed the clock: symbols out = samples / (2 / (1 + ppm))
    bank.close()


@pytest.mark.parametrize('rate,cfo,timing,phase0', [(0, 0.0, 0.0, 0.0), (2, 1e-3, 0.3, 0.4), (4, -5e-4, 0.6, 1.0)])
def test_dvbs_chain_decodes_like_oracle(engine, pkg, rate, cfo, timing, phase0):
    """IQ -> decoded bits through front end, slicer and the self-locking Viterbi: equal to the oracle bit for bit, call by call"""
    nsym = 4096 * 10
    iq, bits = od.dvbs_iq(rate, nsym, seed=5 + rate, esn0_db=12.0, cfo=cfo, timing=timing, phase0=phase0)
    chunk = 16384
    rx = od.OracleQpskAlt()
    o = od.L()
    sl = od.VP(o.orc_dvbs_slicer_create())
    vit = od.OracleViterbi()
    bank = pkg.DvbsDemodBank(engine, 1, max_samples=chunk)
    all_bits = []
    for p in range(0, iq.size, chunk):
        sy = np.ascontiguousarray(rx.process(iq[p:p + chunk]))
        soft = np.zeros(2 * sy.size + 8192, np.int8)
        n = o.orc_dvbs_slicer_process(sl, sy.size, od.P(sy), od.P(soft))
        exp_bits = []
        if n:
            eb, en, es = vit.work(soft[:n].reshape(-1, 8192))
            for b in range(len(en)):
                exp_bits.append(eb[b, :en[b]])
        exp_bits = np.concatenate(exp_bits) if exp_bits else np.zeros(0, np.uint8)
        got_bits = bank.process(iq[p:p + chunk])
        assert np.array_equal(bank.symbols().view(np.uint32), sy.view(np.uint32)), ('symbols', p)
        assert np.array_equal(got_bits, exp_bits), ('decoded bits', p, got_bits.size, exp_bits.size)
        all_bits.append(got_bits)
    all_bits = np.concatenate(all_bits)
    st = bank.stats()[0]
    assert st.state == 1 and st.rate == rate and st.ber < 0.15
    tail = all_bits[-12000:-200]                        # and the decoded stream is the transmitted one
    assert any(np.array_equal(bits[off:off + 64], tail[:64]) and (bits[off:off + len(tail)] == tail).mean() > 0.999
               for off in range(0, len(bits) - len(tail)))
    assert np.array_equal(bank.loop_state().view(np.uint32), rx.state().view(np.uint32))
    bank.close()


def test_dvbs_bank_batch_equals_single(engine, pkg):
    import torch
    S = 5
    iqs = [od.dvbs_iq(r % 5, 12288, seed=40 + r, esn0_db=12.0, timing=0.1 * r)[0] for r in range(S)]
    counts = [24576, 20001, 24576, 777, 16384]
    bank = pkg.DvbsDemodBank(engine, S, max_samples=24576)
    single = [pkg.DvbsDemodBank(engine, 1, max_samples=24576) for _ in range(S)]
    for rep in range(3):
        tin = [torch.from_numpy(iqs[i][:counts[i]]).cuda() for i in range(S)]
        tout = [torch.zeros(4 * 8192 + 24576, dtype=torch.uint8, device='cuda') for _ in range(S)]
        nb = bank.process_batch(tin, tout)
        for i in range(S):
            ref = single[i].process(iqs[i][:counts[i]])
            assert nb[i] == ref.size and np.array_equal(tout[i][:nb[i]].cpu().numpy(), ref), (rep, i)
            assert np.array_equal(bank.symbols(i).view(np.uint32), single[i].symbols().view(np.uint32))
    for b in single:
        b.close()
    bank.close()


@pytest.mark.parametrize('slices', ['1', '3'])
def test_dvbs_bank_kernels_equal_the_wave_per_stream_kernels(engine, pkg, slices):
    """a bank of more than 256 carriers runs the FLL with several streams per wave; forced here for 6 carriers of differing lengths
    (context option dvbs_bank_min = 1), in one time slice and in three: decoded bits, symbols and loop state must equal the wave-per-stream path's
    (which the tests above compare with the oracle), call by call"""
    import os
    import torch
    S = 6
    iqs = [od.dvbs_iq(r % 5, 12288, seed=60 + r, esn0_db=11.0, cfo=(r - 2) * 6e-4, timing=0.13 * r, phase0=0.2 * r)[0] for r in range(S)]
    counts = [[24576, 20001, 24576, 777, 16384, 9000], [1000, 24576, 63, 24576, 130, 24576], [24576, 4575, 24576, 24576, 8192, 15576]]
    single = [pkg.DvbsDemodBank(engine, 1, max_samples=24576) for _ in range(S)]
    e2 = pkg.Engine(0, options={'dvbs_bank_min': 1, 'dvbs_fe_slices': int(slices)})
    bank = pkg.DvbsDemodBank(e2, S, max_samples=24576)
    pos = [0] * S
    for rep in range(3):
        tin = [torch.from_numpy(iqs[i][pos[i]:pos[i] + counts[rep][i]]).cuda() for i in range(S)]
        tout = [torch.zeros(4 * 8192 + 24576, dtype=torch.uint8, device='cuda') for _ in range(S)]
        nb = bank.process_batch(tin, tout)
        for i in range(S):
            ref = single[i].process(iqs[i][pos[i]:pos[i] + counts[rep][i]])
            pos[i] += tin[i].numel()
            assert np.array_equal(bank.symbols(i).view(np.uint32), single[i].symbols().view(np.uint32)), (rep, i)
            assert np.array_equal(bank.loop_state(i).view(np.uint32), single[i].loop_state().view(np.uint32)), (rep, i)
            assert nb[i] == ref.size and np.array_equal(tout[i][:nb[i]].cpu().numpy(), ref), (rep, i)
    for b in single:
        b.close()
    bank.close()
    e2.close()


def test_time_sliced_dvbs_front_end_changes_nothing(engine, pkg):
    """small banks run AGC / FLL + RRC / timing recovery / Costas + soft FIFO + Viterbi as time-sliced stages on three streams (every stage keeps
    its state in the stream record); 1 slice (what a GPU-filling bank uses), 3, 8, the maximum 32 and the default 24 must give the same symbols,
    loop state and decoded bits, call by call
    (context options dvbs_fe_slices, dvbs_agc_stream)"""
    import os
    iq, _ = od.dvbs_iq(2, 30000, seed=7, esn0_db=9.0, cfo=8e-4, timing=0.41, phase0=0.3)
    chunks = [4097, 12288, 900, 20000, iq.size - 4097 - 12288 - 900 - 20000]
    assert min(chunks) > 0

    def run(eng):
        bank = pkg.DvbsDemodBank(eng, 1, max_samples=max(chunks))
        out, pos = [], 0
        for c in chunks:
            bits = bank.process(iq[pos:pos + c])
            out += [bits.copy(), bank.symbols().view(np.uint32).copy(), bank.loop_state().view(np.uint32).copy()]
            pos += c
        bank.close()
        return out

    ref = run(engine)
    assert sum(x.size for x in ref[1::3]) > 20000
    # (dvbs_agc_stream = 0: the AGC slices on the Viterbi stream instead of a stream of their own)
    for opts in ({'dvbs_fe_slices': 1}, {'dvbs_fe_slices': 3}, {'dvbs_fe_slices': 8}, {'dvbs_fe_slices': 32}, {'dvbs_agc_stream': 0},
                 {'dvbs_agc_stream': 0, 'dvbs_fe_slices': 5}):
        e2 = pkg.Engine(0, options=opts)
        got = run(e2)
        e2.close()
        assert len(got) == len(ref)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), opts


def test_dvbs_demod_error_codes(engine, pkg):
    import ctypes as C
    with pytest.raises(pkg.Dvbs2GpuError):
        pkg.DvbsDemodBank(engine, 1, max_samples=1000, rrc_taps=33)
    # rate ratios / clock limits the symbol buffers are not sized for are refused (samples per symbol * (1 - omega_rel_limit) >= 1.5)
    for kw in (dict(samplerate=2.4e6, symbolrate=2e6), dict(omega_rel_limit=0.3), dict(omega_rel_limit=-0.01), dict(symbolrate=0.0)):
        with pytest.raises(pkg.Dvbs2GpuError) as e:
            pkg.DvbsDemodBank(engine, 1, max_samples=1000, **kw)
        assert e.value.code == pkg.ERR_ARG, kw
    ok = pkg.DvbsDemodBank(engine, 1, max_samples=1000, samplerate=3.2e6, symbolrate=2e6)      # 1.6 samples per symbol: accepted
    ok.close()
    bank = pkg.DvbsDemodBank(engine, 2, max_samples=1000)
    assert engine.lib.dvbs2gpu_dvbs_demod_process(bank.h, 10, None, None, 0) == pkg.ERR_ARG      # host entry needs a 1-stream bank
    iq = np.zeros(2000, np.complex64)
    one = pkg.DvbsDemodBank(engine, 1, max_samples=1000)
    out = np.zeros(100, np.uint8)
    assert engine.lib.dvbs2gpu_dvbs_demod_process(one.h, 2000, C.c_void_p(iq.ctypes.data), C.c_void_p(out.ctypes.data), 100) == pkg.ERR_ARG
    assert one.process(iq[:0]).size == 0
    bank.close(); one.close()
    for nseg, own, warm in ((0, 16384, 8192), (2, 8192, 16384), (2, 16384, 4096), (1, 1 << 30, 8192)):   # the last: beyond the bank's int counts
        with pytest.raises(pkg.Dvbs2GpuError) as e:
            pkg.DvbsSegmentReceiver(engine, nseg, own, warm)
        assert e.value.code == pkg.ERR_ARG


def _match_offset(bits, ref):
    """(index in ref where bits[0:256] occurs exactly, inverted?) or (-1, 0)"""
    k = ref.tobytes().find(bits[:256].tobytes())
    if k >= 0:
        return k, 0
    return ref.tobytes().find((bits[:256] ^ 1).tobytes()), 1


@pytest.mark.parametrize('rate', [0, 2])
def test_dvbs_segment_receiver_returns_one_continuous_bit_stream(pkg, engine, rate):
    """dvbs2gpu_dvbs_segrx_*: one continuous DVB-S carrier (carrier, phase and timing offsets) handed over in chunks; the returned bits
    are the transmitted information bits (up to the polarity QPSK leaves open, which the deframer behind accepts), in order and without a
    gap or a repeat, from shortly after the first segment's lock to (almost) the end of what was handed over"""
    import torch
    nseg, own, warm = 4, 49152, 32768
    nsym = 5 * nseg * own // 2 + 20000
    iq, bits = od.dvbs_iq(rate, nsym, seed=31 + rate, esn0_db=12.0, cfo=1e-4, timing=0.3, phase0=0.6)   # (at 9 dB or 5e-4 the reference's FLL / Costas pair
    # still slips a cycle now and then 50 k symbols after a start: a property of the loops, seen as a discontinuity by any receiver)
    rx = pkg.DvbsSegmentReceiver(engine, nseg, own, warm)
    d_iq = torch.from_numpy(iq).cuda()
    out = torch.zeros(2 * nseg * own * 2 + 4 * 65536, dtype=torch.uint8, device='cuda')
    got, a, k, unmatched = [], 0, 0, 0
    sizes = [rx.chunk_samples, rx.chunk_samples // 2 + 777, 100, 20001]      # (short calls re-run the kept history and add what is new)
    while a < iq.size:
        n = min(sizes[k % 4], iq.size - a)
        nb = rx.process(d_iq[a:a + n], out)
        got.append(out[:nb].cpu().numpy().copy())
        unmatched += rx.stats()['unmatched']
        a += n
        k += 1
    rx.close()
    got = np.concatenate(got)
    ref = np.asarray(bits, np.uint8)
    skip = 70000                                          # the first segment of the stream settles (loops, 180-degree slips) like any receiver start
    off, inv = _match_offset(got[skip:], ref)
    assert off >= 0 and unmatched == 0, (off, unmatched)
    n = min(got.size - skip, ref.size - off)
    errs = int(np.count_nonzero((got[skip:skip + n] ^ inv) != ref[off:off + n]))
    assert errs <= 8, (errs, n, off, inv)                # one bit stream: a slip anywhere would make half of the rest differ
    assert n >= ref.size - off - 3 * 8192, (n, ref.size, off)


def test_dvbs_bank_of_4096_carriers_equals_the_oracle_on_sampled_streams(engine, pkg):
    """the receiver bank at the bench's size (4096 carriers: four-streams-per-wave FLL, time-sliced front end, every stage kernel co-resident with the others'
    slices): sixteen sampled carriers go through the CPU restatement of DVBSDemod::process' receive side (module_dvbs_demod.cpp:78-117: QPSK_ALT front end,
    slicer, self-locking Viterbi) with the same inputs, call by call -- decoded bits, symbols and loop state of those carriers must be the oracle's, bit for bit"""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    S, D, calls, chunk = 4096, 8, 3, 16384
    base = [od.dvbs_iq(r % 5, chunk * calls // 2, seed=80 + r, esn0_db=11.0, cfo=(r - 3) * 4e-4, timing=0.11 * r, phase0=0.3 * r)[0] for r in range(D)]
    n = base[0].size
    d_base = [torch.from_numpy(b).cuda() for b in base]
    shift = lambda s: 2 * ((s // D) * 977 % (n // 2))
    iq = torch.empty((S, n), dtype=torch.complex64, device='cuda')
    for s in range(S):
        iq[s] = torch.roll(d_base[s % D], shift(s))
    bank = pkg.DvbsDemodBank(engine, S, max_samples=chunk)
    tout = [torch.zeros(4 * 8192 + chunk, dtype=torch.uint8, device='cuda') for _ in range(S)]
    sample = [0, 1, 7, 8, 9, 255, 256, 1000, 2047, 2048, 2049, 3000, 3333, 4000, 4094, 4095]

    def oracle_stream(s):
        x = np.roll(base[s % D], shift(s))
        rx = od.OracleQpskAlt()
        o = od.L()
        sl = od.VP(o.orc_dvbs_slicer_create())
        vit = od.OracleViterbi()
        out = []
        for c in range(calls):
            sy = np.ascontiguousarray(rx.process(x[c * chunk:(c + 1) * chunk]))
            soft = np.zeros(2 * sy.size + 8192, np.int8)
            k = o.orc_dvbs_slicer_process(sl, sy.size, od.P(sy), od.P(soft))
            bits = []
            if k:
                eb, en, es = vit.work(soft[:k].reshape(-1, 8192))
                bits = [eb[b, :en[b]] for b in range(len(en))]
            out.append((np.concatenate(bits) if bits else np.zeros(0, np.uint8), sy, rx.state().copy()))
        return out

    with ThreadPoolExecutor(max_workers=16) as ex:
        want = dict(zip(sample, ex.map(oracle_stream, sample)))
    nbits = 0
    for c in range(calls):
        nb = bank.process_batch([iq[s, c * chunk:(c + 1) * chunk] for s in range(S)], tout)
        for s in sample:
            bits, sy, st = want[s][c]
            assert np.array_equal(bank.symbols(s).view(np.uint32), sy.view(np.uint32)), ('symbols', c, s)
            assert np.array_equal(bank.loop_state(s).view(np.uint32), st.view(np.uint32)), ('loop state', c, s)
            assert nb[s] == bits.size and np.array_equal(tout[s][:nb[s]].cpu().numpy(), bits), ('decoded bits', c, s)
            nbits += bits.size
    assert nbits > 16 * 8192
    bank.close()
