#!/usr/bin/env python3
"""Generates tests/golden/fec_golden.json (+ small .npy inputs/outputs) from the REFERENCE's own FEC code
compiled into oracle/_ref/libdvbs2ref.so (recipe: oracle/Makefile target `ref`; only works in the
development container where /root/reference exists).  The fixtures are data only: seeded inputs and the
reference's outputs (full arrays for small cases, SHA-256 for large ones).

Run:  python3 tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ldpc_input(rate, short, seed, snr_db):
    """valid codeword (own encoder) + AWGN -> int8 LLRs; fully determined by (rate, short, seed, snr_db)"""
    rng = np.random.default_rng(seed)
    _, bits = orc.encode_frame(rate, short, seed)
    return orc.bits_to_llr(bits, snr_db, rng)


MARGINAL = {0: 0.7, 1: 1.2, 2: 1.3, 3: 2.2, 4: 2.8, 5: 4.1, 6: 4.7, 7: 5.2, 8: 5.7, 9: 6.6, 10: 6.8}


def main():
    R = orc.ref()
    assert R is not None, 'oracle/_ref/libdvbs2ref.so missing: run `make -C oracle ref` where /root/reference exists'
    G = {'ldpc': [], 'bch': [], 'bb_prbs_first_256': None, 'deinterleave': []}
    # ---- LDPC: every code, clean / marginal / noise-only, trials 16 and 50
    for rate, short in orc.ALL_CODES:
        p = orc.fec_params(rate, short)
        for kind, snr in (('clean', MARGINAL[rate] + 4.0), ('marginal', MARGINAL[rate] + 0.3), ('noise', -9.0)):
            for trials in (16, 50):
                if kind != 'marginal' and trials == 50:
                    continue
                seed = 1000 * rate + 100 * short + len(kind)
                llr = ldpc_input(rate, short, seed, snr)
                out = llr.copy()
                ret = R.ref_ldpc_decode(rate, short, out, trials)
                G['ldpc'].append(dict(rate=rate, short=short, kind=kind, seed=seed, snr_db=snr, trials=trials, ret=int(ret),
                                      in_sha=sha(llr), out_sha=sha(out)))
    # one short code with the full vectors stored
    rate, short = 3, 1
    llr = ldpc_input(rate, short, 4242, MARGINAL[rate] + 0.3)
    out = llr.copy()
    ret = R.ref_ldpc_decode(rate, short, out, 16)
    np.save(os.path.join(HERE, 'ldpc_c4_in.npy'), llr)
    np.save(os.path.join(HERE, 'ldpc_c4_out.npy'), out)
    G['ldpc_full'] = dict(rate=rate, short=short, trials=16, ret=int(ret), infile='ldpc_c4_in.npy', outfile='ldpc_c4_out.npy')
    # ---- BCH: each family, 0/1/2/3/t/t+1 flipped bits (+ one in the parity part)
    fams = [(6, 0), (5, 0), (9, 0), (3, 1)]   # N12, N10, N8, S12
    small_in, small_out, small_ret = [], [], []
    for rate, short in fams:
        p = orc.fec_params(rate, short)
        nb = p['K'] // 8
        t = p['t']
        rng = np.random.default_rng(77 + rate + 10 * short)
        for ne in (0, 1, 2, 3, t, t + 1, 2 * t):
            fr = np.zeros(nb, np.uint8)
            orc.lib().orc_make_bbframe(fr, p['kbch'], 500 + ne)
            R.ref_bch_encode(rate, short, fr)
            pos = sorted(int(x) for x in rng.choice(p['K'], ne, replace=False))
            if ne >= 2:
                pos[0] = p['K'] - 5      # one error inside the parity bits
            pos = sorted(set(pos))
            for x in pos:
                fr[x // 8] ^= 1 << (7 - x % 8)
            bad = fr.copy()
            ret = R.ref_bch_decode(rate, short, fr)
            G['bch'].append(dict(rate=rate, short=short, errors=pos, frame_seed=500 + ne, ret=int(ret), in_sha=sha(bad), out_sha=sha(fr)))
            if short:
                small_in.append(bad); small_out.append(fr.copy()); small_ret.append(int(ret))
    np.save(os.path.join(HERE, 'bch_s12_in.npy'), np.stack(small_in))
    np.save(os.path.join(HERE, 'bch_s12_out.npy'), np.stack(small_out))
    G['bch_full'] = dict(rate=3, short=1, ret=small_ret, infile='bch_s12_in.npy', outfile='bch_s12_out.npy')
    # ---- BB scrambler sequence: descrambling zeros yields the PRBS
    z = np.zeros(64800 // 8, np.uint8)
    R.ref_bb_descramble(10, 0, z)     # 9/10 normal: longest kbch
    G['bb_prbs_first_256'] = [int(x) for x in z[:256]]
    G['bb_prbs_sha_7274'] = sha(z[:58192 // 8])
    # ---- bit de-interleaver permutations (index ramps through the reference's S2Deinterleaver)
    for constel, rate, short in [(0, 3, 0), (1, 6, 0), (1, 4, 0), (1, 4, 1), (2, 6, 0), (2, 7, 1), (3, 6, 0), (3, 9, 1)]:
        n = 16200 if short else 64800
        src = (np.arange(n) * 7 % 251).astype(np.int8)
        dst = np.zeros(n, np.int8)
        R.ref_deinterleave(constel, rate, short, src, dst)
        G['deinterleave'].append(dict(constel=constel, rate=rate, short=short, out_sha=sha(dst)))
    with open(os.path.join(HERE, 'fec_golden.json'), 'w') as f:
        json.dump(G, f, indent=1)
    print('wrote', len(G['ldpc']), 'ldpc,', len(G['bch']), 'bch cases')


if __name__ == '__main__':
    main()
