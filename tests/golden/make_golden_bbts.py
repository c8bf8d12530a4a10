#!/usr/bin/env python3
"""Generates tests/golden/bbts_golden.json: known-answer cases for the BBFRAME -> TS / GSE parser.  The reference's translation unit
(dvbs2/bbframe_ts_parser.cpp) cannot be compiled here (it includes SDR++ core's <dsp/stream.h>), so these vectors are NOT reference
outputs: the inputs come from this repo's transmitter side (tests/orc_bbts.py, per EN 302 307-1 5.1.4-5.1.6 and TS 102 606) and the
expected outputs are (a) the transmitted TS packets / PDUs themselves where the case is a clean round trip, (b) the oracle
restatement's output for the fuzzed cases (regression anchors; PARITY UNPINNED for this row).  Inputs are regenerated from seeds.

Run:  python3 tests/golden/make_golden_bbts.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc_bbts as B  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    G = {'ts_round_trip': [], 'fuzz': []}
    for kbch, dfl in ((14232, None), (48408, None), (3072, None), (14232, 1000)):
        rng = np.random.default_rng(kbch + (dfl or 0))
        nfr = 9
        D = dfl if dfl is not None else kbch // 8 - 10
        pk = B.ts_packets(nfr * D // 188 + 2, rng)
        fr = B.bbframes_from_ts(pk, kbch, nfr, dfl)
        p = B.OracleBbTs(kbch)
        out = np.concatenate([p.work(fr[:4]), p.work(fr[4:])])
        n = (nfr * D - 1) // 188
        assert np.array_equal(out.reshape(-1, 188), pk[:n])
        G['ts_round_trip'].append({'kbch': kbch, 'dfl_bytes': dfl, 'seed': kbch + (dfl or 0), 'nframes': nfr, 'packets_out': n,
                                   'sha256_out': sha(out), 'sha256_in': sha(fr)})
    for seed, kbch, choices in ((1, 3072, (3, 3, 0, 2)), (2, 14232, (3, 1, 1, 0)), (3, 3072, (1, 1, 3))):
        rng = np.random.default_rng(seed)
        p = B.OracleBbTs(kbch)
        outs, stats = [], []
        for call in range(6):
            fr = B.fuzz_frames(rng, kbch, int(rng.integers(0, 6)), ts_gs_choices=choices, p_bad=0.2)
            o = p.work(fr, cap=fr.size + 376)
            outs.append(sha(o))
            st = p.stats()
            stats.append([st['synched'], st['last_bb_proc'], st['last_gse_crc_err'], st['ts_gs'], int(o.size)])
        G['fuzz'].append({'seed': seed, 'kbch': kbch, 'ts_gs_choices': list(choices), 'calls': 6, 'sha256_out_per_call': outs, 'state_per_call': stats})
    with open(os.path.join(HERE, 'bbts_golden.json'), 'w') as f:
        json.dump(G, f, indent=1)
    print('wrote bbts_golden.json')


if __name__ == '__main__':
    main()
