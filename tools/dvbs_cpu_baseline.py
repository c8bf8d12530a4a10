#!/usr/bin/env python3
"""BASELINE config D on the host cores: the CPU restatement of DVBSDemod::process (oracle/: QPSK_ALT front end -> slicer -> Viterbi_DVBS ->
TS deframer -> Forney -> RS(204,188) -> energy dispersal), ONE stream on ONE thread, timed stage by stage -- the number the GPU
figures of DESIGN.md §8 (DVB-S) stand beside.  kind = "port": the front end and the inner decoder are restatements (SDR++ core and VOLK are
not available), the tail stages are the reference's own sources (oracle/_ref)."""
import os
import sys
import time
import json
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import orc_dvbs as od
import orc_dvbs_tail as ot
from test_gpu_dvbs_tail import OracleTail


def main():
    npk = int(os.environ.get('PACKETS', 640))
    obits, ts = ot.dvbs_outer_tx(npk, seed=5)
    enc = od.cc_encode(obits)
    nsym = enc.size // 2
    iq = np.zeros(2 * nsym, np.complex64)
    od.LF().orc_dvbs_modulate(od.P(np.ascontiguousarray(enc)), nsym, 9.0, 5e-4, 0.3, 0.2, 7, od.P(iq))
    chunk = 65536
    rx, o = od.OracleQpskAlt(), od.L()
    sl, vit, tail = od.VP(o.orc_dvbs_slicer_create()), od.OracleViterbi(), OracleTail()
    t = {'front_end': 0.0, 'slicer': 0.0, 'viterbi': 0.0, 'tail': 0.0}
    npkts = 0
    sent = {bytes(p) for p in ts}
    hits = 0
    for p in range(0, iq.size, chunk):
        t0 = time.perf_counter()
        sy = np.ascontiguousarray(rx.process(iq[p:p + chunk]))
        t1 = time.perf_counter()
        soft = np.zeros(2 * sy.size + 8192, np.int8)
        n = o.orc_dvbs_slicer_process(sl, sy.size, od.P(sy), od.P(soft))
        t2 = time.perf_counter()
        bits = []
        if n:
            eb, en, es = vit.work(soft[:n].reshape(-1, 8192))
            bits = [eb[b, :en[b]] for b in range(len(en))]
        t3 = time.perf_counter()
        out, nf = tail.process(np.concatenate(bits) if bits else np.zeros(0, np.uint8))
        t4 = time.perf_counter()
        t['front_end'] += t1 - t0; t['slicer'] += t2 - t1; t['viterbi'] += t3 - t2; t['tail'] += t4 - t3
        pk = out.reshape(-1, 188)
        npkts += len(pk)
        hits += sum(bytes(x) in sent for x in pk)
    total = sum(t.values())
    print(json.dumps({'workload': 'DVB-S QPSK 1/2, one stream, %d symbols, IQ -> TS packets' % nsym, 'cores': 1, 'kind': 'port',
                      'Msymbols_per_s': round(nsym / total / 1e6, 3), 'seconds': round(total, 2),
                      'share': {k: round(v / total, 3) for k, v in t.items()}, 'ts_packets': npkts, 'transmitted_ones': hits}))


if __name__ == '__main__':
    main()
