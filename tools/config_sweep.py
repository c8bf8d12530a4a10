#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json through the same bench code path (bench.main with other constants):
  config 2: DVB-S2 QPSK 1/2 normal frames @50 forced LDPC iterations
  config 5 stand-in: DVB-S2 32APSK 8/9 SHORT frames, pilots ON (MODCOD 27; 32APSK 9/10 short does not exist in the reference)
Each prints the bench's JSON line (metric string unchanged, config.workload names the configuration)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONFIGS = {
    'qpsk12': dict(MODCOD=4, RATE=3, SHORT=0, PILOTS=0, ESN0_DB=8.0, PREROLL=24,
                   WORKLOAD='DVB-S2 QPSK 1/2 normal FECFRAME (MODCOD 4), pilots off, 2 sps, Es/N0 8 dB, 50 forced LDPC iterations'),
    '32apsk89s': dict(MODCOD=27, RATE=9, SHORT=1, PILOTS=1, ESN0_DB=20.0,
                      WORKLOAD='DVB-S2 32APSK 8/9 SHORT FECFRAME (MODCOD 27), pilots on, 2 sps, Es/N0 20 dB, 50 forced LDPC iterations'),
}

if __name__ == '__main__':
    name = sys.argv[1] if len(sys.argv) > 1 else 'qpsk12'
    for k, v in CONFIGS[name].items():
        setattr(bench, k, v)
    sys.argv = [sys.argv[0]] + sys.argv[2:] + ["--no-cpu-baseline", "--no-aux"]
    bench.main()
