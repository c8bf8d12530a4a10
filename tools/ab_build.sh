#!/bin/bash
# Development aid: bench.py against variants of s2_rx_kernels.hip built with compile-time switches (on the GPU box):
#   bash tools/ab_build.sh "-DFL_LPS_N=4" "-DFE_PRIO=1" ...   one bench line (value, ms/step, stage times) per variant; restores nothing (scratch copy)
cd $GRAFT_REPO_ROOT/sdrpp-dvbs-demodulator_amd/csrc
for V in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $V -c s2_rx_kernels.hip -o /tmp/s2_rx_v.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdvbs2gpu.so bbts.o bch_kernel.o capi.o dvbs_capi.o dvbs_demod.o dvbs_kernels.o dvbs_segrx.o ldpc_kernel.o s2_demod.o /tmp/s2_rx_v.o segrx.o
  (cd $GRAFT_REPO_ROOT && python bench.py --steps ${STEPS:-6} --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['config']['fraction_equal_to_transmitted'])")
done
