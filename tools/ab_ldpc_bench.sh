#!/bin/bash
# Development aid (on the GPU box): the whole bench for variants of ldpc_kernel.hip built with compile-time switches
#   bash tools/ab_ldpc_bench.sh "" "-DLDPC_REC_NT=1"
cd $GRAFT_REPO_ROOT/sdrpp-dvbs-demodulator_amd/csrc
for V in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $V -c ldpc_kernel.hip -o /tmp/ldpc_v.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdvbs2gpu.so bbts.o bch_kernel.o capi.o dvbs_capi.o dvbs_demod.o dvbs_kernels.o dvbs_segrx.o /tmp/ldpc_v.o s2_demod.o s2_rx_kernels.o segrx.o
  (cd $GRAFT_REPO_ROOT && python bench.py --steps ${STEPS:-8} --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$V]', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['kernel_ms_alone'])")
done
