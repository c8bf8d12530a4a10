#!/bin/bash
# Development aid (on the GPU box): the LDPC kernel alone for variants of ldpc_kernel.hip built with compile-time switches
#   bash tools/ab_ldpc.sh "" "-DLDPC_EXP=1" ...      (CODES="6,0 4,0" selects the codes, default the headline code)
cd $GRAFT_REPO_ROOT/sdrpp-dvbs-demodulator_amd/csrc
for V in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $V -c ldpc_kernel.hip -o /tmp/ldpc_v.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdvbs2gpu.so bbts.o bch_kernel.o capi.o dvbs_capi.o dvbs_demod.o dvbs_kernels.o dvbs_segrx.o /tmp/ldpc_v.o s2_demod.o s2_rx_kernels.o segrx.o
  echo "== variant [$V]"
  (cd $GRAFT_REPO_ROOT && FRAMES=4096 python tools/ldpc_sweep.py ${CODES:-6,0} 2>/dev/null | cut -c1-24,200-)
done
