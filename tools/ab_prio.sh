set -e
cd $GRAFT_REPO_ROOT/sdrpp-dvbs-demodulator_amd/csrc
for P in 0 1; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -DFE_PRIO=$P -c s2_rx_kernels.hip -o /tmp/s2_rx_p$P.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libdvbs2gpu.so bbts.o bch_kernel.o capi.o dvbs_capi.o dvbs_demod.o dvbs_kernels.o dvbs_segrx.o ldpc_kernel.o s2_demod.o /tmp/s2_rx_p$P.o segrx.o
  cd $GRAFT_REPO_ROOT
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FE_PRIO=$P', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
  cd $GRAFT_REPO_ROOT/sdrpp-dvbs-demodulator_amd/csrc
done
