#!/bin/bash
# Development aid, on the GPU box: the bench against VARIANTS of one csrc/*.hip file built with compile-time switches.  Every variant
# is compiled and linked under /tmp (the in-tree library and objects are never touched) and loaded through DVBS2GPU_LIB.
#   gpurun -- bash tools/ab.sh s2_rx_kernels "-DGB_PRIO=0" "-DGB_PRIO=1 -DGB_T_N=8"        (STEPS=6 BENCH_ARGS="..." optional)
#   MODE=ldpc: time the LDPC kernel alone (tools/ldpc_sweep.py, RATES=...) instead of the bench; MODE=cmd CMD="...": any command
set -e  # (a variant that does not compile ends the run)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
SRC=$R/sdrpp-dvbs-demodulator_amd/csrc
F=$1; shift
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
case "$F" in s2_rx_kernels|s2_demod|dvbs_demod) FLAGS="$FLAGS -ffp-contract=off";; esac
OTHERS=$(ls $SRC/*.o | grep -v "/$F.o")
for V in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $V -c $SRC/$F.hip -o /tmp/ab_variant.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdvbs2gpu_variant.so $OTHERS /tmp/ab_variant.o
  if [ "${MODE:-bench}" = cmd ]; then
    echo "== $V"; (cd $R && DVBS2GPU_LIB=/tmp/libdvbs2gpu_variant.so bash -c "$CMD")
  elif [ "${MODE:-bench}" = ldpc ]; then
    echo "== $V"; (cd $R && DVBS2GPU_LIB=/tmp/libdvbs2gpu_variant.so python tools/ldpc_sweep.py ${RATES:-6,0})
  else
    (cd $R && DVBS2GPU_LIB=/tmp/libdvbs2gpu_variant.so python bench.py --steps ${STEPS:-8} --warmup ${WARMUP:-1} --no-cpu-baseline --no-secondary ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', d['value'], d['ms_per_step'], d['stage_ms_per_step'], 'ldpc alone', d['roofline']['kernel_ms_alone'])")
  fi
done
