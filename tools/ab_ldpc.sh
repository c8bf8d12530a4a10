#!/bin/bash
# Development aid, on the GPU box: the LDPC kernel alone (rate 3/4 normal, FRAMES x 50 forced iterations) for VARIANTS of one csrc file, every variant timed REPS times in turn
# (boxes differ by ~3 %: only figures of one run compare).   gpurun -- bash tools/ab_ldpc.sh ldpc_split_kernel "-DX=0" "-DX=1"
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.."; pwd)}
SRC=$R/sdrpp-dvbs-demodulator_amd/csrc
F=$1; shift
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
OTHERS=$(ls $SRC/*.o | grep -v "/$F.o")
i=0
for V in "$@"; do
  case "$V" in
    @*) /opt/rocm/bin/hipcc $FLAGS -I$SRC -c -x hip $R/${V#@} -o /tmp/ab_variant_$i.o;;        # "@path": another version of the file (e.g. git show HEAD:... > gpurun_in/old.hip)
    *) /opt/rocm/bin/hipcc $FLAGS $V -c $SRC/$F.hip -o /tmp/ab_variant_$i.o;;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libdvbs2gpu_variant_$i.so $OTHERS /tmp/ab_variant_$i.o
  i=$((i+1))
done
[ -n "$TESTS" ] && (cd $R && DVBS2GPU_LIB=/tmp/libdvbs2gpu_variant_${TEST_VARIANT:-0}.so python -m pytest $TESTS -x -q 2>&1 | tail -2)
for rep in $(seq ${REPS:-3}); do
  i=0
  for V in "$@"; do
    echo -n "[$V] "; (cd $R && DVBS2GPU_LIB=/tmp/libdvbs2gpu_variant_$i.so python tools/ldpc_sweep.py ${RATES:-6,0} 2>/dev/null | grep -o "F [0-9]* ms [0-9.]*" | tr '\n' ' '); echo
    i=$((i+1))
  done
done
