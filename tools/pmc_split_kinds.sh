#!/bin/bash
# Development aid, on the GPU box: where the half-row decoder's vector instructions go -- SQ_INSTS_VALU and the kernel time of ONE forced launch (rate 3/4 normal, FRAMES x ITERS)
# for builds in which the pseudo-layers of some kinds do nothing (-DLDPC_SPLIT_SKIP=mask, results wrong on purpose): kind 0 / 7 conflict-free layers, 1 chain layers (walk or attempt), 8 speculative layers.  SNR=8 in the environment: decodable frames instead of noise.
#   gpurun -- bash tools/pmc_split_kinds.sh r06      -> gpurun_out/pmc_split_kinds_<tag>.txt
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
SRC=$R/sdrpp-dvbs-demodulator_amd/csrc
O=$R/gpurun_out/pmc_kinds_$TAG
rm -rf $O; mkdir -p $O
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
OTHERS=$(ls $SRC/*.o | grep -v "/ldpc_split_kernel.o")
export FRAMES=${FRAMES:-4096} ITERS=${ITERS:-50}
for M in 0 0x81 0x02 0x100 0x183; do
  /opt/rocm/bin/hipcc $FLAGS -DLDPC_SPLIT_SKIP=$M -c $SRC/ldpc_split_kernel.hip -o /tmp/kinds_$M.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libkinds_$M.so $OTHERS /tmp/kinds_$M.o || exit 1
  export DVBS2GPU_LIB=/tmp/libkinds_$M.so
  (cd /tmp; export TMPDIR=/tmp; timeout 200 rocprofv3 --kernel-trace --stats --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/m$M -- python3 $R/tools/pmc_ldpc.py 6 > $O/m$M.log 2>&1)
  echo "== LDPC_SPLIT_SKIP=$M"; python $R/tools/rocpd_summary.py $(find $O/m$M -name "*.db" | head -1) 2>&1 | grep -i "ldpc_split"
done > $R/gpurun_out/pmc_split_kinds_$TAG.txt
find $O -name "*.db" -delete
cat $R/gpurun_out/pmc_split_kinds_$TAG.txt
