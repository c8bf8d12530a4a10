#!/bin/bash
# Development aid: device ISA of one csrc/*.hip file -> /tmp/isa/<name>.s; optional second argument = a mangled-name substring,
# whose kernel is cut out into /tmp/isa/<name>.kernel.s.   usage: tools/isa_dump.sh ldpc_kernel [Li12ELi4ELb0ELi2]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/sdrpp-dvbs-demodulator_amd/csrc
mkdir -p /tmp/isa
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include"
case "$1" in s2_rx_kernels|s2_demod|dvbs_demod) FLAGS="$FLAGS -ffp-contract=off";; esac
/opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only -o /tmp/isa/$1.s $SRC/$1.hip 2>&1 | grep -v "argument unused" || true
if [ -n "$2" ]; then
  awk -v pat="$2" '$0 ~ "^_Z.*" pat ".*:" {p=1} p{print} /^\.Lfunc_end/{if(p)exit}' /tmp/isa/$1.s > /tmp/isa/$1.kernel.s
  grep -c "" /tmp/isa/$1.kernel.s
  grep -m4 "NumVgprs\|ScratchSize\|Occupancy\|NumSgprs" /tmp/isa/$1.kernel.s || grep -A12 "$2" /tmp/isa/$1.s | grep -m4 "vgpr_count\|scratch" || true
fi
