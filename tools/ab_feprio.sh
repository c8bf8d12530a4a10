# A/B of the front-end wave priority (FE_PRIO) under the pipelined bench; variants built into sdrpp-dvbs-demodulator_amd/ab/
for lib in sdrpp-dvbs-demodulator_amd/libdvbs2gpu.so sdrpp-dvbs-demodulator_amd/ab/libdvbs2gpu_feprio1.so sdrpp-dvbs-demodulator_amd/ab/libdvbs2gpu_feprio0.so; do
  echo "== $lib"
  timeout 300 python tools/ab_bench.py $lib --steps 30 --warmup 3 --no-cpu-baseline --no-aux 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['output_bit_exact'])"
done
