#!/bin/bash
# Development aid (GPU box): the bench line's main numbers under different environment switches.  usage: ab_bench.sh "VAR=1 VAR2=x" "VAR=0" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('headline', d['value'], 'ms/step', round(d['ms_per_step'],1), d.get('stage_ms_per_step'))
r=d['roofline']; print('ldpc in-step', r.get('kernel_ms_in_step'), 'alone', r.get('kernel_ms_alone'))
for x in d.get('secondary',[]):
    print('  ', x['config'][:28], {k:v for k,v in x.items() if k in ('value','ms_per_step','bank_4096_msym_s','bank_1_msym_s','ms_per_call','msym_s_total','msym_s_per_stream','stage_ms_per_step','stage_ms_per_call')})
"
done
