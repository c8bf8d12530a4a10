cd $GRAFT_REPO_ROOT
for o in ldpc_split=1 ldpc_split=0; do
DVBS2GPU_OPTIONS=$o python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', d['value'], d['ms_per_step'], d['stage_ms_per_step'], 'ldpc alone', d['roofline']['kernel_ms_alone'], d.get('value_normal_mode'))"
done
