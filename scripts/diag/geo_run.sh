for i in 1 2; do
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k,v in d['variants'].items(): print(k, v['ms_per_step'], v.get('host_enqueue_ms_per_step'), (v.get('whole_step_graph') or {}).get('ms_per_step'))
"
done
