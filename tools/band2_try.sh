run() { python bench.py --sub --no-overlap --profile-filter '' --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['roofline']['layers']
print('$1', 'fps', d['value'], 'sum_us', round(sum(v['us'] for v in L.values())), ' '.join(f\"{k}={v['us']:.0f}\" for k, v in L.items()))"; }
mkdir -p gpurun_out
for i in 1 2 3; do
EVR_GROUP_STORE=1 run gs1 | tee -a gpurun_out/mx_layers.txt
EVR_GROUP_STORE=0 run gs0 | tee -a gpurun_out/mx_layers.txt
done
EVR_GROUP_STORE=1 python bench.py --sub --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('gs1 overlap fps', d['value'])"
EVR_GROUP_STORE=0 python bench.py --sub --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('gs0 overlap fps', d['value'])"
