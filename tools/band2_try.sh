run() { python bench.py --sub --no-overlap --profile-filter '' --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['roofline']['layers']
print('$1', 'fps', d['value'], 'sum_us', round(sum(v['us'] for v in L.values())), ' '.join(f\"{k}={v['us']:.0f}\" for k, v in L.items()))"; }
mkdir -p gpurun_out
for i in 1 2; do
run base | tee -a gpurun_out/mx_layers.txt
EVR_BAND_PROG_ALL=1 run progall | tee -a gpurun_out/mx_layers.txt
done
