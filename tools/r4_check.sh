EVR_WIDE_MIN=1 EVR_BAND_MIN=1 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -2
run() { python bench.py --sub --no-overlap --profile-filter '' --steps 10 --warmup 3 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['roofline']['layers']
print('$1', 'fps', d['value'], 'sum_us', round(sum(v['us'] for v in L.values())), ' '.join(f\"{k}={v['us']:.0f}\" for k, v in L.items()))"; }
run $1 | tee -a gpurun_out/r4_layer_times.txt
for e in "X=" "X="; do env $e python bench.py --sub --steps 40 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$e overlap', d['value'], (d.get('steady_state') or {}).get('value'), (d.get('score_parity') or {}).get('image_max_abs_err'))"; done
