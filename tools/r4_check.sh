# round 4's A/B call after a kernel change:  bash tools/r4_check.sh <tag>   (on the GPU box; appends to gpurun_out/r4_layer_times.txt)
#   parity of the wide / band kernels on the small golden shapes, per-layer HIP-event times (single stream), the two-stream headline twice
EVR_WIDE_MIN=1 EVR_BAND_MIN=1 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -2
run() { python bench.py --sub --no-overlap --profile-filter '' --steps 10 --warmup 3 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['roofline']['layers']
print('$1', 'fps', d['value'], 'sum_us', round(sum(v['us'] for v in L.values())), ' '.join(f\"{k}={v['us']:.0f}\" for k, v in L.items()))"; }
mkdir -p gpurun_out
run ${1:-run} | tee -a gpurun_out/r4_layer_times.txt
for rep in 1 2; do python bench.py --sub --steps 40 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('overlap', d['value'], (d.get('steady_state') or {}).get('value'), (d.get('score_parity') or {}).get('image_max_abs_err'))"; done
