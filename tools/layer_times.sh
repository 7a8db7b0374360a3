# Per-layer HIP-event times (single stream, 64 sequences 346x260) for A/B comparisons of kernel-selection switches, pairs run
# back to back on one box:   bash tools/layer_times.sh [repeats]   ->  appends to gpurun_out/layer_times.txt
# (edit the pairs below; switches: EVR_ARITH=mx|mx6|h3, EVR_WIDE=0|2|3, EVR_GROUP_STORE=0, EVR_BAND_PROG_ALL=0, EVR_WIDE_DEC=0, EVR_NO_PRED_DOT=1)
run() { python bench.py --sub --no-overlap --profile-filter '' --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['roofline']['layers']
print('$1', 'fps', d['value'], 'err', d.get('score_parity', {}).get('image_max_abs_err'), 'sum_us', round(sum(v['us'] for v in L.values())), ' '.join(f\"{k}={v['us']:.0f}\" for k, v in L.items()))"; }
mkdir -p gpurun_out
for i in $(seq ${1:-2}); do
EVR_ARITH=mx run mx | tee -a gpurun_out/layer_times.txt
EVR_ARITH=mx6 run mx6 | tee -a gpurun_out/layer_times.txt
EVR_ARITH=h3 run h3 | tee -a gpurun_out/layer_times.txt
done
