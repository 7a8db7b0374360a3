# Round 6 (GPU box): the Winograd F(2x2, 3x3) path of the exact-fp32 mode -- parity, then per-layer HIP-event times (single stream,
# 64 sequences 346x260) against the direct form, then the two-stream fp32 step.   bash tools/r6_wino.sh [tag]  -> gpurun_out/<tag>/
tag=${1:-r06_wino}; out=gpurun_out/$tag; mkdir -p $out
layers() { python bench.py --sub --no-overlap --profile-filter '' --steps 5 --warmup 2 2>$out/err_$1.txt | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['roofline']['layers']
print('$1', 'fps', d['value'], 'err', (d.get('score_parity') or {}).get('image_max_abs_err'), 'sum_us', round(sum(v['us'] for v in L.values())), ' '.join(f\"{k}={v['us']:.0f}\" for k, v in L.items()))"; }
if [ "$2" != "skiptests" ]; then
EVR_FP32=1 timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15 | tee $out/pytest_model_fp32.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py::test_exact_fp32_twin_winograd_346x260 -x -q -s -m gpu -p no:cacheprovider 2>&1 | tail -15 | tee $out/pytest_twin.txt
fi
EVR_FP32=1 layers wino | tee -a $out/layer_times.txt
EVR_FP32=1 EVR_WINO=0 layers direct | tee -a $out/layer_times.txt
EVR_FP32=1 EVR_WINO_FASTACT=1 layers wino_fastact | tee -a $out/layer_times.txt
for v in "EVR_WINO=1" "EVR_WINO=0"; do
  env EVR_FP32=1 $v python bench.py --sub --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('two-stream fp32 $v', 'fps', d['value'], 'ms', d['ms_per_step'], 'err', (d.get('score_parity') or {}).get('image_max_abs_err'), 'roofline', {k: d['roofline'].get(k) for k in ('achieved', 'frac', 'kernel_us')})" | tee -a $out/layer_times.txt
done
