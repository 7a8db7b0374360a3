run() { python bench.py --sub --no-overlap --profile-filter '' --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'fps', d['value'], ' '.join(f\"{k}={v['us']:.0f}\" for k, v in d['roofline']['layers'].items()))"; }
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_modes.py -x -q -m gpu 2>&1 | tail -3
run "$1" | tee -a gpurun_out/mx_layers.txt
