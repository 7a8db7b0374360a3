python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_ckpt.py -m gpu -x -q -k "fire or ckpt or checkpoint or gru" 2>&1 | tail -3
for e in "EVR_C16_OUT=2,3" "EVR_C16_OUT=2,4" "EVR_C16_OUT=4,2" "EVR_C16_OUT=4,3" "EVR_C16_OUT=2,3 EVR_C16_BLOCKS=4"; do env $e python bench.py --sub --config firenet --no-overlap --profile-filter '' --steps 10 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('noov $e', d['value'], ' '.join(f\"{k}={v['us']:.0f}\" for k, v in d['roofline']['layers'].items()))"; done
for rep in 1 2; do
python bench.py --sub --config firenet --steps 40 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('firenet |', d['value'], (d.get('steady_state') or {}).get('value'), (d.get('score_parity') or {}).get('image_max_abs_err'), d['roofline'].get('avg_launch_us'), d['roofline'].get('frac'))"; done
