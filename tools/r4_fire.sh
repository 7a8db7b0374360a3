python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_ckpt.py -m gpu -x -q -k "fire or ckpt or checkpoint" 2>&1 | tail -3
for rep in 1 2; do
python bench.py --sub --config firenet --steps 40 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('firenet |', d['value'], (d.get('steady_state') or {}).get('value'), (d.get('score_parity') or {}).get('image_max_abs_err'), d['roofline'].get('avg_launch_us'))"; done
