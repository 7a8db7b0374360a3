python -m pytest tests/test_gpu_prepost.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -2
for e in "EVR_LPIPS_SCORE_SPLIT=0" "EVR_LPIPS_SCORE_SPLIT=1" "EVR_LPIPS_SCORE_SPLIT=0" "EVR_LPIPS_SCORE_SPLIT=1"; do env $e python bench.py --sub --steps 40 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$e e2vid', d['value'], (d.get('steady_state') or {}).get('value'), d['config']['scores']['lpips'])"; done
for e in "EVR_LPIPS_SCORE_SPLIT=0" "EVR_LPIPS_SCORE_SPLIT=1" "EVR_LPIPS_SCORE_SPLIT=0" "EVR_LPIPS_SCORE_SPLIT=1"; do env $e python bench.py --sub --config firenet --steps 40 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$e firenet', d['value'], (d.get('steady_state') or {}).get('value'), d['config']['scores']['lpips'])"; done
