EVR_DEC_BANDK=1 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "e2vid" 2>&1 | tail -2
run() { python bench.py --sub --no-overlap --profile-filter 'dec' --steps 10 --warmup 3 --cpu-frames 0 --parity-frames 3 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['roofline']['layers']
print('$1', 'fps', d['value'], (d.get('score_parity') or {}).get('image_max_abs_err'), ' '.join(f\"{k}={v['us']:.0f}\" for k, v in L.items()))"; }
run base; EVR_DEC_BANDK=1 run bandk
for e in "X=" "EVR_DEC_BANDK=1" "X=" "EVR_DEC_BANDK=1"; do env $e python bench.py --sub --steps 40 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$e overlap', d['value'], (d.get('steady_state') or {}).get('value'))"; done
