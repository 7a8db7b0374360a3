p() { python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1 |', d['value'], (d.get('steady_state') or {}).get('value'))"; }
for i in 1 2; do for g in res0.conv2 res1.conv2; do
EVR_EVAL_GATE=$g python bench.py --sub --cpu-frames 0 --sensor 640x480 2>/dev/null | p "640x480 $g"
EVR_EVAL_GATE=$g python bench.py --sub --cpu-frames 0 --n-seq 1 2>/dev/null | p "nseq1 $g"
EVR_EVAL_GATE=$g python bench.py --sub --cpu-frames 0 --n-seq 8 2>/dev/null | p "nseq8 $g"
EVR_EVAL_GATE=$g python bench.py --sub --cpu-frames 0 --n-seq 32 2>/dev/null | p "nseq32 $g"
EVR_EVAL_GATE=$g python bench.py --sub --cpu-frames 0 --config e2vidplus 2>/dev/null | p "e2vidplus $g"
EVR_EVAL_GATE=$g python bench.py --sub --cpu-frames 0 2>/dev/null | p "headline $g"
done; done
