for rep in 1 2; do
for a in "" "--no-overlap"; do python bench.py --sub --config firenet --steps 40 --cpu-frames 0 $a 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('firenet $a |', d['value'], (d.get('steady_state') or {}).get('value'))"; done
done
python tools/eval_cli_profile.py 2>&1 | head -4
