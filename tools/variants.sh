run() { python bench.py --sub --no-overlap --profile-filter '' --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', 'fps', d['value'], ' '.join(f\"{k}={v['us']:.0f}\" for k, v in d['roofline']['layers'].items()))"; }
run base
EVR_BAND_GCFG=83 run gcfg83
EVR_BAND_GCFG=43 run gcfg43
EVR_BAND_PROG_ALL=1 run progall
EVR_BAND_CFG=43 run cfg43
EVR_BAND_CFG=82 run cfg82
