"""per-layer table of a `bench.py --sub --profile-filter ''` line on stdin:  ... | python tools/layer_table.py [peak_tflops]"""
import json, sys
peak = float(sys.argv[1]) if len(sys.argv) > 1 else 2500.0
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['roofline']['layers']
print('frames/s', d['value'], 'sum_us', round(sum(v['us'] for v in L.values())))
for k, v in L.items():
    print(f"  {k:12s} {v['us']:8.0f} us  {v['tflops']:7.1f} TF/s  {v['tflops'] / peak:.3f}")
