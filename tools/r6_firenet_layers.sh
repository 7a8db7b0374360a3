#!/bin/bash
# FireNet (config 3), per-layer HIP-event times on one stream: row-walking kernel vs tile kernel
for r in 1 0; do
EVR_C16_ROWS=$r python bench.py --config firenet --sub --no-overlap --profile-filter "" --steps 20 --warmup 3 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['roofline']['layers']
print('rows=$r fps', d['value'], ' '.join(k + '=' + str(round(v['us'])) for k, v in L.items()))"
done
