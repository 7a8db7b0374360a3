#!/bin/bash
# per-layer HIP-event times (single stream, 64 sequences) + the two-stream headline:  bash tools/r5_layers.sh <tag> [ENV=VAL ...]
TAG=$1; shift
mkdir -p gpurun_out
env "$@" python bench.py --sub --no-overlap --profile-filter '' --steps 10 --warmup 3 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['roofline']['layers']
print('$TAG', 'single-stream fps', d['value'], 'sum_us', round(sum(v['us'] for v in L.values())), ' '.join(f\"{k}={v['us']:.0f}\" for k, v in L.items()))" | tee -a gpurun_out/r5_layer_times.txt
for rep in 1 2; do env "$@" python bench.py --sub --steps 40 --cpu-frames 0 --parity-frames 4 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$TAG', 'two-stream', d['value'], (d.get('steady_state') or {}).get('value'))" | tee -a gpurun_out/r5_layer_times.txt; done
