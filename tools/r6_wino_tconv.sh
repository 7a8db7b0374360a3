#!/bin/bash
# exact-fp32 mode: the transposed decoders (EVR_WINO_TCONV) and the k5 stride-2 encoders in space-to-depth form (EVR_WINO_S2D) on the
# Winograd kernel (default 1) vs the direct implicit GEMM (0): parity suites, per-layer us on one stream, frames/s on two
mkdir -p gpurun_out/r06d
EVR_FP32=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_eval.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2
for cfg in "1 1" "1 0" "0 0"; do set -- $cfg
EVR_FP32=1 EVR_WINO_TCONV=$1 EVR_WINO_S2D=$2 python bench.py --sub --no-overlap --profile-filter "" --steps 10 --warmup 3 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['roofline']['layers']
print('tconv=$1 s2d=$2 fps', d['value'], ' '.join(k + '=' + str(round(v['us'])) for k, v in L.items()))"
EVR_FP32=1 EVR_WINO_TCONV=$1 EVR_WINO_S2D=$2 python bench.py --sub --steps 40 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('tconv=$1 s2d=$2 two streams fps', d['value'], 'err', (d.get('score_parity') or {}).get('image_max_abs_err'))"
done
