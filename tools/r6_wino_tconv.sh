#!/bin/bash
# exact-fp32 mode: the transposed decoders on the Winograd kernel (EVR_WINO_TCONV=1, default) vs the direct implicit GEMM (=0)
mkdir -p gpurun_out/r06d
EVR_FP32=1 timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_eval.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2
for t in 1 0; do
EVR_FP32=1 EVR_WINO_TCONV=$t python bench.py --sub --no-overlap --profile-filter "" --steps 10 --warmup 3 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
L = d['roofline']['layers']
print('tconv=$t fps', d['value'], 'err', (d.get('score_parity') or {}).get('image_max_abs_err'), ' '.join(k + '=' + str(round(v['us'])) for k, v in L.items()))"
EVR_FP32=1 EVR_WINO_TCONV=$t python bench.py --sub --steps 40 --cpu-frames 0 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('tconv=$t two streams fps', d['value'], 'err', (d.get('score_parity') or {}).get('image_max_abs_err'))"
done
